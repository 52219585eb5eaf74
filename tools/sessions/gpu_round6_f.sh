#!/bin/bash
# round-6 session f: 16-byte memory instructions, continued — the split-K partial tiles of conv2d_nhwc_kernel (P3D_CONV_STORE4=2: the 4-byte stores) and the epilogue
# of torgb_wide_split_kernel (P3D_TORGB_STORE4=1: four-byte skip-tap loads and stores) — parity, then the inference line each way, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_f
timeout 1800 python -m pytest tests/test_conv_gpu.py tests/test_split_acts.py tests/test_conv_grad_gpu.py tests/test_model_gpu.py tests/test_model_full.py tests/test_model_variants.py tests/test_srheads.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2 3; do
  for v in 21 01 00; do
    P3D_CONV_STORE4=${v:0:1} P3D_TORGB_STORE4=${v:1:1} timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('conv_partial4,torgb4=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
echo finished
