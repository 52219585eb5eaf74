#!/bin/bash
# round-5 session r: the fused ToRGB at Co = 256 (conv3x3_h2_f16_kernel<false, true>: a work-group walks both channel blocks of its patch) — parity, then the
# inference line with and without it, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_r
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_srheads.py tests/test_model_gpu.py tests/test_model_full.py -q -m gpu -x > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2; do
  for v in 0 1; do
    P3D_FUSE_TORGB_256=$v timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_fuse256_${v}_${rep}.json 2>/dev/null
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_fuse256_${v}_${rep}.json')); print('fuse256=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'], d['mfma_conv']['conv_f16'])"
  done
done
echo finished
