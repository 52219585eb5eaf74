#!/bin/bash
# round-6 session d: 16-byte stores for the split (bf16 hi | lo) activations — fir4_cl_fused_kernel<float> (lane pairs trade halves) and conv3x3_r2_bf16x3_kernel
# (4 x 4 quad transpose in registers + ds_swizzle) — parity, then the inference line with the old stores (P3D_FIR4_STORE8 / P3D_R2_STORE2 = 1) and the new, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_d
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_split_acts.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_model_full.py tests/test_model_variants.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2 3; do
  for v in 11 01 00; do
    P3D_FIR4_STORE8=${v:0:1} P3D_R2_STORE2=${v:1:1} timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('fir4_store8,r2_store2=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
export TMPDIR=/tmp
for v in 11 00; do
  ( cd /tmp && P3D_FIR4_STORE8=${v:0:1} P3D_R2_STORE2=${v:1:1} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o e -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-train-step --no-exact-fp32 --no-configs 2>/dev/null | tail -1 ) > /dev/null
  find /tmp/prof_$v -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_kernel_stats_${v}.csv \;
  grep -i "fir4_cl_fused_kernel<float>\|conv3x3_r2_bf16x3" gpurun_out/${T}_kernel_stats_${v}.csv | cut -c1-200
done
echo finished
