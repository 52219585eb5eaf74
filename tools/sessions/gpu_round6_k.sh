#!/bin/bash
# round-6 session k: the next chunk's slab one LDS-DMA piece per tap (conv3x3_h2_f16_kernel, conv3x3_r2_bf16x3_kernel) instead of six behind tap 0's rendezvous.
# Parity, then the inference line with the previous build (tools/ab/libp3d_hip_burst.so via P3D_LIB_PATH) and this one (…_spread.so), interleaved, and the eager kernel statistics of both.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_k
timeout 1800 python -m pytest tests/test_conv_gpu.py tests/test_srheads.py tests/test_split_acts.py tests/test_model_gpu.py tests/test_model_full.py tests/test_model_variants.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2 3; do
  for v in burst spread; do
    P3D_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libp3d_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('slab=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'], d['mfma_conv']['conv_f16']['frac_of_peak'], d['mfma_conv']['conv_bf16x3']['frac_of_peak'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
export TMPDIR=/tmp
for v in burst spread; do
  ( cd /tmp && P3D_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libp3d_hip_$v.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o e -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-train-step --no-exact-fp32 --no-configs 2>/dev/null | tail -1 ) > /dev/null
  find /tmp/prof_$v -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_kernel_stats_${v}.csv \;
  grep -i "conv3x3_h2_f16\|conv3x3_r2_bf16x3" gpurun_out/${T}_kernel_stats_${v}.csv | cut -c1-160
done
echo finished
