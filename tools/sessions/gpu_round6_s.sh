#!/bin/bash
# round-6 session s: convT_r2_bf16x3_kernel (the backbone's x2 transposed convolutions, bf16x3 on split activations, one parity class per block on a slab) against the
# (NOT KEPT: the kernel is profiles/round6_s_convT_r2_bf16x3_not_kept.diff; without it P3D_CONVT_R2 is read by nobody)
# generic kernel (P3D_CONVT_R2=0): parity, then the inference line both ways, interleaved, and the eager kernel statistics.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_s
timeout 1800 python -m pytest tests/test_split_acts.py tests/test_conv_gpu.py tests/test_model_gpu.py tests/test_model_full.py tests/test_model_variants.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
grep "transposed slab kernel" gpurun_out/${T}_gputest.log | head
for rep in 1 2 3; do
  for v in 0 1; do
    P3D_CONVT_R2=$v timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('convT_r2=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'], d['mfma_conv']['conv_bf16x3']['frac_of_peak'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
export TMPDIR=/tmp
for v in 0 1; do
  ( cd /tmp && P3D_CONVT_R2=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o e -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-train-step --no-exact-fp32 --no-configs 2>/dev/null | tail -1 ) > /dev/null
  find /tmp/prof_$v -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_kernel_stats_${v}.csv \;
  grep -i "convT_r2\|conv2d_nhwc_kernel<float, true, true\|conv2d_nhwc_kernel<float, true, false, false, true" gpurun_out/${T}_kernel_stats_${v}.csv | cut -c1-170
done
echo finished
