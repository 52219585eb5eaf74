#!/bin/bash
# round-6 session ze: conv3x3_r2_bf16x3_kernel<TR> (p3d_conv3x3_torgb_split): the backbone's last 3x3 layer + wide ToRGB + skip-image sum in one launch.  Parity, then the
# inference line with P3D_FUSE_CONV_WIDE_TORGB=0 / 1, interleaved on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_ze
timeout 900 python -m pytest tests/test_split_acts.py -q -m gpu -x --tb=short -s > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -5 gpurun_out/${T}_gputest.log | cut -c1-400
grep "vs two launches" gpurun_out/${T}_gputest.log | cut -c1-200
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_model_full.py tests/test_conv_gpu.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest2.log 2>&1; echo "tests2 exit $?"
tail -5 gpurun_out/${T}_gputest2.log | cut -c1-400
for rep in 1 2 3; do
  for v in 0 1; do
    P3D_FUSE_CONV_WIDE_TORGB=$v timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs --no-exact-fp32 > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('fused=$v rep $rep:', d['value'], 'img/s,', d['ms_per_step'], 'ms,', d.get('stage_ms'), 'conv_bf16x3', d.get('mfma_conv', {}).get('conv_bf16x3'))" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
echo finished
