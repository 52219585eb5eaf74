#!/bin/bash
# round-6 session m: bf16x6 3x3 convolution on operands split once per work-group (conv3x3_halo_x6p_kernel) against the in-register form, same box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_m
timeout 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -x --tb=short -k "bf16x6 or x6" > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -5 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2; do
  for v in 0 1; do
    P3D_BF16X3=0 P3D_X6_PRESPLIT=$v timeout 300 python tests/gpu_time_x6.py 2>&1 | grep "^x6" | tee -a gpurun_out/${T}_time_x6.log
  done
done
echo finished
