#!/bin/bash
# round-5 session q: LDS row pitch of the fp32 weight-gradient image (128 / 64 floats -> 160 / 96, bf16x6: 132): the kernel alone, old vs new pitch, exact and bf16x6,
# interleaved; then every gradient test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_q
rm -f gpurun_out/wgrad_variants.txt
for rep in 1 2; do
  P3D_WGRAD_PITCH_OLD=1 WGRAD_X6=0 timeout 200 python tests/gpu_time_wgrad.py exact_p128 > /dev/null 2>&1
  WGRAD_X6=0 timeout 200 python tests/gpu_time_wgrad.py exact_new > /dev/null 2>&1
  P3D_WGRAD_PITCH_OLD=1 WGRAD_X6=1 timeout 200 python tests/gpu_time_wgrad.py x6_p128 > /dev/null 2>&1
  WGRAD_X6=1 timeout 200 python tests/gpu_time_wgrad.py x6_new > /dev/null 2>&1
done
grep float32 gpurun_out/wgrad_variants.txt | sort -k3,7 -s | cut -c1-100
cp gpurun_out/wgrad_variants.txt gpurun_out/${T}_wgrad_variants.txt
timeout 900 python -m pytest tests/test_conv_grad_gpu.py -q -m gpu > gpurun_out/${T}_gputest.log 2>&1; echo "grad tests exit $?"
tail -3 gpurun_out/${T}_gputest.log | cut -c1-200
echo finished
