#!/bin/bash
# round-6 session p: conv_wgrad_x6p_kernel (bf16x6 weight gradient, operands split once per work-group) — bit-identity with the in-register kernel, the weight-gradient
# (NOT KEPT: the kernel this script measured is profiles/round6_p_wgrad_x6_presplit_not_kept.diff; without it P3D_WGRAD_X6_PRESPLIT is read by nobody)
# timings both ways, then the training iteration both ways (interleaved, one box).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_p
timeout 1200 python -m pytest tests/test_conv_grad_gpu.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
rm -f gpurun_out/wgrad_variants.txt
for rep in 1 2; do
  for v in 0 1; do
    WGRAD_X6=1 P3D_WGRAD_X6_PRESPLIT=$v timeout 300 python tests/gpu_time_wgrad.py presplit=$v 2>&1 | grep "float32" | head -9
  done
done
cp gpurun_out/wgrad_variants.txt gpurun_out/${T}_wgrad_variants.txt
for rep in 1 2; do
  for v in 0 1; do
    P3D_WGRAD_X6_PRESPLIT=$v timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_${v}_${rep}.json 2>gpurun_out/${T}_train_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_${v}_${rep}.json')); t = d.get('train_step', d); print('wgrad_x6_presplit=$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_${v}_${rep}.err
  done
done
echo finished
