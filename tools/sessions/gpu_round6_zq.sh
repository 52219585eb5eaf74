#!/bin/bash
# round-6 session zq: demod_bwd_styles_kernel (a 128-step weight walk per thread, unrolled by sixteen) and skinny_wgrad_kernel (sixteen conditional loads per step made unconditional):
# the training-side suites, then the training iteration with this build and the previous one (pix2pix3d_amd/libp3d_hip_base.so), interleaved on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zq
( timeout 1500 python -m pytest tests/test_conv_layer_gpu.py tests/test_conv_grad_gpu.py tests/test_loss_phases.py tests/test_train_full.py tests/test_train_step.py tests/test_discriminator.py tests/test_bcast_gpu.py tests/test_small_ops_gpu.py -q -m gpu -x --tb=short 2>&1 | tail -8 ) > gpurun_out/${T}_gputest.log 2>&1
grep -n "passed\|failed" gpurun_out/${T}_gputest.log | tail -2
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_base.so; else unset P3D_LIB_PATH; fi
    timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_${v}_${rep}.json 2>gpurun_out/${T}_train_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_${v}_${rep}.json')); t = d.get('train_step', d); print('$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_${v}_${rep}.err
  done
done
unset P3D_LIB_PATH
echo finished
