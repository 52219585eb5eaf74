#!/bin/bash
# round-6 session zd: conv3x3_ws64_f16_kernel (64 -> 64 fp16, weights resident in LDS): parity, its launch against the halo-slab kernel's, the training iteration either way
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zd
timeout 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -x --tb=short -k "weight_stationary or halo_kernels" > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
timeout 300 python tests/gpu_time_ws64.py > gpurun_out/${T}_time.txt 2>&1; cat gpurun_out/${T}_time.txt | tail -20
timeout 900 python -m pytest tests/test_discriminator.py tests/test_loss_phases.py tests/test_conv_grad_gpu.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest2.log 2>&1; echo "tests2 exit $?"
tail -4 gpurun_out/${T}_gputest2.log | cut -c1-300
for rep in 1 2; do
  for v in 0 1; do
    P3D_CONV_WS64=$v timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_${v}_${rep}.json 2>gpurun_out/${T}_train_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_${v}_${rep}.json')); t = d.get('train_step', d); print('ws64=$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_${v}_${rep}.err
  done
done
echo finished
