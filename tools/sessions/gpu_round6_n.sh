#!/bin/bash
# round-6 session n: conv3x3_halo_x6p_kernel (bf16x6 operands split once per work-group) — parity of the training-side suites, then the training iteration and the
# exact_fp32 leg of the inference line with P3D_X6_PRESPLIT=0 (in-register splits) / 1, interleaved on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_n
timeout 1800 python -m pytest tests/test_conv_gpu.py tests/test_conv_grad_gpu.py tests/test_conv_layer_gpu.py tests/test_discriminator.py tests/test_train_full.py tests/test_loss_phases.py tests/test_train_step.py tests/test_model_gpu.py tests/test_model_full.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2; do
  for v in 0 1; do
    P3D_X6_PRESPLIT=$v timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_${v}_${rep}.json 2>gpurun_out/${T}_train_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_${v}_${rep}.json')); t = d.get('train_step', d); print('x6_presplit=$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_${v}_${rep}.err
  done
done
for rep in 1 2; do
  for v in 0 1; do
    P3D_X6_PRESPLIT=$v timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); e = d['exact_fp32']; print('x6_presplit=$v rep $rep', d['value'], 'exact', e.get('value'), e.get('stage_ms'), 'bf16x6', {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'stage_ms')}) for k, v in e.items() if 'x6' in k})" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
echo finished
