#!/bin/bash
# round-6 session u: layer 1 of the decoder MLPs as bf16x6 in the exact forward (render_forward_kernel<.., L1X6>): parity, then the exact legs of the inference line and the
# training iteration with P3D_MLP_L1X6=0 / 1, interleaved on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_u
timeout 1800 python -m pytest tests/test_render_gpu.py tests/test_render_bwd_gpu.py tests/test_train_full.py tests/test_model_gpu.py tests/test_model_full.py tests/test_loss_phases.py -q -m gpu -x --tb=short -s > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
grep "layer 1 as bf16x6" gpurun_out/${T}_gputest.log | cut -c1-200
for rep in 1 2; do
  for v in 0 1; do
    P3D_MLP_L1X6=$v timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); e = d['exact_fp32']; b = e['backbone_as_bf16x6']; print('l1x6=$v rep $rep', d['value'], 'exact', e.get('value'), e.get('stage_ms'), 'bf16x6 leg', b['value'], b['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
for rep in 1 2; do
  for v in 0 1; do
    P3D_MLP_L1X6=$v timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_${v}_${rep}.json 2>gpurun_out/${T}_train_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_${v}_${rep}.json')); t = d.get('train_step', d); print('l1x6=$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_${v}_${rep}.err
  done
done
echo finished
