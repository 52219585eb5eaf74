#!/bin/bash
# round-6 session e: quad-transposed 16-byte stores in the generic kernel's fp32 epilogue (conv2d_nhwc_kernel<float, ...>; P3D_CONV_STORE4=1 = the 4-byte stores it
# replaces) — parity of the convolution / gradient / layer / model suites, then the inference line and the training iteration both ways, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_e
timeout 1800 python -m pytest tests/test_conv_gpu.py tests/test_conv_grad_gpu.py tests/test_conv_layer_gpu.py tests/test_split_acts.py tests/test_model_gpu.py tests/test_model_full.py tests/test_discriminator.py tests/test_train_full.py tests/test_loss_phases.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2 3; do
  for v in 1 0; do
    P3D_CONV_STORE4=$v timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('conv_store4=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
for rep in 1 2; do
  for v in 1 0; do
    P3D_CONV_STORE4=$v timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_${v}_${rep}.json 2>gpurun_out/${T}_train_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_${v}_${rep}.json')); t = d.get('train_step', d); print('conv_store4=$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_${v}_${rep}.err
  done
done
echo finished
