#!/bin/bash
# round-6 session i: fir4_cl_fused_kernel one output row at a time (112 VGPRs / three waves per SIMD in fp32 where the input-row walk held 235 / two; fp16 216 / two where it held 256 / one).
# Parity, then the inference line and the training iteration with the previous build of the library (tools/ab/libp3d_hip_fir4old.so via P3D_LIB_PATH) and this one, interleaved.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_i
timeout 1800 python -m pytest tests/test_conv_gpu.py tests/test_ops_gpu.py tests/test_split_acts.py tests/test_model_gpu.py tests/test_model_full.py tests/test_discriminator.py tests/test_conv_layer_gpu.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2 3; do
  for v in old new; do
    P3D_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libp3d_hip_fir4$v.so timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('fir4=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
for rep in 1 2; do
  for v in old new; do
    P3D_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libp3d_hip_fir4$v.so timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_${v}_${rep}.json 2>gpurun_out/${T}_train_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_${v}_${rep}.json')); t = d.get('train_step', d); print('fir4=$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_${v}_${rep}.err
  done
done
echo finished
