#!/bin/bash
# round-6 session zj: splitk_epilogue_kernel with its loads batched (partial tiles eight at a time, scales / biases ahead of them) and up2_fir_f16_kernel's noise strength read ahead of the
# K loop: parity suites, then the line with this build and with the previous one (pix2pix3d_amd/libp3d_hip_base.so: the up2 change in, the split-K one not), interleaved on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zj
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_split_acts.py tests/test_srheads.py tests/test_model_gpu.py tests/test_model_full.py tests/test_conv_layer_gpu.py tests/test_conv_grad_gpu.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -3 gpurun_out/${T}_gputest.log | cut -c1-400
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_base.so; else unset P3D_LIB_PATH; fi
    timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs --no-exact-fp32 > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('$v rep $rep:', d['value'], 'img/s,', d['ms_per_step'], 'ms,', d.get('stage_ms'))" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
unset P3D_LIB_PATH
python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; echo "trace exit $?"; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null
grep -n "splitk_epilogue\|up2_fir\|one step" gpurun_out/${T}_step_trace.txt | cut -c1-160
echo finished
