#!/bin/bash
# round-6 session zi: epilogues of the 16 x 16-patch kernels without dependent memory round trips (conv3x3_h2_f16_kernel<TR>: ToRGB weights + biases parked in LDS by the kernel's
# first instructions, the skip image's old values requested in one batch; the LDS-image epilogue and conv3x3_r2_bf16x3_kernel: biases / ToRGB weights / read-modify-writes batched).
# Parity, then the line with this build and with the previous one (pix2pix3d_amd/libp3d_hip_base.so), interleaved on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zi
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_split_acts.py tests/test_srheads.py tests/test_model_gpu.py tests/test_model_full.py tests/test_conv_grad_gpu.py tests/test_discriminator.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-400
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_base.so; else unset P3D_LIB_PATH; fi
    timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs --no-exact-fp32 > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('$v rep $rep:', d['value'], 'img/s,', d['ms_per_step'], 'ms,', d.get('stage_ms'), 'conv_f16', d.get('mfma_conv', {}).get('conv_f16'))" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
unset P3D_LIB_PATH
python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; echo "trace exit $?"; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null
grep -n "r2_bf16x3\|h2_f16\|one step" gpurun_out/${T}_step_trace.txt | cut -c1-160
echo finished
