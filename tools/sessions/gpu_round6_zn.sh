#!/bin/bash
# round-6 session zn: fir4_cl_fused_kernel with every small operand (filter taps, biases, noise strength, the eight noise values) requested first and unconditionally — as written it waited
# for each of its eight noise loads separately and read its biases behind the rendezvous: ~10 dependent memory round trips per block.  Parity suites, then this build against the previous
# one (pix2pix3d_amd/libp3d_hip_base.so) interleaved on one box: the inference line, the training iteration; a step trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zn
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_conv_gpu.py tests/test_split_acts.py tests/test_conv_layer_gpu.py tests/test_model_gpu.py tests/test_model_full.py tests/test_srheads.py tests/test_discriminator.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -3 gpurun_out/${T}_gputest.log | cut -c1-400
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_base.so; else unset P3D_LIB_PATH; fi
    timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs --no-exact-fp32 > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('$v rep $rep:', d['value'], 'img/s,', d['ms_per_step'], 'ms,', d.get('stage_ms'))" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
for rep in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_base.so; else unset P3D_LIB_PATH; fi
    timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_${v}_${rep}.json 2>gpurun_out/${T}_train_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_${v}_${rep}.json')); t = d.get('train_step', d); print('$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_${v}_${rep}.err
  done
done
unset P3D_LIB_PATH
python tests/gpu_step_trace.py > /dev/null 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null
head -1 gpurun_out/${T}_step_trace.txt | cut -c1-90; grep "fir4_cl_fused" gpurun_out/${T}_step_trace.txt | awk '{printf "%s ", $2}'; echo
echo finished
