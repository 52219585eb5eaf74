#!/bin/bash
# round-6 session zp: fc_multi / demod_coefs_multi loops unrolled (independent loads in flight), upfirdn2d_cl_kernel reads the accumulated-into value with its taps, the generic kernel batches its two biases.
# modulate_weights_kernel (a rolled loop: 18 in a row for a 512-channel 3x3 layer), the 8 x 16-patch kernels' epilogue (32 noise values + 2 biases per lane, each under its branch).
# The whole GPU suite, then this build against the previous one (pix2pix3d_amd/libp3d_hip_base.so) interleaved on one box: the inference line WITH its exact legs, the training iteration
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zp
( timeout 1500 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -15 ) > gpurun_out/${T}_gputest.log 2>&1
grep -n "passed\|failed" gpurun_out/${T}_gputest.log | tail -2
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_base.so; else unset P3D_LIB_PATH; fi
    timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); e = d['exact_fp32']; b = e['backbone_as_bf16x6']; print('$v rep $rep:', d['value'], 'img/s,', d.get('stage_ms'), '| exact', e['value'], e['stage_ms'], '| bf16x6', b['value'], b['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
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
echo finished
