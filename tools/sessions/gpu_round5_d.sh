#!/bin/bash
# round-5 session d: ablation builds of the ray-marcher (tools/build_render_variants.py: P3D_LIB_PATH), both decoder arithmetics; the whole -m gpu suite on this tree;
# the short bench line and the step trace after the ATen launches left the inference step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_d
: > gpurun_out/${T}_render_ablation.log
for bits in 0 1 2 4 8 16 3; do
  for mode in 1 0; do
    echo "variant $bits P3D_MLP_BF16X3=$mode: $(P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_rv$bits.so P3D_MLP_BF16X3=$mode ITERS=10 timeout 120 python tests/gpu_profile_render.py 2>/dev/null | tail -1)" >> gpurun_out/${T}_render_ablation.log
  done
done
cat gpurun_out/${T}_render_ablation.log | cut -c1-200
( timeout 900 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -15 ) > gpurun_out/${T}_gputest.log 2>&1
tail -3 gpurun_out/${T}_gputest.log | cut -c1-600
( timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs --steps 40 2>/dev/null | tail -1 ) > gpurun_out/${T}_bench_short.json
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_short.json')); print(d['value'], d['ms_per_step'], d['stage_ms'], d['mfma_conv']['conv_f16'], d.get('mfma_real_data_ceiling'))"
timeout 300 python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null; head -3 gpurun_out/${T}_step_trace.txt | cut -c1-200
