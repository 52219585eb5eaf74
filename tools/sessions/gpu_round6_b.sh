#!/bin/bash
# round-6 session b: two more branches of the captured step — the ray-marcher's prologue (rays, uniform draws, decoder packing) beside the backbone
# (P3D_RENDER_BRANCH) and the super-resolution heads' plans issued from inside the backbone's forward (P3D_SR_PREFETCH_AHEAD) — parity with everything on, then
# the inference line per switch, interleaved (wait elision on throughout; the skip-image branch of session a as its own column).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_b
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_model_full.py tests/test_small_ops_gpu.py tests/test_srheads.py tests/test_model_variants.py tests/test_conv_gpu.py tests/test_render_gpu.py tests/test_checkpoint.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2 3; do
  for v in 000 010 001 011 111; do
    P3D_IMAGE_CHAIN=${v:0:1} P3D_RENDER_BRANCH=${v:1:1} P3D_SR_PREFETCH_AHEAD=${v:2:1} timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('chain,render,sr=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
P3D_IMAGE_CHAIN=0 timeout 300 python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null; head -1 gpurun_out/${T}_step_trace.txt | cut -c1-200
echo finished
