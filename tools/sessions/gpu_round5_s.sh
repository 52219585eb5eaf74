#!/bin/bash
# round-5 session s: the super-resolution heads' plans made before the ray-marcher (superresolution.prefetch_early) — parity (bit-identical), then the inference
# (NOT KEPT: the code this session measured — P3D_SR_EARLY_PREFETCH — was removed again: profiles/round5_s_early_prefetch_not_kept.log)
# line with and without it, interleaved, and the step trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_s
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_srheads.py tests/test_model_api.py tests/test_small_ops_gpu.py -q -m gpu -x > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2; do
  for v in 0 1; do
    P3D_SR_EARLY_PREFETCH=$v timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_early_${v}_${rep}.json 2>/dev/null
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_early_${v}_${rep}.json')); print('early=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'])"
  done
done
timeout 300 python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null; head -1 gpurun_out/${T}_step_trace.txt | cut -c1-200
echo finished
