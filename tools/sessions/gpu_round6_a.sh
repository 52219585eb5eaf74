#!/bin/bash
# round-6 session a: the captured step's graph edges — (1) take_plan waits only for positions of the prefetch stream its stream does not already stand behind,
# (2) the low-resolution blocks' ToRGB / skip-image launches on a branch stream (modconv.ImageChain).  Parity of the model-level suites with both on, then the
# inference line with each switch off / on, interleaved, and the step trace of the final form.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_a
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_model_full.py tests/test_small_ops_gpu.py tests/test_srheads.py tests/test_model_variants.py tests/test_conv_gpu.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2; do
  for v in 00 10 11; do
    P3D_PLAN_WAIT_ELISION=${v:0:1} P3D_IMAGE_CHAIN=${v:1:1} timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('elision,chain=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
timeout 300 python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null; head -1 gpurun_out/${T}_step_trace.txt | cut -c1-200
echo finished
