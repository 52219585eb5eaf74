#!/bin/bash
# round-6 session g: ToRGB layers' weight modulation on the prefetch stream (P3D_PREMODULATE_RGB; eleven 3-6 us launches per step leave the main path) — parity,
# then the inference line both ways, interleaved, and the step trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_g
timeout 1800 python -m pytest tests/test_model_gpu.py tests/test_model_full.py tests/test_model_variants.py tests/test_srheads.py tests/test_small_ops_gpu.py tests/test_checkpoint.py tests/test_train_nograd_gpu.py tests/test_split_acts.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2 3; do
  for v in 0 1; do
    P3D_PREMODULATE_RGB=$v timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('premodulate_rgb=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
timeout 300 python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null; head -1 gpurun_out/${T}_step_trace.txt | cut -c1-200
echo finished
