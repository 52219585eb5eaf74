#!/bin/bash
# round-6 session zl: conv2d_nhwc_kernel<.., NST = 4> — the split-K launches of the low-resolution bf16x3 layers with three tiles in flight (counted vmcnt waits) instead of a drain per K step.
# Parity suites, then the line with P3D_CONV_DEEP=0 / 1 interleaved on one box, a step trace either way, and the split-K plan's two knobs under the deep form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zl
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_split_acts.py tests/test_conv_layer_gpu.py tests/test_model_gpu.py tests/test_model_full.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -3 gpurun_out/${T}_gputest.log | cut -c1-400
for rep in 1 2 3; do
  for v in 0 1; do
    P3D_CONV_DEEP=$v timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs --no-exact-fp32 > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('deep=$v rep $rep:', d['value'], 'img/s,', d['ms_per_step'], 'ms,', d.get('stage_ms'))" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
for knobs in "1 4" "1 8" "2 8" "2 3"; do
  set -- $knobs
  P3D_SPLITK_PER_CU=$1 P3D_SPLITK_MIN_STEPS=$2 timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs --no-exact-fp32 > gpurun_out/${T}_bench_k$1_$2.json 2>/dev/null
  python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_k$1_$2.json')); print('deep=1 per_cu=$1 min_steps=$2:', d['value'], 'img/s,', d['ms_per_step'], 'ms,', d.get('stage_ms'))"
done
P3D_CONV_DEEP=0 python tests/gpu_step_trace.py > /dev/null 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace_deep0.txt 2>/dev/null
python tests/gpu_step_trace.py > /dev/null 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace_deep1.txt 2>/dev/null
for v in 0 1; do echo "deep=$v:"; head -1 gpurun_out/${T}_step_trace_deep$v.txt | cut -c1-80; grep "conv2d_nhwc_kernel" gpurun_out/${T}_step_trace_deep$v.txt | awk '{printf "%s ", $2}'; echo; done
echo finished
