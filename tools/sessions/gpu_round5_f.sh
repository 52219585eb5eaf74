#!/bin/bash
# round-5 session f: batched demodulation coefficients — parity, step trace (kernel count), short bench; the training iteration with the frozen passes' arithmetic both ways
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_f
( timeout 600 python -m pytest tests/test_small_ops_gpu.py tests/test_conv_gpu.py tests/test_model_gpu.py tests/test_model_full.py -q -m gpu --tb=short 2>&1 | tail -12 ) > gpurun_out/${T}_gputest.log 2>&1
grep -n "passed\|failed" gpurun_out/${T}_gputest.log | tail -2
( timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs --steps 40 2>/dev/null | tail -1 ) > gpurun_out/${T}_bench_short.json
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_short.json')); print(d['value'], d['ms_per_step'], d['stage_ms'])"
timeout 300 python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null; head -1 gpurun_out/${T}_step_trace.txt | cut -c1-200
for mode in 1 0; do
  ( P3D_FROZEN_EXACT_FP32=$mode timeout 400 python bench.py --train-step --steps 3 --warmup 2 2>/dev/null | tail -1 ) > gpurun_out/${T}_train_frozen_exact_$mode.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_train_frozen_exact_$mode.json'))['train_step']; print('P3D_FROZEN_EXACT_FP32=$mode', d['ms_per_iteration'], d['phase_ms'], d['lazy_schedule']['ms_per_iteration'])"
done
