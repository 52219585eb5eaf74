#!/bin/bash
# round-5 session e: the shared-weight layers with x * styles inside the convolution kernel — parity (bit-identical to the pass it replaces), same-box A/B, step trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_e
( timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py tests/test_split_acts.py -q -m gpu --tb=short 2>&1 | tail -12 ) > gpurun_out/${T}_gputest.log 2>&1
grep -n "passed\|failed" gpurun_out/${T}_gputest.log | tail -2
B="python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs --steps 40"
for rep in 1 2; do
  ( P3D_FUSE_INPUT_SCALE=0 timeout 300 $B 2>/dev/null | tail -1 ) > gpurun_out/${T}_ab_own_pass_$rep.json
  ( timeout 300 $B 2>/dev/null | tail -1 ) > gpurun_out/${T}_ab_in_kernel_$rep.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/round5_e_ab_*.json')):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['stage_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout 300 python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null; head -2 gpurun_out/${T}_step_trace.txt | cut -c1-200
