#!/bin/bash
# round-5 session i: class-major block order of the transposed convolutions (heaviest parity class first) — parity, same-box A/B; the two new small tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_i
( timeout 600 python -m pytest tests/test_small_ops_gpu.py tests/test_conv_gpu.py tests/test_split_acts.py tests/test_model_gpu.py -q -m gpu --tb=short 2>&1 | tail -12 ) > gpurun_out/${T}_gputest.log 2>&1
grep -n "passed\|failed" gpurun_out/${T}_gputest.log | tail -2
B="python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs --steps 40"
for rep in 1 2; do
  ( P3D_CONV_CLASS_MAJOR=0 timeout 300 $B 2>/dev/null | tail -1 ) > gpurun_out/${T}_ab_image_major_$rep.json
  ( timeout 300 $B 2>/dev/null | tail -1 ) > gpurun_out/${T}_ab_class_major_$rep.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/round5_i_ab_*.json')):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['stage_ms'], d['mfma_conv']['conv_bf16x3'])
    except Exception as e: print(f, 'ERR', e)
PY
