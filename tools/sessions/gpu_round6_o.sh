#!/bin/bash
# round-6 session o: how far each fp32 arithmetic sits from the recorded reference gradients in tests/test_loss_phases.py (device leg): the f32-input MFMA,
# bf16x6 with in-register splits, bf16x6 with operands split once per work-group (another summation order over K).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_o
for v in "P3D_F32_BF16X6=0" "P3D_X6_PRESPLIT=0" "P3D_X6_PRESPLIT=1"; do
  rm -f gpurun_out/parity_errors.json
  env $v timeout 900 python -m pytest tests/test_loss_phases.py -q -m gpu --tb=line 2>&1 | tail -3 | cut -c1-400
  python - <<PY
import json
d = json.load(open('gpurun_out/parity_errors.json'))
json.dump(d, open('gpurun_out/${T}_loss_phases_${v//=/_}.json', 'w'), indent=1, sort_keys=True)
rows = sorted(((v['worst_rel'][0][1], k, v['worst_rel'][0][0], v['largest_norm_error_over_scale']) for k, v in d.items()), reverse=True)
print('$v')
for r in rows[:6]: print('   %.3e  %s  %s  (largest norm error / scale %.2e)' % r)
PY
done
echo finished
