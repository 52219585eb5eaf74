#!/bin/bash
# round 4, session b2: the re-toleranced tests, then the default line (counter passes match the tree) and the training line with / without the fused no-grad passes
tag=${1:-round4_b2}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for f in test_discriminator test_dp_two_ranks_gpu test_loss_phases test_train_nograd_gpu; do
  echo "== $f" | tee -a gpurun_out/${tag}_tests.log
  timeout 600 python -m pytest tests/$f.py -m gpu -q -s --tb=short -rf -p no:cacheprovider >> gpurun_out/${tag}_tests.log 2>&1
  echo "rc=$?" | tee -a gpurun_out/${tag}_tests.log
done
grep -E "conv-family|TWO_RANKS|worst parameters|passed|failed" gpurun_out/${tag}_tests.log | cut -c1-1000
grep -E "^E  |Error" gpurun_out/${tag}_tests.log | head -30 | cut -c1-2500
echo "== bench"
timeout 900 python bench.py > gpurun_out/${tag}_bench_line_default.json 2> gpurun_out/${tag}_bench.err; head -c 300 gpurun_out/${tag}_bench_line_default.json; echo; tail -n 3 gpurun_out/${tag}_bench.err
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/${tag}_bench_line_default.json'))
    t = d.get('train_step') or {}
    print('TRAIN', t.get('ms_per_iteration'), t.get('phase_ms'), t.get('lazy_schedule', {}).get('ms_per_iteration'), t.get('error'))
    print('STAGES', d.get('stage_ms'), 'roofline', d['roofline']['bound'], d['roofline']['frac'], d['roofline']['ms_per_launch'], 'exact', (d.get('exact_fp32') or {}).get('value'))
except Exception as e:
    print('no line', e)
PY
P3D_NO_GRAD_FUSED=0 timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/${tag}_bench_line_train_unfused_nograd.json 2>> gpurun_out/${tag}_bench.err
python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_line_train_unfused_nograd.json')); print('TRAIN P3D_NO_GRAD_FUSED=0', d['ms_per_step'], d['train_step']['phase_ms'])"
