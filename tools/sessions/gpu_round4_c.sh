#!/bin/bash
# round 4, session c: the tests touched since session b2, the whole -m gpu suite, the default line (frozen-generator passes under no_grad, generator-scoped
# bf16x3 training leg) and the per-kernel counter passes of the six-phase training iteration (tests/gpu_pmc_kernels.py train6).
tag=${1:-round4_c}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for f in test_dp_two_ranks_gpu test_loss_phases test_train_nograd_gpu test_train_full; do
  echo "== $f" | tee -a gpurun_out/${tag}_tests.log
  timeout 600 python -m pytest tests/$f.py -m gpu -q -s --tb=short -rf -p no:cacheprovider >> gpurun_out/${tag}_tests.log 2>&1
  echo "rc=$?" | tee -a gpurun_out/${tag}_tests.log
done
grep -E "conv-family|TWO_RANKS|worst |passed|failed|---- rank" gpurun_out/${tag}_tests.log | cut -c1-1200
grep -E "^E  |Error" gpurun_out/${tag}_tests.log | head -30 | cut -c1-2500
echo "== suite"
timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_gputest.log 2>&1; tail -n 8 gpurun_out/${tag}_gputest.log | cut -c1-400
echo "== bench"
timeout 900 python bench.py > gpurun_out/${tag}_bench_line_default.json 2> gpurun_out/${tag}_bench.err; head -c 300 gpurun_out/${tag}_bench_line_default.json; echo; tail -n 3 gpurun_out/${tag}_bench.err
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/${tag}_bench_line_default.json'))
    t = d.get('train_step') or {}
    print('TRAIN', t.get('ms_per_iteration'), t.get('phase_ms'), t.get('lazy_schedule', {}).get('ms_per_iteration'), t.get('error'))
    print('TRAIN G bf16x3', json.dumps(t.get('generator_bf16x3'))[-500:])
    print('STAGES', d.get('stage_ms'), 'roofline', d['roofline']['bound'], d['roofline']['frac'], d['roofline']['ms_per_launch'], 'exact', (d.get('exact_fp32') or {}).get('value'))
except Exception as e:
    print('no line', e)
PY
echo "== training kernels: counters"
PMC_PASS_TIMEOUT=120 timeout 800 python tests/gpu_pmc_kernels.py train6 > gpurun_out/${tag}_kernel_pmc_train6.log 2>&1; head -n 30 gpurun_out/${tag}_kernel_pmc_train6.log | cut -c1-200
