#!/bin/bash
# round 4, session b: the four tests session a left red (tolerances / metrics reworked), the no-grad training-pass test, steps in flight as one
# graph per lane, the default line again (counter passes now match the tree) and the training line with the fused no-grad passes.
tag=${1:-round4_b}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
echo "== tests"
timeout 900 python -m pytest tests/test_loss_phases.py tests/test_discriminator.py tests/test_dp_two_ranks_gpu.py tests/test_train_nograd_gpu.py tests/test_train_full.py tests/test_train_step.py \
    -m gpu -q -s --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1; tail -n 30 gpurun_out/${tag}_tests.log | cut -c1-1200
grep -E "fp16-top-4|fp32 \{|conv-family|TWO_RANKS|worst parameters" gpurun_out/${tag}_tests.log | cut -c1-900
echo "== streams"
for s in 1 2 3 4; do
  timeout 300 python -X faulthandler bench.py --streams $s --steps 24 --no-train-step --no-cpu-baseline --no-exact-fp32 > gpurun_out/${tag}_bench_line_streams$s.json 2> gpurun_out/${tag}_streams$s.err
  python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_line_streams$s.json')); print('STREAMS $s', d['value'], d['ms_per_step'], d['config']['launch'])" || tail -n 15 gpurun_out/${tag}_streams$s.err
done
echo "== bench"
timeout 900 python bench.py > gpurun_out/${tag}_bench_line_default.json 2> gpurun_out/${tag}_bench.err; head -c 400 gpurun_out/${tag}_bench_line_default.json; echo; tail -n 3 gpurun_out/${tag}_bench.err
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
