#!/bin/bash
# round 4, closing session: the whole -m gpu suite, smoke(), the default benchmark line, the eager step under rocprofv3, the training iteration under rocprofv3,
# and the per-kernel counter passes of the training iteration.   usage: bash tools/sessions/gpu_round4_ab.sh <tag>
tag=${1:-round4_ab}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_gputest.log 2>&1; tail -n 6 gpurun_out/${tag}_gputest.log | cut -c1-500
grep -E "^E  " gpurun_out/${tag}_gputest.log | head -12 | cut -c1-1500
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -n 1 gpurun_out/${tag}_smoke.log | cut -c1-400
timeout 900 python bench.py > gpurun_out/${tag}_bench_line_default.json 2> gpurun_out/${tag}_bench.err; head -c 300 gpurun_out/${tag}_bench_line_default.json; echo; tail -n 2 gpurun_out/${tag}_bench.err
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/${tag}_bench_line_default.json'))
    t = d.get('train_step') or {}
    print('TRAIN', t.get('ms_per_iteration'), t.get('phase_ms'), t.get('lazy_schedule', {}).get('ms_per_iteration'), t.get('error'))
    print('TRAIN G bf16x3', (t.get('generator_bf16x3') or {}).get('ms_per_iteration'), (t.get('generator_bf16x3') or {}).get('lazy_schedule_ms'), 'roofline', json.dumps((t.get('roofline') or {}).get('dominant'))[:300])
    print('STAGES', d.get('stage_ms'), 'roofline', d['roofline']['bound'], d['roofline']['frac'], d['roofline']['ms_per_launch'], 'exact', (d.get('exact_fp32') or {}).get('value'), 'cpu', d['cpu_baseline']['value'])
except Exception as e:
    print('no line', e)
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-train-step --no-exact-fp32 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_line_eager.json 2> /tmp/prof_bench.err)
cp $(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_bench_kernel_stats.csv; head -n 4 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-170
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o t -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_line_train_profiled.json 2> /tmp/prof_train.err)
cp $(find /tmp/prof_train -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_train_kernel_stats.csv; grep -E "bias_act_kernel<__half, 3" gpurun_out/${tag}_train_kernel_stats.csv | cut -c1-170
PMC_PASS_TIMEOUT=120 timeout 800 python tests/gpu_pmc_kernels.py train6 > gpurun_out/${tag}_kernel_pmc_train6.log 2>&1; head -n 10 gpurun_out/${tag}_kernel_pmc_train6.log | cut -c1-200
cp gpurun_out/kernel_pmc_train6.json gpurun_out/${tag}_kernel_pmc_train6.json; cp gpurun_out/kernel_pmc_train6.txt gpurun_out/${tag}_kernel_pmc_train6.txt
rm -rf gpurun_out/kpmc gpurun_out/rpmc gpurun_out/rpmc_exact /tmp/prof_bench /tmp/prof_train; du -sh gpurun_out | tail -1
