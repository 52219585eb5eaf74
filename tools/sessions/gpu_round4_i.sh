#!/bin/bash
# round 4, session i: plain 4x4 FIR on the LDS-tiled kernel — the whole suite, then the training line with and without it
tag=${1:-round4_i}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_gputest.log 2>&1; tail -n 6 gpurun_out/${tag}_gputest.log | cut -c1-500
grep -E "^E  " gpurun_out/${tag}_gputest.log | head -12 | cut -c1-1500
for v in on off; do
  if [ $v = off ]; then export P3D_UPFIRDN_NO_FIR4=1; fi
  timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/${tag}_train_fir4_$v.json 2>> gpurun_out/${tag}_bench.err
  python -c "import json; d=json.load(open('gpurun_out/${tag}_train_fir4_$v.json')); print('FIR4 tiled $v', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/${tag}_bench.err
done
unset P3D_UPFIRDN_NO_FIR4
timeout 300 python bench.py --no-train-step --no-cpu-baseline --no-exact-fp32 > gpurun_out/${tag}_bench_line_hipgraph.json 2>> gpurun_out/${tag}_bench.err
python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_line_hipgraph.json')); print('INFER', d['value'], d['ms_per_step'], d['stage_ms'])"
