#!/bin/bash
# round-6 closing session: the whole -m gpu suite, smoke, the default bench line, rocprofv3 kernel statistics of the eager step, the step trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=${1:-round6_y}
( timeout 900 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -15 ) > gpurun_out/${T}_gputest.log 2>&1
grep -n "passed\|failed" gpurun_out/${T}_gputest.log | tail -2
cp gpurun_out/parity_errors.json gpurun_out/${T}_parity_errors.json 2>/dev/null
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 ) > gpurun_out/${T}_smoke.log 2>&1; tail -2 gpurun_out/${T}_smoke.log | cut -c1-300
( timeout 900 python bench.py 2>gpurun_out/${T}_bench.err | tail -1 ) > gpurun_out/${T}_bench_line_default.json
cut -c1-300 gpurun_out/${T}_bench_line_default.json
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o e -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-train-step --no-exact-fp32 --no-configs 2>/dev/null | tail -1 ) > gpurun_out/${T}_bench_eager.json
find /tmp/prof_g -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_bench_kernel_stats.csv \;
head -14 gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-160
timeout 300 python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null; head -1 gpurun_out/${T}_step_trace.txt | cut -c1-200
