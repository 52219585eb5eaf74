# End-of-round GPU pass: [full -m gpu suite,] smoke, the bench line (hipGraph, with roofline + cpu_baseline), the eager rocprofv3 profile.
# usage: bash tools/sessions/gpu_final_pass.sh [nosuite]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
if [ "$1" != "nosuite" ]; then ( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/final/pytest.log 2>&1; fi
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -5 ) > gpurun_out/final/smoke.log 2>&1
( timeout 600 python bench.py 2>/dev/null | tail -1 ) > gpurun_out/final/bench_hipgraph.json
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o e -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph 2>/dev/null | tail -1 ) > gpurun_out/final/bench_eager.json
find /tmp/prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/final/kernel_stats.csv \;
[ -f gpurun_out/final/pytest.log ] && tail -3 gpurun_out/final/pytest.log; tail -2 gpurun_out/final/smoke.log; cut -c1-300 gpurun_out/final/bench_hipgraph.json
