set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/final/pytest.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -5 ) > gpurun_out/final/smoke.log 2>&1
( timeout 600 python bench.py 2>/dev/null | tail -1 ) > gpurun_out/final/bench_hipgraph.json
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/final/bench_torchrun.json
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o e -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph 2>/dev/null | tail -1 ) > gpurun_out/final/bench_eager.json
find /tmp/prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/final/kernel_stats.csv \;
tail -3 gpurun_out/final/pytest.log; cat gpurun_out/final/smoke.log | tail -2; cat gpurun_out/final/bench_hipgraph.json | cut -c1-400
