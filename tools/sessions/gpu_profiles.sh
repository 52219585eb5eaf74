#!/bin/bash
# rocprofv3 summaries for profiles/: the benchmark step (eager launches, every kernel visible by name) and the two training passes.
# usage: bash tools/sessions/gpu_profiles.sh <tag>     -> gpurun_out/<tag>_*
tag=${1:-round2}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-train-step --no-exact-fp32 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_line_eager.json 2> /tmp/prof_bench.err)
cp $(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o t -- python $GRAFT_REPO_ROOT/tests/gpu_train_census.py 4 128 --no-census > $GRAFT_REPO_ROOT/gpurun_out/${tag}_train_passes.log 2> /tmp/prof_train.err)
cp $(find /tmp/prof_train -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_train_kernel_stats.csv
python tests/gpu_train_census.py 4 128 --json gpurun_out/${tag}_train_census.json > gpurun_out/${tag}_train_census.log 2>&1
python bench.py --no-train-step --no-cpu-baseline > gpurun_out/${tag}_bench_line_hipgraph.json 2> /dev/null
head -c 600 gpurun_out/${tag}_bench_line_hipgraph.json; echo; grep pass gpurun_out/${tag}_train_census.log | cut -c1-200; head -12 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-160
