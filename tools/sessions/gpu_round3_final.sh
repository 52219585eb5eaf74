#!/bin/bash
# round 3, closing session: the whole -m gpu suite, smoke(), the default benchmark line (CPU baseline, exact-fp32 leg, all-phases train step),
# and the rocprofv3 kernel statistics of the eager step + the training passes.   usage: bash tools/sessions/gpu_round3_final.sh <tag>
tag=${1:-round3_h}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_gputest.log 2>&1; tail -6 gpurun_out/${tag}_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -2 gpurun_out/${tag}_smoke.log
timeout 900 python bench.py > gpurun_out/${tag}_bench_line_default.json 2> gpurun_out/${tag}_bench.err; head -c 1500 gpurun_out/${tag}_bench_line_default.json; echo
timeout 600 python bench.py --train-step --steps 5 --warmup 2 > gpurun_out/${tag}_bench_line_train.json 2>> gpurun_out/${tag}_bench.err
timeout 900 bash tools/sessions/gpu_profiles.sh ${tag}
