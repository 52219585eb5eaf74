#!/bin/bash
# A/B on one box: the two SR heads on two streams inside the captured step
tag=${1:-round4_g}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 0 1 0 1; do
  P3D_CONCURRENT_SR=$v timeout 300 python bench.py --no-train-step --no-cpu-baseline --no-exact-fp32 > gpurun_out/${tag}_sr$v.json 2>> gpurun_out/${tag}_bench.err
  python -c "import json; d=json.load(open('gpurun_out/${tag}_sr$v.json')); print('CONCURRENT_SR=$v', d['value'], d['ms_per_step'], d['stage_ms'], d['config']['launch'])" || tail -n 5 gpurun_out/${tag}_bench.err
done
P3D_CONCURRENT_SR=1 timeout 600 python -m pytest tests/test_model_gpu.py tests/test_model_full.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -n 3
