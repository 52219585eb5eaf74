#!/bin/bash
# round 4, session e: the loss-phase test with the ATen-fallback log, the model-level inference tests (renderer preparation on its own side stream), the default line
tag=${1:-round4_e}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_loss_phases.py -m gpu -q -s --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_loss_phases.log 2>&1; tail -n 4 gpurun_out/${tag}_loss_phases.log | cut -c1-300
grep -E "^E  " gpurun_out/${tag}_loss_phases.log | head -5 | cut -c1-4000
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_model_full.py tests/test_model_api.py tests/test_model_variants.py tests/test_split_acts.py tests/test_checkpoint.py tests/test_dp_two_ranks_gpu.py -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_model_tests.log 2>&1; tail -n 4 gpurun_out/${tag}_model_tests.log | cut -c1-300
timeout 300 python bench.py --no-train-step --no-cpu-baseline --no-exact-fp32 > gpurun_out/${tag}_bench_line_hipgraph.json 2> gpurun_out/${tag}_bench.err
python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_line_hipgraph.json')); print('HOIST lane 1', d['value'], d['ms_per_step'], d['stage_ms'])"
P3D_HOIST_RENDER_PREP=0 timeout 300 python bench.py --no-train-step --no-cpu-baseline --no-exact-fp32 > gpurun_out/${tag}_bench_line_nohoist.json 2>> gpurun_out/${tag}_bench.err
python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_line_nohoist.json')); print('NO HOIST    ', d['value'], d['ms_per_step'], d['stage_ms'])"
timeout 300 python bench.py --no-train-step --no-cpu-baseline --no-exact-fp32 > gpurun_out/${tag}_bench_line_hipgraph2.json 2>> gpurun_out/${tag}_bench.err
python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_line_hipgraph2.json')); print('HOIST lane 1', d['value'], d['ms_per_step'], d['stage_ms'])"
