#!/bin/bash
# round 3, session d: renderer parity after the register-only importance sampler + the tests fixed since session c; quick benchmark line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_small_ops_gpu.py tests/test_semrenderer_gpu.py tests/test_model_full.py tests/test_model_gpu.py tests/test_render_bwd_gpu.py tests/test_train_full.py tests/test_train_step.py tests/test_discriminator.py -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/r3d_tests.log 2>&1; tail -8 gpurun_out/r3d_tests.log
timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'img/s', d['ms_per_step'], 'ms', d['stage_ms'], 'render', d['roofline']['ms_per_launch'], d['config']['launch'])" | tee gpurun_out/r3d_bench.log
