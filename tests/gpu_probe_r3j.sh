#!/bin/bash
# round 3, session j: the renderer's backward with one ray per wave-tile + run-combined plane atomics: parity, then same-box A/B against the previous build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_render_bwd_gpu.py tests/test_train_step.py tests/test_train_full.py tests/test_model_variants.py -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/r3j_tests.log 2>&1; tail -6 gpurun_out/r3j_tests.log
for i in 1 2; do
  timeout 120 python tests/gpu_time_render_bwd.py 2>&1 | grep "^render backward"
  P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_rbwd_old.so timeout 120 python tests/gpu_time_render_bwd.py 2>&1 | grep "^render backward"
done | tee gpurun_out/r3j_rbwd_ab.log
