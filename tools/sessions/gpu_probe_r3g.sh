#!/bin/bash
# round 3, session g: per-kernel SQ counters of the inference step and of the training passes; counter passes of the exact-fp32 ray-marcher
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 500 python tests/gpu_pmc_kernels.py infer > gpurun_out/r3g_kpmc_infer.log 2>&1; head -32 gpurun_out/kernel_pmc_infer.txt
timeout 500 python tests/gpu_pmc_kernels.py train > gpurun_out/r3g_kpmc_train.log 2>&1; head -24 gpurun_out/kernel_pmc_train.txt
P3D_MLP_BF16X3=0 timeout 700 python tests/gpu_pmc_render.py > gpurun_out/r3g_pmc_exact.log 2>&1; tail -c 1500 gpurun_out/r3g_pmc_exact.log
