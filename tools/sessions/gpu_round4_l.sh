#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tests/gpu_time_train_convs.py > gpurun_out/round4_l_convs.log 2>&1; head -n 48 gpurun_out/round4_l_convs.log | cut -c1-160
