#!/bin/bash
# round-4 session w: where the training iteration's time goes after the weight-gradient / 64-column work — per-geometry rows and the kernel statistics
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tests/gpu_time_train_convs.py > gpurun_out/w_convs.log 2>&1; head -3 gpurun_out/train_conv_geometries.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o t -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/w_train_profiled.json 2> /tmp/prof_train.err)
f=$(find /tmp/prof_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/w_train_kernel_stats.csv
head -12 gpurun_out/w_train_kernel_stats.csv | cut -c1-150
