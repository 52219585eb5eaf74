#!/bin/bash
# round-6 session q: where the training iteration's time is after the pre-split bf16x6 3x3 kernel: convolution calls by geometry, ATen glue by issuing line, kernel statistics.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_q
timeout 600 python tests/gpu_time_train_convs.py > /dev/null 2>gpurun_out/${T}_convs.err; cp gpurun_out/train_conv_geometries.txt gpurun_out/${T}_train_conv_geometries.txt
timeout 600 python tests/gpu_aten_census.py > /dev/null 2>gpurun_out/${T}_census.err; cp gpurun_out/aten_census.txt gpurun_out/${T}_train_aten_census.txt
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o e -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 ) > /dev/null
find /tmp/prof_t -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_train_kernel_stats.csv \;
head -30 gpurun_out/${T}_train_conv_geometries.txt | cut -c1-150
echo finished
