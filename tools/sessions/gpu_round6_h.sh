#!/bin/bash
# round-6 session h: is the training iteration host-bound now?  Host enqueue time against device time per phase (tests/gpu_cpu_vs_gpu_phases.py), and the
# ATen-op census of one iteration
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_h
timeout 900 python tests/gpu_cpu_vs_gpu_phases.py > gpurun_out/${T}_cpu_vs_gpu_phases.txt 2>&1; tail -15 gpurun_out/${T}_cpu_vs_gpu_phases.txt | cut -c1-200
timeout 900 python tests/gpu_train_census.py > gpurun_out/${T}_train_census.log 2>&1; tail -5 gpurun_out/${T}_train_census.log | cut -c1-200
ls gpurun_out | grep -i census | head
echo finished
