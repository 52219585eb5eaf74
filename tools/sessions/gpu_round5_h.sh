#!/bin/bash
# round-5 session h: per-kernel counter passes of this tree — the inference step (eager) and the six-phase training iteration (train_step.roofline reads the latter)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_h
timeout 700 python tests/gpu_pmc_kernels.py infer > gpurun_out/${T}_kernel_pmc_infer.log 2>&1; echo "infer pmc exit $?"
cp gpurun_out/kernel_pmc_infer.txt gpurun_out/${T}_kernel_pmc_infer.txt 2>/dev/null; head -16 gpurun_out/${T}_kernel_pmc_infer.txt | cut -c1-220
timeout 1100 python tests/gpu_pmc_kernels.py train6 > gpurun_out/${T}_kernel_pmc_train6.log 2>&1; echo "train6 pmc exit $?"
cp gpurun_out/kernel_pmc_train6.txt gpurun_out/${T}_kernel_pmc_train6.txt 2>/dev/null; head -12 gpurun_out/${T}_kernel_pmc_train6.txt | cut -c1-220
