#!/bin/bash
# round-6 closing session of the final tree: gpu_round6_y.sh (suite, smoke, default line, kernel statistics, step trace) + the per-kernel counter passes (inference step, six-phase iteration)
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-round6_z}
bash tools/sessions/gpu_round6_y.sh $T
timeout 700 python tests/gpu_pmc_kernels.py infer > gpurun_out/${T}_kernel_pmc_infer.log 2>&1; echo "infer pmc exit $?"
cp gpurun_out/kernel_pmc_infer.txt gpurun_out/${T}_kernel_pmc_infer.txt 2>/dev/null; head -12 gpurun_out/${T}_kernel_pmc_infer.txt | cut -c1-200
timeout 1100 python tests/gpu_pmc_kernels.py train6 > gpurun_out/${T}_kernel_pmc_train6.log 2>&1; echo "train6 pmc exit $?"
cp gpurun_out/kernel_pmc_train6.txt gpurun_out/${T}_kernel_pmc_train6.txt 2>/dev/null; head -10 gpurun_out/${T}_kernel_pmc_train6.txt | cut -c1-200
echo finished
