#!/bin/bash
# round-6 session v: the ray-marcher's counter passes re-taken for this tree's render_device.h (layer 1 as bf16x6 added; the bf16x3 and the f32-input kernels' code is
# what it was, the hash that pins profiles/render_pmc*.json is not): bf16x3 decoder, the f32-input MFMA decoder (P3D_MLP_L1X6=0), the edge2car launch, and the new
# kernel (layer 1 as bf16x6) into a file of its own.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_v
timeout 900 python tests/gpu_pmc_render.py > gpurun_out/${T}_render_pmc.log 2>&1; echo "pmc bf16x3 exit $?"
P3D_MLP_BF16X3=0 P3D_MLP_L1X6=0 timeout 900 python tests/gpu_pmc_render.py > gpurun_out/${T}_render_pmc_exact.log 2>&1; echo "pmc exact exit $?"
cp gpurun_out/render_pmc_exact_fp32.json gpurun_out/${T}_keep_exact.json; cp gpurun_out/render_sq_pmc_exact_fp32.txt gpurun_out/${T}_keep_exact.txt
P3D_MLP_BF16X3=0 P3D_MLP_L1X6=1 P3D_PMC_GROUPS=0,1,2,3,4 timeout 900 python tests/gpu_pmc_render.py > gpurun_out/${T}_render_pmc_l1x6.log 2>&1; echo "pmc l1x6 exit $?"
cp gpurun_out/render_pmc_exact_fp32.json gpurun_out/render_pmc_l1x6.json; cp gpurun_out/render_sq_pmc_exact_fp32.txt gpurun_out/render_sq_pmc_l1x6.txt
cp gpurun_out/${T}_keep_exact.json gpurun_out/render_pmc_exact_fp32.json; cp gpurun_out/${T}_keep_exact.txt gpurun_out/render_sq_pmc_exact_fp32.txt
P3D_PMC_DATASET=edge2car P3D_PMC_GROUPS=0,1,2,3,4,5 timeout 400 python tests/gpu_pmc_render.py > gpurun_out/${T}_render_pmc_edge2car.log 2>&1; echo "pmc edge2car exit $?"
tail -3 gpurun_out/${T}_render_pmc.log; head -30 gpurun_out/render_sq_pmc_l1x6.txt | cut -c1-160
echo finished
