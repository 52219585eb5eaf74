#!/bin/bash
# where the renderer's point-wise backward spends its time: ablation builds of csrc/render_bwd.hip (-DP3D_RBWD_DEBUG=<bits>, see the file), same box.
# Build the variants first: bash tools/sessions/build_rbwd_variants.sh 16 1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for d in "" 16 1 ""; do
  if [ -z "$d" ]; then timeout 120 python tests/gpu_time_render_bwd.py 2>&1 | grep "^render backward"
  else P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_rbwd$d.so timeout 120 python tests/gpu_time_render_bwd.py 2>&1 | grep "^render backward"; fi
done | tee gpurun_out/r3k_rbwd_ablation2.log
