#!/bin/bash
# same-box A/B of the ray-marcher: this tree's kernel vs round 2's (libp3d_hip_r2render.so = this tree with round 2's render.hip / render_device.h)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for i in 1 2 3; do
  REPS=5 ITERS=40 timeout 120 python tests/gpu_profile_render.py 2>&1 | grep "^render"
  REPS=5 ITERS=40 P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_r2render.so timeout 120 python tests/gpu_profile_render.py 2>&1 | grep "^render"
done | tee gpurun_out/r3e_render_ab.log
P3D_MLP_BF16X3=0 REPS=5 ITERS=40 timeout 120 python tests/gpu_profile_render.py 2>&1 | grep "^render" | tee -a gpurun_out/r3e_render_ab.log
P3D_MLP_BF16X3=0 REPS=5 ITERS=40 P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_r2render.so timeout 120 python tests/gpu_profile_render.py 2>&1 | grep "^render" | tee -a gpurun_out/r3e_render_ab.log
