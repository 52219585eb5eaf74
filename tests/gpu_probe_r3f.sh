#!/bin/bash
# ray-marcher: hold the second wave of every SIMD back by n x 512 cycles (P3D_RENDER_DESYNC) — same-box sweep
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for n in 0 4 8 12 16 24 32 0; do
  echo -n "desync $n: "; P3D_RENDER_DESYNC=$n REPS=5 ITERS=40 timeout 120 python tests/gpu_profile_render.py 2>&1 | grep "^render"
done | tee gpurun_out/r3f_render_desync.log
