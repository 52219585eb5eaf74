#!/bin/bash
# round-5 session w: the compiler's MFMA-interleaving strategies in the ray-marcher's decoder regions (tools/build_render_variants.py 0 2048 4096 = none / iglp_opt(0) / iglp_opt(1)), three rounds interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_w
: > gpurun_out/${T}_render_variants.log
for rep in 1 2 3; do
  for bits in 0 2048 4096; do
    echo "round $rep variant $bits: $(P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_rv$bits.so ITERS=20 timeout 120 python tests/gpu_profile_render.py 2>/dev/null | tail -1 | cut -c1-60)" >> gpurun_out/${T}_render_variants.log
  done
done
cat gpurun_out/${T}_render_variants.log
echo finished
