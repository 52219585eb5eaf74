#!/bin/bash
# Builds the ablation variants of csrc/render_bwd.hip that tools/sessions/gpu_probe_rbwd.sh times (run here, before gpurun: the .so files travel with the tree).
#   usage: bash tools/sessions/build_rbwd_variants.sh 16 1        -> pix2pix3d_amd/libp3d_hip_rbwd16.so, ...rbwd1.so   (delete them afterwards)
cd "$(dirname "$0")/../pix2pix3d_amd/csrc" || exit 1
python -m pix2pix3d_amd.build > /dev/null 2>&1 || (cd ../.. && python -c "import __graft_entry__ as g; g.build()")
for d in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -DP3D_RBWD_DEBUG=$d -c render_bwd.hip -o /tmp/rb_$d.o || exit 1
  hipcc -shared -fPIC --offload-arch=gfx950 -o ../libp3d_hip_rbwd$d.so $(ls _obj/*.o | grep -v "render_bwd.o") /tmp/rb_$d.o || exit 1
  echo built ../libp3d_hip_rbwd$d.so
done
