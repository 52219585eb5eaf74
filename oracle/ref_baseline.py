"""cpu_baseline leg of bench.py, kind "reference": the REFERENCE implementation itself (imported from the checkout named by
P3D_REFERENCE, default /root/reference), timed on the host cores in its own process — its packages (``training``, ``torch_utils``,
``dnnlib``) share names with this repository's mirrors, so the two cannot live in one interpreter.

TEST / MEASUREMENT INFRASTRUCTURE ONLY: nothing in pix2pix3d_amd/ imports or launches this.  It runs the reference's
``TriPlaneSemanticEntangleGenerator.synthesis`` (training/triplane_cond.py:1020-1061) with its CPU fallbacks (``_bias_act_ref``,
``_upfirdn2d_ref``, ``F.conv2d``, ``F.grid_sample``; forced fp32 by networks_stylegan2.py:423-425), batch 1, on the workload bench.py
names, and prints one JSON object.

    python oracle/ref_baseline.py <dataset> <nrr> <S_coarse> <S_fine> <reps>
"""
import importlib.util
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('P3D_REFERENCE', '/root/reference')


def main():
    dataset, nrr, sc, sf, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    sys.path.insert(0, REF)
    import numpy as np
    import torch
    import dnnlib                                                             # the reference's
    spec = importlib.util.spec_from_file_location('p3d_configs', os.path.join(os.path.dirname(HERE), 'pix2pix3d_amd', 'configs.py'))
    configs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(configs)                                          # pure data: constructor arguments as train.py assembles them
    torch.set_num_threads(os.cpu_count() or 1)
    kw = configs.generator_kwargs(dataset, depth=(sc, sf))
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**kw).eval().requires_grad_(False)
    rk = kw['rendering_kwargs']
    g = torch.Generator().manual_seed(1234)
    ws = torch.randn(1, G.backbone.num_ws, 512, generator=g)
    c = torch.tensor(np.stack([configs.orbit_camera(3, radius=rk['avg_camera_radius'], pivot=rk['avg_camera_pivot'])]))
    times = []
    with torch.no_grad():
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='const')
            times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:])) if len(times) > 1 else times[0]
    print(json.dumps({'seconds_per_image': t, 'threads': torch.get_num_threads(), 'logical_cores': os.cpu_count(), 'reps': max(len(times) - 1, 1),
                      'image_shape': list(out['image'].shape), 'reference': REF}))


if __name__ == '__main__':
    main()
