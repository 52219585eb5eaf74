"""One rank of tests/test_dp_two_ranks_gpu.py: two processes, ONE GPU, backend gloo (RCCL refuses two ranks on one device), every tensor on the device
and every kernel on libp3d_hip.so.  (a) dp.broadcast_module + dp.allreduce_gradients on a real generator's gradients == the single-process gradient
of the full batch; (b) rank-sharded G.synthesis replayed as a hipGraph (what bench.py --gpus N times) == the unsharded batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    on_gpu = os.environ.get('P3D_TWO_RANK_DEVICE', 'cuda') == 'cuda'          # ('cpu': a logic dry run of this script where there is no GPU)
    dev = torch.device('cuda', 0) if on_gpu else torch.device('cpu')
    if on_gpu:
        torch.cuda.set_device(dev)
    from pix2pix3d_amd import _lib, configs, dnnlib, dp
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    from model_cases import weights, replay_uniforms
    if on_gpu:
        _lib.lib()
        conv2d_gradfix.enabled = True
        rmod.fused_policy = 'require'
    gkw, _, _ = configs.small_train_kwargs()
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**gkw).train().requires_grad_(True)
    weights.seed_module(G, seed=1)
    G = G.to(dev)
    if rank == 1:                                              # the broadcast has something to repair
        with torch.no_grad():
            for p in G.parameters():
                p.add_(0.5)
    dp.broadcast_module(G, src=0)
    n, nrr = 2, 32
    gen = torch.Generator().manual_seed(3)
    ws = torch.randn(n, G.backbone.num_ws, 512, generator=gen).to(dev)
    c = torch.tensor(np.stack([configs.orbit_camera(k, radius=1.7, focal=1.7074) for k in (9, 77)])).to(dev)
    target = torch.randn(n, 3, 128, 128, generator=gen).to(dev)
    rk = G.rendering_kwargs
    m = nrr * nrr
    u_c = torch.rand([n, m, rk['depth_resolution'], 1], generator=gen)
    u_f = torch.rand([n * m, rk['depth_resolution_importance']], generator=gen)

    def loss_on(idx):
        sel = torch.tensor(idx, device=dev)
        uf = torch.cat([u_f[i * m:(i + 1) * m] for i in idx])
        with replay_uniforms(u_c[idx], uf):
            out = G.synthesis(ws[sel], c[sel], neural_rendering_resolution=nrr, noise_mode='const')
        return (out['image'].float() - target[sel]).square().mean() + out['semantic'].float().square().mean() * 0.1 + out['image_raw'].square().mean()

    mine = dp.shard_indices(n, rank, world)
    for p in G.parameters():
        p.grad = None
    loss_on(mine).backward()
    buf = {}
    flat = dp.allreduce_gradients(G, out=buf)                   # world size from the process group
    assert flat.is_cuda == on_gpu
    shared = {k: p.grad.clone() for k, p in G.named_parameters() if p.grad is not None}
    for p in G.parameters():
        p.grad = None
    loss_on(list(range(n))).backward()                         # every rank also computes the full batch alone
    # fp32 summation order: the batch-2 pass and the two batch-1 passes take different tilings / atomics orders (the renderer's backward adds ~10^5
    # contributions per texel with atomics).  Per parameter: L2 error relative to the larger of its own norm and 1e-3 of the largest gradient norm.
    full_norms = {k: float(p.grad.double().norm()) for k, p in G.named_parameters() if p.grad is not None}
    top = max(full_norms.values())
    errs_p = []
    for k, p in G.named_parameters():
        if p.grad is None:
            assert k not in shared, k
            continue
        errs_p.append((float((p.grad - shared[k]).double().norm()) / max(full_norms[k], 1e-3 * top), k, full_norms[k]))
    errs_p.sort(reverse=True)
    worst = errs_p[0][0]
    if True:
        print(f'rank {rank}: worst parameters (rel L2 err, name, norm):', [(f'{e:.2e}', k, f'{nrm:.2e}') for e, k, nrm in errs_p[:4]], 'largest norm', f'{top:.3e}', flush=True)
    # fp16 blocks (storage rounding of activations and their gradients) and the scalar noise strengths (one signed sum over a whole activation tensor,
    # heavy cancellation): the looser class
    print(f'rank {rank}: worst fp32 (non-scalar) parameter error', max([e for e, k, _ in errs_p if not (k.startswith('superresolution') or k.endswith('noise_strength'))] + [0.0]), flush=True)
    fp16_part = lambda k: k.startswith('superresolution') or k.endswith('noise_strength')
    worst32 = max([e for e, k, _ in errs_p if not fp16_part(k)] + [0.0])
    assert worst < 3e-2 and worst32 < 2e-2, (worst32, errs_p[:4])                    # (measured on an MI355X: 1.4e-2 in the fp16 SR heads, 1.1e-2 on a noise strength; the rest of the fp32 part is printed as `worst fp32`)

    # (b) sharded inference as one hipGraph per rank
    G.eval().requires_grad_(False)
    sel = torch.tensor(mine, device=dev)
    uf = torch.cat([u_f[i * m:(i + 1) * m] for i in mine]).to(dev)
    uc = u_c[mine].to(dev)

    def step(ws_, c_, uc_, uf_):
        with rmod._replay_draws(uc_, uf_), torch.no_grad():
            return G.synthesis(ws_, c_, neural_rendering_resolution=nrr, noise_mode='const')
    ws_s, c_s = ws[sel].clone(), c[sel].clone()
    if on_gpu:
        for _ in range(2):
            step(ws_s, c_s, uc, uf)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step(ws_s, c_s, uc, uf)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = step(ws_s, c_s, uc, uf)
        graph.replay(); graph.replay()
        torch.cuda.synchronize()
    else:
        out = step(ws_s, c_s, uc, uf)
    full = step(ws, c, u_c.to(dev), u_f.to(dev))
    errs = {}
    for k in ('image', 'semantic', 'image_raw', 'image_depth'):
        a, b = out[k].float(), full[k][sel].float()
        errs[k] = float((a - b).abs().max() / b.abs().max())
        assert errs[k] < (2e-3 if k in ('image', 'semantic') else 2e-5), (k, errs[k])     # fp16 SR heads amplify a last-bit difference of their input
    gathered = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor([worst]))
    dist.barrier()
    if rank == 0:
        print('TWO_RANKS_OK', dict(grad_err=[float(g) for g in gathered], sharded=errs, flat_bytes=flat.numel() * 4), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
