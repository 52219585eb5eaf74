"""The N > 1 path on CPU: world-size-2 gloo processes.  (a) the flat gradient all-reduce equals the single-process
gradient of the full batch; (b) rank-sharded inference (what bench.py --gpus N does) reproduces the unsharded result."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from pix2pix3d_amd import dp
    from pix2pix3d_amd.training.networks_stylegan2 import SynthesisNetwork
    torch.manual_seed(0)                                       # same init everywhere, then perturb rank 1 to test the broadcast
    net = SynthesisNetwork(w_dim=32, img_resolution=16, img_channels=3, channel_base=256, channel_max=16, num_fp16_res=0)
    for p in net.parameters():                                 # the 3x3 weights channels-last, as fp16_channels_last lays out the fp16 blocks' (networks_stylegan2.py:403-409)
        if p.ndim == 4 and p.shape[2] == 3:
            p.data = p.data.contiguous(memory_format=torch.channels_last)
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    dp.broadcast_module(net, src=0)
    g = torch.Generator().manual_seed(1)
    ws = torch.randn(4, net.num_ws, 32, generator=g)
    target = torch.randn(4, 3, 16, 16, generator=g)
    mine = dp.shard_indices(4, rank, world)
    # gradient of the mean loss over the FULL batch == average over ranks of the per-shard mean-loss gradients
    net.train()
    loss = (net(ws[mine], noise_mode='const') - target[mine]).square().mean()
    loss.backward()
    flat = dp.allreduce_gradients(net)
    assert flat is not None and flat.ndim == 1
    assert all(p.grad.stride() == p.stride() for p in net.parameters() if p.grad is not None)      # pieces of the flat vector in each parameter's own memory order
    assert any(not p.is_contiguous() for p in net.parameters())
    grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    # sharded inference: every rank renders its own images; nothing is exchanged
    net.eval()
    with torch.no_grad():
        img = net(ws[mine], noise_mode='const')
    torch.save({'grads': grads, 'img': img, 'idx': mine, 'state': {k: v.clone() for k, v in net.state_dict().items()}}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_and_sharded_inference(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'r{r}.pt') for r in (0, 1))
    # broadcast made the replicas identical
    for k in r0['state']:
        assert torch.equal(r0['state'][k], r1['state'][k]), k
    # both ranks hold the same averaged gradient
    for k in r0['grads']:
        assert torch.allclose(r0['grads'][k], r1['grads'][k], atol=0, rtol=0), k
    # single-process reference on the full batch
    from pix2pix3d_amd.training.networks_stylegan2 import SynthesisNetwork
    torch.manual_seed(0)
    net = SynthesisNetwork(w_dim=32, img_resolution=16, img_channels=3, channel_base=256, channel_max=16, num_fp16_res=0)
    g = torch.Generator().manual_seed(1)
    ws = torch.randn(4, net.num_ws, 32, generator=g)
    target = torch.randn(4, 3, 16, 16, generator=g)
    net.train()
    (net(ws, noise_mode='const') - target).square().mean().backward()
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, r0['grads'][n], atol=2e-5, rtol=1e-4), n
    net.eval()
    with torch.no_grad():
        full = net(ws, noise_mode='const')
    for r in (r0, r1):
        assert torch.allclose(full[r['idx']], r['img'], atol=1e-5)


def test_flat_gradients_keep_each_parameters_strides():
    """The pieces of the flat vector come back with the PARAMETER's strides (channels-last weights of the fp16 blocks, size-1 dimensions included, a
    transposed matrix): the optimizer's multi-tensor kernels only take lists whose (param, grad, state) strides agree.  Values are those of the reference's
    flatten / reshape round trip (training_loop.py:531-542) whatever layout the incoming gradient has, with and without the persistent buffer."""
    from pix2pix3d_amd import dp
    g = torch.Generator().manual_seed(5)
    ps = [torch.nn.Parameter(torch.randn(8, 4, 3, 3, generator=g).to(memory_format=torch.channels_last)), torch.nn.Parameter(torch.randn(5, 7, generator=g)),
          torch.nn.Parameter(torch.randn(3, 6, 1, 1, generator=g).to(memory_format=torch.channels_last)), torch.nn.Parameter(torch.randn(4, generator=g)),
          torch.nn.Parameter(torch.randn(1, 6, 1, 1, generator=g)), torch.nn.Parameter(torch.randn(6, 5, generator=g).t())]
    assert ps[2].stride() == (6, 1, 6, 6)
    ref = [torch.randn(p.shape, generator=g) for p in ps]
    for own_layout in (False, True):
        for buf in (None, {}):
            for p, r in zip(ps, ref):
                p.grad = torch.empty_like(p).copy_(r) if own_layout else r.clone().contiguous()
            for _ in range(2):                                   # the second exchange meets gradients that are views of the buffer
                flat = dp.allreduce_gradients(ps, world_size=1, out=buf)
                assert flat.numel() == sum(p.numel() for p in ps)
                for p, r in zip(ps, ref):
                    assert torch.equal(p.grad, r) and p.grad.stride() == p.stride()
    opt = torch.optim.Adam(ps, lr=1e-3, betas=(0.0, 0.99))
    before = [p.detach().clone() for p in ps]
    opt.step()
    assert all(not torch.equal(a, p.detach()) for a, p in zip(before, ps))
