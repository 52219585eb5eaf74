"""One rank of tests/test_bench_two_ranks_gpu.py: bench.py's OWN training iteration (train_setup / train_iteration: six phases through Pix2Pix3DLoss, flat
gradient exchange + Adam per phase) with world size 2 on one GPU, backend gloo.  Ranks start from the same weights and see different data; after the
iteration every parameter of G, D and D_semantic must be bit-identical across the ranks (same averaged gradients into the same Adam state) and differ from
the initial weights."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    import bench
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    _lib.lib()
    conv2d_gradfix.enabled = True
    rmod.fused_policy = 'require'

    class Args:
        dataset, batch, train_nrr = 'seg2cat', 2, 64            # config 3's networks at full size; fewer rays and images keep the test short
    st = bench.train_setup(Args, dev, world)
    before = {k: float(sum(p.double().sum() for p in net.parameters())) for k, net in st['nets'].items()}
    assert float((st['batch']['image'] - st['batch']['image']).abs().max()) == 0
    sizes = bench.train_iteration(st, {})
    torch.cuda.synchronize()
    assert set(sizes) == set(bench.PHASE_ORDER) and sizes['Gmain'] > 300e6 and sizes['D_semanticreg'] > 100e6, sizes
    sums = torch.tensor([float(sum(p.double().sum() for p in net.parameters())) for net in st['nets'].values()], dtype=torch.float64)
    absum = torch.tensor([float(sum(p.double().abs().sum() for p in net.parameters())) for net in st['nets'].values()], dtype=torch.float64)
    gathered = [torch.zeros(6, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.cat([sums, absum]))
    for g in gathered[1:]:
        assert torch.equal(g, gathered[0]), (gathered[0].tolist(), g.tolist())          # replicas stayed in lock-step, bit for bit
    for (k, b), a in zip(before.items(), sums.tolist()):
        assert a != b, f'{k} did not move'
    data = torch.tensor([float(st['batch']['image'].double().sum())], dtype=torch.float64)
    seen = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(seen, data)
    assert float(seen[0]) != float(seen[1]), 'the ranks were supposed to see different data'
    # the N > 1 line's self-evidence object (bench.collective_info; backend gloo here, so the bandwidth is host staging and says so)
    info = bench.collective_info(dist, world, rank, dev, sizes=(8 << 20,), reps=2)
    assert info['ranks'] == world and len(info['devices']) == world and {d['rank'] for d in info['devices']} == set(range(world)), info
    assert info['distinct_devices'] == 1 and 'gloo' in info['backend'] and info['env']['NCCL_P2P_DISABLE'] in (None, '', '0'), info       # two ranks on ONE device: visible in the record
    assert info['allreduce_alone'][str(8 << 20)]['bus_GBps'] > 0 and info['xgmi']['links_per_gpu'] == 7
    dist.barrier()
    if rank == 0:
        print('BENCH_TWO_RANKS_OK', {k: int(v) for k, v in sizes.items()}, flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
