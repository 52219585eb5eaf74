"""The RCCL leg of the gradient exchange on the one GPU a test box has.  RCCL refuses two ranks on one device ("Duplicate GPU
detected", tools/sessions/gpu_rccl_smoke.sh records it), so what can be proven here is that `backend='nccl'` (= RCCL on ROCm) initialises
from this package's entry points and that dp.allreduce_gradients drives it on a flat buffer of the real size class; the world-size-2
arithmetic is covered by tests/test_dp_gloo.py and the 8-GPU timing by bench.py --gpus N (--train-step)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flat_allreduce_runs_on_rccl(hip_lib):
    import torch.distributed as dist
    from pix2pix3d_amd import dp
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 300))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        net = torch.nn.Sequential(torch.nn.Linear(1024, 4096), torch.nn.Linear(4096, 4096)).cuda()
        for p in net.parameters():
            p.grad = torch.full_like(p, 2.0)
        net[0].bias.grad[0] = float('nan')
        flat = dp.allreduce_gradients(net)                  # world_size read from the process group (1): SUM, / 1, nan_to_num, scatter
        assert flat.is_cuda and flat.numel() == sum(p.numel() for p in net.parameters())
        assert net[0].bias.grad[0].item() == 0.0 and torch.all(net[1].weight.grad == 2.0)
        big = torch.ones(336_402_180 // 4, device='cuda')   # G's flat gradient vector (training_loop.py:531-542)
        dist.all_reduce(big)
        torch.cuda.synchronize()
        assert big[0].item() == 1.0 and dist.get_backend() == 'nccl'
    finally:
        dist.destroy_process_group()
