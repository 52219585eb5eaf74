"""Debug aid: error pattern of p3d_conv2d_bwd_weight (fp16) on tiny structured problems."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd.torch_utils.ops.conv2d_gradfix import _Cfg, _weight_grad_impl

torch.manual_seed(0)
for dtype in (torch.float16, torch.float32):
    for (n, c, h, k) in ((1, 128, 8, 1), (1, 128, 16, 1), (2, 128, 8, 3), (1, 64, 8, 1)):
        x = torch.randint(-3, 4, (n, c, h, h), device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
        gy = torch.randint(-3, 4, (n, c, h, h), device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
        cfg = _Cfg(False, (c, c, k, k), 1, k // 2, 0, 1, 1)
        gw = _weight_grad_impl(gy, x, cfg).float().cpu()
        xp = torch.nn.functional.pad(x.float(), (k // 2,) * 4)
        ref = torch.stack([torch.stack([torch.einsum('nohw,nihw->oi', gy.float(), xp[:, :, ky:ky + h, kx:kx + h]) for kx in range(k)], -1) for ky in range(k)], -2).cpu()
        bad = (gw - ref).abs() > 0.5
        print(dtype, (n, c, h, k), 'bad fraction', bad.float().mean().item())
        if bad.any():
            b2 = bad.any(-1).any(-1)
            rows = b2.any(1).nonzero().flatten().tolist(); cols = b2.any(0).nonzero().flatten().tolist()
            print('  bad rows (cs):', rows[:40], '...' if len(rows) > 40 else '', len(rows))
            print('  bad cols (cb):', cols[:40], '...' if len(cols) > 40 else '', len(cols))
            print('  row-wise bad counts:', b2.sum(1)[:16].tolist(), ' col-wise:', b2.sum(0)[:16].tolist())
            # does the wrong value equal a reference value somewhere else?  (a permutation bug)
            r0, c0 = b2.nonzero()[0].tolist()
            v = gw[r0, c0, 0, 0].item()
            hits = (ref[:, :, 0, 0] == v).nonzero()[:6].tolist()
            print(f'  gw[{r0},{c0}]={v} ref there {ref[r0, c0, 0, 0].item()}; ref equals that value at {hits}')
