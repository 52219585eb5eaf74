"""PMC target: a few launches of the big fp16 3x3 conv (run under rocprofv3 --pmc)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd.torch_utils.ops import modconv
N, ci, co, r = 4, 256, 256, 256
x = torch.randn(N, ci, r, r, device='cuda').half().to(memory_format=torch.channels_last)
weight = torch.randn(co, ci, 3, 3, device='cuda'); styles = torch.randn(N, ci, device='cuda') + 1
wmod = modconv.modulate_weights(weight, styles)
bias = torch.randn(co, device='cuda')
for _ in range(5):
    y = modconv.conv2d(x, wmod, bias=bias, act=1, gain=1.414, clamp=256.0)
torch.cuda.synchronize()
