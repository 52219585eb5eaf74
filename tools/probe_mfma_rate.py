"""What an MFMA-only loop sustains on this MI355X, in conv3x3_h2_f16_kernel's register blocking (csrc/probes/mfma_rate_probe.hip):
independent accumulator chains per wave x waves per SIMD x operand data x grid size, with the clock the chip held in each run.
Writes gpurun_out/<tag>_mfma_rate_probe.json and a table on stdout.   usage: python tools/probe_mfma_rate.py [tag]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd import diagnostics  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'probe'
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(5)
SHAPE = [8, 2, 6, 64, 8]


def operands(kind):
    if kind == 'zeros':
        t = torch.zeros(SHAPE)
    elif kind == 'uniform':                                       # full-range uniform [-1, 1): the worst case for power
        t = torch.rand(SHAPE, generator=g) * 2 - 1
    elif kind == 'sr_layer':                                      # what the SR heads multiply: A = lrelu(N(0,1)) * sqrt(2), B = demodulated weights ~ N(0, 1/sqrt(9 * 256))
        t = torch.randn(SHAPE, generator=g)
        a = torch.nn.functional.leaky_relu(t[:, :, :2], 0.2) * 2 ** 0.5
        b = t[:, :, 2:] / (9 * 256) ** 0.5
        t = torch.cat([a, b], dim=2)
    else:
        raise ValueError(kind)
    return t.half().to(dev).contiguous()


rows = []


def run(kind, ops, **kw):
    r = diagnostics.mfma_rate(ops, **kw)
    r['data'] = kind
    rows.append(r)
    print(f"{kind:9s} chains {r['chains']} waves/SIMD {r['waves_per_simd']} blocks {r['blocks']:5d}: {r['tflops']:7.1f} TF = {r['frac_of_2p5pf']:.3f} of 2.5 PF | "
          f"{r['cycles_per_mfma_per_simd']:.2f} cycles per MFMA and SIMD | clock {r['clock_ghz']:.3f} GHz (pipe peak there {r['tflops_peak_at_that_clock']:.0f} TF) | {r['us_per_launch']:.0f} us", flush=True)


data = {k: operands(k) for k in ('zeros', 'uniform', 'sr_layer')}
# a long warm-up at the heaviest setting: the power controller's state, not a cold boost clock
for _ in range(3):
    diagnostics.mfma_rate(data['uniform'], chains=8, waves_per_simd=2, iters=2000, launches=40)
for kind, ops in data.items():
    for wps in (1, 2, 4):
        for chains in (1, 2, 4, 8):
            if wps == 4 and chains == 8:
                continue
            run(kind, ops, chains=chains, waves_per_simd=wps, iters=2000 if chains > 1 else 1000, launches=30)
# grid size at the kernel's own setting (8 chains, two waves per SIMD = two blocks per CU): one round, partial rounds, many rounds
for kind in ('uniform', 'sr_layer'):
    for iters in (72, 576):                                       # 72 x 16 = the 1152 MFMAs a wave issues for a 256-channel 3x3 layer: the real block's life
        for blocks in (256, 512, 640, 1024, 2048, 2304, 4096):
            run(kind, data[kind], chains=8, waves_per_simd=2, blocks=blocks, iters=iters, launches=60)
os.makedirs('gpurun_out', exist_ok=True)
with open(f'gpurun_out/{tag}_mfma_rate_probe.json', 'w') as f:
    json.dump({'device': torch.cuda.get_device_name(0), 'rows': rows,
               'note': 'tflops over HIP-event wall time of back-to-back launches; cycles from the waves own s_memtime stamps; clock = s_memtime / s_memrealtime x 100 MHz'}, f, indent=1)
