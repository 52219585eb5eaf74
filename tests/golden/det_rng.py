"""Deterministic stand-in for the global random generator, shared by the golden recorder (reference modules, CPU) and the tests (this
package's modules, CPU or cuda): while active, ``torch.randn`` / ``torch.rand`` / ``torch.randn_like`` / ``torch.rand_like`` return
values that depend only on (normal | uniform, shape, how many draws of that kind and shape came before) — not on the device, the
generator state or the order in which draws of DIFFERENT shapes interleave.  Two implementations of the same function that draw
the same shapes in the same relative order therefore see the same numbers, wherever they run.

torch + zlib only: loaded by path from tests/golden/make_golden.py (which must not import this repository's package).
"""
import zlib

import torch


def _shape_of(args):
    if len(args) == 1 and isinstance(args[0], (list, tuple, torch.Size)):
        return tuple(int(v) for v in args[0])
    return tuple(int(v) for v in args)


class DetRNG:
    def __init__(self, seed=0):
        self.seed, self.counts, self.log = int(seed), {}, []

    def _draw(self, kind, shape, device, dtype):
        k = self.counts.get((kind, shape), 0)
        self.counts[(kind, shape)] = k + 1
        self.log.append((kind, shape))
        g = torch.Generator().manual_seed(zlib.crc32(f'{self.seed}:{kind}:{shape}:{k}'.encode()))
        raw = self._orig['randn' if kind == 'n' else 'rand'](shape, generator=g, dtype=torch.float32)          # always drawn on the CPU
        return raw.to(device=device, dtype=dtype or torch.float32)

    def __enter__(self):
        self._orig = {n: getattr(torch, n) for n in ('randn', 'rand', 'randn_like', 'rand_like')}

        def make(kind):
            def fn(*size, device=None, dtype=None, generator=None, **kw):
                if generator is not None:                       # an explicitly seeded draw is already deterministic
                    return self._orig['randn' if kind == 'n' else 'rand'](*size, device=device, dtype=dtype, generator=generator, **kw)
                return self._draw(kind, _shape_of(size), device, dtype)

            def like(t, *a, dtype=None, device=None, **kw):
                return self._draw(kind, tuple(t.shape), device if device is not None else t.device, dtype or t.dtype)
            return fn, like
        torch.randn, torch.randn_like = make('n')
        torch.rand, torch.rand_like = make('u')
        return self

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(torch, n, f)
