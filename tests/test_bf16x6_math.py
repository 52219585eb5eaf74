"""The arithmetic claim behind P3D_F32_BF16X6 (csrc/bf16_split.h), checked on the host with bit-exact bf16 rounding:
hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) reproduce an fp32 value EXACTLY (3 x 8 significand bits), and the six products the kernels
keep — (h,h) (h,m) (m,h) (h,l) (l,h) (m,m) — leave a relative error of the size of one fp32 rounding, where bf16x3's three products leave 2^-16."""
import numpy as np


def bf16(x):
    """Round-to-nearest-even to bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 does to finite values)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    hi = bf16(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16(r1)
    lo = bf16((r1 - mid).astype(np.float32))
    return hi, mid, lo


def _values(n, seed):
    rng = np.random.default_rng(seed)
    mant = rng.standard_normal(n).astype(np.float32)
    expo = rng.integers(-40, 40, n)
    return (mant * np.exp2(expo).astype(np.float32)).astype(np.float32)


def test_three_pieces_are_the_value():
    x = np.concatenate([_values(200000, 0), np.float32([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.1754944e-38, 65504.0, 1 + 2 ** -23, 1 - 2 ** -24])])
    hi, mid, lo = split3(x)
    for piece in (hi, mid, lo):
        assert np.array_equal(bf16(piece), piece)                                        # every piece IS a bf16 value
    assert np.array_equal((hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))     # exactly, not just after rounding
    # both subtractions of the split are exact in fp32 (Sterbenz-like: the subtrahend is the rounded minuend)
    assert np.array_equal((x - hi).astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))


def test_six_products_are_an_fp32_product():
    a, b = _values(200000, 1), _values(200000, 2)
    keep = np.isfinite(a.astype(np.float64) * b.astype(np.float64)) & (np.abs(a.astype(np.float64) * b.astype(np.float64)) > 1e-30) & (np.abs(a.astype(np.float64) * b.astype(np.float64)) < 1e30)
    a, b = a[keep], b[keep]
    (ah, am, al), (bh, bm, bl) = split3(a), split3(b)
    d = lambda v: v.astype(np.float64)
    exact = d(a) * d(b)
    six = d(ah) * d(bh) + d(ah) * d(bm) + d(am) * d(bh) + d(ah) * d(bl) + d(al) * d(bh) + d(am) * d(bm)      # each bf16 x bf16 product is exact in the MFMA's fp32 accumulate
    three = d(ah) * d(bh) + d(ah) * d(bm) + d(am) * d(bh)                                                    # bf16x3: hi / lo(= mid here) pieces, three products
    e6 = np.abs(six - exact) / np.abs(exact)
    e3 = np.abs(three - exact) / np.abs(exact)
    e32 = np.abs(d((a * b).astype(np.float32)) - exact) / np.abs(exact)                                       # one fp32 rounding of the product
    print('bf16x6 max rel', e6.max(), 'bf16x3', e3.max(), 'one fp32 rounding', e32.max())
    assert e6.max() < 2.0 ** -22 and e6.mean() < 2.0 ** -25                  # the dropped terms (m,l) (l,m) (l,l): <= 2^-8 * 2^-16 * 2 + 2^-32 relative to hi*hi
    assert e32.max() <= 2.0 ** -24 * 1.0001
    assert e3.max() > 2.0 ** -18 and e3.max() < 2.0 ** -14                   # what the third piece buys: two orders of magnitude
