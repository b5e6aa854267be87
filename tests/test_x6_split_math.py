"""CPU: the arithmetic healnet_amd/csrc/gemm_x6.hip relies on, restated in numpy (no GPU, no library).

An fp32 number splits EXACTLY into three bf16 numbers, x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
(round to nearest even, both differences exact in fp32); of the nine bf16 x bf16 products of x y the three that gemm_x6 drops
(m l', l m', l l') are at most 2^-23 |x y| and 2^-27.4 rms -- the rounding of one fp32 product is 2^-24 at most, 2^-25.2 rms.  The GPU side of the claim
(the six-product MFMA sum against fp64, beside the fp32 MFMA) is tests/test_gpu_x6.py."""
import numpy as np


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32: what v_cvt_pk_bf16_f32 does for finite inputs."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    l = bf16_rne(r2)
    return h, m, l, r1, r2


def _samples(n, seed):
    g = np.random.default_rng(seed)
    mant = g.standard_normal(n).astype(np.float32)
    expo = g.integers(-60, 60, n)                        # (|x| in ~[1e-19, 1e19]: far inside the range where all three planes are normal)
    x = (mant * np.exp2(expo.astype(np.float64))).astype(np.float32)
    # the corners: every bit of the significand set, powers of two, values one ulp around a bf16 rounding boundary
    extra = np.array([1.0, -1.0, 1.9999999, 1.00390625, 1.0039063, 1.0039062, 3.38e38, -3.3e38, 1e-30, 255.99998, 0.0], dtype=np.float32)
    return np.concatenate([x, extra])


def test_three_bf16_planes_hold_an_fp32_number_exactly():
    x = _samples(2_000_000, 0)
    h, m, l, r1, r2 = split3(x)
    assert np.all(np.isfinite(h)) and np.all(np.isfinite(m)) and np.all(np.isfinite(l))
    # the two differences are exact (Sterbenz-like: h is within half a bf16 ulp of x), so the fp32 subtractions lose nothing
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
    total = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64)), "h + m + l != x for %d samples" % int((total != x).sum())
    # sizes of the planes: |m| <= 2^-8 |x| (half a bf16 ulp of x, which is at most 2^-8 |x|), |l| <= 2^-16 |x|
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(m) <= ax * 2.0 ** -8) and np.all(np.abs(l) <= ax * 2.0 ** -16)


def test_the_documented_limits_of_the_split():
    """|x| above the largest bf16 (3.3895e38, the top 0.4 % of fp32's last binade) rounds to Inf in the first plane and the result
    is NaN where the fp32 product might have been finite; a third plane below the smallest subnormal flushes.  Both are stated in
    gemm_x6.hip / DESIGN.md 4.4 and irrelevant for LayerNorm-ed rows and gradients."""
    h, m, l, _, _ = split3(np.array([3.4e38], dtype=np.float32))
    assert np.isinf(h[0])
    # below fp32's normal range the planes are bf16 subnormals (spacing 2^-133): the split is then off by at most half of that
    tiny = np.array([1.0000001e-38, 1e-40, -3.3e-39], dtype=np.float32)
    h, m, l, _, _ = split3(tiny)
    err = np.abs(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64) - tiny.astype(np.float64))
    assert np.all(err <= 2.0 ** -134)


def test_the_three_dropped_products_are_below_the_rounding_of_one_fp32_product():
    x, y = _samples(500_000, 1), _samples(500_000, 2)[::-1].copy()
    hx, mx, lx, _, _ = split3(x)
    hy, my, ly, _, _ = split3(y)
    f = lambda a: a.astype(np.float64)
    kept = f(hx) * f(hy) + f(hx) * f(my) + f(mx) * f(hy) + f(hx) * f(ly) + f(mx) * f(my) + f(lx) * f(hy)
    dropped = f(mx) * f(ly) + f(lx) * f(my) + f(lx) * f(ly)
    exact = f(x) * f(y)
    ok = exact != 0
    assert np.allclose(kept + dropped, exact, rtol=0, atol=0) or np.max(np.abs((kept + dropped - exact)[ok] / exact[ok])) < 2.0 ** -50
    rel = np.abs(dropped[ok] / exact[ok])
    assert rel.max() <= 2.0 ** -23, rel.max()            # worst case: 2 x 2^-8 x 2^-16 + 2^-32
    rms = float(np.sqrt((rel ** 2).mean()))
    p32 = (x * y).astype(np.float32)                     # what one fp32 product rounds away
    sane = ok & np.isfinite(p32) & (np.abs(exact) > 1e-30)
    r32 = np.abs((f(p32) - exact)[sane] / exact[sane])
    rms32 = float(np.sqrt((r32 ** 2).mean()))
    assert rms <= 2.0 ** -27 and rms32 >= 2.0 ** -25.5 and rms <= 0.3 * rms32, (rms, rms32)
    # every kept product is exact in fp32: 8 x 8 significand bits
    for a, b in ((hx, hy), (hx, my), (mx, hy), (hx, ly), (mx, my), (lx, hy)):
        p = (a.astype(np.float32) * b.astype(np.float32))
        fin = np.isfinite(p) & (np.abs(f(a) * f(b)) > 1e-37)
        assert np.array_equal(f(p)[fin], (f(a) * f(b))[fin])
