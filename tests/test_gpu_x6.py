"""GPU: the fp32-exact patch-bag GEMMs on the bf16 matrix pipe (healnet_amd/csrc/gemm_x6.hip) -- the K/V projection of
healnet/models/healnet.py:405 on a large bag and its weight gradient G = dKV^T z, each fp32 operand as three bf16 planes, six
bf16 products per fp32 product.

  * kernel level, through tests/_build/libhn_x6_check.so (the product's own sources behind plain C entry points): BOTH routes --
    the split kernels and the fp32-MFMA kernels of gemm_nt.hip -- against an fp64 product of the same operands.  The claim under
    test: the split route's error is NOT larger than the fp32 MFMA's (it is slightly smaller: the six partial products are exact
    and only their sum rounds).  Shapes: BASELINE configs[3]'s (32 768 x 773 -> 1024), ragged ones (rows / columns / contraction
    off every tile size), operands with a wide dynamic range, exactly representable operands (bit-exact result required: any
    layout or index slip shows), NaNs in the row pads (must not enter);
  * model level: route A/B in two subprocesses (HN_FORCE_X6_GEMM=1 against HN_NO_X6_GEMM=1) -- logits and EVERY parameter gradient
    of small ragged models (padded head width, two bags, odd row counts), and of BASELINE configs[3] at its real size through the
    default gates; the two routes must agree to fp32 rounding and must NOT be bit-identical (the switch is proven to act);
  * the same against the oracle is tests/test_gpu_fullsize.py::test_cfg4_full_size_gradients_vs_oracle_autograd (x6 is its default
    route) and the suite's other bag models.
"""
import ctypes as C
import os
import subprocess
import sys

import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "_build", "libhn_x6_check.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build_test_helpers()
    L = C.CDLL(LIB)
    L.x6_check_ws_bytes.restype = C.c_size_t
    L.x6_check_ws_bytes.argtypes = [C.c_long, C.c_int, C.c_int]
    L.x6_check_nt.restype = C.c_int
    L.x6_check_nt.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.x6_check_tn.restype = C.c_int
    L.x6_check_tn.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                              C.c_void_p]
    return L


def _operands(M, N, K, dist, seed):
    g = torch.Generator().manual_seed(seed)
    if dist == "uniform":
        A = torch.rand(M, K, generator=g) * 2 - 1
        W = (torch.rand(N, K, generator=g) * 2 - 1) * 0.05
    elif dist == "wide":      # 6 decades of magnitude inside every row: the low planes carry real weight
        A = torch.randn(M, K, generator=g) * torch.exp(7.0 * (torch.rand(M, K, generator=g) - 0.5))
        W = torch.randn(N, K, generator=g) * torch.exp(7.0 * (torch.rand(N, K, generator=g) - 0.5)) * 0.05
    else:                     # "exact": small integers / 8 -- every product and every partial sum is exact in fp32
        A = torch.randint(-8, 9, (M, K), generator=g).float() / 8
        W = torch.randint(-8, 9, (N, K), generator=g).float() / 8
    return A, W


def _nt(lib, A, W, route, pad=0.0):
    M, K = A.shape
    N = W.shape[0]
    lda = (K + 3) // 4 * 4
    Ad = torch.full((M, lda), pad, device=DEV)
    Ad[:, :K] = A.to(DEV)
    Wd = W.to(DEV).contiguous()
    Cd = torch.full((M, N), float("nan"), device=DEV)
    ws = torch.empty(lib.x6_check_ws_bytes(M, N, K), dtype=torch.uint8, device=DEV)
    rc = lib.x6_check_nt(Ad.data_ptr(), lda, Wd.data_ptr(), M, N, K, Cd.data_ptr(), route, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, f"x6_check_nt(route {route}) returned {rc}"
    torch.cuda.synchronize()
    return Cd


NT_SHAPES = [(32768, 1024, 773), (4100, 260, 70), (16421, 512, 1029), (2048, 32, 64)]


@pytest.mark.parametrize("dist", ["uniform", "wide"])
@pytest.mark.parametrize("shape", NT_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_nt_error_is_not_larger_than_the_fp32_mfma_s(lib, shape, dist):
    M, N, K = shape
    A, W = _operands(M, N, K, dist, seed=M + N + K)
    ref = A.to(DEV).double() @ W.to(DEV).double().T
    scale = float(ref.abs().max())
    # (the fp32 route wants finite row pads -- gemm_nt.hip's contract; the split route is given NaNs there: they must never enter)
    c32 = _nt(lib, A, W, 0)
    c6 = _nt(lib, A, W, 1, pad=float("nan"))
    assert torch.isfinite(c6).all(), "a NaN of the row pad entered the product"
    e32, e6 = (c32.double() - ref).abs(), (c6.double() - ref).abs()
    m32, m6 = float(e32.max()), float(e6.max())
    r32, r6 = float(e32.pow(2).mean().sqrt()), float(e6.pow(2).mean().sqrt())
    print(f"NT {M}x{K}->{N} ({dist}): |C|max {scale:.3g}; fp32 MFMA max {m32:.2e} rms {r32:.2e}; x6 max {m6:.2e} rms {r6:.2e}")
    assert r6 <= 1.05 * r32 + 1e-12 * scale, f"rms error {r6:.3e} of the split route above the fp32 MFMA's {r32:.3e}"
    assert m6 <= 1.5 * m32 + 1e-12 * scale, f"max error {m6:.3e} of the split route above the fp32 MFMA's {m32:.3e}"


@pytest.mark.parametrize("shape", [(4100, 260, 70), (2304, 1024, 773)], ids=lambda s: "x".join(map(str, s)))
def test_nt_exactly_representable_operands_give_the_exact_product(lib, shape):
    """Operands on a grid of 1/8 with |x| <= 1: every product is a multiple of 1/64 below 1, every partial sum is exact in fp32 --
    both routes must return the fp64 product BIT FOR BIT; a swapped plane, a shifted k-step or a wrong row of a fragment cannot."""
    M, N, K = shape
    A, W = _operands(M, N, K, "exact", seed=5)
    ref = (A.to(DEV).double() @ W.to(DEV).double().T).float()
    c6 = _nt(lib, A, W, 1)
    bad = (c6 != ref).nonzero()
    assert bad.numel() == 0, f"{bad.shape[0]} elements differ, first at {bad[0].tolist()}: {float(c6[tuple(bad[0])])} vs {float(ref[tuple(bad[0])])}"


def _tn(lib, A, B, route):
    R, M = A.shape
    N = B.shape[1]
    ldb = (N + 3) // 4 * 4
    Ad = A.to(DEV).contiguous()
    Bd = torch.zeros(R, ldb, device=DEV)
    Bd[:, :N] = B.to(DEV)
    G = torch.full((M, N), float("nan"), device=DEV)
    cs = torch.full((M,), float("nan"), device=DEV)
    ws = torch.empty(lib.x6_check_ws_bytes(R, M, N), dtype=torch.uint8, device=DEV)
    rc = lib.x6_check_tn(Ad.data_ptr(), Bd.data_ptr(), ldb, R, M, N, G.data_ptr(), cs.data_ptr(), route, ws.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, f"x6_check_tn(route {route}) returned {rc}"
    torch.cuda.synchronize()
    return G, cs


TN_SHAPES = [(32768, 1024, 773), (16391, 432, 773), (20000, 1024, 100)]


@pytest.mark.parametrize("dist", ["uniform", "wide"])
@pytest.mark.parametrize("shape", TN_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_tn_error_is_not_larger_than_the_fp32_mfma_s(lib, shape, dist):
    R, M, N = shape
    A, _ = _operands(R, 4, M, dist, seed=R + M)      # (R, M): stands for dKV
    B, _ = _operands(R, 4, N, "uniform", seed=R + N)  # (R, N): the normalised bag
    ref = A.to(DEV).double().T @ B.to(DEV).double()
    ref_cs = A.to(DEV).double().sum(0)
    g32, cs32 = _tn(lib, A, B, 0)
    g6, cs6 = _tn(lib, A, B, 1)
    assert torch.isfinite(g6).all() and torch.isfinite(cs6).all()
    e32, e6 = (g32.double() - ref).abs(), (g6.double() - ref).abs()
    r32, r6 = float(e32.pow(2).mean().sqrt()), float(e6.pow(2).mean().sqrt())
    m32, m6 = float(e32.max()), float(e6.max())
    c32, c6 = float((cs32.double() - ref_cs).abs().max()), float((cs6.double() - ref_cs).abs().max())
    scale, cscale = float(ref.abs().max()), float(ref_cs.abs().max())
    print(f"TN {M}x{N} over {R} rows ({dist}): |G|max {scale:.3g}; fp32 MFMA max {m32:.2e} rms {r32:.2e}; x6 max {m6:.2e} rms {r6:.2e}; "
          f"column sums (|cs|max {cscale:.3g}): {c32:.2e} / {c6:.2e}")
    # (both routes cut the contraction into slices and add the partials in fp32; the slice counts differ, so the comparison is
    # statistical: same order of magnitude, and both far inside the gradient tolerances of the suite)
    assert r6 <= 1.3 * r32 + 1e-12 * scale, f"rms error {r6:.3e} of the split route above the fp32 MFMA's {r32:.3e}"
    assert m6 <= 2.0 * m32 + 1e-12 * scale
    assert c6 <= 4.0 * c32 + 2e-7 * cscale


def test_tn_exactly_representable_operands_give_the_exact_product(lib):
    R, M, N = 16400, 432, 773
    g = torch.Generator().manual_seed(9)
    A = torch.randint(-4, 5, (R, M), generator=g).float() / 4
    B = torch.randint(-4, 5, (R, N), generator=g).float() / 4
    ref = (A.to(DEV).double().T @ B.to(DEV).double()).float()
    ref_cs = A.to(DEV).double().sum(0).float()
    g6, cs6 = _tn(lib, A, B, 1)
    assert torch.equal(g6, ref), f"{int((g6 != ref).sum())} elements of G differ"
    assert torch.equal(cs6, ref_cs), "the ones column does not return the exact column sums"


# ---- model level: the two routes of the same forward + backward in two processes
CASES = {
    # name: (constructor kwargs, input shapes, batch, forced)
    "ragged_padded_heads": (dict(n_modalities=2, channel_dims=[300, 60], num_spatial_axes=[1, 1], out_dims=3, depth=2, x_heads=2,
                                 cross_dim_head=27), [(1, 300), (701, 60)], 3, True),
    "two_bags": (dict(n_modalities=3, channel_dims=[120, 64, 90], num_spatial_axes=[1, 1, 1], out_dims=4, depth=2, x_heads=4,
                      cross_dim_head=32), [(1, 120), (1100, 64), (530, 90)], 4, True),
    "masked_single_bag": (dict(n_modalities=1, channel_dims=[96], num_spatial_axes=[1], out_dims=2, depth=2, x_heads=4),
                          [(803, 96)], 5, True),      # key-padding mask on the bag (healnet.py:412-416): masked rows carry no gradient
    "cfg4_b8_default_gates": (dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4),
                              [(1, 2000), (4096, 768)], 8, False),
}

_SCRIPT = """
import sys, torch
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests")
import healnet_amd as hn
from test_gpu_x6 import CASES
res = {{}}
for name, (kw, shapes, b, forced) in CASES.items():
    torch.manual_seed(3)
    model = hn.HealNet(**kw).train().to("cuda:0")
    flat = hn.train.flatten_parameters(model)
    gen = torch.Generator().manual_seed(5)
    ins = [torch.rand(b, *s, generator=gen).to("cuda:0") for s in shapes]
    mask = None
    if name.startswith("masked"):
        mask = (torch.rand(b, shapes[0][0], generator=gen) > 0.3).to("cuda:0")
        mask[:, 0] = True

    def step():
        flat.zero_grad()
        y = model(list(ins), mask=mask)
        wts = torch.linspace(0.5, 1.5, y.numel(), device=y.device).view_as(y)
        (y * wts).sum().backward()
        return y.detach().clone(), wts, {{k: p.grad.detach().clone() for k, p in model.named_parameters()}}

    y, wts, g = step()
    out = {{"logits": y.cpu()}}
    for k, v in g.items():
        out["grad." + k] = v.cpu()
    # the same step again: the route is deterministic (fixed-order split-k, no atomics)
    y2, _, g2 = step()
    out["deterministic"] = torch.tensor(torch.equal(y2, y) and all(torch.equal(g2[k], g[k]) for k in g))
    # ... and as ONE graph replay (healnet_amd.train.GraphedStep): the captured launches are the eager ones
    gstep = hn.train.GraphedStep(model, lambda logits, w: (logits * w).sum(), list(ins), (wts,), mask=mask)
    gstep(ins, (wts,), mask=mask)
    torch.cuda.synchronize()
    out["graph_equals_eager"] = torch.tensor(all(torch.equal(p.grad, g[k]) for k, p in model.named_parameters()))
    gstep.close()
    model.eval()
    with torch.no_grad():
        out["logits_eval"] = model(list(ins), mask=mask).cpu()
    res[name] = out
torch.save(res, sys.argv[1])
"""


@pytest.fixture(scope="module")
def routes(tmp_path_factory):
    d = tmp_path_factory.mktemp("x6")
    script = d / "run.py"
    script.write_text(_SCRIPT.format(root=ROOT))
    res = {}
    for tag, env in (("x6", {"HN_FORCE_X6_GEMM": "1"}), ("fp32", {"HN_NO_X6_GEMM": "1"}), ("default", {})):
        e = dict(os.environ, **env)
        for k in ("HN_FORCE_X6_GEMM", "HN_NO_X6_GEMM"):
            if k not in env:
                e.pop(k, None)
        out = d / f"{tag}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = torch.load(out)
    return res


@pytest.mark.parametrize("name", list(CASES))
def test_the_two_routes_agree_to_fp32_rounding(routes, name):
    forced = CASES[name][3]
    a, p = routes["x6" if forced else "default"][name], routes["fp32"][name]
    assert a.keys() == p.keys()
    differs = 0
    for flag in ("deterministic", "graph_equals_eager"):
        for r in (a, p):
            assert flag not in r or bool(r[flag]), f"{name}: {flag} is false"
    for k in a:
        if k in ("deterministic", "graph_equals_eager"):
            continue
        assert torch.isfinite(a[k]).all(), f"{name}: {k} of the split route is not finite"
        # logits to 2e-5; gradients to 1e-4 of their tensor's scale (LeakyReLU / SELU kinks may flip on single elements downstream of
        # a last-bit difference -- tests/test_gpu_fullsize.py's allowance: a handful of elements beyond 5e-4, none beyond 5e-3)
        if k.startswith("grad."):
            scale = float(p[k].abs().max().clamp_min(1e-30))
            d = (a[k].double() - p[k].double()).abs() / scale
            assert float(d.max()) <= 5e-3 and int((d > 5e-4).sum()) <= max(4, int(1e-4 * d.numel())), \
                f"{name}: {k}: max {float(d.max()):.2e} of scale, {int((d > 5e-4).sum())} elements beyond 5e-4"
        else:
            assert_close(a[k], p[k], rel=2e-5, floor=2e-6, what=f"{name}: {k}, split route vs fp32 MFMA route")
        differs += int(not torch.equal(a[k], p[k]))
    assert differs > 0, f"{name}: both runs took the same route (the switches had no effect)"


def test_the_default_route_of_small_bags_is_the_fp32_mfma(routes):
    """Below the size gates (8192 rows) nothing changes: the default run of the small models is bit-equal to HN_NO_X6_GEMM=1."""
    for name, (_, _, _, forced) in CASES.items():
        if not forced:
            continue
        for k, v in routes["default"][name].items():
            assert torch.equal(v, routes["fp32"][name][k]), f"{name}: {k} differs between the default and the fp32 route"


def test_the_first_forward_of_a_process_may_be_a_capture_on_this_route():
    """HealNet.capture() as the very first call of a process at BASELINE configs[3]'s size: the route's one-time set-up (dynamic-LDS
    attributes of its kernels) happens inside the warm-up run of the capture, the replay holds the image kernels and both GEMMs, and
    equals the eager forward bit for bit."""
    script = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
import healnet_amd as hn
torch.manual_seed(0)
model = hn.HealNet(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4).eval().to("cuda:0")
gen = torch.Generator().manual_seed(2)
ins = [torch.rand(8, 1, 2000, generator=gen).cuda(), torch.rand(8, 4096, 768, generator=gen).cuda()]
with torch.no_grad():
    g = model.capture(list(ins))
    a = g().clone()
    b = g([t.clone() for t in ins]).clone()
    e = model(list(ins))
print("EQUAL", bool(torch.equal(a, e) and torch.equal(b, e) and torch.isfinite(e).all()))
"""
    env = {k: v for k, v in os.environ.items() if k not in ("HN_FORCE_X6_GEMM", "HN_NO_X6_GEMM")}
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    assert "EQUAL True" in r.stdout, r.stdout + r.stderr[-800:]


def test_batch_sizes_on_both_sides_of_the_size_gate_in_one_process():
    """b = 8 (32 768 bag rows: this route on 256 x 256 tiles), 3 (12 288: the same), 2 (8192: this route on 256 x 128 tiles, exactly at
    the gate), 1 (4096: the fp32-MFMA kernels), 8 again, on one model in one process, inference and training forward + backward: the per-sample results do not depend on which route or
    batch they were computed in beyond fp32 rounding, and the last b = 8 run is bit-equal to the first (no state survives a call)."""
    import healnet_amd as hn
    torch.manual_seed(1)
    kw = dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4)
    model = hn.HealNet(**kw).to(DEV)
    gen = torch.Generator().manual_seed(6)
    full = [torch.rand(8, 1, 2000, generator=gen).to(DEV), torch.rand(8, 4096, 768, generator=gen).to(DEV)]

    def run(b, train):
        model.train(train)
        ins = [t[:b].contiguous() for t in full]
        if not train:
            with torch.no_grad():
                return model(list(ins)).clone(), None
        model.zero_grad(set_to_none=True)
        y = model(list(ins))
        y.sum().backward()
        return y.detach().clone(), model.layers[0][2].fn.to_kv.weight.grad.clone()      # layer 0, the bag's cross-attention: to_kv

    for train in (False, True):
        first, g_first = run(8, train)
        assert torch.isfinite(first).all()
        for b in (3, 2, 1):
            y, _ = run(b, train)
            assert_close(y, first[:b], rel=2e-5, floor=2e-6, what=f"b={b} ({'training' if train else 'inference'}) against the first {b} samples of b=8")
        again, g_again = run(8, train)
        assert torch.equal(again, first), "the second b = 8 run differs from the first"
        if g_first is not None:
            assert torch.equal(g_again, g_first)
