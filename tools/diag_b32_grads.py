#!/usr/bin/env python3
"""Diagnostic: cfg2 b=32 full-size gradients -- fused / unfused GPU routes vs the oracle in fp32 and fp64 (which differences are
the oracle's own fp32 rounding + kink flips, which are the build's).   python tools/diag_b32_grads.py [gpu-fused|gpu-unfused]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

KW = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
B = 32


def inputs():
    gen = torch.Generator().manual_seed(4132)
    ins = [torch.rand(B, 1, 2000, generator=gen), torch.rand(B, 224, 224, 3, generator=gen)]
    dl = torch.randn(B, 4, generator=gen)
    return ins, dl


def gpu(tag):
    import healnet_amd as hn
    ins, dl = inputs()
    torch.manual_seed(43)
    model = hn.HealNet(**KW).train().to("cuda:0")
    got = model([t.to("cuda:0") for t in ins])
    (got * dl.to("cuda:0")).sum().backward()
    torch.save({k: p.grad.cpu() for k, p in model.named_parameters()}, f"/tmp/g_{tag}.pt")


def oracle(dtype, tag):
    import healnet_amd as hn
    from oracle import healnet_cpu as O
    ins, dl = inputs()
    torch.manual_seed(43)
    model = hn.HealNet(**KW).train()
    sd = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in model.state_dict().items()}
    torch.set_num_threads(32)
    t = time.time()
    for i in range(B):
        out = O.fusion_forward(sd, O.FusionConfig(**KW), [x[i:i + 1].to(dtype) for x in ins])
        (out * dl[i:i + 1].to(dtype)).sum().backward()
    print(f"oracle {tag}: {time.time() - t:.0f} s", flush=True)
    torch.save({k: (v.grad if v.grad is not None else torch.zeros_like(v)).double() for k, v in sd.items()}, f"/tmp/g_{tag}.pt")


def compare(a, b):
    A, Bm = torch.load(f"/tmp/g_{a}.pt"), torch.load(f"/tmp/g_{b}.pt")
    rows = []
    for k in A:
        if k not in Bm:
            continue
        x, y = A[k].double(), Bm[k].double()
        scale = float(y.abs().max().clamp_min(1e-30))
        d = x - y
        linf = float(d.abs().max()) / scale
        l2 = float(d.norm() / y.norm().clamp_min(1e-30))
        rank = ""
        if d.dim() == 2 and min(d.shape) >= 8:
            s = torch.linalg.svdvals(d)
            e = (s * s).cumsum(0) / (s * s).sum()
            rank = f"energy in top 1/4/16 singular values {float(e[0]):.2f}/{float(e[3]):.2f}/{float(e[min(15, len(e) - 1)]):.2f}"
        rows.append((l2, linf, k, int((d.abs() > 5e-4 * scale).sum()), rank))
    rows.sort(reverse=True)
    print(f"== {a} vs {b}: worst L2 {rows[0][0]:.2e}, worst max-norm {max(r[1] for r in rows):.2e}")
    for r in rows[:6]:
        print(f"   L2 {r[0]:.2e} linf {r[1]:.2e} out {r[3]:5d}  {r[2]}  {r[4]}")


if len(sys.argv) > 1:
    gpu(sys.argv[1])
    sys.exit(0)
subprocess.run([sys.executable, __file__, "fused"], check=True)
subprocess.run([sys.executable, __file__, "unfused"], check=True, env=dict(os.environ, HN_NO_BCHAIN="1", HN_NO_CHAIN="1"))
oracle(torch.float32, "o32")
oracle(torch.float64, "o64")
for a, b in (("fused", "o32"), ("unfused", "o32"), ("fused", "unfused"), ("o32", "o64"), ("fused", "o64"), ("unfused", "o64")):
    compare(a, b)
