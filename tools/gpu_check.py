#!/usr/bin/env python3
"""Development aid: run a handful of op-level comparisons against the oracle and PRINT the errors
(no early exit), so that one gpurun call gives a full picture."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
from oracle import healnet_cpu as O

DEV = "cuda:0"
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

def run(name, fn):
    try:
        t = time.time(); r = fn(); torch.cuda.synchronize()
        print(f"[{name}] {r}  ({time.time()-t:.2f}s)", flush=True)
    except Exception as e:
        print(f"[{name}] EXC {type(e).__name__}: {e}", flush=True); traceback.print_exc()

def enc():
    out = []
    for shape in [(2, 6, 5, 3), (2, 1, 20), (2, 224, 224, 3), (2, 64, 768)]:
        x = torch.rand(*shape)
        out.append(rel(hn.fourier_encode_concat(x.to(DEV)), O.encode_modality(x, 2, 10.0)))
    return out

def ff():
    blk = hn.PreNorm(128, hn.FeedForward(128, snn=True)).to(DEV)
    x = torch.randn(2, 128, 128)
    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    want = O.feed_forward(O.layer_norm(x, sd["norm.weight"], sd["norm.bias"]), sd["fn.net.0.weight"], sd["fn.net.0.bias"], sd["fn.net.2.weight"], sd["fn.net.2.bias"], True)
    return rel(blk(x.to(DEV)), want)

def attn(b, L, N, D, heads, dh, qd=128):
    def f():
        blk = hn.PreNorm(qd, hn.Attention(qd, D, heads=heads, dim_head=dh), context_dim=D).to(DEV)
        x = torch.randn(b, L, qd); ctx = torch.rand(b, N, D)
        with torch.no_grad():
            blk.fn.to_q.weight.mul_(2.0); blk.fn.to_kv.weight.mul_(2.0)
        sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
        want, pw = O.attention(O.layer_norm(x, sd["norm.weight"], sd["norm.bias"]), O.layer_norm(ctx, sd["norm_context.weight"], sd["norm_context.bias"]),
                           sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"], heads, return_weights=True)
        got = blk(x.to(DEV), context=ctx.to(DEV))
        return rel(got, want), rel(blk.fn.attn_weights, pw)
    return f

def selfattn():
    blk = hn.PreNorm(128, hn.Attention(128, heads=8, dim_head=64)).to(DEV)
    x = torch.randn(2, 128, 128)
    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    want = O.attention(O.layer_norm(x, sd["norm.weight"], sd["norm.bias"]), None, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"], 8)
    return rel(blk(x.to(DEV)), want)

def model(kw, shapes):
    def f():
        torch.manual_seed(0)
        m = hn.HealNet(**kw).eval()
        ins = [torch.rand(*s) for s in shapes]
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        with torch.no_grad():
            want = O.fusion_forward(sd, O.FusionConfig(**kw), ins)
        m.to(DEV)
        got = m([t.to(DEV) for t in ins])
        return rel(got, want), got[0].tolist(), want[0].tolist()
    return f

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    run("encode", enc)
    run("ff", ff)
    run("self", selfattn)
    run("attn rankD img", attn(2, 128, 5000, 13, 8, 64))
    run("attn rankD odd", attn(2, 25, 333, 18, 2, 63, qd=32))
    run("attn explicit wsi", attn(1, 128, 1024, 773, 8, 64))
    run("attn explicit tab", attn(3, 128, 1, 2005, 8, 64))
    run("attn explicit odd", attn(2, 17, 65, 40, 4, 27, qd=32))
    run("model tiny", model(dict(n_modalities=2, channel_dims=[20, 3], num_spatial_axes=[1, 2], out_dims=3, l_c=8, l_d=16, x_heads=2, l_heads=2, cross_dim_head=4, latent_dim_head=4), [(2, 1, 20), (2, 6, 5, 3)]))
    run("model cfg1", model(dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4), [(4, 1, 2000), (4, 224, 224, 3)]))
    # quick timing of cfg2
    torch.manual_seed(0)
    m = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to(DEV)
    tab, img = torch.rand(32, 1, 2000, device=DEV), torch.rand(32, 224, 224, 3, device=DEV)
    for _ in range(3): m([tab, img])
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): m([tab, img])
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print(f"[cfg2 b=32] {dt*1e3:.2f} ms/forward  {32/dt:.0f} samples/s", flush=True)
