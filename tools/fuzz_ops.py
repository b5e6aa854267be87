#!/usr/bin/env python3
"""Randomised op-level parity through the module surface of the reference (`PreNorm(Attention)`, `PreNorm(FeedForward)`):
random batch / latent rows / token counts / context widths / head counts and dims, optional key mask, self- and
cross-attention, `attn_weights` on demand -- against the CPU oracle.  These calls take the granular C-ABI entry points
(hn_attn_fwd, hn_ff_fwd, hn_attn_probs), not the fused forward tools/fuzz_forward.py exercises.

    python tools/fuzz_ops.py [--n 200] [--seed 0]
"""
import argparse, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
from oracle import healnet_cpu as O

DEV = "cuda:0"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    rng = random.Random(args.seed)
    bad, worst = 0, 0.0
    with torch.no_grad():
        for case in range(args.n):
            gen = torch.Generator().manual_seed(7000 + case)
            b = rng.choice([1, 2, 3, 5, 33])
            L = rng.choice([1, 4, 16, 17, 25, 128, 130])
            qd = rng.choice([16, 17, 32, 119, 128])
            heads = rng.choice([1, 2, 4, 8])
            dh = rng.choice([4, 8, 16, 27, 32, 63, 64, 103])
            kind = rng.choice(["cross", "cross", "cross", "self", "ff"])
            if kind == "ff":
                snn = rng.random() < 0.6
                blk = hn.PreNorm(qd, hn.FeedForward(qd, snn=snn)).to(DEV)
                for p_ in blk.parameters():
                    if p_.dim() == 1:
                        p_.add_(0.2 * torch.randn(p_.shape, generator=gen).to(DEV))
                x = torch.randn(b, L, qd, generator=gen)
                sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
                xn = O.layer_norm(x, sd["norm.weight"], sd["norm.bias"])
                want = O.feed_forward(xn, sd["fn.net.0.weight"], sd["fn.net.0.bias"], sd["fn.net.2.weight"], sd["fn.net.2.bias"], snn=snn)
                e = rel(blk(x.to(DEV)), want)
                desc = f"ff b={b} L={L} d={qd} snn={snn}"
            else:
                cross = kind == "cross"
                N = rng.choice([1, 2, 15, 16, 33, 65, 300, 1000]) if cross else L
                D = rng.choice([3, 6, 12, 13, 15, 16, 17, 18, 29, 31, 32, 40, 96, 130, 773]) if cross else qd
                masked = cross and rng.random() < 0.3
                att = hn.Attention(qd, D if cross else None, heads=heads, dim_head=dh)
                blk = (hn.PreNorm(qd, att, context_dim=D) if cross else hn.PreNorm(qd, att)).to(DEV)
                for p_ in blk.parameters():
                    if p_.dim() == 1:
                        p_.add_(0.2 * torch.randn(p_.shape, generator=gen).to(DEV))
                blk.fn.to_q.weight.mul_(2.0)
                x = torch.randn(b, L, qd, generator=gen)
                ctx = torch.rand(b, N, D, generator=gen) * 2 if cross else None
                mask = None
                if masked:
                    mask = torch.rand(b, N, generator=gen) > 0.4
                    mask[:, 0] = True
                sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
                xn = O.layer_norm(x, sd["norm.weight"], sd["norm.bias"])
                cn = O.layer_norm(ctx, sd["norm_context.weight"], sd["norm_context.bias"]) if cross else None
                want, pw = O.attention(xn, cn, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"],
                                       heads, mask=mask, return_weights=True)
                got = blk(x.to(DEV), **({"context": ctx.to(DEV), "mask": None if mask is None else mask.to(DEV)} if cross else {}))
                e = rel(got, want)
                e = max(e, 0.4 * rel(blk.fn.attn_weights, pw))
                desc = f"{kind} b={b} L={L} N={N} D={D} qd={qd} h={heads}x{dh} mask={masked}"
            worst = max(worst, e)
            flag = "" if e <= 2e-4 else "   <<<<<< FAIL"
            bad += bool(flag)
            print(f"[{case}] {desc}: {e:.1e}{flag}", flush=True)
    print(f"worst error {worst:.2e}; {bad} failing case(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
