#!/usr/bin/env python3
"""Randomised parity sweep aimed at the layer chains (lchain.hip) and at the host's all-or-nothing eligibility plan
(api_fusion.hip): default latent widths (l_c = l_d = 128, 8 x 64 self-attention) so that the route applies, random modality mixes
(one-token tabular, 2-D image on the shared-context binding, small patch bag on the explicit binding = NOT eligible -> the per-block
route must be chosen), random order, random missing modalities, depth, gate, batch sizes around the size gate (with
HN_FORCE_SELF_IN_CHAIN=1 also below it), with / without the attention trace -- logits against the CPU oracle.

    HN_FORCE_SELF_IN_CHAIN=1 python tools/fuzz_layer_chain.py --n 40 --seed 0"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    import healnet_amd as hn
    from oracle import healnet_cpu as O
    gen = torch.Generator().manual_seed(1000 + args.seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=gen))

    worst, bad = 0.0, 0
    for case in range(args.n):
        M = ri(1, 4)
        kinds = [["tab", "img", "tab", "img", "bag"][ri(0, 4)] for _ in range(M)]
        dims, axes, shapes = [], [], []
        for k in kinds:
            if k == "tab":
                c = [2000, 700, 1500][ri(0, 2)]
                dims.append(c); axes.append(1); shapes.append((1, c))
            elif k == "img":
                dims.append(3); axes.append(2); shapes.append((ri(8, 40), ri(8, 40), 3))
            else:
                dims.append(96); axes.append(1); shapes.append((ri(30, 200), 96))
        kw = dict(n_modalities=M, channel_dims=dims, num_spatial_axes=axes, out_dims=ri(2, 5), depth=ri(1, 3), snn=bool(ri(0, 1)))
        b = [3, 17, 18, 24, 32, 33][ri(0, 5)]
        present = [bool(ri(0, 4)) for _ in range(M)]
        if not any(present):
            present[ri(0, M - 1)] = True
        torch.manual_seed(case * 7 + args.seed)
        model = hn.HealNet(**kw).eval()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        ins = [torch.rand(b, *s, generator=gen) if p else None for s, p in zip(shapes, present)]
        with torch.no_grad():
            want = O.fusion_forward(sd, O.FusionConfig(**kw), [None if t is None else t.clone() for t in ins])
        model.to("cuda:0")
        model.keep_attention_stats = bool(ri(0, 1))
        with torch.no_grad():
            got = model([None if t is None else t.to("cuda:0") for t in ins]).cpu()
            got2 = model([None if t is None else t.to("cuda:0") for t in ins]).cpu()
        err = float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))
        ok = bool(torch.isfinite(got).all()) and err <= 1e-3 and bool(torch.equal(got, got2))
        worst = max(worst, err)
        if not ok:
            bad += 1
            print(f"FAIL case {case}: kinds {kinds} present {present} depth {kw['depth']} b {b} snn {kw['snn']} trace {model.keep_attention_stats}: rel err {err:.3e}", flush=True)
    print(f"fuzz_layer_chain: {args.n} cases, {bad} failed, worst rel err {worst:.2e}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
