#!/usr/bin/env python3
"""Randomised whole-model parity: random modality mixes (axes, channels, token counts), head counts / dims, latent sizes,
batch sizes, key masks, missing modalities, on BOTH forwards (inference under no_grad, tape-recording with grad mode on) and
the backward, against the CPU oracle.  Every fast path has entry conditions on these shapes; the BASELINE configs only
exercise a few of them.

    python tools/fuzz_forward.py [--n 60] [--seed 0] [--backward]
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


def random_case_medium(rng):
    """Larger shapes: the size-gated routes (128x128 GEMM of the patch-bag projection, split-K latent GEMMs, Q+KV in one launch,
    one-token look-ahead, many-split merges, row-chunked GEMV above 32 samples)."""
    M = rng.choice([1, 2, 2, 3])
    chans, axes, shapes = [], [], []
    for _ in range(M):
        kind = rng.choice(["tab_wide", "img", "vol", "bag", "seq"])
        if kind == "tab_wide":
            c, sh = rng.choice([600, 2000]), (1,)
        elif kind == "img":
            c, sh = rng.choice([1, 3]), (rng.choice([24, 40, 64]), rng.choice([24, 56]))
        elif kind == "vol":
            c, sh = rng.choice([1, 3]), (rng.choice([4, 6]), rng.choice([12, 16]), rng.choice([12, 16]))
        elif kind == "bag":
            c, sh = rng.choice([300, 768]), (rng.choice([300, 512, 1200]),)
        else:
            c, sh = rng.choice([5, 24]), (rng.choice([700, 2500]),)
        chans.append(c); axes.append(len(sh)); shapes.append(sh)
    kw = dict(n_modalities=M, channel_dims=chans, num_spatial_axes=axes, out_dims=4, depth=rng.choice([1, 2]),
              l_c=rng.choice([64, 128, 256]), l_d=rng.choice([64, 128]), x_heads=8, l_heads=rng.choice([4, 8]),
              cross_dim_head=64, latent_dim_head=rng.choice([32, 64]), num_freq_bands=2, max_freq=10.0, snn=True,
              weight_tie_layers=False, self_per_cross_attn=1, fourier_encode_data=True, final_classifier_head=True)
    b = rng.choice([4, 9, 20, 33, 48])
    if any(s[0] >= 1200 for s in shapes if len(s) == 1) or any(c >= 768 for c in chans):
        b = min(b, 9)
    return kw, shapes, b, False


def random_case_chain(rng):
    """Shapes the fused latent chain (chain.hip) accepts -- l_d = 128, l_c a multiple of 16, heads * dim_head a multiple of 128
    with dim_head in {16, 32, 64, 128} -- mixed with ones it must leave to the per-block launches, over every head kind of a
    chain (attention out-projection, one-token broadcast add) and projection kind (rank-D: Q only; explicit / self: scaled Q
    (+ K|V); nothing in front of a one-token block), with key masks, GELU / SELU, missing modalities, no self block."""
    small = random_case(rng)
    kw, shapes, b, masked = small
    inner = rng.choice([128, 256, 512])
    dh = rng.choice([16, 32, 64, 128])
    li = rng.choice([128, 256, 384, 512])
    ldh = rng.choice([32, 64, 128])
    kw.update(l_d=128, l_c=rng.choice([16, 32, 48, 128]), x_heads=max(1, inner // dh), cross_dim_head=dh,
              l_heads=max(1, li // ldh), latent_dim_head=ldh)
    if rng.random() < 0.2:      # not eligible: odd head dim -> the unfused route inside an otherwise eligible model
        kw.update(cross_dim_head=24, x_heads=4)
    return kw, shapes, min(b, 9), masked


def random_case(rng):
    M = rng.choice([1, 2, 2, 3])
    chans, axes, shapes = [], [], []
    masked = rng.random() < 0.25
    n_tokens = None
    for _ in range(M):
        kind = rng.choice(["tab", "tab_wide", "seq", "img", "vol", "bag"])
        if masked:       # one mask for every modality: equal token counts
            kind = rng.choice(["seq", "bag"])
        if kind == "tab":
            c, sh = rng.choice([3, 6, 20, 40]), (1,)
        elif kind == "tab_wide":
            c, sh = rng.choice([600, 2000]), (1,)
        elif kind == "seq":
            c, sh = rng.choice([1, 2, 5, 9, 24]), (rng.choice([2, 7, 33, 200]),)
        elif kind == "img":
            c, sh = rng.choice([1, 3, 4]), (rng.choice([3, 9, 17]), rng.choice([4, 8, 21]))
        elif kind == "vol":
            c, sh = rng.choice([1, 3]), (rng.choice([2, 3]), rng.choice([3, 5]), rng.choice([4, 6]))
        else:
            c, sh = rng.choice([70, 96, 130]), (rng.choice([5, 64, 300]),)
        if masked:
            if n_tokens is None:
                n_tokens = sh[0]
            sh = (n_tokens,)
        chans.append(c); axes.append(len(sh)); shapes.append(sh)
    l_d = rng.choice([16, 32, 64, 128])
    heads = rng.choice([1, 2, 4, 8])
    kw = dict(n_modalities=M, channel_dims=chans, num_spatial_axes=axes, out_dims=rng.choice([2, 4]), depth=rng.choice([1, 2, 3]),
              l_c=rng.choice([4, 16, 24, 128]), l_d=l_d, x_heads=heads, l_heads=rng.choice([1, 2, 4]),
              cross_dim_head=rng.choice([4, 8, 16, 32, 64]), latent_dim_head=rng.choice([8, 16, 64]),
              num_freq_bands=rng.choice([1, 2, 2, 4]), max_freq=rng.choice([2.0, 10.0]), snn=rng.random() < 0.7,
              weight_tie_layers=rng.random() < 0.2, self_per_cross_attn=rng.choice([0, 1, 1, 1]),
              fourier_encode_data=rng.random() < 0.85, final_classifier_head=rng.random() < 0.85)
    b = rng.choice([1, 2, 3, 5, 33, 40])
    if max(max(s) for s in shapes) >= 200 or max(chans) >= 600:
        b = min(b, 5)
    return kw, shapes, b, masked


def random_case_staged(rng):
    """Shapes outside the latent chains' own that FIT them after zero padding (staged models) -- odd latent widths
    up to 128, any latent count, odd head widths with heads * padded width <= 512 -- over the small case's modality mixes."""
    kw, shapes, b, masked = random_case(rng)

    def heads_dim():
        dh = rng.choice([5, 11, 16, 20, 27, 31, 48, 63, 64, 103, 127])
        dhp = 16 if dh <= 16 else 32 if dh <= 32 else 64 if dh <= 64 else 128
        return rng.choice([h for h in (1, 2, 3, 4, 8) if h * dhp <= 512]), dh
    xh, xd = heads_dim()
    lh, ld = heads_dim()
    kw.update(l_d=rng.choice([9, 30, 62, 65, 100, 119, 126, 128]), l_c=rng.choice([3, 8, 16, 17, 25, 40]), x_heads=xh, cross_dim_head=xd,
              l_heads=lh, latent_dim_head=ld)
    return kw, shapes, min(b, 9), masked


def random_case_tuned(rng):
    """The reference's tuned regime (config/best_hyperparams.yml): ONE or two narrow cross heads of odd width over a long patch bag
    (>= 2048 context rows per batch: the narrow LDS-DMA projection and weight-gradient kernels of gemm_nt.hip, the skinny GEMV's 8-row
    form, the one-element split merge) beside a wide one-token modality, staged latent sizes."""
    def heads_dim():
        dh = rng.choice([16, 27, 48, 63, 64, 103])
        dhp = 16 if dh <= 16 else 32 if dh <= 32 else 64 if dh <= 64 else 128
        return rng.choice([h for h in (1, 1, 2) if h * dhp <= 256]), dh
    xh, xd = heads_dim()
    lh, ld = heads_dim()
    b = rng.choice([2, 5, 8])
    n_bag = rng.choice([600, 1100, 2500]) if b > 2 else rng.choice([1100, 2500, 4200])
    chans, axes, shapes = [rng.choice([600, 2000]), rng.choice([127, 296, 768])], [1, 1], [(1,), (n_bag,)]
    if rng.random() < 0.3:
        chans, axes, shapes = chans[::-1], axes[::-1], shapes[::-1]
    kw = dict(n_modalities=2, channel_dims=chans, num_spatial_axes=axes, out_dims=4, depth=rng.choice([1, 2]),
              l_c=rng.choice([16, 17, 25]), l_d=rng.choice([62, 65, 119, 126]), x_heads=xh, l_heads=lh,
              cross_dim_head=xd, latent_dim_head=ld, num_freq_bands=2, max_freq=10.0, snn=rng.random() < 0.7,
              weight_tie_layers=False, self_per_cross_attn=rng.choice([0, 1]), fourier_encode_data=True, final_classifier_head=True)
    return kw, shapes, b, False


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backward", action="store_true")
    ap.add_argument("--data-seed", type=int, default=0, help="offset of the input / weight seeds (same configurations, other numbers)")
    ap.add_argument("--dropout", action="store_true", help="training mode with random attention / feed-forward dropout; the oracle replays the build's exported Philox masks")
    ap.add_argument("--attn", action="store_true", help="also compare get_attention_weights() / get_attention_importance() of the inference forward (untied weights, nothing missing)")
    ap.add_argument("--scale", default="small", choices=["small", "medium", "chain", "staged", "tuned"])
    ap.add_argument("--core-precision", default="fp32", choices=["fp32", "bf16", "bf16x3"], help="attention core of the inference forward")
    ap.add_argument("--only", type=int, nargs="*", default=None, help="case indices to run (the others are generated and skipped)")
    args = ap.parse_args(argv)
    rng = random.Random(args.seed)
    worst = 0.0
    bad = 0
    for case in range(args.n):
        kw, shapes, b, masked = (random_case_medium(rng) if args.scale == "medium" else random_case_chain(rng) if args.scale == "chain"
                                 else random_case_staged(rng) if args.scale == "staged" else random_case_tuned(rng) if args.scale == "tuned"
                                 else random_case(rng))
        if args.dropout:
            kw["attn_dropout"] = rng.choice([0.0, 0.1, 0.3])
            kw["ff_dropout"] = rng.choice([0.0, 0.2]) if kw["attn_dropout"] > 0 else 0.2
        missing_draw = rng.random(), rng.random()
        verbose = rng.random() < 0.15                  # the reference's `continue` quirk: skipped modalities also skip the self block
        embeddings = rng.random() < 0.15      # drawn for every case so that --only reproduces the same sequence
        if args.dropout:
            verbose = False                            # (the mask numbering helper of the tests follows the plain schedule)
        if args.only is not None and case not in args.only:
            continue
        torch.manual_seed(1000 + case + 7919 * args.data_seed)
        try:
            model = hn.HealNet(**kw, core_precision=args.core_precision).eval()
        except Exception as e:      # invalid combination for the reference constructor as well
            print(f"[{case}] skipped at construction: {type(e).__name__}: {e}")
            continue
        gen = torch.Generator().manual_seed(case + 104729 * args.data_seed)
        ins = [torch.rand(b, *s, c, generator=gen) for s, c in zip(shapes, kw["channel_dims"])]
        # narrow transports: some modalities arrive as bf16 (read in place) or uint8 (byte / 255 in the encode kernel); the
        # oracle gets the same values widened to fp32
        dev_ins = list(ins)
        for i in range(len(ins)):
            u = rng.random()
            if u < 0.12:
                dev_ins[i] = ins[i].to(torch.bfloat16)
                ins[i] = dev_ins[i].float()
            elif u < 0.20:
                dev_ins[i] = (ins[i] * 255).round().to(torch.uint8)
                ins[i] = dev_ins[i].float().div(255)
        missing = None
        if kw["n_modalities"] > 1 and missing_draw[0] < 0.2:
            missing = int(missing_draw[1] * kw["n_modalities"])
            ins[missing] = None
            dev_ins[missing] = None
        mask = None
        if masked and ins[0] is not None:
            n = ins[0].shape[1]
            mask = torch.rand(b, n, generator=gen) > 0.3
            mask[:, 0] = True
        # tied parameters (weight_tie_layers) stay ONE leaf tensor under all their keys, as in the reference's module tree
        leaves = {}
        sd = {}
        for k, v in model.state_dict().items():
            if v.data_ptr() not in leaves:
                leaves[v.data_ptr()] = v.detach().clone().requires_grad_(args.backward)
            sd[k] = leaves[v.data_ptr()]
        cfg = O.FusionConfig(**kw)
        # smallest |pre-activation| over every LeakyReLU of the oracle forward: below ~1e-5 two correct fp32 forwards can
        # disagree on its sign and the gradients through that element differ by O(1) -- such cases are
        # reported as "kink", not as failures
        margins = []
        orig_leaky = O.F.leaky_relu
        def recording_leaky(x, *a, **k):
            margins.append(float(x.detach().abs().min()))
            return orig_leaky(x, *a, **k)
        O.F.leaky_relu = recording_leaky
        try:
            with torch.set_grad_enabled(args.backward):
                want = O.fusion_forward(sd, cfg, ins, mask=mask, verbose=verbose, return_embeddings=embeddings)
        finally:
            O.F.leaky_relu = orig_leaky
        margin = min(margins) if margins else 1.0
        model.to(DEV)
        dins = [None if t is None else t.to(DEV) for t in dev_ins]
        dmask = None if mask is None else mask.to(DEV)
        if args.dropout:
            # training mode: draw masks on the device, export them, replay them in the oracle
            import importlib.util
            spec = importlib.util.spec_from_file_location("tgd", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_gpu_dropout.py"))
            tgd = importlib.util.module_from_spec(spec); sys.path.insert(0, os.path.dirname(spec.origin)); spec.loader.exec_module(tgd)
            model.train()
            got = model(list(dins), mask=dmask, verbose=verbose, return_embeddings=embeddings)
            seed_, offset_ = model._last_rng
            n_tok = [1 if t is None else int(t.numel() // (t.shape[0] * t.shape[-1])) for t in ins]
            drop = tgd._oracle_masks(hn, model, kw, b, n_tok, seed_, offset_, [t is not None for t in ins])
            with torch.set_grad_enabled(args.backward):
                want = O.fusion_forward(sd, cfg, ins, mask=mask, drop=drop, verbose=verbose, return_embeddings=embeddings)
            e_inf = e_tape = rel(got.detach(), want.detach())
        else:
            with torch.no_grad():
                e_inf = rel(model(list(dins), mask=dmask, verbose=verbose, return_embeddings=embeddings), want.detach())
                check_attn = args.attn and missing is None and not kw["weight_tie_layers"] and not verbose
                ref_p = None
                if check_attn:
                    tr = O.FusionTrace()
                    O.fusion_forward({k: v.detach() for k, v in sd.items()}, cfg, ins, mask=mask, trace=tr, return_embeddings=embeddings)
                    ref_p = O.attention_weights_in_module_order(tr, cfg)

                def attn_error():
                    got_p = [p for p in model.get_attention_weights() if p is not None]
                    got_i = [p for p in model.get_attention_importance() if p is not None]
                    assert len(ref_p) == len(got_p) == len(got_i), (len(ref_p), len(got_p), len(got_i))
                    e = 0.0
                    for rp, gp, gi in zip(ref_p, got_p, got_i):
                        ok_rows = torch.isfinite(rp).all(dim=-1, keepdim=True)       # fully masked rows are NaN on both sides
                        e = max(e, rel(torch.where(ok_rows.to(gp.device), gp, torch.zeros_like(gp)), torch.where(ok_rows, rp, torch.zeros_like(rp))))
                        e = max(e, rel(torch.nan_to_num(gi), torch.nan_to_num(rp).mean(dim=1)))
                    return 0.4 * e        # tolerance 5e-4 on the probabilities against 2e-4 on the logits

                if check_attn:
                    e_inf = max(e_inf, attn_error())
            got = model(list(dins), mask=dmask, verbose=verbose, return_embeddings=embeddings)
            e_tape = rel(got.detach(), want.detach())
            if check_attn:
                with torch.no_grad():
                    e_tape = max(e_tape, attn_error())      # after a taping forward: views of the tape
        e_grad = 0.0
        if args.backward:
            dl = torch.randn(want.shape, generator=gen)
            (want * dl).sum().backward()
            (got * dl.to(DEV)).sum().backward()
            refs = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k, _ in model.named_parameters()}
            gmax = max(float(r.abs().max()) for r in refs.values())
            for k, p in model.named_parameters():       # (named_parameters lists a tied parameter once, under its first key)
                ref = refs[k]
                # per-parameter max-norm error; gradients that vanish analytically (|ref| ~ 1e-6 of the largest one) are
                # measured against the model's gradient scale, not against their own rounding noise
                scale = max(float(ref.abs().max()), 1e-3 * gmax, 1e-12)
                e_k = float((p.grad.cpu().double() - ref.double()).abs().max()) / scale
                if args.only is not None and e_k > 5e-3:
                    print(f"      {k}: rel err {e_k:.2e} (|ref| max {float(ref.abs().max()):.2e}, model max {gmax:.2e})")
                e_grad = max(e_grad, e_k)
        tol_inf = {"fp32": 2e-4, "bf16": 2e-2, "bf16x3": 1e-3}[args.core_precision]
        fwd_ok = e_inf <= tol_inf and e_tape <= 2e-4
        flag = "" if fwd_ok and e_grad <= 5e-3 else ("   (kink: min |pre| %.1e)" % margin if fwd_ok and margin < 2e-5 else "   <<<<<< FAIL")
        bad += flag.endswith("FAIL")
        worst = max(worst, e_inf, e_tape)
        print(f"[{case}] M={kw['n_modalities']} ch={kw['channel_dims']} shapes={shapes} b={b} l=({kw['l_c']},{kw['l_d']}) h={kw['x_heads']}x{kw['cross_dim_head']} "
              f"depth={kw['depth']} tie={int(kw['weight_tie_layers'])} self={kw['self_per_cross_attn']} bands={kw['num_freq_bands']} mask={masked} missing={missing} verbose={int(verbose)} emb={int(embeddings)}: inference {e_inf:.1e} taping {e_tape:.1e}"
              + (f" grad {e_grad:.1e}" if args.backward else "") + flag, flush=True)
    print(f"worst forward error {worst:.2e}; {bad} failing case(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
