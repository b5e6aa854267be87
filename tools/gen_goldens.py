#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Runs only in the build container (needs /root/reference, imported read-only by file path:
healnet/models/healnet.py depends on torch + einops only).  The reference never travels:
what is committed is data (inputs, weights, expected outputs) plus this script.

While generating, every fixture is also replayed through oracle/healnet_cpu.py and the
deviation is printed / asserted, which is what pins the oracle.

    python tools/gen_goldens.py            # rewrites tests/golden/*.npz + manifest.json
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF_FILE = "/root/reference/healnet/models/healnet.py"

from oracle import healnet_cpu as O  # noqa: E402


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_healnet", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def np_(t):
    return t.detach().cpu().numpy()


MANIFEST = {}


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: (np_(v) if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    MANIFEST[name] = {"bytes": os.path.getsize(path), "keys": sorted(arrays.keys())}


# ------------------------------------------------------------------------------------------
def g1_fourier(ref):
    out = {}
    for S in (1, 3, 224):
        for (mf, nb) in ((10.0, 2), (2.0, 2), (10.0, 4)):
            pos = torch.linspace(-1.0, 1.0, S)[:, None]
            r = ref.fourier_encode(pos, mf, nb)
            o = O.fourier_features(pos, mf, nb)
            assert torch.equal(r, o), ("fourier", S, mf, nb)
            out[f"S{S}_mf{int(mf)}_nb{nb}"] = r
    save("g1_fourier", **out)


def g2_preprocess(ref):
    out = {}
    shapes = {"tab": (2, 1, 20), "img": (2, 6, 5, 3), "vol": (2, 3, 4, 5, 2), "one": (1, 1, 1, 4)}
    for salt, (name, shp) in enumerate(shapes.items()):
        data = O.filler_input(shp, salt)
        axes = len(shp) - 2
        model = ref.HealNet(n_modalities=1, channel_dims=[shp[-1]], num_spatial_axes=[axes], out_dims=2,
                            l_c=4, l_d=8, x_heads=1, l_heads=1, cross_dim_head=4, latent_dim_head=4, depth=1)
        lst = [data.clone()]
        model(lst)                      # the reference writes the encoded context back into the list (:222)
        enc_ref = lst[0]
        enc_o = O.encode_modality(data, 2, 10.0, True)
        assert torch.equal(enc_ref, enc_o), name
        out[name + "_in"] = data
        out[name + "_enc"] = enc_ref
    save("g2_preprocess", **out)


def g3_attention(ref):
    cases = [
        dict(name="cross_default_small", qd=16, cd=13, heads=8, dh=64, L=12, N=50, b=2, mask=False),
        dict(name="cross_odd", qd=17, cd=13, heads=2, dh=27, L=9, N=37, b=3, mask=False),
        dict(name="cross_masked", qd=16, cd=21, heads=2, dh=8, L=5, N=19, b=2, mask=True),
        dict(name="cross_one_token", qd=16, cd=45, heads=8, dh=16, L=8, N=1, b=2, mask=False),
        dict(name="self_h1", qd=24, cd=None, heads=1, dh=16, L=10, N=None, b=2, mask=False),
        dict(name="self_default", qd=32, cd=None, heads=8, dh=64, L=16, N=None, b=2, mask=False),
    ]
    out = {}
    meta = {}
    for i, c in enumerate(cases):
        torch.manual_seed(100 + i)
        att = ref.Attention(c["qd"], c["cd"], heads=c["heads"], dim_head=c["dh"])
        # sharpen the weights so softmax is far from uniform
        with torch.no_grad():
            att.to_q.weight.mul_(3.0)
            att.to_kv.weight.mul_(3.0)
        x = torch.randn(c["b"], c["L"], c["qd"])
        ctx = torch.randn(c["b"], c["N"], c["cd"]) if c["cd"] is not None else None
        mask = None
        if c["mask"]:
            mask = torch.rand(c["b"], c["N"]) > 0.3
            mask[:, 0] = True
        y = att(x, context=ctx, mask=mask)
        p = att.attn_weights
        yo, po = O.attention(x, ctx, att.to_q.weight, att.to_kv.weight, att.to_out[0].weight, att.to_out[0].bias,
                             c["heads"], mask, return_weights=True)
        assert rel_err(yo, y) < 2e-6 and rel_err(po, p) < 2e-6, (c["name"], rel_err(yo, y), rel_err(po, p))
        n = c["name"]
        out[n + "_x"] = x
        if ctx is not None:
            out[n + "_ctx"] = ctx
        if mask is not None:
            out[n + "_mask"] = mask
        out[n + "_wq"] = att.to_q.weight
        out[n + "_wkv"] = att.to_kv.weight
        out[n + "_wo"] = att.to_out[0].weight
        out[n + "_bo"] = att.to_out[0].bias
        out[n + "_y"] = y
        out[n + "_p"] = p
        meta[n] = {k: v for k, v in c.items() if k != "name"}
    save("g3_attention", **out)
    MANIFEST["g3_attention"]["cases"] = meta


def g4_feedforward(ref):
    out = {}
    for snn in (True, False):
        torch.manual_seed(7)
        ffn = ref.FeedForward(16, snn=snn)
        x = torch.randn(3, 5, 16) * 2.0
        y = ffn(x)
        yo = O.feed_forward(x, ffn.net[0].weight, ffn.net[0].bias, ffn.net[2].weight, ffn.net[2].bias, snn)
        assert rel_err(yo, y) < 2e-6
        tag = "selu" if snn else "gelu"
        out.update({f"{tag}_x": x, f"{tag}_w1": ffn.net[0].weight, f"{tag}_b1": ffn.net[0].bias,
                    f"{tag}_w2": ffn.net[2].weight, f"{tag}_b2": ffn.net[2].bias, f"{tag}_y": y})
    save("g4_feedforward", **out)


TINY = dict(l_c=8, l_d=16, x_heads=2, l_heads=2, cross_dim_head=4, latent_dim_head=4)


def tiny_inputs(M, b=2):
    shapes = [(b, 1, 20), (b, 6, 5, 3), (b, 3, 4, 5, 2)]
    return [O.filler_input(s, 10 + i) for i, s in enumerate(shapes[:M])]


def g5_tiny_models(ref):
    """Whole-model tiny configurations with full state_dict, per-block traces, attention weights, grads."""
    variants = [
        dict(name="m1_d1", M=1, depth=1),
        dict(name="m2_d3", M=2, depth=3),
        dict(name="m3_d3", M=3, depth=3),
        dict(name="m2_d3_tied", M=2, depth=3, weight_tie_layers=True),
        dict(name="m2_d2_noself", M=2, depth=2, self_per_cross_attn=0),
        dict(name="m2_d2_nofourier", M=2, depth=2, fourier_encode_data=False),
        dict(name="m2_d2_gelu", M=2, depth=2, snn=False),
        dict(name="m2_d2_nohead", M=2, depth=2, final_classifier_head=False),
        dict(name="m2_d2_bands4", M=2, depth=2, num_freq_bands=4, max_freq=2.0),
    ]
    cds_all, axes_all = [20, 3, 2], [1, 2, 3]
    for vi, v in enumerate(variants):
        v = dict(v)
        name, M = v.pop("name"), v.pop("M")
        kw = dict(n_modalities=M, channel_dims=cds_all[:M], num_spatial_axes=axes_all[:M], out_dims=3, **TINY, **v)
        torch.manual_seed(1000 + vi)
        model = ref.HealNet(**kw)
        with torch.no_grad():                      # sharpen attention so softmax is not ~uniform
            for k, p in model.named_parameters():
                if k.endswith("to_q.weight") or k.endswith("to_kv.weight"):
                    p.mul_(2.5)
        sd = {k: t.detach().clone() for k, t in model.state_dict().items()}
        cfg = O.FusionConfig(**kw)
        ins = tiny_inputs(M)
        arrays = {"sd::" + k: t for k, t in sd.items()}
        for i, t in enumerate(ins):
            arrays[f"in{i}"] = t

        # plain forward, embeddings, attention weights, block trace via hooks
        blocks = []
        hooks = []
        for L, layer in enumerate(model.layers):
            mods = list(layer[:-1]) + list(layer[-1])
            for mod in mods:
                hooks.append(mod.register_forward_hook(lambda m_, i_, o_: blocks.append(o_.detach().clone())))
        logits = model([t.clone() for t in ins])
        for h in hooks:
            h.remove()
        attn = [a.detach().clone() for a in model.get_attention_weights()]
        emb = model([t.clone() for t in ins], return_embeddings=True)
        tr = O.FusionTrace()
        lo = O.fusion_forward(sd, cfg, ins, trace=tr)
        eo = O.fusion_forward(sd, cfg, ins, return_embeddings=True)
        assert rel_err(lo, logits) < 5e-6 and rel_err(eo, emb) < 5e-6, (name, rel_err(lo, logits))
        arrays["logits"], arrays["emb"] = logits, emb
        for i, a in enumerate(attn):
            arrays[f"attn{i}"] = a
        MANIFEST.setdefault("_notes", {})[name + "_n_attn"] = len(attn)

        # gradients of logits.sum() (or emb.sum()) w.r.t. every parameter
        model.zero_grad()
        out = model([t.clone() for t in ins])
        (out * O.filler_input(out.shape, 77)).sum().backward()
        for k, p in model.named_parameters():
            arrays["grad::" + k] = p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)

        # missing-modality behaviour (Appendix B-1) and shorter lists (B-7)
        if M >= 2:
            ins_missing = [ins[0], None] + ins[2:]
            y_quiet = model([None if t is None else t.clone() for t in ins_missing])
            y_verbose = model([None if t is None else t.clone() for t in ins_missing], verbose=True)
            arrays["logits_missing1"] = y_quiet
            arrays["logits_missing1_verbose"] = y_verbose
            oq = O.fusion_forward(sd, cfg, ins_missing)
            ov = O.fusion_forward(sd, cfg, ins_missing, verbose=True)
            assert rel_err(oq, y_quiet) < 5e-6 and rel_err(ov, y_verbose) < 5e-6, name
            if M == 2:
                y_short = model([ins[0].clone()])
                assert rel_err(y_short, y_quiet) < 1e-6
                ins_m0 = [None, ins[1]]
                y_m0 = model([None, ins[1].clone()])
                arrays["logits_missing0"] = y_m0
                assert rel_err(O.fusion_forward(sd, cfg, ins_m0), y_m0) < 5e-6
        save("g5_" + name, **arrays)
        MANIFEST["g5_" + name]["kwargs"] = {k: (list(x) if isinstance(x, (list, tuple)) else x) for k, x in kw.items()}

    # masked whole-model case: both modalities must share N (Appendix B-5)
    kw = dict(n_modalities=2, channel_dims=[6, 5], num_spatial_axes=[1, 1], out_dims=3, depth=2, **TINY)
    torch.manual_seed(4242)
    model = ref.HealNet(**kw)
    sd = {k: t.detach().clone() for k, t in model.state_dict().items()}
    ins = [O.filler_input((2, 11, 6), 3), O.filler_input((2, 11, 5), 4)]
    mask = torch.tensor([[1, 1, 0, 1, 1, 1, 0, 0, 1, 1, 1], [1, 0, 1, 1, 0, 1, 1, 1, 1, 0, 1]], dtype=torch.bool)
    y = model([t.clone() for t in ins], mask=mask)
    yo = O.fusion_forward(sd, O.FusionConfig(**kw), ins, mask=mask)
    assert rel_err(yo, y) < 5e-6
    arrays = {"sd::" + k: t for k, t in sd.items()}
    arrays.update(in0=ins[0], in1=ins[1], mask=mask, logits=y)
    save("g5_m2_d2_masked", **arrays)
    MANIFEST["g5_m2_d2_masked"]["kwargs"] = {k: (list(x) if isinstance(x, (list, tuple)) else x) for k, x in kw.items()}


def g6_default_size(ref):
    """Default hyper-parameters at the BASELINE configs' shapes with closed-form weights/inputs;
    only outputs are stored (weights are regenerated from oracle.filler_* on the test side)."""
    configs = {
        "cfg1": dict(kw=dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4),
                     shapes=[(4, 1, 2000), (4, 224, 224, 3)]),
        "cfg3s": dict(kw=dict(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4),
                      shapes=[(1, 1, 2000), (1, 224, 224, 3), (1, 4, 56, 56, 3)]),
        "cfg4": dict(kw=dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4),
                     shapes=[(2, 1, 2000), (2, 4096, 768)]),
        "tuned": dict(kw=dict(n_modalities=2, channel_dims=[300, 96], num_spatial_axes=[1, 1], out_dims=4,
                              depth=2, l_c=25, l_d=119, x_heads=1, l_heads=4, cross_dim_head=63,
                              latent_dim_head=27, self_per_cross_attn=0, max_freq=2.0),
                      shapes=[(3, 1, 300), (3, 500, 96)]),
    }
    for name, c in configs.items():
        kw = c["kw"]
        cfg = O.FusionConfig(**kw)
        model = ref.HealNet(**kw).eval()
        sd = O.filler_state_dict(cfg, gain=2.0 if name != "cfg4" else 1.0)
        missing, unexpected = model.load_state_dict(sd, strict=True), None
        ins = [O.filler_input(s, 20 + i) for i, s in enumerate(c["shapes"])]
        t0 = time.time()
        with torch.no_grad():
            logits = model([t.clone() for t in ins])
            emb = model([t.clone() for t in ins], return_embeddings=True)
            attn = model.get_attention_weights()
            # latent-mean of attention rows for the largest-N cross block of layer 0 (explainer usage)
            big = max(range(kw["n_modalities"]), key=lambda i: attn[i].shape[-1])
            attn_mean = attn[big].mean(dim=1)[:, :4096]
            lo = O.fusion_forward(sd, cfg, ins)
        dt = time.time() - t0
        e = rel_err(lo, logits)
        print(f"[g6] {name}: ref fwd x2 {dt:.1f}s  oracle-vs-ref rel {e:.2e}  logits[0]={np_(logits[0])}")
        assert e < 2e-5, (name, e)
        save("g6_" + name, logits=logits, emb=emb, attn_mean=attn_mean, attn_mean_index=np.int64(big))
        MANIFEST["g6_" + name]["kwargs"] = {k: (list(x) if isinstance(x, (list, tuple)) else x) for k, x in kw.items()}
        MANIFEST["g6_" + name]["shapes"] = [list(s) for s in c["shapes"]]
        MANIFEST["g6_" + name]["gain"] = 2.0 if name != "cfg4" else 1.0
        MANIFEST["g6_" + name]["oracle_rel"] = e


def kat0_seed_route(ref):
    """torch.manual_seed(0) default-init model + torch.rand inputs and the per-key
    checksums that pin the RNG consumption order of the constructor (a1)."""
    torch.manual_seed(0)
    kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
    model = ref.HealNet(**kw).eval()
    tab = torch.rand(4, 1, 2000)
    img = torch.rand(4, 224, 224, 3)
    with torch.no_grad():
        logits = model([tab.clone(), img.clone()])
        emb = model([tab.clone(), img.clone()], return_embeddings=True)
    sd = model.state_dict()
    lo = O.fusion_forward({k: v for k, v in sd.items()}, O.FusionConfig(**kw), [tab, img])
    print("[kat0] logits row0", np_(logits[0]), " oracle rel", rel_err(lo, logits))
    assert rel_err(lo, logits) < 2e-5
    keys = sorted(sd.keys())
    sums = np.array([float(sd[k].double().sum()) for k in keys])
    abssums = np.array([float(sd[k].double().abs().sum()) for k in keys])
    save("kat0", logits=logits, emb_mean=emb.mean(), emb_absmax=emb.abs().max(), emb_row0=emb[0, :4, :8],
         key_sums=sums, key_abssums=abssums)
    MANIFEST["kat0"]["state_keys"] = keys
    MANIFEST["kat0"]["kwargs"] = kw

    # tied + 3-modality constructor checksums (sharing map, Appendix B-4)
    for tag, kw2 in {"tied": dict(n_modalities=2, channel_dims=[20, 3], num_spatial_axes=[1, 2], out_dims=3,
                                  weight_tie_layers=True, **TINY),
                     "m3": dict(n_modalities=3, channel_dims=[20, 3, 2], num_spatial_axes=[1, 2, 3], out_dims=3, **TINY),
                     "noself": dict(n_modalities=2, channel_dims=[20, 3], num_spatial_axes=[1, 2], out_dims=3,
                                    self_per_cross_attn=0, **TINY)}.items():
        torch.manual_seed(5)
        m = ref.HealNet(**kw2)
        sd2 = m.state_dict()
        keys2 = sorted(sd2.keys())
        ptr_groups = {}
        for k in keys2:
            ptr_groups.setdefault(sd2[k].data_ptr(), []).append(k)
        save("kat_init_" + tag, key_sums=np.array([float(sd2[k].double().sum()) for k in keys2]),
             key_abssums=np.array([float(sd2[k].double().abs().sum()) for k in keys2]))
        MANIFEST["kat_init_" + tag]["state_keys"] = keys2
        MANIFEST["kat_init_" + tag]["kwargs"] = kw2
        MANIFEST["kat_init_" + tag]["shared_groups"] = [g for g in ptr_groups.values() if len(g) > 1]
        MANIFEST["kat_init_" + tag]["n_params"] = int(sum(p.numel() for p in m.parameters()))


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    ref = load_reference()
    g1_fourier(ref)
    g2_preprocess(ref)
    g3_attention(ref)
    g4_feedforward(ref)
    g5_tiny_models(ref)
    kat0_seed_route(ref)
    g6_default_size(ref)
    MANIFEST["_torch"] = torch.__version__
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(MANIFEST, f, indent=1, sort_keys=True)
    total = sum(v.get("bytes", 0) for v in MANIFEST.values() if isinstance(v, dict))
    print(f"wrote {len(MANIFEST)} entries, {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
