#!/usr/bin/env python3
"""Full-size expectations from the REFERENCE itself: the logits -- and, compressed, the gradients -- that
/root/reference/healnet/models/healnet.py produces at the BASELINE sizes, so that the GPU box compares the HIP path with committed
reference outputs instead of re-running 0.6 GB-per-sample oracle forwards and autograd passes (which were 40 % of the GPU suite's
wall clock, all of it host time).

Runs only in the build container (imports the reference by file path; nothing of it is copied).  Writes tests/golden/g7_*.npz:

  g10_cfg2_b32_bench     logits (32, 4) of the seed-0 default model on bench.py's own inputs (the driver's headline workload)
  g10_cfg2_b32_train     logits + compressed per-parameter gradients of sum(logits * dl), seed-43 model, b = 32, full 224 x 224 image
  g10_cfg5_cut_b2        logits of the depth-8 four-modality model, volume cut to 4 x 224 x 224, b = 2 -- all present / bag 2 missing
  g10_cfg5_full_b1       the same model at the config's full sizes (volume 12 x 224 x 224), b = 1
  g10_cfg3_b16_s0        fp32 reference logits of sample 0 of the b = 16 bf16-rounded cfg3 inputs

Each case also checks, here, that the seeded healnet_amd.HealNet constructor yields the reference's state dict bit for bit (the
fixtures then need no weights) and that oracle/healnet_cpu.py reproduces the reference's logits (the oracle stays pinned at these
sizes too).   python tools/gen_goldens_fullsize.py [case ...]"""
from __future__ import annotations

import importlib.util
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_FILE = "/root/reference/healnet/models/healnet.py"

import fullsize_fixtures as F  # noqa: E402
from oracle import healnet_cpu as O  # noqa: E402


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_healnet", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"  wrote {name}.npz ({os.path.getsize(path)} bytes)", flush=True)


def ref_model(ref, kw, seed, train=False):
    """The reference model under `seed`, and the proof that healnet_amd's constructor consumes the generator the same way."""
    import healnet_amd
    torch.manual_seed(seed)
    model = ref.HealNet(**kw)
    torch.manual_seed(seed)
    mine = healnet_amd.HealNet(**kw)
    sd_r, sd_m = model.state_dict(), mine.state_dict()
    assert list(sd_r.keys()) == list(sd_m.keys())
    for k in sd_r:
        assert torch.equal(sd_r[k], sd_m[k]), k
    return model.train() if train else model.eval()


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def per_sample(model, ins, fn=None):
    """The reference forward sample by sample (nothing couples samples; the materialised scores are 0.6 GB per sample and layer)."""
    outs = []
    b = next(t.shape[0] for t in ins if t is not None)
    for i in range(b):
        out = model([None if t is None else t[i:i + 1].clone() for t in ins])
        if fn is not None:
            fn(i, out)
        outs.append(out.detach())
    return torch.cat(outs)


def cfg2_b32_bench(ref):
    model = ref_model(ref, F.CFG2, 0)
    ins = F.bench_inputs(32)
    with torch.no_grad():
        want = per_sample(model, ins)
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        orc = O.fusion_forward(sd, O.FusionConfig(**F.CFG2), [t[:2] for t in ins])
    print("  oracle vs reference (2 samples):", rel_err(orc, want[:2]))
    assert rel_err(orc, want[:2]) < 1e-6
    save("g10_cfg2_b32_bench", logits=want)


def cfg2_b32_train(ref):
    model = ref_model(ref, F.CFG2, 43, train=True)
    ins, dl = F.cfg2_train_inputs(32)
    t0 = time.time()

    def back(i, out):
        (out * dl[i:i + 1]).sum().backward()
        if i % 4 == 0:
            print(f"    sample {i}: {time.time() - t0:.0f} s", flush=True)

    want = per_sample(model, ins, back)
    arrays = {"logits": want}
    keys = []
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        for name, v in F.compress_grad(g, k).items():
            arrays[f"{k}::{name}"] = v
        keys.append(k)
    arrays["keys"] = np.array(keys)
    save("g10_cfg2_b32_train", **arrays)


def cfg5(ref):
    model = ref_model(ref, F.CFG5, 51)
    ins = F.cfg5_cut_inputs()
    with torch.no_grad():
        want = per_sample(model, ins)
        want_m = per_sample(model, [ins[0], ins[1], None, ins[3]])
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        orc = O.fusion_forward(sd, O.FusionConfig(**F.CFG5), [t[:1] for t in ins])
    print("  cfg5 cut: oracle vs reference (sample 0):", rel_err(orc, want[:1]))
    assert rel_err(orc, want[:1]) < 1e-6
    save("g10_cfg5_cut_b2", logits=want, logits_bag2_missing=want_m)
    model = ref_model(ref, F.CFG5, 52)
    with torch.no_grad():
        want = model([t.clone() for t in F.cfg5_full_inputs()])
    save("g10_cfg5_full_b1", logits=want)


def cfg3(ref):
    model = ref_model(ref, F.CFG3, 0)
    ins = F.cfg3_inputs(16)
    with torch.no_grad():
        want = model([t[:1].float() for t in ins])
    save("g10_cfg3_b16_s0", logits=want)


CASES = {"cfg2_b32_bench": cfg2_b32_bench, "cfg2_b32_train": cfg2_b32_train, "cfg5": cfg5, "cfg3": cfg3}

if __name__ == "__main__":
    ref = load_reference()
    for name in (sys.argv[1:] or list(CASES)):
        t0 = time.time()
        print(name, flush=True)
        CASES[name](ref)
        print(f"  {time.time() - t0:.0f} s", flush=True)
