#!/usr/bin/env python3
"""GPU: the layer-chain route (lchain.hip) against the reference's committed logits and against the per-block route.

    python tools/layer_chain_check.py [b ...]          # default 32; HN_NO_SELF_IN_CHAIN=1 in the environment selects the per-block route
Prints, per batch size: max-norm relative error against tests/golden/g10_cfg2_b32_bench.npz (the REFERENCE on bench.py's inputs),
ms per forward (CUDA events over 50 forwards), and writes the logits to gpurun_out/layer_chain_<tag>_b<b>.pt for an A/B diff."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fullsize_fixtures as F  # noqa: E402
import healnet_amd  # noqa: E402

tag = "perblock" if os.environ.get("HN_NO_SELF_IN_CHAIN") else "layer"
want = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "g10_cfg2_b32_bench.npz"))["logits"])
torch.manual_seed(0)
model = healnet_amd.HealNet(**F.CFG2).eval().to("cuda:0")
ins = F.bench_inputs(32)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for b in [int(a) for a in sys.argv[1:]] or [32]:
    rep = (b + 31) // 32
    x = [torch.cat([t] * rep)[:b].to("cuda:0") for t in ins]
    w = torch.cat([want] * rep)[:b]
    with torch.no_grad():
        y = model(x)
        torch.cuda.synchronize()
        err = float((y.cpu() - w).abs().max() / w.abs().max())
        for _ in range(5):
            model(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            model(x)
        e1.record()
        torch.cuda.synchronize()
        y2 = model(x)
        # the latent self-attention probabilities rebuilt from the kept statistics + trace (hn_attn_probs): rows must sum to 1
        pw = [model.layers[l][-1][0].fn.attn_weights for l in range(model.depth)]
        psum = max(float((w.sum(-1) - 1).abs().max()) for w in pw)
    same = bool(torch.equal(y, y2))
    print(f"{tag} b={b}: rel err vs reference {err:.3e}  finite {bool(torch.isfinite(y).all())}  repeatable {same}  self-attn row sums off by {psum:.2e}  "
          f"{e0.elapsed_time(e1) / 50:.4f} ms/forward  cluster {healnet_amd._capi.cluster_status(0)}", flush=True)
    torch.save({"y": y.cpu(), "p": [w[:16].cpu() for w in pw]}, os.path.join(ROOT, "gpurun_out", f"layer_chain_{tag}_b{b}.pt"))
