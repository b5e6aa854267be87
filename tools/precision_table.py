#!/usr/bin/env python3
"""Max-norm error of the fused forward against the reference-generated fixtures (tests/golden/g6_*), per core precision.

    python tools/precision_table.py            # prints one line per (config, precision)
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import healnet_amd as hn
from oracle import healnet_cpu as O          # closed-form weights / inputs of the fixtures (checker-side data only)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
man = json.load(open(os.path.join(ROOT, "tests/golden/manifest.json")))
with torch.no_grad():
    for name in ["cfg1", "cfg3s", "cfg4", "tuned"]:
        m = man["g6_" + name]
        g = dict(np.load(os.path.join(ROOT, "tests/golden/g6_%s.npz" % name)))
        cfg = O.FusionConfig(**m["kwargs"])
        for prec in ["fp32", "bf16x3", "bf16"]:
            model = hn.HealNet(**m["kwargs"], core_precision=prec).eval()
            model.load_state_dict(O.filler_state_dict(cfg, gain=m["gain"]), strict=True)
            model.to("cuda:0")
            ins = [O.filler_input(s, 20 + i).to("cuda:0") for i, s in enumerate(m["shapes"])]
            y = model(list(ins)).cpu().numpy()
            e = model(list(ins), return_embeddings=True).cpu().numpy()
            print(json.dumps(dict(config=name, core_precision=prec,
                                  logits_maxnorm_err=float(abs(y - g["logits"]).max() / abs(g["logits"]).max()),
                                  emb_maxnorm_err=float(abs(e - g["emb"]).max() / abs(g["emb"]).max()))), flush=True)
