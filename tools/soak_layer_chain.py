#!/usr/bin/env python3
"""Soak of the layer-chain route: N back-to-back cfg2 b = 32 forwards (the bench's loop), every output checked for NaN on the device,
the cluster counters at the end -- a lost exchange (co-residency) would show as a NaN row + `lost` > 0.   python tools/soak_layer_chain.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
from healnet_amd import _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
torch.manual_seed(0)
m = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to("cuda:0")
tab, img = torch.rand(32, 1, 2000, device="cuda:0"), torch.rand(32, 224, 224, 3, device="cuda:0")
bad = torch.zeros((), dtype=torch.int64, device="cuda:0")
with torch.no_grad():
    ref = m([tab, img]).clone()
    t0 = time.time()
    for i in range(n):
        y = m([tab, img])
        bad += (~torch.isfinite(y)).any().long() + (y != ref).any().long()
    torch.cuda.synchronize()
dt = time.time() - t0
print(f"{n} forwards in {dt:.1f} s ({dt / n * 1e3:.3f} ms each incl. the checks): {int(bad)} outputs differed from the first / held NaN; cluster {_capi.cluster_status(0)}")
sys.exit(1 if int(bad) else 0)
