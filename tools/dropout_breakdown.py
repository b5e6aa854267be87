#!/usr/bin/env python3
"""cfg2 b=32 training step with a given (attn_dropout, ff_dropout): run under rocprofv3 --kernel-trace --stats to see where dropout
costs (profiles/r03_*_dropout_breakdown.txt).   python tools/dropout_breakdown.py 0.25 0.25"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import healnet_amd as hn
kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
pa, pf = float(sys.argv[1]), float(sys.argv[2])
torch.manual_seed(0)
m = hn.HealNet(**kw, attn_dropout=pa, ff_dropout=pf).train().cuda()
flat = hn.train.flatten_parameters(m)
ins = [torch.rand(32, 1, 2000, device="cuda"), torch.rand(32, 224, 224, 3, device="cuda")]


def step():
    flat.zero_grad()
    m(list(ins)).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.time()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"dropout=({pa},{pf}) fwd+bwd {(time.time() - t) / 10 * 1e3:.2f} ms")
