#!/usr/bin/env python3
"""Cost of keep_attention_stats (softmax statistics + block inputs kept for Attention.attn_weights): forward time at cfg2 b=32
with the flag off / on, alternating.  Measured: 3.365 vs 3.373 ms (the blocks write both in place, no copies)."""
import sys, time, torch
sys.path.insert(0, '.')
import healnet_amd as hn
torch.manual_seed(0)
m = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to("cuda:0")
tab, img = torch.rand(32, 1, 2000, device="cuda:0"), torch.rand(32, 224, 224, 3, device="cuda:0")
with torch.no_grad():
    for keep in (False, True, False, True):
        m.keep_attention_stats = keep
        for _ in range(5): m([tab, img])
        torch.cuda.synchronize(); t = time.time()
        for _ in range(50): m([tab, img])
        torch.cuda.synchronize(); print(keep, (time.time() - t) / 50 * 1e3, "ms")
