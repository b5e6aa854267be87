#!/usr/bin/env python3
"""How do the latent-side GEMMs scale with the number of rows?  (fixed per-kernel latency vs throughput)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
torch.manual_seed(0)
blk = hn.PreNorm(128, hn.FeedForward(128, snn=True)).to("cuda:0")
att = hn.PreNorm(128, hn.Attention(128, heads=8, dim_head=64)).to("cuda:0")
for rows_b in (2, 8, 32, 128):
    x = torch.randn(rows_b, 128, 128, device="cuda:0")
    with torch.no_grad():
        for _ in range(5): blk(x); att(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): blk(x)
        e1.record(); torch.cuda.synchronize()
        t_ff = e0.elapsed_time(e1) / 20
        e0.record()
        for _ in range(20): att(x)
        e1.record(); torch.cuda.synchronize()
        t_at = e0.elapsed_time(e1) / 20
    print(f"rows={rows_b*128:6d}  ff block {t_ff*1e3:7.1f} us   self-attn block {t_at*1e3:7.1f} us")
