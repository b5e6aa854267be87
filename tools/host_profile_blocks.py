#!/usr/bin/env python3
"""Host profile of the block-by-block training route (healnet_amd.dist.context_parallel_forward under autograd, one rank):
where the time of a step goes on the host.   python tools/host_profile_blocks.py [--depth 12]"""
import argparse, cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
from healnet_amd import dist as hd

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=12)
ap.add_argument("--world", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = hn.HealNet(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4).eval().to(dev)
gen = torch.Generator().manual_seed(1234)
ins = [torch.rand(1, 1, 2000, generator=gen).to(dev), torch.rand(1, 224, 224, 3, generator=gen).to(dev),
       torch.rand(1, args.depth, 224, 224, 3, generator=gen).to(dev)]
G = args.world


def fake_gather(o, st):
    return o.unsqueeze(0).expand(G, *o.shape).contiguous(), st.unsqueeze(0).expand(G, *st.shape).contiguous()


def step():
    model.zero_grad(set_to_none=True)
    hd.context_parallel_forward(model, ins, rank=0, world=G, gather=fake_gather, reduce=lambda ts: None).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    step()
host = (time.perf_counter() - t) / 10 * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t) / 10 * 1e3
print(f"step: host {host:.2f} ms, with the GPU drained {total:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:60]))
