#!/usr/bin/env python3
"""Where the HOST time of a tuned-shape training step goes (development): cProfile of 200 steps of the blca config, and the
host-only time per step (launch calls return without waiting for the GPU)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
TUNED = {"blca": dict(depth=2, l_c=25, l_d=119, cross_dim_head=16, latent_dim_head=127, attn_dropout=0.0830, ff_dropout=0.4733),
         "kirp": dict(depth=5, l_c=17, l_d=62, cross_dim_head=27, latent_dim_head=113, attn_dropout=0.3179, ff_dropout=0.0474)}
kw = TUNED[sys.argv[1] if len(sys.argv) > 1 else "blca"]
extra = dict(x_heads=1, l_heads=8, self_per_cross_attn=0, num_freq_bands=2, max_freq=2.0)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = hn.HealNet(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4, **kw, **extra).to(dev).train()
flat = hn.train.flatten_parameters(model)
gen = torch.Generator().manual_seed(1)
ins = [torch.rand(8, 1, 2000, generator=gen).to(dev), torch.rand(8, 4096, 768, generator=gen).to(dev)]


def step():
    flat.zero_grad()
    model(list(ins)).sum().backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
host = []
for _ in range(50):
    torch.cuda.synchronize()
    t = time.perf_counter()
    step()
    host.append(time.perf_counter() - t)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(100):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / 100
host.sort()
print(f"host-only per step (median) {host[len(host) // 2] * 1e3:.3f} ms, wall per step {wall * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(14)
