#!/usr/bin/env python3
"""Where the HOST time of a cfg4 training step goes: per-step enqueue time (no synchronisation) against the synchronised step time,
and a cProfile of 20 steps (forward thread only; the backward runs on the autograd engine's thread and shows up as time inside
`backward`)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn

dev = torch.device("cuda", 0)
kw = dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4)
torch.manual_seed(0)
model = hn.HealNet(**kw).train().to(dev)
gen = torch.Generator().manual_seed(1)
ins = [torch.rand(8, 1, 2000, generator=gen).to(dev), torch.rand(8, 4096, 768, generator=gen).to(dev)]
y = torch.randint(0, 4, (8,), generator=gen).to(dev)
c = torch.randint(0, 2, (8,), generator=gen).to(dev)
flat = hn.train.flatten_parameters(model)
opt = hn.train.FusedL1Adam(flat, lr=1e-4, l1=1e-4)


def step():
    opt.zero_grad()
    out = hn.train.surv_nll_loss(model(list(ins)), y, c)
    out.loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"enqueue {t_enq / n * 1e3:.3f} ms/step   synchronised {t_all / n * 1e3:.3f} ms/step")
# forward and backward host time separately, GPU idle in between
tf = tb = 0.0
for _ in range(10):
    opt.zero_grad(); torch.cuda.synchronize()
    t = time.perf_counter(); out = hn.train.surv_nll_loss(model(list(ins)), y, c); tf += time.perf_counter() - t
    torch.cuda.synchronize()
    t = time.perf_counter(); out.loss.backward(); tb += time.perf_counter() - t
    torch.cuda.synchronize()
print(f"host: forward+loss call {tf / 10 * 1e3:.3f} ms, backward call {tb / 10 * 1e3:.3f} ms (GPU idle at entry)")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:40]))
