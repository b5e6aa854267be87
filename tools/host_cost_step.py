#!/usr/bin/env python3
"""What the HOST spends per cfg4 training step: the same step at a toy size (b = 1, 64 patches: the GPU work vanishes, the ~120
launches and the Python / autograd path stay) with a cProfile of it, then at the real size with the enqueue time beside the wall
time.  Round 6: 2.2 ms of host time against 4.06 ms of GPU time per step -- the step is GPU-bound with a 1.8 x margin.
    python tools/host_cost_step.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import healnet_amd as hn
from healnet_amd import dist as hdist
dev = torch.device("cuda", 0)
kw = dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4)
torch.manual_seed(0)
model = hn.HealNet(**kw).train().to(dev)
for b, n in ((1, 64), (8, 4096)):
    gen = torch.Generator().manual_seed(1)
    ins = [torch.rand(b, 1, 2000, generator=gen).to(dev), torch.rand(b, n, 768, generator=gen).to(dev)]
    y = torch.randint(0, 4, (b,), generator=gen).to(dev); c = torch.randint(0, 2, (b,), generator=gen).to(dev)
    flat = hn.train.flatten_parameters(model)
    opt = hn.train.FusedL1Adam(flat, lr=1e-4, l1=1e-4)
    sync = hdist.GradReadyAllReduce(model, flat)
    def step():
        opt.zero_grad()
        out = hn.train.surv_nll_loss(model(list(ins)), y, c)
        out.loss.backward(); sync.wait(); opt.step()
    for _ in range(10): step()
    torch.cuda.synchronize()
    import cProfile, pstats
    t0 = time.perf_counter()
    for _ in range(100): step()
    host = (time.perf_counter() - t0) / 100
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 100
    print(f"b={b} N={n}: host enqueue {host*1e3:.3f} ms/step, wall {tot*1e3:.3f} ms/step")
    if b == 1:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(50): step()
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    sync.close()
