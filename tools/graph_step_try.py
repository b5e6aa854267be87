#!/usr/bin/env python3
"""Can one training step (zero_grad + tape forward + survival NLL + fused backward) be captured into a HIP graph as it is?"""
import os, sys, time
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


def fwd_bwd():
    flat.zero_grad()
    out = hn.train.surv_nll_loss(model(list(ins)), y, c)
    out.loss.backward()
    return out.loss


loss_e = fwd_bwd().detach().clone()
g_e = flat.grads.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fwd_bwd()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss_g = fwd_bwd()
flat.grads.zero_()
graph.replay()
torch.cuda.synchronize()
print("loss eager", float(loss_e), "graph", float(loss_g), "grad max diff", float((flat.grads - g_e).abs().max()), "grad scale", float(g_e.abs().max()))
for name, fn in (("eager", fwd_bwd), ("graph", graph.replay)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: enqueue {(t1 - t0) / 40 * 1e3:.3f} ms  synchronised {(t2 - t0) / 40 * 1e3:.3f} ms per fwd+bwd")
