#!/usr/bin/env python3
"""Does splitting the batch over two HIP streams (two concurrent forwards of b/2) beat one forward of b?
The latent-side kernels are latency-bound and leave most of the chip idle; the image core fills it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to("cuda:0")
torch.set_grad_enabled(False)
tab, img = torch.rand(b, 1, 2000, device="cuda:0"), torch.rand(b, 224, 224, 3, device="cuda:0")
def one(n=50):
    for _ in range(5): m([tab, img])
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): y = m([tab, img])
    torch.cuda.synchronize(); return (time.time() - t) / n, y
def two(n=50, parts=2):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    chunks = [(tab[i::parts].contiguous(), img[i::parts].contiguous()) for i in range(parts)]
    def step():
        outs = []
        for s, (t_, i_) in zip(streams, chunks):
            with torch.cuda.stream(s):
                outs.append(m([t_, i_]))
        return outs
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): outs = step()
    torch.cuda.synchronize(); return (time.time() - t) / n, outs
t1, y = one()
print(f"one stream  b={b}: {t1*1e3:.3f} ms  {b/t1:.0f} samples/s")
for parts in (2, 4):
    t2, outs = two(parts=parts)
    ok = all(torch.allclose(outs[i], y[i::parts], rtol=1e-4, atol=1e-5) for i in range(parts))
    print(f"{parts} streams  b={b}/{parts}: {t2*1e3:.3f} ms  {b/t2:.0f} samples/s  outputs match: {ok}")
