#!/usr/bin/env python3
"""Where does the host time of one forward go?  (call time without sync vs GPU time; hipGraph replay)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to("cuda:0")
m.keep_attention_stats = False
tab, img = torch.rand(b, 1, 2000, device="cuda:0"), torch.rand(b, 224, 224, 3, device="cuda:0")
for _ in range(3): m([tab, img])
torch.cuda.synchronize()
n = 20
t = time.perf_counter()
for _ in range(n): m([tab, img])
t_call = (time.perf_counter() - t) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t) / n
print(f"b={b}: host enqueue {t_call*1e3:.3f} ms/forward, wall {t_all*1e3:.3f} ms/forward")
# descriptor cost alone
t = time.perf_counter()
for _ in range(100): m._descriptor()
print(f"descriptor build {(time.perf_counter()-t)/100*1e6:.0f} us")
# graph capture
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): out = m([tab, img])
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g):
        out = m([tab, img])
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    print(f"graph replay wall {(time.perf_counter()-t)/n*1e3:.3f} ms/forward; out finite {bool(torch.isfinite(out).all())}")
    ref = m([tab, img]); torch.cuda.synchronize()
    print("graph vs eager equal:", bool(torch.equal(ref, out)))
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])
