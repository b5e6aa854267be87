#!/usr/bin/env python3
"""Times hn.latent_block (chain -> self-attention core -> chain; host-bound when called in a loop) and the cfg2 forward.  HN_LIB_PATH
selects another build of the library: this is how the variants in the header of healnet_amd/csrc/chain.hip were compared (e.g. a copy
of chain.hip with the weight loads / the LDS stores / the MFMAs compiled out, linked against the other objects under healnet_amd/build/)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
torch.set_grad_enabled(False)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to("cuda:0")
att, ff = m.layers[0][-1][0], m.layers[0][-1][1]
x = torch.randn(b, 128, 128, device="cuda:0")
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t_blk = timeit(lambda: hn.latent_block(att, ff, x))
m.keep_attention_stats = False
tab, img = torch.rand(b, 1, 2000, device="cuda:0"), torch.rand(b, 224, 224, 3, device="cuda:0")
t_fwd = timeit(lambda: m([tab, img]), 50)
print(f"{os.environ.get('HN_LIB_PATH', 'default'):40s} b={b} latent_block {t_blk:7.1f} us   forward {t_fwd:8.1f} us")
