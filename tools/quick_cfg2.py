#!/usr/bin/env python3
"""cfg2 forward loop for profiling (rocprofv3 --kernel-trace --stats -- python tools/quick_cfg2.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
m = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4,
               core_precision=os.environ.get("HN_QUICK_PRECISION", "fp32")).eval().to("cuda:0")
m.keep_attention_stats = False
torch.set_grad_enabled(False)     # the inference forward (hn_fusion_forward), not the tape-recording one
tab, img = torch.rand(b, 1, 2000, device="cuda:0"), torch.rand(b, 224, 224, 3, device="cuda:0")
for _ in range(3): m([tab, img])
torch.cuda.synchronize(); t = time.time()
for _ in range(steps): m([tab, img])
torch.cuda.synchronize(); dt = (time.time() - t) / steps
print(f"b={b} {dt*1e3:.3f} ms/forward {b/dt:.0f} samples/s")
