#!/usr/bin/env python3
"""A/B the image cross-attention block (rank-D path) under the development knobs of the attention core.
Each variant runs in its own process (the knobs are read once)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
import healnet_amd as hn
b = 32
torch.manual_seed(0)
blk = hn.PreNorm(128, hn.Attention(128, 13, heads=8, dim_head=64), context_dim=13).to("cuda:0")
x = torch.randn(b, 128, 128, device="cuda:0")
img = torch.rand(b, 224, 224, 3, device="cuda:0")
ctx = hn.fourier_encode_concat(img)
for _ in range(3): blk(x, context=ctx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(5):
    e0.record()
    for _ in range(5): y = blk(x, context=ctx)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
print("RESULT", min(ts), sorted(ts)[len(ts)//2], float(y.abs().sum()))
''' % ROOT
variants = [dict(), dict(HN_CORE_NQ="8"), dict(HN_CORE_NQ="2"), dict(HN_CORE_WAVES="8192"), dict(HN_CORE_WAVES="2048"),
            dict(HN_CORE_NQ="8", HN_CORE_WAVES="2048"), dict(HN_CORE_NQ="8", HN_CORE_WAVES="3072"), dict(HN_CORE_NQ="2", HN_CORE_WAVES="8192")]
if len(sys.argv) > 1:
    variants = [json.loads(a) for a in sys.argv[1:]]
for v in variants:
    env = dict(os.environ); env.update(v)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print(v, line[0] if line else out.stderr[-300:], flush=True)
