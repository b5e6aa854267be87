#!/usr/bin/env python3
"""Training-step timing: forward + backward (+ bucketed RCCL gradient all-reduce when launched with torchrun) +
Adam step, on synthetic data.  `--config cfg2|cfg4`.

    python tools/train_step.py --config cfg4 --steps 20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_step.py --config cfg4
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
from healnet_amd import dist as hd

CFG = {
    "cfg2": (dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4), [(1, 2000), (224, 224, 3)], 32),
    "cfg4": (dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4), [(1, 2000), (4096, 768)], 8),
}
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg4")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--batch", type=int, default=0)
args = ap.parse_args()
rank, world, local = hd.init_from_env("nccl")
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
kw, shapes, b = CFG[args.config]
b = args.batch or b
torch.manual_seed(0)
model = hn.HealNet(**kw).train().to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
gen = torch.Generator().manual_seed(1234 + rank)
ins = [torch.rand(b, *s, generator=gen).to(dev) for s in shapes]
target = torch.randn(b, kw["out_dims"], generator=gen).to(dev)

def step():
    opt.zero_grad(set_to_none=True)
    out = model(list(ins))
    loss = ((out - target) ** 2).mean()
    loss.backward()
    hd.allreduce_mean_([p.grad for p in model.parameters() if p.grad is not None])
    opt.step()
    return loss

for _ in range(args.warmup):
    step()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize(dev)
dt = hd.max_over_ranks((time.perf_counter() - t0) / args.steps, dev)
# forward-only / backward-only split
torch.cuda.synchronize(dev); t1 = time.perf_counter()
for _ in range(args.steps):
    out = model(list(ins))
torch.cuda.synchronize(dev); tf = (time.perf_counter() - t1) / args.steps
if rank == 0:
    print(json.dumps({"config": args.config, "batch_per_gpu": b, "n_gpus": world, "ms_per_step": round(dt * 1e3, 3),
                      "samples_per_s": round(b * world / dt, 1), "fwd_train_ms": round(tf * 1e3, 3), "loss": float(loss)}))
