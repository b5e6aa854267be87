#!/usr/bin/env python3
"""Training-step timing on synthetic data, `--config cfg2|cfg4`: the reference's loop body (healnet/main.py:425-467) --
forward, survival NLL, backward, L1 regulariser over all parameters, Adam under OneCycleLR -- plus the data-parallel
gradient average when launched with torchrun.

  --tail fused   (default) hn_surv_nll + gradients accumulated into one flat buffer + ONE all-reduce + hn_l1_adam_step
  --tail torch   the same step written with stock PyTorch ops (torch loss, l1 * sum|p| through autograd, bucketed
                 all-reduce, torch.optim.Adam): what the reference's loop would run on top of the fused forward/backward

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
ap.add_argument("--tail", choices=["fused", "torch"], default="fused")
ap.add_argument("--l1", type=float, default=1e-4)
args = ap.parse_args()
rank, world, local = hd.init_from_env("nccl")
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
kw, shapes, b = CFG[args.config]
b = args.batch or b
torch.manual_seed(0)
model = hn.HealNet(**kw).train().to(dev)
gen = torch.Generator().manual_seed(1234 + rank)
ins = [torch.rand(b, *s, generator=gen).to(dev) for s in shapes]
y_disc = torch.randint(0, kw["out_dims"], (b,), generator=gen).to(dev)
censorship = torch.randint(0, 2, (b,), generator=gen).to(dev)
total_steps = args.steps + args.warmup + 1

if args.tail == "fused":
    flat = hn.train.flatten_parameters(model)
    opt = hn.train.FusedL1Adam(flat, lr=1e-4, l1=args.l1)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total_steps)

    def step():
        opt.zero_grad()
        out = hn.train.surv_nll_loss(model(list(ins)), y_disc, censorship)
        out.loss.backward()
        hd.allreduce_mean_([flat.grads])
        opt.step()
        sched.step()
        return out.loss
else:
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total_steps)

    def step():
        opt.zero_grad(set_to_none=True)
        logits = model(list(ins))
        hazards = torch.sigmoid(logits)
        surv = torch.cumprod(1 - hazards, dim=1)
        Y, c = y_disc.view(b, 1), censorship.view(b, 1).float()
        Sp = torch.cat([torch.ones_like(c), surv], 1)
        unc = -(1 - c) * (torch.log(torch.gather(Sp, 1, Y).clamp(min=1e-7)) + torch.log(torch.gather(hazards, 1, Y).clamp(min=1e-7)))
        cen = -c * torch.log(torch.gather(Sp, 1, Y + 1).clamp(min=1e-7))
        loss = (0.6 * (cen + unc) + 0.4 * unc).mean()
        reg = args.l1 * sum(p.abs().sum() for p in model.parameters())
        (loss + reg).backward()
        hd.allreduce_mean_([p.grad for p in model.parameters() if p.grad is not None])
        opt.step()
        sched.step()
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
    print(json.dumps({"config": args.config, "tail": args.tail, "batch_per_gpu": b, "n_gpus": world, "ms_per_step": round(dt * 1e3, 3),
                      "samples_per_s": round(b * world / dt, 1), "fwd_train_ms": round(tf * 1e3, 3), "loss": float(loss)}))
