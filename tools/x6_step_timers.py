#!/usr/bin/env python3
"""cfg4 training step (bench.py's record) with hipEvent pairs around the patch-bag GEMM classes of BOTH routes, in the live loop
(no profiler): what the fp32-exact bf16 route (gemm_x6.hip) costs per launch inside the step, beside the step time itself.
    python tools/x6_step_timers.py            # HN_NO_X6_GEMM=1 for the fp32-MFMA route"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import healnet_amd as hn
from healnet_amd import dist as hdist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.manual_seed(0)
model = hn.HealNet(**bench.TRAIN_KW).train().to(dev)
gen = torch.Generator().manual_seed(4321)
b = int(os.environ.get("X6_BATCH", bench.TRAIN_BATCH))
ins = [torch.rand(b, *s, generator=gen).to(dev) for s in bench.TRAIN_SHAPES]
y = torch.randint(0, bench.TRAIN_KW["out_dims"], (b,), generator=gen).to(dev)
c = torch.randint(0, 2, (b,), generator=gen).to(dev)
flat = hn.train.flatten_parameters(model)
opt = hn.train.FusedL1Adam(flat, lr=1e-4, l1=1e-4)
sync = hdist.GradReadyAllReduce(model, flat)
sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=4000) if os.environ.get("X6_SCHED") else None
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def step():
    opt.zero_grad()
    out = hn.train.surv_nll_loss(model(list(ins)), y, c)
    out.loss.backward()
    sync.wait()
    opt.step()
    if sched is not None:
        sched.step()
    return out.loss


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
names = ["gemm_nt_x6", "gemm_tn_x6", "x6_split", "x6_split_t", "x6_tn_reduce", "gemm_nt_glds", "gemm_tn_glds"]
with bench.KernelTimers(names, 4 * steps + 8) as kt:
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
avg = kt.averages_ms()
print(json.dumps({"route": "fp32-mfma" if os.environ.get("HN_NO_X6_GEMM") else "x6", "ms_per_step": round(dt * 1e3, 4), "host_enqueue_ms_per_step": round(host * 1e3, 4),
                  "per_launch_us": {k: (round(v[0] * 1e3, 1) if v[0] is not None else None, v[1] // steps) for k, v in avg.items()}}))

# ---- the same loop as ONE HIP graph per gradient half (no host enqueue), and from a secondary thread (bench.py's watchdog runs the
# record in one): where the host stops keeping up with a 4.1 ms step
sync.close()
gstep = hn.train.GraphedStep(model, lambda logits, yy, cc: hn.train.surv_nll_loss(logits, yy, cc).loss, list(ins), (y, c))
for _ in range(5):
    gstep(ins, (y, c)); opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    gstep(ins, (y, c)); opt.step()
torch.cuda.synchronize()
print(json.dumps({"graphed_ms_per_step": round((time.perf_counter() - t0) / steps * 1e3, 4)}))
gstep.close()
import threading
sync = hdist.GradReadyAllReduce(model, flat)
res = {}
def worker():
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    res["thread_ms_per_step"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
th = threading.Thread(target=worker)
th.start()
while th.is_alive():
    th.join(0.05)
print(json.dumps(res))
