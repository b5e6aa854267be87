#!/usr/bin/env python3
"""Compute side of the context split (healnet_amd.dist.context_parallel_forward, SURVEY.md 8(e) second axis) on ONE GPU: the
work of one rank out of G for a b = 1 forward of a cfg3-shaped model (tab 1 x 2000 + image 224 x 224 x 3 + volume
12 x 224 x 224 x 3), G = 1, 2, 4, 8.  The all-gather is replaced by a stand-in that repeats the rank's own partials G times (same
sizes, same merge work, no communication) -- this measures what a rank computes, NOT a multi-GPU run: the exchange is one
all-gather of b * l_c * (inner + 2 * heads) floats per rank and split cross block (270 KB per sample), whose cost on xGMI is unmeasured.

    python tools/bench_context_split.py [--batch 1] [--json out.json]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn
from healnet_amd import dist as hd

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--json", default="")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--depth", type=int, default=12, help="slices of the volume modality (cfg3: 12)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
b = args.batch
torch.manual_seed(0)
model = hn.HealNet(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4).eval().to(dev)
gen = torch.Generator().manual_seed(1234)
ins = [torch.rand(b, 1, 2000, generator=gen).to(dev), torch.rand(b, 224, 224, 3, generator=gen).to(dev),
       torch.rand(b, args.depth, 224, 224, 3, generator=gen).to(dev)]


def timeit(fn, n=args.steps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    fused = timeit(lambda: model(ins))
    ref = model(ins)
rows = [dict(path="fused forward (hn_fusion_forward, one GPU holds the whole context)", ranks=1, ms=round(fused, 3))]
for G in (1, 2, 4, 8):
    gathered = {}

    def fake_gather(o, st, G=G):
        gathered["floats_per_rank"] = o.numel() + st.numel()
        return o.unsqueeze(0).expand(G, *o.shape).contiguous(), st.unsqueeze(0).expand(G, *st.shape).contiguous()

    with torch.no_grad():
        ms = timeit(lambda: hd.context_parallel_forward(model, ins, rank=0, world=G, gather=fake_gather))
    row = dict(path="block-by-block, rank 0 of G (stand-in gather)", ranks=G, ms=round(ms, 3),
               gathered_kb_per_rank_and_block=round(gathered.get("floats_per_rank", 0) * 4 / 1024, 1),
               speedup_vs_one_rank=None)
    rows.append(row)
    # the fused route (hn_fusion_forward_cp: latent chains and all, one C call, the exchange as a callback)
    ms = timeit(lambda: hd.context_parallel_forward(model, ins, rank=0, world=G, fused=True,
                                                    gather_flat=lambda lo, pa, G=G: pa.copy_(lo.repeat(G))))
    rows.append(dict(path="fused (hn_fusion_forward_cp), rank 0 of G (stand-in gather)", ranks=G, ms=round(ms, 3), speedup_vs_one_rank=None))

    # TRAINING (ABI v11): forward + backward of rank 0 of G under autograd, block by block; the all-reduce of the backward is a no-op
    # stand-in (same launches, no communication)
    def train_step(G=G):
        model.zero_grad(set_to_none=True)
        hd.context_parallel_forward(model, ins, rank=0, world=G, gather=fake_gather, reduce=lambda ts: None).sum().backward()

    rows.append(dict(path="training step, block-by-block, rank 0 of G (stand-in gather / reduce)", ranks=G, ms=round(timeit(train_step, n=10), 3),
                     speedup_vs_one_rank=None))
def plain_train():
    model.zero_grad(set_to_none=True)
    model(ins).sum().backward()


rows.insert(1, dict(path="plain training step (hn_fusion_forward_train + hn_fusion_backward, one GPU holds the whole context)", ranks=1,
                    ms=round(timeit(plain_train, n=10), 3)))
for kind in ("block", "fused (", "training step, block"):
    one = next(r["ms"] for r in rows if r["ranks"] == 1 and r["path"].startswith(kind))
    for r in rows:
        if r["path"].startswith(kind):
            r["speedup_vs_one_rank"] = round(one / r["ms"], 2)
# G = 1 of the block-by-block path must reproduce the fused forward
with torch.no_grad():
    got = hd.context_parallel_forward(model, ins, rank=0, world=1, fused=False)
    got_f = hd.context_parallel_forward(model, ins, rank=0, world=1, fused=True)
err = float((got - ref).abs().max() / ref.abs().max())
err_f = float((got_f - ref).abs().max() / ref.abs().max())
doc = dict(workload=f"cfg3-shaped model, b = {b}: tab (b,1,2000) + img (b,224,224,3) + vol (b,{args.depth},224,224,3), fp32, eval", rows=rows,
           block_by_block_vs_fused_rel_err=err, fused_cp_one_part_vs_fused_rel_err=err_f,
           note="compute of ONE rank on one GPU; no communication is measured (stand-in gather); no multi-GPU number is claimed")
for r in rows:
    print(r)
print("rel err block-by-block vs fused:", err, " hn_fusion_forward_cp (one part) vs fused:", err_f)
if args.json:
    with open(args.json, "w") as f:
        json.dump(doc, f, indent=1)
