#!/usr/bin/env python3
"""PCIe-inclusive forward throughput at cfg2: host batches -> DeviceLoader -> model, against the
HBM-resident rate bench.py reports.  `--transport fp32|bf16|uint8` (uint8: the image as 8-bit pixels).

    python tools/bench_staging.py --steps 50
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to(dev)
model.keep_attention_stats = False
gen = torch.Generator().manual_seed(1234)
b = args.batch
pool = [([torch.rand(b, 1, 2000, generator=gen), torch.rand(b, 224, 224, 3, generator=gen)],) for _ in range(4)]
pool8 = [([p[0][0], (p[0][1] * 255).round().to(torch.uint8)],) for p in pool]
pool16 = [([t.to(torch.bfloat16) for t in p[0]],) for p in pool]      # a dataset that is stored in bf16


def run(batches, transport, staged=True):
    seq = [batches[i % len(batches)] for i in range(args.steps + 5)]
    with torch.no_grad():
        it = hn.etl.DeviceLoader(seq, dev, depth=2, transport=transport) if staged else ([(  [f.to(dev) for f in s[0]],) for s in [x]][0] for x in seq)
        n = 0
        for (features,) in it:
            if n == 5:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            model(features)
            n += 1
        torch.cuda.synchronize()
    return b * args.steps / (time.perf_counter() - t0)


res = {"resident": None}
with torch.no_grad():
    ins = [t.to(dev) for t in pool[0][0]]
    for _ in range(5): model(ins)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps): model(ins)
    torch.cuda.synchronize(); res["resident"] = round(b * args.steps / (time.perf_counter() - t0), 1)
res["blocking_to_device_fp32"] = round(run(pool, None, staged=False), 1)
res["device_loader_fp32"] = round(run(pool, None), 1)
res["device_loader_bf16_stored"] = round(run(pool16, None), 1)           # half the PCIe bytes, K1 reads bf16 in place
res["device_loader_bf16_cast_on_host"] = round(run(pool, "bf16"), 1)     # fp32 dataset cast per batch by the producer thread: host-bound
res["device_loader_uint8_image"] = round(run(pool8, None), 1)
res["unit"] = "samples/s, cfg2 b=%d, 1 GPU" % b
print(json.dumps(res))
