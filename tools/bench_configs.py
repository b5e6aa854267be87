#!/usr/bin/env python3
"""Forward timing of the BASELINE.json configs on one GPU (the headline bench.py line is cfg2 only).

    python tools/bench_configs.py [--cfg 2 3 4 5] [--core-precision fp32 bf16] [--steps 10] [--json out.json]

cfg2  tab (b,1,2000) + img (b,224,224,3), b=32                            fp32 tensors
cfg3  + vol (b,12,224,224,3), b=16                                         bf16 tensors (configs[2])
cfg4  omic (b,1,2000) + WSI bag (b,4096,768), b=8                          fp32 tensors
cfg5  tab + 2x WSI bag + vol, depth=8, b=4 (the per-GPU share of b=32/8)    fp32 tensors
For every (cfg, precision) prints ms/forward and samples/s; with both precisions given also the max-norm
difference of the bf16-core logits from the fp32-core logits on the same inputs."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn

DEV = "cuda:0"
CFG = {
    2: dict(kw=dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4), b=32,
            shapes=[(1, 2000), (224, 224, 3)], dtype=torch.float32),
    3: dict(kw=dict(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4), b=16,
            shapes=[(1, 2000), (224, 224, 3), (12, 224, 224, 3)], dtype=torch.bfloat16),
    4: dict(kw=dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4), b=8,
            shapes=[(1, 2000), (4096, 768)], dtype=torch.float32),
    5: dict(kw=dict(n_modalities=4, channel_dims=[2000, 768, 768, 3], num_spatial_axes=[1, 1, 1, 3], out_dims=4, depth=8), b=4,
            shapes=[(1, 2000), (4096, 768), (4096, 768), (12, 224, 224, 3)], dtype=torch.float32),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, nargs="+", default=[2, 3, 4, 5])
    ap.add_argument("--core-precision", nargs="+", default=["fp32", "bf16"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    rows = []
    for c in args.cfg:
        spec = CFG[c]
        gen = torch.Generator().manual_seed(1234)
        ins = [torch.rand(spec["b"], *s, generator=gen).to(spec["dtype"]).to(DEV) for s in spec["shapes"]]
        outs = {}
        for prec in args.core_precision:
            torch.manual_seed(0)
            m = hn.HealNet(**spec["kw"], core_precision=prec).eval().to(DEV)
            m.keep_attention_stats = False
            with torch.no_grad():
                for _ in range(3):
                    y = m(list(ins))
                torch.cuda.synchronize()
                t = time.time()
                for _ in range(args.steps):
                    y = m(list(ins))
                torch.cuda.synchronize()
            dt = (time.time() - t) / args.steps
            outs[prec] = y.float()
            row = dict(cfg=c, core_precision=prec, tensors=str(spec["dtype"]).replace("torch.", ""), batch=spec["b"],
                       ms_per_forward=round(dt * 1e3, 3), samples_per_s=round(spec["b"] / dt, 1))
            if prec == "bf16" and "fp32" in outs:
                row["maxnorm_diff_vs_fp32_core"] = float((outs["bf16"] - outs["fp32"]).abs().max() / outs["fp32"].abs().max())
            rows.append(row)
            print(json.dumps(row), flush=True)
            del m
            torch.cuda.empty_cache()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
