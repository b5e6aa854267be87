#!/usr/bin/env python3
"""Small-batch latency (Next 4): host enqueue vs wall time of one forward at b = 1 / 4 for cfg1 (tab + image) and the
reference's README call (tab + image + volume, README.md:80-110), eager vs the model-owned graph replay (HealNet.capture).

    python tools/small_batch.py [--profile]          # --profile: cProfile of the eager host path at b = 1
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn

DEV = "cuda:0"
CASES = {
    "cfg1": dict(kw=dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4), shapes=[(1, 2000), (224, 224, 3)]),
    "readme3": dict(kw=dict(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4),
                    shapes=[(1, 2000), (224, 224, 3), (12, 224, 224, 3)]),
}


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    t_call = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return t_call * 1e3, (time.perf_counter() - t) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--json", default=None)
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 4])
    args = ap.parse_args()
    rows = []
    torch.set_grad_enabled(False)
    for name, case in CASES.items():
        for keep in (True, False):
            torch.manual_seed(0)
            m = hn.HealNet(**case["kw"]).eval().to(DEV)
            m.keep_attention_stats = keep
            for b in args.batches:
                gen = torch.Generator().manual_seed(7)
                ins = [torch.rand(b, *s, generator=gen).to(DEV) for s in case["shapes"]]
                host, wall = timeit(lambda: m(list(ins)), args.n)
                ref = m(list(ins)).clone()
                g = m.capture(ins)
                ghost, gwall = timeit(lambda: g(), args.n)
                gcopy_host, gcopy_wall = timeit(lambda: g(ins), args.n)
                same = bool(torch.equal(g(ins), ref))
                row = dict(case=name, b=b, keep_attention_stats=keep, eager_host_ms=round(host, 4), eager_wall_ms=round(wall, 4),
                           graph_host_ms=round(ghost, 4), graph_wall_ms=round(gwall, 4), graph_with_input_copy_wall_ms=round(gcopy_wall, 4),
                           graph_equals_eager_bitwise=same)
                rows.append(row)
                print(json.dumps(row), flush=True)
                del g
    if args.profile:
        import cProfile, pstats
        m = hn.HealNet(**CASES["cfg1"]["kw"]).eval().to(DEV)
        ins = [torch.rand(1, *s).to(DEV) for s in CASES["cfg1"]["shapes"]]
        for _ in range(10):
            m(list(ins))
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(300):
            m(list(ins))
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
