#!/usr/bin/env python3
"""time HealNet with the reference's tuned TCGA hyper-parameters (config/best_hyperparams.yml of the reference:
odd widths, ONE cross head, no latent self-attention, dropout on) on a cfg4-shaped batch -- omic (b, 1, 2000) + WSI patch bag
(b, 4096, 768): inference forward, training forward + backward (dropout active), and the default-width model beside them.

    python tools/bench_tuned.py [--batch 8] [--json out.json]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn

TUNED = {   # dataset: depth, num_latents, latent_dim, cross_dim_head, latent_dim_head, attn_dropout, ff_dropout
    "blca": dict(depth=2, l_c=25, l_d=119, cross_dim_head=16, latent_dim_head=127, attn_dropout=0.0830, ff_dropout=0.4733),
    "brca": dict(depth=2, l_c=17, l_d=126, cross_dim_head=63, latent_dim_head=20, attn_dropout=0.4553, ff_dropout=0.3647),
    "kirp": dict(depth=5, l_c=17, l_d=62, cross_dim_head=27, latent_dim_head=113, attn_dropout=0.3179, ff_dropout=0.0474),
    "ucec": dict(depth=2, l_c=16, l_d=65, cross_dim_head=103, latent_dim_head=51, attn_dropout=0.2488, ff_dropout=0.0571),
}
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--json", default="")
ap.add_argument("--configs", nargs="*", default=None, help="subset of blca brca kirp ucec default")
ap.add_argument("--per-param-grads", action="store_true", help="the reference loop's `p.grad = None` per step (one zero tensor per parameter "
                "per backward) instead of healnet_amd.train.flatten_parameters (gradients accumulate into one flat buffer)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
b = args.batch
gen = torch.Generator().manual_seed(1)
ins = [torch.rand(b, 1, 2000, generator=gen).to(dev), torch.rand(b, 4096, 768, generator=gen).to(dev)]


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


rows = []
for name, kw in [(n, k) for n, k in list(TUNED.items()) + [("default", dict())] if args.configs is None or n in args.configs]:
    torch.manual_seed(0)
    extra = dict(x_heads=1, l_heads=8, self_per_cross_attn=0, num_freq_bands=2, max_freq=2.0) if kw else {}
    model = hn.HealNet(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4, **kw, **extra).to(dev)
    model.eval()
    with torch.no_grad():
        t_fwd = timeit(lambda: model(list(ins)))
    model.train()
    flat = None if args.per_param_grads else hn.train.flatten_parameters(model)

    def step():
        if flat is None:
            for p in model.parameters():
                p.grad = None
        else:
            flat.zero_grad()
        model(list(ins)).sum().backward()
    t_train = timeit(step, n=15, warm=3)
    rows.append(dict(config=name, batch=b, grads="per-parameter" if flat is None else "flat buffer", forward_ms=round(t_fwd, 3), fwd_bwd_ms=round(t_train, 3),
                     params=sum(p.numel() for p in model.parameters())))
    print(rows[-1], flush=True)
if args.json:
    json.dump(rows, open(args.json, "w"), indent=1)
