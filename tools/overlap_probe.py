#!/usr/bin/env python3
"""How much of the small-kernel "tail" of a training step hides behind a chip-filling GEMM on a second stream?
(feasibility probe for the side-stream weight-gradient / K/V-projection GEMMs)

Stream A: k fp32 products of the patch-bag K/V projection's size (32768 x 773) @ (773 x 1024) (rocBLAS through torch.mm: a stand-in
for gemm_nt_glds / gemm_tn_glds).  Stream B: forward + backward of a one-modality (tab 1 x 2000) depth-3 HealNet at b = 8 -- latent
chains, self-attention pairs, bchains, batched weight gradients: the launches that make up the tail of the cfg4 step.
Prints A alone, B alone, A || B (wall until both are done), with default priorities and with B on a high-priority stream."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import healnet_amd as hn

DEV = "cuda:0"


def main():
    torch.manual_seed(0)
    model = hn.HealNet(n_modalities=1, channel_dims=[2000], num_spatial_axes=[1], out_dims=4, depth=3).train().to(DEV)
    flat = hn.train.flatten_parameters(model)
    x = torch.rand(8, 1, 2000, device=DEV)
    a = torch.rand(32768, 773, device=DEV)
    w = torch.rand(773, 1024, device=DEV)
    out = torch.empty(32768, 1024, device=DEV)
    reps_a = 6

    def work_b(n=4):
        for _ in range(n):
            flat.zero_grad()
            model([x]).sum().backward()

    def work_a():
        for _ in range(reps_a):
            torch.mm(a, w, out=out)

    def wall(fn, n=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    res = {"A_alone_ms": round(wall(work_a), 3), "B_alone_ms": round(wall(work_b), 3)}
    for name, hi in (("default", False), ("B_high_priority", True)):
        side = torch.cuda.Stream(DEV)
        main = torch.cuda.Stream(DEV, priority=-1) if hi else torch.cuda.current_stream()

        def both():
            with torch.cuda.stream(main):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    work_a()
                work_b()
                main.wait_stream(side)
        res["A_and_B_%s_ms" % name] = round(wall(both), 3)
    res["sum_ms"] = round(res["A_alone_ms"] + res["B_alone_ms"], 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
