import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import healnet_amd as hn
for name, kw, shapes, b in [("cfg2", dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4), [(1, 2000), (224, 224, 3)], 32),
                            ("cfg4", dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4), [(1, 2000), (4096, 768)], 8)]:
    for pa, pf in [(0.0, 0.0), (0.25, 0.25)]:
        torch.manual_seed(0)
        m = hn.HealNet(**kw, attn_dropout=pa, ff_dropout=pf).train().cuda()
        ins = [torch.rand(b, *s, device="cuda") for s in shapes]
        def step():
            for p in m.parameters(): p.grad = None
            m(list(ins)).sum().backward()
        for _ in range(3): step()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(10): step()
        torch.cuda.synchronize()
        print(f"{name} dropout=({pa},{pf}) fwd+bwd {(time.time()-t)/10*1e3:.2f} ms", flush=True)
