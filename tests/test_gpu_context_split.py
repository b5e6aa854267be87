"""GPU: the context split over ranks (the optional second axis: b < #GPUs).

  * hn_encode_norm_slab: a slab of a modality along its first spatial axis gets, bit for bit, the rows hn_encode_norm writes for
    the same tokens of the whole tensor;
  * hn_attn_partial_fwd over G shards + hn_attn_merge_fwd == hn_attn_fwd over the whole context == the CPU oracle
    (shared-context / rank-D binding, explicit K/V binding, with and without a key mask, even and ragged shards, a shard whose keys
    are all masked);
  * healnet_amd.dist.context_parallel_forward with 2 and 3 processes on one GPU (gloo moves the partials): logits equal to the plain
    forward and to the oracle, identical on every rank.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape,parts", [((2, 9, 5, 3), 2), ((1, 7, 4, 6, 2), 3), ((3, 40, 11), 4), ((2, 224, 8, 3), 8)],
                         ids=["image", "volume", "bag", "image-8"])
def test_encode_slabs_are_rows_of_the_whole_encode(shape, parts):
    import healnet_amd  # noqa: F401
    from healnet_amd import dist as hd
    hip = torch.ops.healnet_hip
    gen = torch.Generator().manual_seed(1)
    data = torch.rand(*shape, generator=gen).to(DEV)
    pitch = 32
    whole = hip.encode_norm(data, 2, 10.0, True, pitch)
    per_row = int(np.prod(shape[2:-1])) if len(shape) > 3 else 1
    for r in range(parts):
        lo, hi = hd.slab_bounds(shape[1], r, parts)
        slab = hip.encode_norm_slab(data[:, lo:hi].contiguous(), 2, 10.0, True, pitch, lo, shape[1])
        assert torch.equal(slab, whole[:, lo * per_row:hi * per_row]), f"slab {r} of {parts}"


def _block(qd, D, heads, dh, seed):
    import healnet_amd as hn
    torch.manual_seed(seed)
    pn = hn.healnet.PreNorm(qd, hn.Attention(qd, D, heads=heads, dim_head=dh), context_dim=D).to(DEV).eval()
    with torch.no_grad():
        for p in pn.parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn_like(p))
    return pn


CASES = [
    dict(b=2, L=32, N=900, D=13, heads=4, dh=32, qd=64, parts=3),                    # shared-context (rank-D) binding, ragged shards
    dict(b=1, L=128, N=6000, D=13, heads=8, dh=64, qd=128, parts=8),                 # the default model's image block, 8 ranks
    dict(b=2, L=16, N=700, D=96, heads=2, dh=32, qd=32, parts=2),                    # explicit K/V binding
    dict(b=1, L=128, N=4400, D=200, heads=8, dh=64, qd=128, parts=4),                # explicit binding on the LDS-DMA projection
    dict(b=2, L=24, N=500, D=18, heads=2, dh=27, qd=40, parts=2, masked=True),       # padded head width + key mask
    dict(b=2, L=16, N=600, D=13, heads=2, dh=16, qd=32, parts=3, masked=True, dead_shard=1),   # every key of one shard masked
]


@pytest.mark.parametrize("kw", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_partial_plus_merge_equals_the_whole_block(kw):
    from healnet_amd import _capi
    from healnet_amd import dist as hd
    from oracle import healnet_cpu as O
    hip = torch.ops.healnet_hip
    b, L, N, D, heads, dh, qd, parts = (kw[k] for k in ("b", "L", "N", "D", "heads", "dh", "qd", "parts"))
    pn = _block(qd, D, heads, dh, 7)
    a = pn.fn
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(b, L, qd, generator=gen).to(DEV)
    ctx = torch.randn(b, N, D, generator=gen)
    mask = None
    if kw.get("masked"):
        mask = torch.rand(b, N, generator=gen) > 0.3
        if "dead_shard" in kw:
            lo, hi = hd.slab_bounds(N, kw["dead_shard"], parts)
            mask[:, lo:hi] = False
    pitch = _capi.lib().hn_context_pitch(D, dh)
    z = hip.encode_norm(ctx.to(DEV).unsqueeze(2).reshape(b, N, D), 0, 0.0, False, pitch)      # affine-free LayerNorm of the context
    m_dev = None if mask is None else mask.to(DEV).to(torch.uint8)
    wts = (pn.norm.weight, pn.norm.bias, pn.norm_context.weight, pn.norm_context.bias, a.to_q.weight, a.to_kv.weight, a.to_out[0].weight,
           a.to_out[0].bias)
    with torch.no_grad():
        whole, st_whole, _ = hip.attention_fwd(x, z, m_dev, *wts, heads, True, False)
        o_parts, st_parts = [], []
        for r in range(parts):
            lo, hi = hd.slab_bounds(N, r, parts)
            o, st = hip.attention_partial(x, z[:, lo:hi].contiguous(), None if m_dev is None else m_dev[:, lo:hi].contiguous(), *wts, heads)
            o_parts.append(o)
            st_parts.append(st)
        got, st = hip.attention_merge(x, torch.stack(o_parts), torch.stack(st_parts), a.to_q.weight, a.to_out[0].weight, a.to_out[0].bias,
                                      heads, True)
        got2, _ = hip.attention_merge(x, torch.stack(o_parts), torch.stack(st_parts), a.to_q.weight, a.to_out[0].weight, a.to_out[0].bias,
                                      heads, True)
    assert torch.equal(got, got2)
    assert_close(got.cpu(), whole.cpu(), rel=2e-5, floor=2e-6, what="merged shards vs the whole block")
    # merged statistics describe the same softmax as the whole block's: log2-sum-exp M + log2(l) agrees
    lse = lambda s_: (s_[..., 0].double() + torch.log2(s_[..., 1].double()))      # noqa: E731
    live = torch.isfinite(lse(st_whole.cpu())) & (st_whole[..., 0].cpu() > -1e30)
    assert_close(lse(st.cpu())[live].float(), lse(st_whole.cpu())[live].float(), rel=1e-5, floor=1e-4, what="merged log-sum-exp")
    # the oracle on the same numbers
    xc, cc = x.cpu(), ctx
    sd = {k: v.detach().cpu() for k, v in pn.state_dict().items()}
    xn = O.layer_norm(xc, sd["norm.weight"], sd["norm.bias"])
    cn = O.layer_norm(cc, sd["norm_context.weight"], sd["norm_context.bias"])
    want = O.attention(xn, cn, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"], heads=heads,
                       mask=mask) + xc
    assert_close(got.cpu(), want, rel=2e-4, floor=2e-5, what="merged shards vs the oracle")


TRAIN_CASES = [c for c in CASES if not c.get("masked")] + [
    dict(b=2, L=24, N=500, D=18, heads=2, dh=27, qd=40, parts=2),                    # shared-context binding on 32-column rows, padded head width
    dict(b=3, L=17, N=333, D=40, heads=3, dh=20, qd=30, parts=4),                    # explicit binding, odd sizes everywhere, four ragged shards
]


@pytest.mark.parametrize("kw", TRAIN_CASES, ids=[f"case{i}" for i in range(len(TRAIN_CASES))])
def test_shard_backwards_sum_to_the_whole_block_backward(kw):
    """Context split, TRAINING (ABI v11): the training forward per shard, the merge of the shards' (O or P z, statistics) pairs, the
    output from the merged tape, and hn_attn_bwd_cp per shard with the GLOBAL statistics -- the shards' gradients (replicated terms on
    the owner only) sum to the gradients of the whole block: against the plain block's autograd on the GPU and oracle autograd."""
    from healnet_amd import _capi, ops
    from healnet_amd import dist as hd
    from oracle import healnet_cpu as O
    hip = torch.ops.healnet_hip
    b, L, N, D, heads, dh, qd, parts = (kw[k] for k in ("b", "L", "N", "D", "heads", "dh", "qd", "parts"))
    pn = _block(qd, D, heads, dh, 7).train()
    a = pn.fn
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(b, L, qd, generator=gen).to(DEV)
    ctx = torch.randn(b, N, D, generator=gen)
    dy = torch.randn(b, L, qd, generator=gen).to(DEV)
    pitch = _capi.lib().hn_context_pitch(D, dh)
    z = hip.encode_norm(ctx.to(DEV).unsqueeze(2).reshape(b, N, D), 0, 0.0, False, pitch)
    names = ("norm.weight", "norm.bias", "norm_context.weight", "norm_context.bias", "fn.to_q.weight", "fn.to_kv.weight", "fn.to_out.0.weight",
             "fn.to_out.0.bias")
    params = dict(pn.named_parameters())
    wts = tuple(params[n] for n in names)
    # the whole block through its own autograd
    xw = x.clone().requires_grad_(True)
    whole = hip.attention(xw, z, None, *wts, heads, True)
    g_whole = torch.autograd.grad(whole, (xw,) + wts, dy)
    # shards
    wd = tuple(t.detach() for t in wts)
    slabs = [z[:, lo:hi].contiguous() for lo, hi in (hd.slab_bounds(N, r, parts) for r in range(parts))]
    local = [ops.cp_local_forward(x, sl, wd, heads) for sl in slabs]
    width = local[0][3]
    all_parts = torch.stack([lc[2].clone() for lc in local])
    all_stats = torch.stack([lc[0].clone() for lc in local])
    outs = []
    for (stats, saved, _, _), sl in zip(local, slabs):
        ops.cp_merge(all_parts, all_stats, saved, stats, b, heads, L, width)
        outs.append(ops.cp_finish(x, sl, wd, heads, saved, True))
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "the ranks' block outputs differ"
    assert_close(outs[0].cpu(), whole.detach().cpu(), rel=2e-5, floor=2e-6, what="context-split training forward vs the whole block")
    total = None
    for r, ((stats, saved, _, _), sl) in enumerate(zip(local, slabs)):
        g = ops.cp_backward(dy, x, outs[0], sl, wd, heads, stats, saved, owner=(r == 1 % parts))
        total = g if total is None else [t + u for t, u in zip(total, g)]
    total[0] = total[0] + dy
    scale = max(float(t.abs().max()) for t in g_whole)
    for name, got, ref in zip(("x",) + names, total, g_whole):
        assert_close(got.cpu(), ref.cpu(), rel=3e-4, floor=1e-4, abs_floor=2e-6 * scale, what=f"sum over shards: d {name}")
    # oracle autograd on the same numbers
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in pn.state_dict().items()}
    xc = x.cpu().clone().requires_grad_(True)
    xn = O.layer_norm(xc, sd["norm.weight"], sd["norm.bias"])
    cn = O.layer_norm(ctx, sd["norm_context.weight"], sd["norm_context.bias"])
    want = O.attention(xn, cn, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"], heads=heads) + xc
    want.backward(dy.cpu())
    for name, got in zip(("x",) + names, total):
        ref = xc.grad if name == "x" else sd[name].grad
        assert_close(got.cpu(), ref, rel=2e-3, floor=1e-3, abs_floor=1e-5 * scale, what=f"sum over shards vs oracle: d {name}")


def test_context_split_training_refusals():
    """What the training entry points do not take says so: a one-token shard has nothing to merge (hn_attn_saved_part_width = 0), a
    dropping block is refused by hn_attn_finish_fwd / hn_attn_bwd_cp with HN_E_UNSUPPORTED (never a silent dropout-free result)."""
    import ctypes as C
    from healnet_amd import _capi, ops
    pn = _block(32, 13, 2, 16, 3)
    a = pn.fn
    wts = tuple(t.detach() for t in (pn.norm.weight, pn.norm.bias, pn.norm_context.weight, pn.norm_context.bias, a.to_q.weight, a.to_kv.weight,
                                     a.to_out[0].weight, a.to_out[0].bias))
    x = torch.randn(2, 16, 32, device=DEV)
    z = torch.ops.healnet_hip.encode_norm(torch.randn(2, 40, 13, device=DEV), 0, 0.0, False, 16)
    with pytest.raises(RuntimeError, match="cannot be split"):
        ops.cp_local_forward(x, z[:, :1].contiguous(), wts, 2)
    stats, saved, _, _ = ops.cp_local_forward(x, z, wts, 2)
    p, (b, L, N, D, ld) = ops._attn_params(x, z, *wts, 2)
    p.dropout = 0.25
    lib = _capi.lib()
    ws = torch.empty(lib.hn_attn_bwd_workspace_bytes(C.byref(p), 1, ld, b, L, N, D, 0) // 4 + 64, device=DEV)
    out = torch.empty_like(x)
    rc = lib.hn_attn_finish_fwd(C.byref(p), x.data_ptr(), out.data_ptr(), 1, ld, b, L, N, D, saved.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                torch.cuda.current_stream().cuda_stream)
    assert rc == _capi.HN_E_UNSUPPORTED
    g = [torch.zeros_like(t) for t in wts]
    grads = _capi.AttnGrads(norm_w=g[0].data_ptr(), norm_b=g[1].data_ptr(), ctx_gamma=g[2].data_ptr(), ctx_beta=g[3].data_ptr(),
                            w_q=g[4].data_ptr(), w_kv=g[5].data_ptr(), w_out=g[6].data_ptr(), b_out=g[7].data_ptr())
    rc = lib.hn_attn_bwd_cp(C.byref(p), x.data_ptr(), out.data_ptr(), z.data_ptr(), ld, b, L, N, D, stats.data_ptr(), saved.data_ptr(),
                            x.data_ptr(), out.data_ptr(), C.byref(grads), 1, ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
    assert rc == _capi.HN_E_UNSUPPORTED
    assert all(float(t.abs().max()) == 0.0 for t in g)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


KW = dict(n_modalities=3, channel_dims=[200, 3, 64], num_spatial_axes=[1, 2, 1], out_dims=4, depth=2, l_c=32, l_d=64, x_heads=4, l_heads=4,
          cross_dim_head=32, latent_dim_head=16)      # one-token tabular (replicated) + image (rank-D binding) + bag (explicit binding)


KW_CHAIN = dict(n_modalities=3, channel_dims=[200, 3, 64], num_spatial_axes=[1, 2, 1], out_dims=4, depth=2)      # default widths: the
# latent chains take it, so hn_fusion_forward_cp (the fused route) runs


def _inputs(b):
    gen = torch.Generator().manual_seed(21)
    return [torch.rand(b, 1, 200, generator=gen), torch.rand(b, 37, 20, 3, generator=gen), torch.rand(b, 301, 64, generator=gen)]


def _cp_worker(rank, world, port, b, q, kw=None, fused=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import healnet_amd as hn
    from healnet_amd import dist as hd
    try:
        torch.cuda.set_device(0)
        hd.init_from_env("gloo")
        dev = torch.device("cuda", 0)
        torch.manual_seed(5)
        model = hn.HealNet(**(kw or KW)).eval().to(dev)
        ins = [t.to(dev) for t in _inputs(b)]
        with torch.no_grad():      # (with gradients enabled the call takes the training route: the test below)
            out = hd.context_parallel_forward(model, ins, fused=fused)
            emb = hd.context_parallel_forward(model, ins, return_embeddings=True, fused=fused)
        torch.cuda.synchronize()
        q.put((rank, "ok", out.cpu().numpy(), emb.cpu().numpy()))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__)), None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world,b,route", [(2, 1, "blocks"), (3, 2, "blocks"), (2, 2, "fused"), (3, 1, "fused")],
                         ids=["2-ranks-b1", "3-ranks-b2", "2-ranks-b2-fused", "3-ranks-b1-fused"])
def test_context_parallel_forward_matches_the_plain_forward(world, b, route):
    import healnet_amd as hn
    from oracle import healnet_cpu as O
    KW_ = KW_CHAIN if route == "fused" else KW
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cp_worker, args=(r, world, port, b, q, KW_, True if route == "fused" else False)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    assert [r[1] for r in results] == ["ok"] * world, [r[1] for r in results]
    torch.manual_seed(5)
    model = hn.HealNet(**KW_).eval().to(DEV)
    ins = _inputs(b)
    with torch.no_grad():
        plain = model([t.to(DEV) for t in ins]).cpu()
        plain_emb = model([t.to(DEV) for t in ins], return_embeddings=True).cpu()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = O.fusion_forward(sd, O.FusionConfig(**KW_), [t.clone() for t in ins])
    for rank, _, out, emb in results:
        assert_close(torch.from_numpy(out), plain, rel=2e-5, floor=2e-6, what=f"rank {rank}: context-parallel logits vs the plain forward")
        assert_close(torch.from_numpy(emb), plain_emb, rel=2e-5, floor=2e-6, what=f"rank {rank}: embeddings")
        assert_close(torch.from_numpy(out), want, rel=1e-4, floor=1e-5, what=f"rank {rank}: context-parallel logits vs the oracle")
        assert np.array_equal(out, results[0][2]) and np.array_equal(emb, results[0][3]), "ranks diverged"


def _cp_train_worker(rank, world, port, b, q, missing=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import healnet_amd as hn
    from healnet_amd import dist as hd
    try:
        torch.cuda.set_device(0)
        hd.init_from_env("gloo")
        dev = torch.device("cuda", 0)
        torch.manual_seed(5)
        model = hn.HealNet(**KW).eval().to(dev)          # (eval: no dropout modules in play; the parameters still require grad)
        ins = [None if i == missing else t.to(dev) for i, t in enumerate(_inputs(b))]
        dl = torch.randn(b, KW["out_dims"], generator=torch.Generator().manual_seed(9)).to(dev)
        out = hd.context_parallel_forward(model, ins)
        (out * dl).sum().backward()
        torch.cuda.synchronize()
        q.put((rank, "ok", out.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in model.named_parameters() if p.grad is not None}))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__)), None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world,b,missing", [(2, 2, None), (3, 1, None), (2, 1, 1)], ids=["2-ranks-b2", "3-ranks-b1", "2-ranks-b1-image-missing"])
def test_context_parallel_training_step_matches_the_plain_backward(world, b, missing):
    """Training with the contexts split over 2 / 3 processes (one GPU, gloo): after loss.backward() EVERY rank holds the complete
    gradients -- equal to the plain model's fused backward and to oracle autograd, and equal across ranks up to the all-reduce."""
    import healnet_amd as hn
    from oracle import healnet_cpu as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cp_train_worker, args=(r, world, port, b, q, missing)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    assert [r[1] for r in results] == ["ok"] * world, [r[1] for r in results]
    torch.manual_seed(5)
    model = hn.HealNet(**KW).eval().to(DEV)
    ins = [None if i == missing else t for i, t in enumerate(_inputs(b))]
    dl = torch.randn(b, KW["out_dims"], generator=torch.Generator().manual_seed(9))
    plain = model([None if t is None else t.to(DEV) for t in ins])
    (plain * dl.to(DEV)).sum().backward()
    g_plain = {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    want = O.fusion_forward(sd, O.FusionConfig(**KW), [None if t is None else t.clone() for t in ins])
    (want * dl).sum().backward()
    scale = max(float(v.abs().max()) for v in g_plain.values())
    for rank, _, out, grads in results:
        assert_close(torch.from_numpy(out), plain.detach().cpu(), rel=2e-5, floor=2e-6, what=f"rank {rank}: logits")
        # (a skipped modality's blocks: no gradient on the block route, zeros from the fused backward)
        assert set(grads) <= set(g_plain) and all(float(g_plain[k].abs().max()) == 0.0 for k in set(g_plain) - set(grads))
        for k, g in grads.items():
            assert_close(torch.from_numpy(g), g_plain[k], rel=5e-4, floor=2e-4, abs_floor=2e-6 * scale, what=f"rank {rank}: grad[{k}] vs the plain backward")
            ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
            assert_close(torch.from_numpy(g), ref, rel=2e-3, floor=1e-3, abs_floor=1e-5 * scale, what=f"rank {rank}: grad[{k}] vs oracle autograd")
            assert_close(torch.from_numpy(g), torch.from_numpy(results[0][3][k]), rel=1e-5, floor=1e-6, abs_floor=1e-7 * scale, what=f"rank {rank} vs rank 0: grad[{k}]")


def test_single_rank_and_validation():
    """world == 1 degenerates to the block-by-block forward (no collective); refused inputs say why."""
    import healnet_amd as hn
    from healnet_amd import dist as hd
    torch.manual_seed(5)
    model = hn.HealNet(**KW).eval().to(DEV)
    ins = [t.to(DEV) for t in _inputs(2)]
    with torch.no_grad():
        plain = model(ins)
    got = hd.context_parallel_forward(model, ins, rank=0, world=1, fused=False)      # (gradients enabled: the autograd route)
    assert got.requires_grad
    assert_close(got.detach().cpu(), plain.cpu(), rel=2e-5, floor=2e-6, what="block-by-block forward vs the fused forward")
    # the fused route: refused for this model (l_d = 64: not a chain shape) with fused=True, silently replaced by default
    with pytest.raises(RuntimeError):
        hd.context_parallel_forward(model, ins, rank=1, world=2, fused=True, gather_flat=lambda lo, pa: pa.copy_(lo.repeat(2)))
    # ... and taken by a default-width model: one rank of two with a stand-in exchange (its own part twice) still runs end to end,
    # and world = 1 through the fused entry point reproduces the plain forward
    torch.manual_seed(5)
    big = hn.HealNet(**KW_CHAIN).eval().to(DEV)
    with torch.no_grad():
        plain_big = big(ins)
    got_big = hd.context_parallel_forward(big, ins, rank=0, world=1, fused=True)
    assert_close(got_big.cpu(), plain_big.cpu(), rel=2e-5, floor=2e-6, what="hn_fusion_forward_cp with one part vs hn_fusion_forward")
    half = hd.context_parallel_forward(big, ins, rank=1, world=2, fused=True, gather_flat=lambda lo, pa: pa.copy_(lo.repeat(2)))
    assert torch.isfinite(half).all()
    # a missing modality is skipped as in the plain forward (block-by-block route)
    with torch.no_grad():
        plain_missing = model([ins[0], None, ins[2]])
        got_missing = hd.context_parallel_forward(model, [ins[0], None, ins[2]], rank=0, world=1)
        got_short = hd.context_parallel_forward(model, [ins[0], ins[1]], rank=0, world=1)
        plain_short = model([ins[0], ins[1]])
    assert_close(got_missing.cpu(), plain_missing.cpu(), rel=2e-5, floor=2e-6, what="missing modality")
    assert_close(got_short.cpu(), plain_short.cpu(), rel=2e-5, floor=2e-6, what="shorter list")
    with pytest.raises(ValueError):
        hd.context_parallel_forward(model, [None, None, None], rank=0, world=1)
    with pytest.raises(ValueError):
        hd.context_parallel_forward(model, ins + [ins[0]], rank=0, world=1)
