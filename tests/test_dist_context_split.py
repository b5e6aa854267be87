"""CPU, world_size 2 and 3, gloo: the exchange step of the context split (second axis) -- slab planning, the ONE
all-gather of per-rank (normalised output, statistics) pairs, and the merge algebra hn_attn_merge_fwd implements, checked against
the oracle's attention over the whole context.  (The per-rank attention itself is HIP work: tests/test_gpu_context_split.py.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from healnet_amd import dist as hd
from oracle import healnet_cpu as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _partial(x, ctx, w_q, w_kv, heads):
    """What hn_attn_partial_fwd returns for one shard: normalised P V (b, L, inner) and {M, l} in log2 units (b, heads, L, 2)."""
    b, L, _ = x.shape
    inner = w_q.shape[0]
    e = inner // heads
    q = (x @ w_q.t()).reshape(b, L, heads, e).permute(0, 2, 1, 3)
    kv = ctx @ w_kv.t()
    k = kv[..., :inner].reshape(b, -1, heads, e).permute(0, 2, 1, 3)
    v = kv[..., inner:].reshape(b, -1, heads, e).permute(0, 2, 1, 3)
    s = (q @ k.transpose(-1, -2)) * (e ** -0.5) / 0.5 * 1.4426950408889634       # log2 units
    M = s.amax(-1)
    p = torch.exp2(s - M[..., None])
    l = p.sum(-1)
    o = (p @ v) / l[..., None]
    return o.permute(0, 2, 1, 3).reshape(b, L, inner), torch.stack([M, l], dim=-1)


def _merge(o_parts, st_parts, heads):
    """hn_attn_merge_fwd's fold: w_r = 2^(M_r - max M) l_r, o = sum w_r o_r / sum w_r."""
    G, b, L, inner = o_parts.shape
    M = st_parts[..., 0].amax(0)
    w = torch.exp2(st_parts[..., 0] - M) * st_parts[..., 1]                      # (G, b, heads, L)
    w = w / w.sum(0)
    w = w.permute(0, 1, 3, 2).repeat_interleave(inner // heads, dim=-1)          # (G, b, L, inner)
    return (w * o_parts).sum(0)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    hd.init_from_env("gloo")
    try:
        torch.manual_seed(0)                                                    # every rank holds the same latents and weights
        b, L, qd, N, D, heads, e = 2, 8, 16, 53, 11, 2, 4
        x, ctx = torch.randn(b, L, qd), torch.randn(b, N, D)
        w_q, w_kv, w_out, b_out = torch.randn(heads * e, qd) * 0.3, torch.randn(2 * heads * e, D) * 0.3, torch.randn(qd, heads * e) * 0.3, torch.randn(qd)
        lo, hi = hd.slab_bounds(N, rank, world)
        assert (lo, hi) == hd.shard_bounds(N, rank, world) and hi - lo in (N // world, N // world + 1)
        o, st = _partial(x, ctx[:, lo:hi], w_q, w_kv, heads)
        o_all, st_all = hd.gather_partials(o, st)
        assert o_all.shape == (world, b, L, heads * e) and st_all.shape == (world, b, heads, L, 2)
        assert torch.equal(o_all[rank], o) and torch.equal(st_all[rank], st)
        merged = _merge(o_all, st_all, heads)
        got = torch.nn.functional.leaky_relu(merged @ w_out.t() + b_out, 0.01)
        want = O.attention(x, ctx, w_q, w_kv, w_out, b_out, heads=heads)
        err = float((got - want).abs().max() / want.abs().max())
        assert err < 2e-6, err
        q.put((rank, "ok", got.numpy()))
    except Exception as e_:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e_), e_, e_.__traceback__)), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_partials_merge_to_the_whole_attention(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in results] == ["ok"] * world, [r[1] for r in results]
    for r in results[1:]:
        assert (r[2] == results[0][2]).all(), "ranks folded the same parts in the same order: bit-identical"


def _train_worker(rank, world, port, q):
    """The backward of the split block as hn_attn_bwd_cp + allreduce_sum_ arrange it, restated in torch: with the GLOBAL statistics
    and the global O, a rank's slab yields its dK / dV exactly and a PARTIAL dQ; dW_out / db_out come from replicated quantities and
    are produced by the owner only; ONE sum over ranks completes every gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    hd.init_from_env("gloo")
    try:
        torch.manual_seed(0)
        b, L, qd, N, D, heads, e = 2, 8, 16, 53, 11, 2, 4
        inner = heads * e
        x, ctx, dy = torch.randn(b, L, qd), torch.randn(b, N, D), torch.randn(b, L, qd)
        w_q, w_kv, w_out, b_out = torch.randn(inner, qd) * 0.3, torch.randn(2 * inner, D) * 0.3, torch.randn(qd, inner) * 0.3, torch.randn(qd)
        # reference: autograd through the oracle's whole-context attention
        leaf = [t.clone().requires_grad_(True) for t in (x, w_q, w_kv, w_out, b_out)]
        O.attention(leaf[0], ctx, leaf[1], leaf[2], leaf[3], leaf[4], heads=heads).backward(dy)
        want = [t.grad for t in leaf]
        # forward exchange
        lo, hi = hd.slab_bounds(N, rank, world)
        slab = ctx[:, lo:hi]
        o, st = _partial(x, slab, w_q, w_kv, heads)
        o_all, st_all = hd.gather_partials(o, st)
        o_glob = _merge(o_all, st_all, heads)                                   # (b, L, inner), replicated
        M = st_all[..., 0].amax(0)                                              # (b, heads, L)
        l = (torch.exp2(st_all[..., 0] - M) * st_all[..., 1]).sum(0)
        pre = o_glob @ w_out.t() + b_out
        # backward on the slab with the global (M, l)
        dpre = dy * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, 0.01))
        owner = rank == 0
        d_wout = dpre.reshape(-1, qd).t() @ o_glob.reshape(-1, inner) if owner else torch.zeros_like(w_out)
        d_bout = dpre.sum((0, 1)) if owner else torch.zeros_like(b_out)
        heads_of = lambda t: t.reshape(b, -1, heads, e).permute(0, 2, 1, 3)     # noqa: E731
        dO, Og = heads_of(dpre @ w_out), heads_of(o_glob)
        qh = heads_of(x @ w_q.t())
        kv = slab @ w_kv.t()
        kh, vh = heads_of(kv[..., :inner]), heads_of(kv[..., inner:])
        c = (e ** -0.5) / 0.5
        P = torch.exp2((qh @ kh.transpose(-1, -2)) * c * 1.4426950408889634 - M[..., None]) / l[..., None]
        dS = P * (dO @ vh.transpose(-1, -2) - (dO * Og).sum(-1, keepdim=True))
        unheads = lambda t: t.permute(0, 2, 1, 3).reshape(b, -1, inner)         # noqa: E731
        dq, dk, dv = unheads(dS @ kh) * c, unheads(dS.transpose(-1, -2) @ qh) * c, unheads(P.transpose(-1, -2) @ dO)
        part = [dq @ w_q, dq.reshape(-1, inner).t() @ x.reshape(-1, qd),
                torch.cat([dk, dv], -1).reshape(-1, 2 * inner).t() @ slab.reshape(-1, D), d_wout, d_bout]
        hd.allreduce_sum_(part)
        for name, got, ref in zip(("x", "w_q", "w_kv", "w_out", "b_out"), part, want):
            err = float((got - ref).abs().max() / ref.abs().max())
            assert err < 2e-5, (name, err)
        q.put((rank, "ok", part[2].numpy()))
    except Exception as e_:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e_), e_, e_.__traceback__)), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_backwards_with_global_statistics_sum_to_the_whole_gradient(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in results] == ["ok"] * world, [r[1] for r in results]
    for r in results[1:]:
        assert (r[2] == results[0][2]).all(), "every rank holds the same summed gradient"


def test_slab_bounds_cover_the_axis():
    for n, world in [(224, 8), (37, 3), (5, 8), (12, 1)]:
        spans = [hd.slab_bounds(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
