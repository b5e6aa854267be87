"""CPU, world_size 2, gloo: the N>1 plumbing (batch sharding, ragged gather, bucketed gradient averaging,
max-over-ranks timing).  The forward itself needs no collective (samples are independent)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from healnet_amd import dist as hd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = hd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    try:
        # 1. sharding: a 7-sample, 2-modality batch with a missing third modality
        tab = torch.arange(7 * 1 * 5, dtype=torch.float32).reshape(7, 1, 5)
        img = torch.arange(7 * 2 * 3 * 3, dtype=torch.float32).reshape(7, 2, 3, 3)
        mine = hd.shard_batch([tab, img, None], rank, world)
        lo, hi = hd.shard_bounds(7, rank, world)
        assert mine[2] is None and torch.equal(mine[0], tab[lo:hi]) and torch.equal(mine[1], img[lo:hi])
        # a per-sample function stands in for the (sample-independent) forward
        local = mine[0].sum(dim=(1, 2), keepdim=False).unsqueeze(-1) + mine[1].flatten(1).sum(1, keepdim=True)
        full = hd.gather_outputs(local, 7)
        want = tab.sum(dim=(1, 2)).unsqueeze(-1) + img.flatten(1).sum(1, keepdim=True)
        assert torch.equal(full, want)
        # 2. gradient averaging over buckets (sizes straddle the bucket limit, mixed shapes)
        torch.manual_seed(100 + rank)
        grads = [torch.randn(3, 5), torch.randn(1000), torch.randn(7), torch.randn(64, 64)]
        mine_g = [g.clone() for g in grads]
        hd.allreduce_mean_(mine_g, bucket_bytes=4096)
        ref = []
        for i in range(len(grads)):
            acc = torch.zeros_like(grads[i])
            for rr in range(world):
                torch.manual_seed(100 + rr)
                acc += [torch.randn(3, 5), torch.randn(1000), torch.randn(7), torch.randn(64, 64)][i]
            ref.append(acc / world)
        for a, b in zip(mine_g, ref):
            assert torch.allclose(a, b, atol=1e-6)
        # 2b. one flat buffer larger than a bucket (healnet_amd.train.FlatParameters.grads): reduced in place, no staging copy
        torch.manual_seed(200 + rank)
        flat = torch.randn(5000)
        ptr = flat.data_ptr()
        hd.allreduce_mean_([flat], bucket_bytes=4096)
        acc = torch.zeros(5000)
        for rr in range(world):
            torch.manual_seed(200 + rr)
            acc += torch.randn(5000)
        assert flat.data_ptr() == ptr and torch.allclose(flat, acc / world, atol=1e-6)
        # 3. timing contract
        assert hd.max_over_ranks(1.0 + rank) == float(world)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_shard_bounds_cover_everything():
    for n in (1, 2, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [hd.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


@pytest.mark.parametrize("tie", [False, True])
def test_grad_ready_bucket_plan_covers_the_flat_buffer(tie):
    """The overlapped all-reduce releases one contiguous range of the flat gradient buffer per hn_grad_ready signal; together
    the ranges must cover every parameter exactly once, and a parameter may only be released by a signal that fires after the
    last layer using it has run its backward (lowest user layer; tied blocks -> layer 1; layer 0 / latents -> end of call)."""
    from healnet_amd import HealNet
    model = HealNet(n_modalities=2, channel_dims=[7, 3], num_spatial_axes=[1, 2], out_dims=3, depth=4, l_c=8, l_d=16, x_heads=2,
                    l_heads=2, cross_dim_head=8, latent_dim_head=8, weight_tie_layers=tie)
    views = [p for p in model.parameters()]
    offsets, off = [], 0
    for p in views:
        offsets.append(off)
        off += (p.numel() + 3) // 4 * 4
    buckets = hd.grad_ready_buckets(model, views, offsets)
    spans = sorted(r for rs in buckets.values() for r in rs)
    assert spans[0][0] == 0 and spans[-1][1] == off and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert all(len(rs) == 1 for rs in buckets.values()), buckets          # one message per signal
    assert set(buckets) == ({-1, 1} if tie else {-1, 1, 2, 3})
    where = {}
    for idx, rs in buckets.items():
        for lo, hi in rs:
            for p, o in zip(views, offsets):
                if lo <= o < hi:
                    where[id(p)] = idx
    for layer in range(model.depth):
        for p in model.layers[layer].parameters():
            assert where[id(p)] <= layer, "released before a layer that still accumulates into it has finished"
    assert where[id(model.latents)] == -1
    assert hd.shard_loss_scale(4, 7, 2) + hd.shard_loss_scale(3, 7, 2) == pytest.approx(2.0)
