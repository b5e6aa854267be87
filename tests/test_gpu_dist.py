"""GPU: the data-parallel training step with the REAL model.

  * hn_grad_ready contract on one GPU: a gradient range snapshotted on a side stream at the moment its signal fires equals the
    final gradient bit for bit (nothing accumulated into it afterwards), with and without weight tying;
  * data-parallel equivalence: 2 processes on one GPU (gloo moves the CUDA tensors), each running its shard of a cfg4-like batch
    through HealNet + surv_nll_loss + the overlapped all-reduce (GradReadyAllReduce) or the blocking one (allreduce_mean_) +
    FusedL1Adam, reproduce the 1-process full-batch step: averaged flat gradient and updated parameters to fp32 noise, with an
    even (8 = 4 + 4) and a ragged (7 = 4 + 3) batch.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

KW = dict(n_modalities=2, channel_dims=[200, 96], num_spatial_axes=[1, 1], out_dims=4, depth=3, l_c=32, l_d=64, x_heads=4, l_heads=4,
          cross_dim_head=32, latent_dim_head=16)      # cfg4-like: one-token omic + a patch bag on the explicit K/V binding


def _inputs(n, seed):
    gen = torch.Generator().manual_seed(seed)
    return ([torch.rand(n, 1, 200, generator=gen), torch.rand(n, 300, 96, generator=gen)],
            torch.randint(0, 4, (n,), generator=gen), torch.randint(0, 2, (n,), generator=gen))


@pytest.mark.parametrize("tie", [False, True], ids=["untied", "tied"])
def test_grad_ready_signals_release_final_gradients(tie):
    import healnet_amd as hn
    from healnet_amd import dist as hd
    torch.manual_seed(3)
    model = hn.HealNet(**KW, weight_tie_layers=tie).train().to(DEV)
    flat = hn.train.flatten_parameters(model)
    snaps = []

    def snapshot(view):                      # runs on the side stream, after the signal's event
        snaps.append((view.data_ptr(), view.clone()))

    sync = hd.GradReadyAllReduce(model, flat, reduce_fn=snapshot)
    ins, y, c = _inputs(6, 5)
    for _ in range(2):                       # second pass: events are re-recorded, buffers reused
        snaps.clear()
        flat.zero_grad()
        out = hn.train.surv_nll_loss(model([t.to(DEV) for t in ins]), y.to(DEV), c.to(DEV))
        out.loss.backward()
        sync.wait()
        torch.cuda.synchronize()
        order = [idx for idx, _, _ in sync.launched]
        assert order == sorted(order, reverse=True) and order[-1] == -1, order        # top layer first, latents + layer 0 last
        assert len(snaps) == len(sync.launched) == (2 if tie else 3)
        covered = 0
        base = flat.grads.data_ptr()
        for (idx, lo, hi), (ptr, snap) in zip(sync.launched, snaps):
            assert ptr == base + 4 * lo
            assert torch.equal(snap, flat.grads[lo:hi]), f"signal {idx}: range [{lo}, {hi}) changed after it was released"
            assert float(snap.abs().max()) > 0
            covered += hi - lo
        assert covered == flat.numel
    # the same gradients without the hook
    want = flat.grads.clone()
    sync.close()
    flat.zero_grad()
    hn.train.surv_nll_loss(model([t.to(DEV) for t in ins]), y.to(DEV), c.to(DEV)).loss.backward()
    assert torch.equal(flat.grads, want)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _step(model, flat, opt, ins, y, c, scale, reduce):
    import healnet_amd as hn
    opt.zero_grad()
    out = hn.train.surv_nll_loss(model(ins), y, c)
    (out.loss * scale).backward()
    reduce()
    g = flat.grads.clone()
    opt.step()
    return g


def _dp_worker(rank, world, port, n_total, overlapped, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import healnet_amd as hn
    from healnet_amd import dist as hd
    try:
        torch.cuda.set_device(0)
        hd.init_from_env("gloo")
        dev = torch.device("cuda", 0)
        torch.manual_seed(11)
        model = hn.HealNet(**KW).train().to(dev)
        flat = hn.train.flatten_parameters(model)
        opt = hn.train.FusedL1Adam(flat, lr=1e-3, l1=1e-4)
        ins, y, c = _inputs(n_total, 17)
        lo, hi = hd.shard_bounds(n_total, rank, world)
        mine = [t.to(dev) for t in hd.shard_batch(ins, rank, world)]
        scale = hd.shard_loss_scale(hi - lo, n_total, world)
        if overlapped:
            sync = hd.GradReadyAllReduce(model, flat)
            reduce = sync.wait
        else:
            reduce = lambda: hd.allreduce_mean_([flat.grads])      # noqa: E731
        g = _step(model, flat, opt, mine, y[lo:hi].to(dev), c[lo:hi].to(dev), scale, reduce)
        if overlapped:
            assert [i for i, _, _ in sync.launched] == [2, 1, -1], sync.launched
        torch.cuda.synchronize()
        q.put((rank, "ok", g.cpu().numpy(), flat.params.detach().cpu().numpy()))      # numpy: pickled by value (no fd passing)
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__)), None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7], ids=["even", "ragged"])
@pytest.mark.parametrize("overlapped", [True, False], ids=["grad_ready", "blocking"])
def test_two_rank_step_equals_the_full_batch_step(n_total, overlapped):
    import healnet_amd as hn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, n_total, overlapped, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    assert [r[1] for r in results] == ["ok", "ok"], [r[1] for r in results]
    # the 1-process full-batch step
    torch.manual_seed(11)
    model = hn.HealNet(**KW).train().to(DEV)
    flat = hn.train.flatten_parameters(model)
    opt = hn.train.FusedL1Adam(flat, lr=1e-3, l1=1e-4)
    ins, y, c = _inputs(n_total, 17)
    g = _step(model, flat, opt, [t.to(DEV) for t in ins], y.to(DEV), c.to(DEV), 1.0, lambda: None)
    import numpy as np
    p_ref = flat.params.detach().cpu()
    for rank, _, g_dp, p_dp in results:
        assert_close(torch.from_numpy(g_dp), g.cpu(), rel=2e-5, floor=2e-6,
                     what=f"rank {rank}: averaged flat gradient vs the full-batch gradient")
        # Adam's first step moves a parameter by lr * g / (|g| + eps) ~ lr * sign(g): where the total gradient (data term +
        # l1 * sign(p)) nearly cancels, fp32 summation order decides the size of the move -- bounded by a fraction of lr there,
        # and everything else must agree to fp32 noise
        diff = (torch.from_numpy(p_dp) - p_ref).abs()
        assert float(diff.max()) <= 0.2 * 1e-3, float(diff.max())
        assert float((diff > 1e-6).float().mean()) < 1e-4, float((diff > 1e-6).float().mean())
    assert np.array_equal(results[0][2], results[1][2]) and np.array_equal(results[0][3], results[1][3]), "ranks diverged"
