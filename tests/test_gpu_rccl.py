"""GPU: the RCCL code path itself, on a ONE-rank `nccl` process group (Next 1).

Every other multi-rank test of this repo talks over gloo (two processes sharing the box's single GPU; RCCL needs one device per
rank).  Here the default process group is `nccl` (= RCCL on ROCm) with world size 1 and HN_FORCE_COLLECTIVES=1, which turns
the `world == 1` early returns of healnet_amd.dist off -- so the calls the driver's 8-GPU run will make all really execute on
the box:

  * `init_process_group("nccl", device_id=cuda:0)` (eager communicator), `dist.barrier()`;
  * `GradReadyAllReduce` on a real cfg4-like backward: `ncclAllReduce(AVG)` enqueued under the side stream from INSIDE the
    hn_grad_ready host callback (a ctypes callback running in the middle of hn_fusion_backward), `sync.wait()`, and the reduced
    flat gradient == the gradient of the same step without any hook, bit for bit (an average over one rank is the identity);
  * `allreduce_mean_` (flat in-place bucket and the cat / scatter bucket route), `max_over_ranks` (MAX all-reduce of a double on
    the device), `gather_outputs` (all_gather with the ragged-shard padding);
  * `gather_partials` (the context split's one exchange per cross block: `all_gather_into_tensor`);
  * bench.py's `distributed` branch end to end on RCCL (`HN_BENCH_FORCE_DIST=1 python bench.py --gpus 1`), and `--gpus N`
    beyond the visible GPUs exiting with one clear line instead of a rendezvous hang.

Serves the reference's only multi-GPU artefacts: healnet/main.py:464-465 (the optimizer step the all-reduce feeds) and
run_plan.sh:17-21.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KW = dict(n_modalities=2, channel_dims=[200, 96], num_spatial_axes=[1, 1], out_dims=4, depth=3, l_c=32, l_d=64, x_heads=4, l_heads=4,
          cross_dim_head=32, latent_dim_head=16)      # cfg4-like: one-token omic + a patch bag on the explicit K/V binding
KW_DEFAULT = dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4)      # cfg4 itself (11.9 M parameters)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(kw, n, seed, bag):
    gen = torch.Generator().manual_seed(seed)
    return ([torch.rand(n, 1, kw["channel_dims"][0], generator=gen), torch.rand(n, bag, kw["channel_dims"][1], generator=gen)],
            torch.randint(0, 4, (n,), generator=gen), torch.randint(0, 2, (n,), generator=gen))


def _worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HN_FORCE_COLLECTIVES="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import healnet_amd as hn
    from healnet_amd import dist as hd
    report = {}
    try:
        rank, world, local = hd.init_from_env()            # backend None -> nccl on a GPU box
        assert (rank, world, local) == (0, 1, 0)
        assert dist.is_initialized() and dist.get_backend() == "nccl" and hd.force_collectives()
        dev = torch.device("cuda", 0)
        calls = {"all_reduce": 0, "all_gather": 0}
        real_ar, real_ag = dist.all_reduce, dist.all_gather

        def counting_ar(t, *a, **k):
            calls["all_reduce"] += 1
            return real_ar(t, *a, **k)

        def counting_ag(parts, t, *a, **k):
            calls["all_gather"] += 1
            return real_ag(parts, t, *a, **k)

        dist.all_reduce, dist.all_gather = counting_ar, counting_ag
        dist.barrier()
        torch.cuda.synchronize(dev)

        # ---- plain helpers ------------------------------------------------------------------------------------------
        gen = torch.Generator().manual_seed(1)
        ts = [torch.randn(n, generator=gen).to(dev) for n in (7, 1000, 33, 4096 * 3 + 1)]
        want = [t.clone() for t in ts]
        hd.allreduce_mean_(ts, bucket_bytes=8000)          # several cat / scatter buckets + one single-tensor bucket
        torch.cuda.synchronize(dev)
        assert all(torch.equal(a, b) for a, b in zip(ts, want)), "allreduce_mean_ over one rank must be the identity"
        n_bucketed = calls["all_reduce"]
        assert n_bucketed >= 2, calls
        flat1 = torch.randn(1 << 20, generator=gen).to(dev)
        w1 = flat1.clone()
        hd.allreduce_mean_([flat1])                        # the in-place flat route (FlatParameters.grads)
        assert torch.equal(flat1, w1) and calls["all_reduce"] == n_bucketed + 1
        assert hd.max_over_ranks(3.25, dev) == 3.25
        assert calls["all_reduce"] == n_bucketed + 2
        loc = torch.randn(5, 4, generator=gen).to(dev)
        got = hd.gather_outputs(loc, 5)
        assert torch.equal(got, loc) and calls["all_gather"] == 1
        # the context split's exchange (healnet_amd.dist.gather_partials): ncclAllGather into one flat tensor, then the merge kernel
        # over the gathered parts -- one rank, so the merged block must equal the whole block bit for bit up to the merge's own rounding
        o = torch.randn(2, 32, 128, generator=gen).to(dev)
        st = torch.rand(2, 4, 32, 2, generator=gen).to(dev) + 0.5
        o_all, st_all = hd.gather_partials(o, st)
        torch.cuda.synchronize(dev)
        assert o_all.shape == (1, 2, 32, 128) and st_all.shape == (1, 2, 4, 32, 2)
        assert torch.equal(o_all[0], o) and torch.equal(st_all[0], st), "all_gather_into_tensor over one rank must return the input"
        report["context_split_gather_floats"] = int(o.numel() + st.numel())
        # the context split's TRAINING exchange (ABI v11): ncclAllGather of the shard's (P z | statistics) in the forward, ONE
        # ncclAllReduce(SUM) of the partial dx + the block's parameter gradients in the backward -- one rank holding the whole context,
        # so the gradients must be those of the plain block
        before = dict(calls)
        ts2 = [torch.randn(n, generator=gen).to(dev) for n in (5, 300)]
        w2 = [t.clone() for t in ts2]
        hd.allreduce_sum_(ts2)
        assert all(torch.equal(a, b_) for a, b_ in zip(ts2, w2)) and calls["all_reduce"] == before["all_reduce"] + 1
        torch.manual_seed(11)
        pn = hn.healnet.PreNorm(64, hn.Attention(64, 13, heads=4, dim_head=32), context_dim=13).to(dev)
        a_ = pn.fn
        wts = (pn.norm.weight, pn.norm.bias, pn.norm_context.weight, pn.norm_context.bias, a_.to_q.weight, a_.to_kv.weight, a_.to_out[0].weight,
               a_.to_out[0].bias)
        xq = torch.randn(2, 32, 64, generator=gen).to(dev)
        z = torch.ops.healnet_hip.encode_norm(torch.randn(2, 700, 1, 13, generator=gen).to(dev).reshape(2, 700, 13), 0, 0.0, False, 16)
        dyq = torch.randn(2, 32, 64, generator=gen).to(dev)
        x1 = xq.clone().requires_grad_(True)
        g_plain = torch.autograd.grad(torch.ops.healnet_hip.attention(x1, z, None, *wts, 4, True), (x1,) + wts, dyq)
        x2 = xq.clone().requires_grad_(True)
        y2 = hn.ops.ContextSplitAttentionFn.apply(lambda part, st_: hd.gather_partials(part.reshape(2, -1), st_), hd.allreduce_sum_, True, 4, z, x2, *wts)
        g_cp = torch.autograd.grad(y2, (x2,) + wts, dyq)
        torch.cuda.synchronize(dev)
        assert calls["all_reduce"] == before["all_reduce"] + 2
        for ga, gb in zip(g_cp, g_plain):
            assert float((ga - gb).abs().max()) <= 3e-4 * float(gb.abs().max()) + 1e-6, "context-split training step over RCCL"
        report["context_split_training_allreduce_floats"] = int(sum(t.numel() for t in g_cp))
        report["helpers"] = dict(calls)

        # ---- the overlapped gradient all-reduce on real backwards ----------------------------------------------------
        for name, kw, n, bag in (("cfg4_like", KW, 6, 300), ("cfg4_b8", KW_DEFAULT, 8, 4096)):
            torch.manual_seed(3)
            model = hn.HealNet(**kw).train().to(dev)
            flat = hn.train.flatten_parameters(model)
            ins, y, c = _inputs(kw, n, 5, bag)
            ins, y, c = [t.to(dev) for t in ins], y.to(dev), c.to(dev)
            # reference: the same step with no hook registered
            flat.zero_grad()
            hn.train.surv_nll_loss(model(list(ins)), y, c).loss.backward()
            torch.cuda.synchronize(dev)
            want = flat.grads.clone()
            assert float(want.abs().max()) > 0
            sync = hd.GradReadyAllReduce(model, flat)      # default reduce_fn: dist.all_reduce(AVG) on the side stream
            for it in range(3):                            # events re-recorded, communicator reused
                before = calls["all_reduce"]
                flat.zero_grad()
                hn.train.surv_nll_loss(model(list(ins)), y, c).loss.backward()
                sync.wait()
                torch.cuda.synchronize(dev)
                order = [i for i, _, _ in sync.launched]
                assert order == [2, 1, -1], order
                assert calls["all_reduce"] - before == 3, (calls, before)
                assert sum(hi - lo for _, lo, hi in sync.launched) == flat.numel
                assert torch.equal(flat.grads, want), f"{name} iteration {it}: AVG over one rank changed the gradient"
            # the blocking route on the same buffer, then a barrier between "timed loops" as bench.py does
            sync.close()
            flat.zero_grad()
            hn.train.surv_nll_loss(model(list(ins)), y, c).loss.backward()
            hd.allreduce_mean_([flat.grads])
            dist.barrier()
            torch.cuda.synchronize(dev)
            assert torch.equal(flat.grads, want)
            # and one optimizer step behind the overlapped reduce: the loss moves
            hn.ops.register_backward_hook(flat.grads, sync)
            opt = hn.train.FusedL1Adam(flat, lr=1e-3, l1=1e-5)
            losses = []
            for _ in range(4):
                opt.zero_grad()
                out = hn.train.surv_nll_loss(model(list(ins)), y, c)
                out.loss.backward()
                sync.wait()
                opt.step()
                losses.append(float(out.loss))
            assert all(l == l for l in losses) and losses[-1] < losses[0], losses
            sync.close()
            report[name] = {"gradient_floats": int(flat.numel), "buckets": [hi - lo for _, lo, hi in sync.launched], "losses": losses}
            del model, flat, opt, sync
        q.put(("ok", report))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("".join(traceback.format_exception(type(e), e, e.__traceback__)), report))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_one_rank_nccl_group_drives_the_real_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    status, report = q.get(timeout=900)
    p.join(timeout=120)
    assert status == "ok", status
    print("rccl 1-rank report:", json.dumps(report))
    assert report["helpers"]["all_reduce"] >= 4 and report["helpers"]["all_gather"] == 1
    assert len(report["cfg4_b8"]["buckets"]) == 3


def _clean_env(**extra):
    env = dict(os.environ, **extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bench_distributed_branch_on_rccl():
    """`HN_BENCH_FORCE_DIST=1 python bench.py --gpus 1`: nccl process group of one rank, dist.barrier() inside the timed regions,
    the MAX all-reduce of the step time, the overlapped AND the blocking gradient all-reduce and the all-reduce alone."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                          "--no-cpu-baseline", "--train-steps", "3"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=_clean_env(HN_BENCH_FORCE_DIST="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["config"]["global_batch"] == 32
    assert abs(j["value"] - 32 * 1000.0 / j["ms_per_step"]) <= 1e-3 * j["value"]
    t = j["train_step"]
    assert t["world_size"] == 1 and t["backend"] == "nccl"
    assert t["ms_per_step_blocking_allreduce"] > 0 and t["allreduce_alone_ms"] > 0
    assert sum(t["allreduce_buckets_floats"]) == t["gradient_floats"]
    print("bench (forced distributed, RCCL, 1 rank):", json.dumps({k: t[k] for k in ("ms_per_step", "ms_per_step_blocking_allreduce",
                                                                                       "allreduce_alone_ms", "backend")}))


def test_bench_more_gpus_than_visible_exits_with_one_clear_line():
    n = torch.cuda.device_count() + 2
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], cwd=ROOT, capture_output=True, text=True,
                         timeout=300, env=_clean_env())
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    err = [l for l in out.stderr.splitlines() if l.strip()]
    assert err and f"--gpus {n}" in err[-1] and "visible" in err[-1] and "Traceback" not in out.stderr, out.stderr[-2000:]
    # the same under an external launcher's environment (the driver's torchrun line): every rank refuses before any rendezvous
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], cwd=ROOT, capture_output=True, text=True,
                         timeout=300, env=dict(_clean_env(), RANK="0", WORLD_SIZE=str(n), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                                               MASTER_PORT=str(_free_port())))
    assert out.returncode != 0 and "visible" in out.stderr


def test_bench_under_the_drivers_launcher_line_on_rccl():
    """The driver's N > 1 command shape -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with N = 1 and HN_BENCH_FORCE_DIST=1: rendezvous environment from the launcher,
    nccl process group, barriers inside the timed regions, rank 0 prints ONE JSON line."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                          "--no-cpu-baseline", "--train-steps", "3"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=_clean_env(HN_BENCH_FORCE_DIST="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["train_step"]["backend"] == "nccl" and j["scaling"] == "weak"
