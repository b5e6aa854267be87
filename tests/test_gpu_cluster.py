"""GPU: the cluster-mode latent chains beside what N > 1 adds, and their failure signal (VERDICT r4 "next round" item 1).

Small batches run the latent chains as clusters of 2 / 4 workgroups per row tile that spin on each other's flags (chain.hip,
bchain.hip).  With more than one rank an RCCL kernel sits on a side stream WHILE those launches run (healnet_amd/dist.py
GradReadyAllReduce; the step it serves is healnet/main.py:464-467), i.e. part of the chip is not available to the grid.

  * a FOREIGN persistent kernel (tests/csrc/occupy.hip: N workgroups, each owning a CU's whole LDS) parked on a side stream for the
    whole duration of a cfg4-shaped b = 8 training step and of a cfg1 b = 4 forward: no NaN, nothing reported, results BIT-equal to
    the quiet run (the dispatch order of a cluster grid keeps a tile's members adjacent: chain_common.h cluster_decode);
  * the failure path itself, forced by fault injection (hn_cluster_config enable = 2): the loss is reported (hn_cluster_status), the next entry point returns
    HN_E_CORESIDENCY once, the Python op warns once and re-runs without clusters, a training loop survives with finite parameters,
    hn_l1_adam_step skips on the DEVICE while the word is set, GraphedStep drops and re-captures its graphs;
  * four processes started against a deliberately stale library: one relinks under the build lock, all load the fresh one
    (also run on the CPU box: tests/test_host_logic.py imports the same function).
"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import textwrap
import warnings

import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OCCUPY = os.path.join(ROOT, "tests", "_build", "libhn_occupy.so")


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


# ------------------------------------------------------------------------------------------------
# the foreign kernel
# ------------------------------------------------------------------------------------------------
class Occupier:
    """`workgroups` workgroups x `lds_bytes` of LDS parked on a side stream until `release()`.

    HIP maps its streams onto a handful of hardware queues (round-robin); a side stream that lands on the COMPUTE stream's queue would
    put every kernel of the test body behind the parked one (seen in the first full-suite run: the fifth stream of the process
    did).  After the launch a probe kernel on the compute stream must complete within a second; otherwise the foreign kernel is
    released and parked again on another stream."""

    def __init__(self, workgroups, lds_bytes=160 * 1024, max_ms=20000):
        import time
        if not os.path.exists(OCCUPY):
            import __graft_entry__ as g
            g.build_test_helpers()
        self.lib = C.CDLL(OCCUPY)
        self.lib.hn_occupy_alloc.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        self.lib.hn_occupy_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        self.lib.hn_occupy_free.argtypes = [C.c_void_p]
        self.n = workgroups
        probe = torch.zeros(64, device=DEV)
        torch.cuda.current_stream().synchronize()
        for attempt in range(12):
            host, dev = C.c_void_p(), C.c_void_p()
            assert self.lib.hn_occupy_alloc(C.byref(host), C.byref(dev)) == 0
            self.host, self.dev = host, dev
            self.ctl = (C.c_uint * 16).from_address(host.value)
            self.stream = torch.cuda.Stream(DEV)
            assert self.lib.hn_occupy_launch(dev, workgroups, lds_bytes, max_ms, self.stream.cuda_stream) == 0
            t0 = time.time()
            while self.ctl[1] < workgroups:                   # every workgroup is resident before the test body starts
                assert time.time() - t0 < 10.0, f"only {self.ctl[1]} of {workgroups} foreign workgroups started"
                time.sleep(0.001)
            probe.add_(1.0)
            ev = torch.cuda.Event()
            ev.record()
            t0 = time.time()
            while not ev.query() and time.time() - t0 < 1.0:
                time.sleep(0.001)
            if ev.query():
                return
            self._stop()                                      # same hardware queue as the compute stream: try the next stream
        raise AssertionError("no side stream with a hardware queue of its own")

    def _stop(self):
        self.ctl[0] = 1
        self.stream.synchronize()
        gave_up = int(self.ctl[2])
        self.lib.hn_occupy_free(self.host)
        return gave_up

    def release(self):
        gave_up = self._stop()
        assert gave_up == 0, f"{gave_up} foreign workgroups ran into their own time bound: the body took too long"


def _status():
    from healnet_amd import _capi
    return _capi.cluster_status(0)


def _bag_step_setup(hn, b=8, bag=4096, feat=768, depth=2):
    """BASELINE configs[3] shape (omic 1 x 2000 + patch bag 4096 x 768) at b = 8: 64 row tiles -> clusters of 4 both ways."""
    torch.manual_seed(901)
    model = hn.HealNet(n_modalities=2, channel_dims=[2000, feat], num_spatial_axes=[1, 1], out_dims=4, depth=depth).train().to(DEV)
    gen = torch.Generator().manual_seed(902)
    ins = [torch.rand(b, 1, 2000, generator=gen).to(DEV), torch.rand(b, bag, feat, generator=gen).to(DEV)]
    y = torch.randint(0, 4, (b,), generator=gen).to(DEV)
    c = torch.randint(0, 2, (b,), generator=gen).to(DEV)
    return model, ins, y, c


def _step(hn, model, ins, y, c):
    model.zero_grad(set_to_none=True)
    out = hn.train.surv_nll_loss(model(list(ins)), y, c)
    out.loss.backward()
    torch.cuda.current_stream().synchronize()                 # (never a DEVICE synchronize here: the foreign kernel is still running)
    return out.loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize("workgroups", [16, 32, 64])
def test_training_step_beside_a_foreign_kernel(hn, workgroups):
    model, ins, y, c = _bag_step_setup(hn)
    before = _status()
    assert before["enabled"], "cluster mode was switched off by an earlier test"
    _step(hn, model, ins, y, c)                                # allocator / workspaces / descriptor caches settle (no allocation below)
    loss_q, grads_q = _step(hn, model, ins, y, c)
    occ = Occupier(workgroups)
    try:
        loss_o, grads_o = _step(hn, model, ins, y, c)
        loss_o2, grads_o2 = _step(hn, model, ins, y, c)
    finally:
        occ.release()
    after = _status()
    assert not after["pending"] and after["lost"] == before["lost"] and after["fallbacks"] == before["fallbacks"], after
    assert after["enabled"]
    assert torch.isfinite(loss_o) and torch.equal(loss_o, loss_q) and torch.equal(loss_o2, loss_q)
    for k, g in grads_q.items():
        assert torch.isfinite(grads_o[k]).all(), k
        assert torch.equal(grads_o[k], g) and torch.equal(grads_o2[k], g), f"gradient of {k} differs beside {workgroups} foreign workgroups"


@pytest.mark.parametrize("workgroups", [16, 64])
def test_forward_cfg1_beside_a_foreign_kernel(hn, workgroups):
    """BASELINE configs[0]: tab 1 x 2000 + image 224 x 224 x 3 at b = 4 (32 row tiles: clusters of 4 in the inference chains)."""
    torch.manual_seed(911)
    model = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to(DEV)
    gen = torch.Generator().manual_seed(912)
    ins = [torch.rand(4, 1, 2000, generator=gen).to(DEV), torch.rand(4, 224, 224, 3, generator=gen).to(DEV)]
    before = _status()
    with torch.no_grad():
        model(list(ins))
        quiet = model(list(ins)).clone()
        torch.cuda.current_stream().synchronize()
        occ = Occupier(workgroups)
        try:
            outs = [model(list(ins)).clone() for _ in range(3)]
            torch.cuda.current_stream().synchronize()
        finally:
            occ.release()
    after = _status()
    assert not after["pending"] and after["lost"] == before["lost"], after
    for o in outs:
        assert torch.isfinite(o).all() and torch.equal(o, quiet)


# ------------------------------------------------------------------------------------------------
# the failure path, forced
# ------------------------------------------------------------------------------------------------
@pytest.fixture
def tiny_wait(hn):
    """Fault injection (hn_cluster_config enable = 2: the last member of every tile withholds its flag) with a 200 us wait bound:
    every cluster launch loses an exchange and reports it; restored afterwards."""
    from healnet_amd import _capi
    torch.cuda.synchronize()
    _capi.cluster_status(0, acknowledge=True)
    _capi.cluster_config(0, enable=True, timeout_us=200, inject_loss=True)
    yield _capi
    torch.cuda.synchronize()
    _capi.cluster_status(0, acknowledge=True)
    _capi.cluster_config(0, enable=True, timeout_us=0)
    _capi._cluster_events["warned"] = False


def _small(hn, train):
    torch.manual_seed(921)
    kw = dict(n_modalities=2, channel_dims=[300, 64], num_spatial_axes=[1, 1], out_dims=4, depth=2)
    model = hn.HealNet(**kw)
    model = (model.train() if train else model.eval()).to(DEV)
    gen = torch.Generator().manual_seed(922)
    ins = [torch.rand(8, 1, 300, generator=gen).to(DEV), torch.rand(8, 40, 64, generator=gen).to(DEV)]
    return kw, model, ins


def test_lost_exchange_is_reported_and_the_forward_re_runs_without_clusters(hn, tiny_wait):
    _capi = tiny_wait
    kw, model, ins = _small(hn, train=False)
    base = _capi.cluster_status(0)
    with torch.no_grad():
        poisoned = model(list(ins))
        torch.cuda.synchronize()
        st = _capi.cluster_status(0)
        assert st["pending"] and st["enabled"], f"the injected fault did not trip any wait: {st}"
        assert not torch.isfinite(poisoned).all(), "a tile that gave up must not deliver an incomplete sum"
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            good = model(list(ins))                            # HN_E_CORESIDENCY -> warn, clusters off, run again
            torch.cuda.synchronize()
        assert any("cluster" in str(x.message) for x in w), [str(x.message) for x in w]
        st = _capi.cluster_status(0)
        assert not st["pending"] and not st["enabled"] and st["lost"] > base["lost"] and st["fallbacks"] == base["fallbacks"] + 1, st
        assert torch.isfinite(good).all()
        again = model(list(ins))
        assert torch.equal(again, good)
    from oracle import healnet_cpu as O
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.cpu() for t in ins])
    assert_close(good.cpu(), want, rel=1e-3, floor=0.0, abs_floor=1e-5, what="forward after the cluster fallback")


def test_c_abi_returns_coresidency_once(hn, tiny_wait):
    """The status the C caller sees: HN_E_CORESIDENCY (-6) from the next fused entry point, with a message; then HN_OK."""
    _capi = tiny_wait
    kw, model, ins = _small(hn, train=False)
    with torch.no_grad():
        model(list(ins))
        torch.cuda.synchronize()
        assert _capi.cluster_status(0)["pending"]
        # straight through the ABI: a workspace-less call is enough -- the poll comes before every other check
        rc = _capi.lib().hn_l1_adam_step(None, None, None, None, 0, 0.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1, None, None, 0, None)
        assert rc == _capi.HN_E_CORESIDENCY
        msg = _capi.lib().hn_last_error_string().decode()
        assert "co-residency" in msg and "hn_l1_adam_step" in msg, msg
        st = _capi.cluster_status(0)
        assert not st["pending"] and not st["enabled"]
        rc = _capi.lib().hn_l1_adam_step(None, None, None, None, 0, 0.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1, None, None, 0, None)
        assert rc != _capi.HN_E_CORESIDENCY               # (a plain argument error now)
        out = model(list(ins))
        assert torch.isfinite(out).all()


def test_training_loop_survives_a_lost_exchange(hn, tiny_wait):
    _capi = tiny_wait
    _, model, ins = _small(hn, train=True)
    flat = hn.train.flatten_parameters(model)
    opt = hn.train.FusedL1Adam(flat, lr=1e-3, l1=1e-5)
    y = torch.tensor([0, 1, 2, 3, 0, 1, 2, 3], device=DEV)
    c = torch.tensor([0, 1, 0, 1, 0, 1, 0, 1], device=DEV)
    p0 = flat.params.clone()

    def body():
        opt.zero_grad()
        out = hn.train.surv_nll_loss(model(list(ins)), y, c)
        out.loss.backward()
        return out.loss

    losses = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(4):
            loss = hn.train.retry_step(body, retries=2)
            opt.step()
            torch.cuda.synchronize()
            losses.append(float(loss))
    st = _capi.cluster_status(0)
    assert st["lost"] >= 1 and not st["enabled"] and not st["pending"], st
    assert torch.isfinite(flat.params).all(), "a poisoned gradient reached the parameters"
    assert torch.isfinite(opt.exp_avg).all() and torch.isfinite(opt.exp_avg_sq).all()
    assert not torch.equal(flat.params, p0), "no step was applied at all"
    assert losses[-1] == losses[-1] and losses[-1] < 10.0, losses


def _set_word(_capi, value):
    info = _capi.ClusterInfo()
    _capi.check(_capi.lib().hn_cluster_status(0, 0, C.byref(info)), "hn_cluster_status")
    assert bool(info.status_word), "the device has no status word yet"
    info.status_word[0] = value


def test_adam_step_skips_on_the_device_while_the_word_is_set(hn):
    """An hn_l1_adam_step enqueued BEFORE the host learns of a loss (here: a captured one, replayed) must leave parameters and
    moments alone while the status word is non-zero, and apply normally once it is clear."""
    from healnet_amd import _capi
    torch.cuda.synchronize()
    _capi.cluster_status(0, acknowledge=True)
    enabled = _capi.cluster_status(0)["enabled"]
    _, model, ins = _small(hn, train=True)
    flat = hn.train.flatten_parameters(model)
    opt = hn.train.FusedL1Adam(flat, lr=1e-2, l1=1e-5)
    flat.grads.normal_()
    side = torch.cuda.Stream(DEV)
    with torch.cuda.stream(side):
        opt.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        opt.step()
    p0, m0, v0 = flat.params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()
    try:
        _set_word(_capi, 4242)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(flat.params, p0) and torch.equal(opt.exp_avg, m0) and torch.equal(opt.exp_avg_sq, v0)
        reg = float(opt.reg_loss)
        assert abs(reg - 1e-5 * float(flat.params.abs().sum())) <= 1e-4 * abs(reg) + 1e-9      # reg_loss is still produced
    finally:
        _set_word(_capi, 0)
    g.replay()
    torch.cuda.synchronize()
    assert not torch.equal(flat.params, p0) and not torch.equal(opt.exp_avg, m0)
    assert _capi.cluster_status(0)["enabled"] == enabled


def test_graphed_step_re_captures_after_a_reported_loss(hn):
    from healnet_amd import _capi
    torch.cuda.synchronize()
    _capi.cluster_status(0, acknowledge=True)
    _capi.cluster_config(0, enable=True, timeout_us=0)
    _, model, ins = _small(hn, train=True)
    flat = hn.train.flatten_parameters(model)
    y = torch.tensor([0, 1, 2, 3, 0, 1, 2, 3], device=DEV)
    c = torch.tensor([0, 1, 0, 1, 0, 1, 0, 1], device=DEV)
    loss_fn = lambda logits, yy, cc: hn.train.surv_nll_loss(logits, yy, cc).loss      # noqa: E731
    step = hn.train.GraphedStep(model, loss_fn, ins, (y, c))
    loss_a, _ = step(ins, (y, c))
    loss_a, grads_a = loss_a.clone(), flat.grads.clone()
    assert step.captures == 1 and _capi.cluster_status(0)["enabled"]
    try:
        _set_word(_capi, 777)                                  # what a replayed cluster launch that lost an exchange leaves behind
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            loss_b, _ = step(ins, (y, c))
        torch.cuda.synchronize()
        st = _capi.cluster_status(0)
        assert step.captures == 2 and not st["enabled"] and not st["pending"], (step.captures, st)
        assert any("cluster" in str(x.message) for x in w) or _capi._cluster_events["warned"]
        # the same arithmetic without the cluster exchange: equal to fp32 summation noise
        assert_close(loss_b.cpu(), loss_a.cpu(), rel=1e-5, floor=0.0, abs_floor=1e-6, what="loss after re-capture")
        scale = float(grads_a.abs().max())
        assert float((flat.grads - grads_a).abs().max()) <= 2e-5 * scale
    finally:
        _set_word(_capi, 0)
        _capi.cluster_config(0, enable=True, timeout_us=0)
        _capi._cluster_events["warned"] = False
        step.close()


# ------------------------------------------------------------------------------------------------
# N ranks against a stale library
# ------------------------------------------------------------------------------------------------
def run_stale_library_race(tmp_path, procs=4):
    """Copy the package (sources, objects, library) to a private tree, corrupt the library's embedded build id, start `procs`
    processes that all call _capi.lib() at once: every one must end up with the fresh library, exactly one does the (re)link."""
    from healnet_amd import _capi
    tree = os.path.join(str(tmp_path), "tree")
    os.makedirs(os.path.join(tree, "include"))
    shutil.copy(os.path.join(ROOT, "include", "healnet_hip.h"), os.path.join(tree, "include"))
    shutil.copytree(os.path.join(ROOT, "healnet_amd"), os.path.join(tree, "healnet_amd"),
                    ignore=shutil.ignore_patterns("__pycache__", "*.tmp.*", ".lock"))
    lib_path = os.path.join(tree, "healnet_amd", "libhealnet_hip.so")
    if shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None:
        pytest.skip("no hipcc on this machine")
    if not os.path.isdir(os.path.join(tree, "healnet_amd", "build")):
        pytest.skip("no object cache travelled with the tree (a full 15-unit compile is not what this test is about)")
    want = _capi.source_build_id()
    blob = bytearray(open(lib_path, "rb").read())
    at = blob.find(b"HN_BUILD_ID=")
    assert at > 0
    blob[at + 12:at + 28] = b"0" * 16                          # a library "built from other sources"
    with open(lib_path, "wb") as f:
        f.write(blob)
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        from healnet_amd import _capi
        lib = _capi.lib()
        print("ID", lib.hn_build_id().decode(), lib.hn_abi_version())
    """ % tree)
    env = dict(os.environ)
    env.pop("HN_LIB_PATH", None)
    ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=tree)
          for _ in range(procs)]
    outs = [p.communicate(timeout=900) for p in ps]
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0, se[-2000:]
        assert ("ID %s %d" % (want, _capi.HN_ABI_VERSION)) in so, (so, se[-500:])
    built = sum(se.count("healnet_amd: built ") for _, se in outs)
    noticed = sum(se.count("rebuilding") for _, se in outs)
    assert built == 1, f"{built} processes relinked the library (stderr: {[se[-300:] for _, se in outs]})"
    assert noticed >= 1
    assert _capi.library_build_id(lib_path) == want
    leftovers = [f for f in os.listdir(os.path.join(tree, "healnet_amd")) if ".tmp." in f]
    assert not leftovers, leftovers


def test_four_processes_against_a_stale_library(tmp_path):
    run_stale_library_race(tmp_path, procs=4)
