"""GPU: concurrent forwards on several HIP streams of one device.

Small batches run the latent chains in CLUSTER mode (chain.hip / bchain.hip: the workgroups of a row tile exchange partial
sums through flags they spin on, which needs all of them resident at once).  Two such launches from different streams could
starve each other's members (`tools/two_stream.py` ran four streams of b = 16 into the spin limit: 2 s per step and
NaN rows).  `cluster_stream_guard` orders every cluster launch behind the previous one of the device when that went to another
stream; these tests run exactly that pattern and ask for the single-stream bits, and for forward + backward from two threads."""
import threading
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


def _model(hn):
    torch.manual_seed(0)
    return hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).to(DEV)


@pytest.mark.parametrize("parts,b", [(4, 16), (2, 16), (4, 4)])
def test_interleaved_streams_reproduce_the_single_stream_bits(hn, parts, b):
    model = _model(hn).eval()
    torch.manual_seed(1)
    chunks = [(torch.rand(b, 1, 2000, device=DEV), torch.rand(b, 64, 48, 3, device=DEV)) for _ in range(parts)]
    with torch.no_grad():
        want = [model([t, i]).clone() for t, i in chunks]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(parts)]
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        t0 = time.time()
        for _ in range(6):
            got = []
            for s, (t, i) in zip(streams, chunks):
                with torch.cuda.stream(s):
                    got.append(model([t, i]))
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 6
    assert dt < 0.2, f"{dt * 1e3:.1f} ms per round of {parts} forwards: cluster members are starving each other"
    for g, w in zip(got, want):
        assert torch.isfinite(g).all()
        assert torch.equal(g, w)


def test_training_steps_from_two_threads(hn):
    """forward + backward of a small batch (cluster chains both ways) on a stream per thread: gradients of each thread equal the
    ones of the same step run alone"""
    models = [_model(hn).train() for _ in range(2)]
    torch.manual_seed(2)
    data = [(torch.rand(4, 1, 2000, device=DEV), torch.rand(4, 40, 40, 3, device=DEV)) for _ in range(2)]

    def grads(m, d):
        for p in m.parameters():
            p.grad = None
        m(list(d)).square().sum().backward()
        return [p.grad.clone() for p in m.parameters()]

    want = [grads(m, d) for m, d in zip(models, data)]
    torch.cuda.synchronize()
    got, errs = [None, None], []

    def work(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(5):
                    g = grads(models[k], data[k])
            s.synchronize()
            got[k] = g
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for k in range(2):
        for g, w in zip(got[k], want[k]):
            assert torch.equal(g, w)


def test_kernel_timers_armed_on_one_stream_beside_a_second_thread(hn):
    """The library's one process-wide registration (hn_set_kernel_timers, include/healnet_hip.h) beside a concurrent caller
    (VERDICT r5 item 5): thread A arms a table for ITS stream and runs patch-bag forwards (gemm_nt_x6: the K/V projection of a bag of 8192 rows);
    thread B runs the same forwards on another stream at the same time, and a third thread keeps re-arming / clearing the table.
    A's entry must count exactly A's launches (B's are on another stream: neither timed nor counted), both threads must reproduce
    the bits of the quiet run, and nobody may crash on a half-published table."""
    import ctypes as C
    from healnet_amd import _capi
    torch.manual_seed(3)
    model = hn.HealNet(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4, depth=2).eval().to(DEV)
    model.keep_attention_stats = False
    gen = torch.Generator().manual_seed(4)
    ins = [torch.rand(2, 1, 2000, generator=gen).to(DEV), torch.rand(2, 4096, 768, generator=gen).to(DEV)]
    lib = _capi.lib()
    with torch.no_grad():
        quiet = model(list(ins)).clone()
    torch.cuda.synchronize()
    n_ev, rounds = 64, 6
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(n_ev)] for _ in range(2)]
    for row in evs:
        for e in row:
            e.record()                                 # (torch creates the hipEvent_t lazily)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    table = (_capi.KernelTimer * 1)()
    start = (C.c_void_p * n_ev)(*[e.cuda_event for e in evs[0]])
    stop = (C.c_void_p * n_ev)(*[e.cuda_event for e in evs[1]])
    table[0].kernel = b"gemm_nt_x6"
    table[0].ev_start, table[0].ev_stop = C.cast(start, C.POINTER(C.c_void_p)), C.cast(stop, C.POINTER(C.c_void_p))
    table[0].n_events, table[0].n_recorded, table[0].stream = n_ev, 0, sa.cuda_stream
    # how many launches of the class one forward makes: measured alone first
    assert lib.hn_set_kernel_timers(table, 1) == 0
    with torch.no_grad(), torch.cuda.stream(sa):
        model(list(ins))
    sa.synchronize()
    per_forward = int(table[0].n_recorded)
    assert per_forward >= 1, "the patch-bag projection did not run on gemm_nt_x6: the test lost its subject"
    table[0].n_recorded = 0
    outs, errs, stop_flag = {}, [], threading.Event()
    other = (_capi.KernelTimer * 1)()                  # what the third thread arms in between: a class nobody launches
    other[0].kernel = b"no_such_kernel"
    other[0].n_events = 0

    def work(tag, stream):
        try:
            with torch.no_grad(), torch.cuda.stream(stream):
                for _ in range(rounds):
                    y = model(list(ins))
            stream.synchronize()
            outs[tag] = y.clone()
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    def rearm():
        k = 0
        while not stop_flag.is_set():
            lib.hn_set_kernel_timers(other, 1) if k % 3 == 2 else lib.hn_set_kernel_timers(table, 1)
            k += 1
        lib.hn_set_kernel_timers(table, 1)

    # pass 1: A armed, B beside it, table stable -> exact count
    ts = [threading.Thread(target=work, args=("a", sa)), threading.Thread(target=work, args=("b", sb))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert int(table[0].n_recorded) == rounds * per_forward, (int(table[0].n_recorded), rounds, per_forward)
    assert torch.equal(outs["a"], quiet) and torch.equal(outs["b"], quiet)
    # pass 2: the table re-armed continuously from a third thread while both run -> no crash, same bits, never more than A's launches
    table[0].n_recorded = 0
    r = threading.Thread(target=rearm)
    ts = [threading.Thread(target=work, args=("a", sa)), threading.Thread(target=work, args=("b", sb))]
    r.start()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop_flag.set()
    r.join()
    lib.hn_set_kernel_timers(None, 0)
    torch.cuda.synchronize()
    assert not errs, errs
    assert 0 <= int(table[0].n_recorded) <= rounds * per_forward
    assert torch.equal(outs["a"], quiet) and torch.equal(outs["b"], quiet)
    ms = evs[0][0].elapsed_time(evs[1][0])
    assert ms > 0.0
