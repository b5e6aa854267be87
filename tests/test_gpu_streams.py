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
