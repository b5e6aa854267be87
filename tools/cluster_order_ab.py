#!/usr/bin/env python3
"""Cluster-mode latent chains beside a foreign kernel: time and failure reports (evidence for DESIGN.md 4.x "cluster grids").

    python tools/cluster_order_ab.py                                   # the product's grid order
    HN_FORCE_CLUSTER_SPLIT_ORDER=1 python tools/cluster_order_ab.py    # the former order (members `tiles` workgroups apart)

For 0 / 16 / 32 / 64 foreign workgroups (tests/csrc/occupy.hip: each owns a CU's whole LDS) parked on a side stream: the
cfg4-shaped b = 8 training step (64 row tiles, clusters of 4 in forward and backward chains) and the cfg1 b = 4 forward, ms per
call on the compute stream, hn_cluster_status afterwards, and whether the results equal the quiet run bit for bit.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import healnet_amd as hn
from healnet_amd import _capi
from test_gpu_cluster import Occupier, _bag_step_setup, _step

DEV = "cuda:0"


def timed(fn, n):
    s = torch.cuda.current_stream()
    fn(); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    s.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


def main():
    _capi.cluster_config(0, enable=True, timeout_us=int(os.environ.get("AB_TIMEOUT_US", "20000")))
    rows = []
    model, ins, y, c = _bag_step_setup(hn)
    _step(hn, model, ins, y, c)
    _, quiet = _step(hn, model, ins, y, c)
    torch.manual_seed(911)
    fmodel = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to(DEV)
    gen = torch.Generator().manual_seed(912)
    fins = [torch.rand(4, 1, 2000, generator=gen).to(DEV), torch.rand(4, 224, 224, 3, generator=gen).to(DEV)]
    with torch.no_grad():
        fquiet = fmodel(list(fins)).clone()
    for wgs in (0, 16, 32, 64):
        _capi.cluster_status(0, acknowledge=True)
        _capi.cluster_config(0, enable=True)
        occ = Occupier(wgs, max_ms=60000) if wgs else None
        try:
            ms_step, (_, grads) = timed(lambda: _step(hn, model, ins, y, c), 5)
            with torch.no_grad():
                ms_fwd, out = timed(lambda: fmodel(list(fins)), 10)
            out = out.clone()
        except _capi.CoresidencyLost as e:
            ms_step = ms_fwd = float("nan"); grads = None; out = None
        finally:
            if occ:
                occ._stop()
        torch.cuda.synchronize()
        st = _capi.cluster_status(0, acknowledge=True)
        rows.append(dict(foreign_workgroups=wgs, train_step_cfg4_b8_ms=round(ms_step, 3), forward_cfg1_b4_ms=round(ms_fwd, 3),
                         lost=st["lost"], pending_at_end=st["pending"],
                         grads_bit_equal=None if grads is None else all(torch.equal(grads[k], quiet[k]) for k in quiet),
                         forward_bit_equal=None if out is None else bool(torch.equal(out, fquiet))))
    print(json.dumps(dict(order="split (former)" if os.environ.get("HN_FORCE_CLUSTER_SPLIT_ORDER") else "adjacent (product)",
                          timeout_us=int(os.environ.get("AB_TIMEOUT_US", "20000")), rows=rows)))


if __name__ == "__main__":
    main()
