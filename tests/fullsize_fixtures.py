"""Shared by tools/gen_goldens_fullsize.py (build container: runs the REFERENCE at the BASELINE sizes) and tests/test_gpu_fullsize.py
(GPU box: compares the HIP path with what the reference produced): the seeded inputs of every full-size case and the compressed form
in which a full-size GRADIENT is committed.

A gradient of the default model is 9.6 M floats (38 MB): too large for a fixture.  Per parameter tensor the fixture keeps
  norm, scale      the L2 norm and the largest magnitude of the reference gradient
  sketch           a count sketch: P = 128 buckets, bucket j = sum over the elements i = j (mod P) of s_i * g_i with one seeded random
                   sign s_i per element.  The sketch is linear, so sketch(g_hip) - sketch(g_ref) is the sketch of the difference, and
                   the sum of its squares is an unbiased estimate of |g_hip - g_ref|^2 (relative standard deviation ~ sqrt(2 / 128))
  idx, vals        1024 seeded random elements (all of them for smaller tensors), exactly: the element-wise criteria on a sample
which is 125 x ~2.3 K floats = 1.1 MB per case."""
import zlib

import torch

SKETCH_BUCKETS = 128
SAMPLE = 1024

CFG2 = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
CFG3 = dict(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4)
CFG5 = dict(n_modalities=4, channel_dims=[2000, 768, 768, 3], num_spatial_axes=[1, 1, 1, 3], out_dims=4, depth=8)


def _key_seed(key: str) -> int:
    return zlib.crc32(key.encode()) & 0x7FFFFFFF


def signs(key: str, n: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(_key_seed(key))
    return torch.randint(0, 2, (n,), generator=g, dtype=torch.int8).to(torch.float64) * 2 - 1


def sketch(t: torch.Tensor, key: str) -> torch.Tensor:
    """(SKETCH_BUCKETS,) float64 count sketch of a tensor (any device; computed on the CPU in float64)."""
    v = t.detach().double().cpu().reshape(-1)
    n = v.numel()
    v = v * signs(key, n)
    pad = (-n) % SKETCH_BUCKETS
    if pad:
        v = torch.cat([v, v.new_zeros(pad)])
    return v.view(-1, SKETCH_BUCKETS).sum(0)


def sample_index(key: str, n: int) -> torch.Tensor:
    if n <= SAMPLE:
        return torch.arange(n)
    g = torch.Generator().manual_seed(_key_seed(key) ^ 0x5A5A5A)
    return torch.randperm(n, generator=g)[:SAMPLE].sort().values


def compress_grad(t: torch.Tensor, key: str) -> dict:
    v = t.detach().double().cpu().reshape(-1)
    idx = sample_index(key, v.numel())
    return {"norm": v.norm().reshape(1), "scale": v.abs().max().reshape(1), "sketch": sketch(v, key), "vals": v[idx].float()}


# ---- seeded inputs (identical on both sides; torch's CPU generator is deterministic for a given build) ----
def bench_inputs(b: int = 32):
    """bench.py's own inputs at cfg2: generator 1234, tab then img."""
    gen = torch.Generator().manual_seed(1234)
    return [torch.rand(b, 1, 2000, generator=gen), torch.rand(b, 224, 224, 3, generator=gen)]


def cfg2_train_inputs(b: int = 32):
    gen = torch.Generator().manual_seed(4132)
    ins = [torch.rand(b, 1, 2000, generator=gen), torch.rand(b, 224, 224, 3, generator=gen)]
    dl = torch.randn(b, 4, generator=gen)
    return ins, dl


def cfg5_cut_inputs():
    gen = torch.Generator().manual_seed(4105)
    return [torch.rand(2, 1, 2000, generator=gen), torch.rand(2, 4096, 768, generator=gen),
            torch.rand(2, 4096, 768, generator=gen), torch.rand(2, 4, 224, 224, 3, generator=gen)]


def cfg5_full_inputs():
    gen = torch.Generator().manual_seed(4106)
    return [torch.rand(1, 1, 2000, generator=gen), torch.rand(1, 4096, 768, generator=gen),
            torch.rand(1, 4096, 768, generator=gen), torch.rand(1, 12, 224, 224, 3, generator=gen)]


def cfg3_inputs(b: int = 16):
    gen = torch.Generator().manual_seed(1234)
    tab = torch.rand(b, 1, 2000, generator=gen).to(torch.bfloat16)
    img = torch.rand(b, 224, 224, 3, generator=gen).to(torch.bfloat16)
    vol = torch.rand(b, 12, 224, 224, 3, generator=gen).to(torch.bfloat16)
    return [tab, img, vol]
