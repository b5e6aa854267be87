"""Host-side runtime helpers shared by the operator registrations (ops.py) and the module surface (healnet.py):
pointer marshalling rules of the C ABI and the per-(device, stream) scratch the library never allocates itself."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"healnet_amd: {what} must live on a HIP device (got {t.device}); "
                           "the MI355X path has no CPU fallback")


def f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32, contiguous view of an input (positions are always computed in fp32, Appendix B-8); uint8 means byte / 255."""
    if t.dtype == torch.uint8:
        t = t.float().div(255)
    elif t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address of a parameter / auxiliary tensor handed to the C ABI, which reads fp32, dense, row-major memory:
    anything else (model.half() / .bfloat16() / .double(), a transposed view) would be read as garbage -- refuse it."""
    if t is None:
        return None
    if t.is_floating_point() and t.dtype != torch.float32:
        raise TypeError(f"healnet_amd: parameters must be float32 (got {t.dtype}); the kernels read fp32 memory -- keep the module in "
                        "fp32 (bf16 modality TENSORS are fine: pass them to forward as they are)")
    if not t.is_contiguous():
        raise ValueError("healnet_amd: parameters / auxiliary tensors must be contiguous")
    return t.data_ptr()


_POISON = os.environ.get("HN_POISON_WS", "0") == "1"


class Workspace:
    """One growing scratch allocation per (device, stream) (the C ABI never allocates).  Per stream, because calls on
    different streams of one device may run concurrently (two micro-batches, a serving thread per stream) and must not
    share scratch; calls on one stream are ordered, so they can."""

    def __init__(self) -> None:
        self._buf: Dict[tuple, torch.Tensor] = {}

    def get(self, device: torch.device, nbytes: int) -> torch.Tensor:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # Under graph capture an allocation comes from the CAPTURE'S private memory pool.  Caching it here would outlive the
            # graph that owns the pool: the next capture on the same stream found the buffer, baked its address into its kernels,
            # and a replay faulted once the first graph -- and with it the pool -- had been destroyed (round 5: a GraphedForward
            # followed by a GraphedStep in one process; "Memory access fault by GPU node").  A captured call gets scratch of its
            # own, like any temporary a torch op allocates while capturing: it returns to the pool when the call ends, and stream
            # order inside the graph keeps later users of the block behind this call's kernels.
            return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        if _POISON:      # development aid (HN_POISON_WS=1): every call starts from an all-NaN workspace, so a kernel that
            buf.fill_(0xFF)   # reads scratch it has not written shows up deterministically
        return buf


WS = Workspace()
WS_AUX = Workspace()   # for on-demand attention-weight export (must not clobber the forward scratch)


def mask_bytes(mask: Optional[torch.Tensor], b: int, n: int) -> Optional[torch.Tensor]:
    if mask is None:
        return None
    flat = mask.reshape(mask.shape[0], -1)
    if flat.shape[0] != b or flat.shape[1] != n:
        raise ValueError(f"mask of shape {tuple(mask.shape)} does not match context tokens (b={b}, N={n})")
    return flat.to(torch.uint8).contiguous()
