"""Host-side input container mirroring ``healnet.etl.MMDataset`` (reference healnet/etl/loaders.py:21-41):
a list of per-modality tensors indexed by sample, with an optional target."""
from typing import Iterable, List, Optional

import torch
from torch.utils.data import Dataset


class MMDataset(Dataset):
    def __init__(self, tensors: List[torch.Tensor], target: Optional[torch.Tensor] = None):
        if len(tensors) == 0:
            raise ValueError("MMDataset needs at least one modality tensor")
        n = tensors[0].shape[0]
        for t in tensors:
            if t.shape[0] != n:
                raise ValueError("all modalities must hold the same number of samples")
        self.tensors = tensors
        self.target = target

    def __len__(self) -> int:
        return int(self.tensors[0].shape[0])

    def __getitem__(self, idx):
        sample = [t[idx] for t in self.tensors]
        return sample if self.target is None else (sample, self.target[idx])


# ------------------------------------------------------------------------------------------------
# Input staging: DataLoader batch -> device, replacing the blocking
# `features = [feat.to(self.device) for feat in features]` / `censorship.to(...)` of healnet/main.py:415-419
# ------------------------------------------------------------------------------------------------
def _map_tensors(obj, fn):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    return obj


class DeviceLoader:
    """Wraps any iterable of batches (the reference's ``DataLoader(train, pin_memory=True, ...)`` yielding
    ``(features, censorship, event_time, y_disc)``, healnet/main.py:214-223,412) and yields the same structure with every
    tensor resident on ``device``.  A producer thread stages batch k+1 (.. k+depth) into a ring of PRE-ALLOCATED device
    buffers on a dedicated copy stream while batch k computes (no per-batch device allocation, no pinning on the critical
    path); the consumer stream only waits on the copy's event, and a slot is refilled only after the consumer has moved
    past it.  A yielded batch stays valid until the next one is requested (the reference's loop never keeps one longer).

    ``transport`` narrows floating-point modality tensors on the HOST before the copy:
      None / "fp32"  as produced by the dataset
      "bf16"         fp32 -> bf16 through a pinned buffer (half the PCIe bytes); the model reads bf16 tensors in place.
                     The cast runs on the producer thread and is slower than the PCIe time it saves on a 64 GB/s link
                     (tools/bench_staging.py: 1.6k instead of 9.5k samples/s at cfg2) -- only for links that are the
                     bottleneck; a dataset that is already stored in bf16 needs no option and costs nothing.
    uint8 tensors (8-bit images) are always shipped as they are (a quarter of the bytes) and decoded as byte / 255 by the
    encode kernel.  Integer / bool tensors (labels, censorship) are copied unchanged."""

    def __init__(self, loader: Iterable, device, depth: int = 2, transport: Optional[str] = None):
        if transport not in (None, "fp32", "bf16"):
            raise ValueError("transport must be None, 'fp32' or 'bf16'")
        self.loader, self.device, self.depth, self.transport = loader, torch.device(device), max(1, int(depth)), transport
        if self.device.type != "cuda":
            raise RuntimeError("healnet_amd.etl.DeviceLoader stages to a HIP device; there is no CPU path")
        self._stream = torch.cuda.Stream(self.device)
        n = self.depth + 1
        self._dev = [dict() for _ in range(n)]        # slot -> tensor index -> device buffer
        self._pin = [dict() for _ in range(n)]        # slot -> tensor index -> pinned staging buffer (casts only)
        self._consumed = {}                           # batch index -> event on the consumer stream: no longer in use
        self._cv = None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch, k: int, stop):
        n = self.depth + 1
        slot = k % n
        if k >= n:                  # the slot still belongs to batch k - n until the consumer has moved past it
            with self._cv:
                while (k - n) not in self._consumed and not stop.is_set():
                    self._cv.wait(timeout=0.1)
                ev_done = self._consumed.pop(k - n, None)
            if ev_done is not None:
                ev_done.synchronize()
        dev, pin = self._dev[slot], self._pin[slot]
        counter = [0]

        def move(t: torch.Tensor) -> torch.Tensor:
            if t.is_cuda:
                return t
            idx = counter[0]
            counter[0] += 1
            dt = torch.bfloat16 if (self.transport == "bf16" and t.dtype == torch.float32 and t.dim() >= 3) else t.dtype
            d = dev.get(idx)
            if d is None or d.shape != t.shape or d.dtype != dt:
                d = torch.empty(t.shape, dtype=dt, device=self.device)
                dev[idx] = d
            src = t
            if dt != t.dtype:                                      # host-side narrowing, one pass into pinned memory
                h = pin.get(idx)
                if h is None or h.shape != t.shape or h.dtype != dt:
                    h = torch.empty(t.shape, dtype=dt, pin_memory=True)
                    pin[idx] = h
                h.copy_(t)
                src = h
            d.copy_(src, non_blocking=True)
            return d

        with torch.cuda.stream(self._stream):
            out = _map_tensors(batch, move)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return out, ev, k

    def __iter__(self):
        import queue
        import threading
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        self._cv = threading.Condition()
        self._consumed = {}
        # The slot ring is reused across epochs: kernels of the previous epoch's last batches (the backward re-reads the
        # inputs to recompute the contexts) may still be in flight on the consumer's stream when batch 0 of this epoch is
        # copied into slot 0 -- order the copy stream behind everything the consumer has enqueued so far.
        epoch_start = torch.cuda.Event()
        epoch_start.record(torch.cuda.current_stream(self.device))
        self._stream.wait_event(epoch_start)

        def produce():
            torch.cuda.set_device(self.device)
            try:
                for k, batch in enumerate(self.loader):
                    if stop.is_set():
                        return
                    item = self._stage(batch, k, stop)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                q.put(None)
            except BaseException as e:      # surface loader errors in the consumer
                q.put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        prev = None
        try:
            while True:
                item = q.get()
                cur = torch.cuda.current_stream(self.device)
                if prev is not None:                               # everything enqueued for the previous batch
                    done = torch.cuda.Event()
                    done.record(cur)
                    with self._cv:
                        self._consumed[prev] = done
                        self._cv.notify_all()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                batch, ev, prev = item
                cur.wait_event(ev)
                yield batch
        finally:
            stop.set()
            th.join(timeout=5.0)


def bag_padding_mask(bag: torch.Tensor) -> torch.Tensor:
    """(b, N) key mask of a zero-padded patch bag ``(b, N, C)`` (the TCGA loader pads every slide to ``max_patches`` rows of
    zeros, healnet/etl/tasks.py:159-192): True where the row holds a real patch.  OPTIONAL: the reference attends to the
    padding rows as well; pass this as ``mask=`` only when that behaviour is not wanted."""
    return (bag != 0).any(dim=-1)
