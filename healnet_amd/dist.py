"""Data-parallel helpers: one process per GPU (``torch.distributed``; backend ``nccl`` is RCCL over xGMI on
ROCm, ``gloo`` in the CPU tests).

The fusion forward has no cross-sample coupling (per-token LayerNorm, per-row softmax, per-sample mean), so the
batch is sharded contiguously across ranks and the forward needs NO collective.  The only exchange step of the
path is the gradient average of a training step: ``allreduce_mean_`` does it with a few large
flat buckets (the xGMI mesh is point-to-point, 7 links per GPU: a handful of multi-MB messages beat 125 small ones).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def force_collectives() -> bool:
    """HN_FORCE_COLLECTIVES=1: a process group of ONE rank still issues every collective (instead of the world == 1 early
    returns below).  This is how the RCCL code path -- ``init_process_group("nccl", device_id=...)``, ``ReduceOp.AVG`` on a side
    stream from inside the hn_grad_ready host callback, all_gather, MAX all-reduce, barrier -- is exercised on a 1-GPU box
    (tests/test_gpu_rccl.py, ``HN_BENCH_FORCE_DIST=1 python bench.py --gpus 1``)."""
    return os.environ.get("HN_FORCE_COLLECTIVES", "0") == "1"


def _active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives())


_cluster_off_devices = set()


def keep_ranks_in_step(device: Optional[torch.device]) -> None:
    """Data-parallel training with more than one rank: switch the latent chains' CLUSTER mode off on this rank's device.

    The failure handling of a cluster launch is per process (include/healnet_hip.h "failure signal": the rank that lost an
    exchange gets NaN rows, its hn_l1_adam_step skips, the host repeats the step).  Under a gradient all-reduce that is not
    enough: the NaN gradients of that rank have been summed into every peer before any host sees the report, the peers' status
    words are clean, they apply the step, and the ranks diverge for good.  The case the mode's bounded wait exists for
    -- an RCCL kernel holding CUs beside the chain -- is exactly the data-parallel step, so the ranks of a group of more than one
    run the same arithmetic without clusters (slower latent side at b <= 8 per rank, identical results), and a lost exchange can
    no longer happen.  Called by every gradient-averaging entry of this module; idempotent."""
    if device is None or device.type != "cuda" or not dist.is_initialized() or dist.get_world_size() <= 1:
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _cluster_off_devices:
        return
    from . import _capi
    _capi.cluster_config(idx, enable=False)
    _cluster_off_devices.add(idx)


def _free_port() -> int:
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from torchrun's environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port() if world == 1 else 29500)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n samples for `rank`; the first n % world ranks take one extra sample."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors: Sequence[Optional[torch.Tensor]], rank: int, world: int) -> List[Optional[torch.Tensor]]:
    """Slice every modality tensor (batch-first) to this rank's shard; ``None`` (missing modality) passes through."""
    n = next(t.shape[0] for t in tensors if t is not None)
    lo, hi = shard_bounds(n, rank, world)
    return [None if t is None else t[lo:hi] for t in tensors]


def gather_outputs(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather per-rank outputs of a sharded forward back into batch order (ragged shards allowed)."""
    if not _active():
        return local
    world = dist.get_world_size()
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def allreduce_mean_(tensors: Iterable[torch.Tensor], bucket_bytes: int = 32 << 20) -> None:
    """In-place average of a list of tensors (e.g. gradients) across ranks using flat buckets."""
    if not _active():
        return
    world = dist.get_world_size()
    tensors = list(tensors)
    if tensors:
        keep_ranks_in_step(tensors[0].device)
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        if len(bucket) == 1 and bucket[0].is_contiguous():     # already flat (healnet_amd.train.FlatParameters.grads): in place
            dist.all_reduce(bucket[0], op=dist.ReduceOp.SUM)
            bucket[0].div_(world)
            bucket, size = [], 0
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        bucket, size = [], 0

    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if bucket and (size + nbytes > bucket_bytes or t.dtype != bucket[0].dtype):
            flush()
        bucket.append(t)
        size += nbytes
    flush()


def shard_loss_scale(n_local: int, n_total: int, world: int) -> float:
    """Factor for a rank's LOCAL mean loss so that the rank-averaged gradient equals the gradient of the GLOBAL batch mean
    also with ragged shards: (1/world) * sum_r scale_r * grad(mean_r) = sum_r (n_r / n) * grad(mean_r)."""
    return float(n_local) * world / float(n_total)


def grad_ready_buckets(model, views, offsets):
    """{signal index -> [(lo, hi) float ranges of the flat gradient buffer]} for GradReadyAllReduce: a parameter is released by the
    hn_grad_ready signal of the LOWEST layer that uses it (a tied block belongs to layer 1); the head rides with the top layer's
    signal (it is final even earlier), layer 0 and the latent array with the end of the call (index -1).  Adjacent parameters
    released by the same signal merge into one range: with flatten_parameters' layout every bucket is one contiguous range."""
    depth = int(model.depth)
    owner = {}
    for l in range(depth):
        for p in model.layers[l].parameters():
            owner.setdefault(id(p), l)
    head_ids = {id(p) for p in model.to_logits.parameters()}
    top = max(owner.values()) if owner else -1     # the highest layer that owns parameters (1 with weight tying)
    per_param = []
    for p, off in zip(views, offsets):
        if id(p) in head_ids:
            idx = top
        else:
            idx = owner.get(id(p), -1)            # not in any layer: the latent array
        if idx <= 0:
            idx = -1
        per_param.append((off, (p.numel() + 3) // 4 * 4, idx))
    per_param.sort()
    buckets = {}
    for off, n, idx in per_param:
        ranges = buckets.setdefault(idx, [])
        if ranges and ranges[-1][1] == off:
            ranges[-1] = (ranges[-1][0], off + n)
        else:
            ranges.append((off, off + n))
    return buckets


class GradReadyAllReduce:
    """The gradient average of a training step, overlapped with the backward (serves healnet/main.py:464-465).

    ``hn_fusion_backward`` finishes the layers in reverse order and signals, per layer, the moment nothing accumulates into
    that layer's parameter gradients any more (``hn_grad_ready``: an event on the compute stream + a host callback).  With the
    flat layout of ``healnet_amd.train.flatten_parameters`` a layer's gradients are ONE contiguous range of ``flat.grads``, so
    each signal releases one large all-reduce on a side stream (RCCL over xGMI: a few multi-MB messages, not 125 small
    ones) while the lower layers are still running their backward:

        bucket [layer depth-1 + head]  released by signal depth-1
        bucket [layer l]               released by signal l          (l = depth-2 .. 1)
        bucket [latents + layer 0]     released when the backward has been enqueued completely

        flat = healnet_amd.train.flatten_parameters(model)
        sync = healnet_amd.dist.GradReadyAllReduce(model, flat)       # registers itself with the fused backward
        ...
        loss.backward()          # all-reduces are enqueued from inside hn_fusion_backward
        sync.wait()              # the compute stream waits for the side stream; then opt.step()

    ``reduce_fn(view)`` replaces the collective (default: in-place mean over the default process group; a no-op group of one
    still exercises the whole signalling path)."""

    def __init__(self, model, flat, reduce_fn=None):
        from . import _capi, ops
        self.flat = flat
        self.device = flat.grads.device
        keep_ranks_in_step(self.device)                      # world > 1: no cluster launches beside the collectives (see there)
        depth = int(model.depth)
        self.buckets = grad_ready_buckets(model, flat.views, flat.offsets)
        self.depth = depth
        self.reduce_fn = reduce_fn or self._allreduce_mean
        # a single rank with the default collective has nothing to exchange: the signals are still taken (bookkeeping, tests of
        # the bucket plan), but no side-stream hand-off is made -- each cross-stream event wait costs tens of microseconds on this
        # runtime (0.2 ms of a 5.9 ms cfg4 step for the four hops of a depth-3 model)
        self._idle = reduce_fn is None
        self.side = torch.cuda.Stream(self.device)
        self.events = [torch.cuda.Event() for _ in range(depth + 1)]
        for ev in self.events:                               # torch creates the hipEvent_t lazily at the first record
            ev.record(torch.cuda.current_stream(self.device))
        self._ev_arr = (_capi.C.c_void_p * (depth + 1))(*[ev.cuda_event for ev in self.events])
        self._cb = _capi.READY_FN(self._notify)
        self.ready = _capi.GradReady(events=self._ev_arr, notify=self._cb, user=None)
        self._error = None
        self.launched = []                                   # (signal index, lo, hi) in launch order, for tests / logging
        ops.register_backward_hook(flat.grads, self)

    def close(self) -> None:
        from . import ops
        ops.unregister_backward_hook(self.flat.grads)

    def _allreduce_mean(self, view: torch.Tensor) -> None:
        if not _active():
            return
        if dist.get_backend() == "nccl":
            dist.all_reduce(view, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM)
            view.div_(dist.get_world_size())

    def _skip(self) -> bool:
        return self._idle and not _active()

    def _release(self, idx: int) -> None:
        for lo, hi in self.buckets.get(idx, []):
            if not self._skip():
                with torch.cuda.stream(self.side):
                    self.reduce_fn(self.flat.grads[lo:hi])
            self.launched.append((idx, lo, hi))

    # -- called by torch.ops.healnet_hip.fusion_backward around / from inside hn_fusion_backward -------------------
    def begin(self, stream_ptr: int) -> None:
        self._error = None
        self.launched = []

    def _notify(self, idx, user) -> None:                    # host callback: events[idx] has just been recorded
        try:
            if idx in self.buckets and idx >= 1:
                if not self._skip():
                    self.side.wait_event(self.events[idx])
                self._release(idx)
        except BaseException as e:                           # exceptions cannot cross the C frame
            self._error = e

    def end(self, stream_ptr: int) -> None:
        if self._error is not None:
            raise self._error
        if not self._skip():
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self.side.wait_event(done)
        self._release(-1)

    def wait(self) -> None:
        """Make the current stream wait for every all-reduce released by the last backward."""
        if not self._skip():
            torch.cuda.current_stream(self.device).wait_stream(self.side)


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    """bench.py timing contract: the slowest rank defines the step time."""
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------
# Context split: the optional second axis of SURVEY.md 8(e) -- b < #GPUs, one huge volume
# ------------------------------------------------------------------------------------------------
def slab_bounds(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) of a modality's first spatial axis that `rank` attends to (contiguous, the first n_rows % world ranks take
    one extra row; a rank may get none when there are fewer rows than ranks)."""
    return shard_bounds(n_rows, rank, world)


def gather_partials(o_part: torch.Tensor, stats: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """ONE all-gather of a cross-attention block's per-rank (normalised output (b, L, inner), statistics (b, heads, L, 2)) pairs:
    b * L * (inner + 2 * heads) floats per rank (270 KB per sample with the default model) -> (G, b, L, inner), (G, b, heads, L, 2),
    rank-major.  The xGMI mesh is point-to-point: one small message per peer, no ring."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not force_collectives():
        return o_part.unsqueeze(0), stats.unsqueeze(0)
    n_o = o_part.numel()
    flat = torch.cat([o_part.reshape(-1), stats.reshape(-1)])
    out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    out = out.view(world, -1)
    return out[:, :n_o].reshape((world,) + tuple(o_part.shape)).contiguous(), out[:, n_o:].reshape((world,) + tuple(stats.shape)).contiguous()


def allreduce_sum_(tensors: Sequence[torch.Tensor], group=None) -> None:
    """In-place SUM over the ranks of `group` of a list of tensors, as ONE flat collective (the backward of a split cross block:
    its partial dx and parameter gradients, b * l_c * l_d + the block's parameter count floats)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not force_collectives():
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def _gather_flat(local: torch.Tensor, parts: torch.Tensor, group=None) -> None:
    """parts (world * n floats, rank-major) <- all-gather of local (n floats)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not force_collectives():
        parts.copy_(local)
    else:
        dist.all_gather_into_tensor(parts, local, group=group)


def _fused_context_parallel(model, slabs, begins, totals, split, world, return_embeddings, gather_flat):
    """hn_fusion_forward_cp: the fused forward (latent chains and all) over this rank's slabs; the C side calls back once per split
    cross block for the all-gather of its (output | statistics) buffer."""
    import ctypes as C
    from . import _capi, ops
    from ._rt import WS, stream_ptr
    lib = _capi.lib()
    sp = ops.spec_of(model._spec_text)
    params = model._params()
    device = params[0].device
    emb = bool(return_embeddings) or not model.final_classifier_head
    with torch.cuda.device(device), torch.no_grad():
        mdl, _keep = sp.model_cached(params)
        inp, held, b = sp.inputs(slabs)
        need = lib.hn_fusion_workspace_bytes(C.byref(mdl), inp, b)
        if need == 0:
            _capi.check(-1, "hn_fusion_workspace_bytes")
        ws = WS.get(device, need)
        nf = int(lib.hn_context_split_floats(C.byref(mdl), b))
        local = torch.empty(nf, dtype=torch.float32, device=device)
        parts = torch.empty(world * nf, dtype=torch.float32, device=device)
        failure = []

        def exchange(_user, floats, _stream):
            try:                                   # (an exception must not unwind through the C frame)
                gather_flat(local[:floats], parts[: world * floats])
            except BaseException as e:             # noqa: BLE001
                failure.append(e)
                return 1                           # the C loop stops here: no later block calls into a collective the peers left
            return 0

        cb = _capi.CP_EXCHANGE_FN(exchange)
        cp = _capi.ContextSplit(n_parts=world, split_mask=sum(1 << i for i, f in enumerate(split) if f), local=local.data_ptr(),
                                parts=parts.data_ptr(), exchange=cb, user=None)
        for i in range(len(split)):
            cp.axis0_begin[i], cp.axis0_total[i] = int(begins[i]), int(totals[i])
        out = torch.empty(ops._out_shape(sp, b, emb), dtype=torch.float32, device=device)
        status = lib.hn_fusion_forward_cp(C.byref(mdl), inp, b, int(emb), C.byref(cp), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                          stream_ptr(device))
        if failure:
            raise failure[0]
        _capi.check(status, "hn_fusion_forward_cp")
    return out


def context_parallel_forward(model, tensors: Sequence[Optional[torch.Tensor]], group=None, return_embeddings: bool = False,
                             min_rows_per_rank: int = 1, rank: Optional[int] = None, world: Optional[int] = None,
                             gather=None, fused: Optional[bool] = None, gather_flat=None, reduce=None) -> torch.Tensor:
    """Forward of a HealNet whose CONTEXTS are split over the ranks of `group` (every rank passes the SAME full-batch
    `tensors`; a rank reads only its slab of each split modality): the reference's fusion loop (healnet.py:225-250) block by
    block through the C ABI --

      cross block of a split modality   hn_encode_norm_slab (once per modality) -> hn_attn_partial_fwd on the rank's tokens ->
                                        one all-gather of (output, statistics) per block -> hn_attn_merge_fwd (every rank folds
                                        all shards in rank order: bit-identical latents everywhere, no broadcast needed)
      everything else                   replicated: cross blocks of short modalities (fewer than `min_rows_per_rank` rows or two
                                        tokens per rank, e.g. the one-token tabular input), feed-forward blocks, latent self blocks, head.

    Partition = the first spatial axis (image rows, volume slices, bag patches), contiguous slabs.  For b >= #GPUs shard the BATCH
    instead (no forward collective at all).  No mask, no dropout; a missing modality (None) is skipped as in the plain forward, on the
    block-by-block route.

    `fused` (default: try it, fall back): run the whole loop inside ONE C call, hn_fusion_forward_cp -- the fused forward with its
    latent chains, calling back once per split cross block for the all-gather -- instead of block by block (models whose shapes
    the chains do not take, staged models and bf16 cores stay on the block-by-block route).

    TRAINING (gradients enabled and a parameter that requires them; ABI v11): the block-by-block route under autograd.  A split cross
    block runs `ops.ContextSplitAttentionFn` -- the training forward on the slab, the all-gather of (O or P z | statistics), the
    merge into the tape, the output from the merged tape; in the backward the ordinary block backward on the slab with the GLOBAL
    statistics, then ONE all-reduce (`reduce`, default `allreduce_sum_`) of the partial dx and the block's parameter gradients, the
    gradients that do not pass through the core computed on rank 0 only.  After `loss.backward()` every rank holds the same,
    complete gradients: no further reduction (the replicated blocks computed identical ones everywhere).  No dropout."""
    from . import healnet as hm                              # (late: healnet.py imports this package's ops)
    hip = torch.ops.healnet_hip
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    custom_gather = gather is not None
    gather = gather or (lambda o, st: gather_partials(o, st, group))
    M = model.modalities
    if len(tensors) > M:
        raise ValueError(f"{len(tensors)} tensors passed to a model with {M} modalities")
    # a missing modality (a None entry, or a list shorter than the model: healnet.py:193,238) skips its cross-attention and
    # feed-forward blocks; the latent self block of that iteration still runs (the verbose=True quirk is the plain forward's)
    present = [i < len(tensors) and tensors[i] is not None for i in range(M)]
    if not any(present):
        raise ValueError("at least one modality must be present")
    tensors = [tensors[i] if present[i] else None for i in range(M)]
    if model.training and model._any_dropout:
        raise NotImplementedError("context_parallel_forward runs without dropout (eval mode, or attn_dropout = ff_dropout = 0)")
    b = next(t for t in tensors if t is not None).shape[0]
    ctx: List[Optional[torch.Tensor]] = [None] * M          # the rank's normalised slab, or the whole context of a replicated modality
    split = [False] * M
    train = fused is not True and torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())      # (fused=True: the inference entry point, as before)
    can_fuse = not train and all(present) and not custom_gather and not model.runs_staged() and getattr(model, "core_precision", "fp32") == "fp32"
    if fused and not can_fuse:
        raise RuntimeError("healnet_amd: hn_fusion_forward_cp (the fused context split) takes unstaged fp32-core models and the flat "
                           "gather; this model / call runs block by block (fused=None or False)")
    if fused is not False and can_fuse:
        slabs, begins, totals = [], [], []
        for m, data in enumerate(tensors):
            rows = data.shape[1]
            tokens_per_row = data[0, 0].numel() // data.shape[-1]
            split[m] = world > 1 and rows >= world * max(1, min_rows_per_rank) and (rows // world) * tokens_per_row >= 2
            lo, hi = slab_bounds(rows, rank, world) if split[m] else (0, rows)
            slabs.append(data[:, lo:hi].contiguous() if split[m] else data)
            begins.append(lo)
            totals.append(rows)
        try:
            return _fused_context_parallel(model, slabs, begins, totals, split, world, return_embeddings,
                                           gather_flat or (lambda lo_, pa_: _gather_flat(lo_, pa_, group)))
        except _capi_mod().HealnetHipError as e:
            # only "the chains do not take this model" falls back to the block-level route; launch / workspace / pointer errors (which
            # also carry the entry point's name in their text -- the old `"fusion" not in str(e)` filter swallowed them all) propagate
            if fused or e.status != _capi_mod().HN_E_UNSUPPORTED:
                raise
        split = [False] * M
    from . import ops as _ops
    reduce = reduce or (lambda ts: allreduce_sum_(ts, group))
    with torch.no_grad():
        for m, data in enumerate(tensors):
            if data is None:
                continue
            att = model.layers[0][2 * m].fn
            pitch = _capi_lib().hn_context_pitch(model.context_dims[m], att.dim_head)
            rows = data.shape[1]
            tokens_per_row = data[0, 0].numel() // data.shape[-1]
            split[m] = world > 1 and rows >= world * max(1, min_rows_per_rank) and (rows // world) * tokens_per_row >= 2
            if split[m]:
                lo, hi = slab_bounds(rows, rank, world)
                ctx[m] = hip.encode_norm_slab(data[:, lo:hi].contiguous(), model.num_freq_bands, model.max_freq, model.fourier_encode_data,
                                              pitch, lo, rows)
            else:
                ctx[m] = hip.encode_norm(data, model.num_freq_bands, model.max_freq, model.fourier_encode_data, pitch)
    with torch.set_grad_enabled(train):
        x = (model.latents if train else model.latents.detach()).unsqueeze(0).expand(b, -1, -1).contiguous()
        for layer in model.layers:
            for m in range(M):
                if not present[m]:
                    if model.self_per_cross_attn > 0:
                        x = hm.latent_block(layer[2 * M][0], layer[2 * M][1], x)
                    continue
                pn, ff = layer[2 * m], layer[2 * m + 1]
                a = pn.fn
                wts = (pn.norm.weight, pn.norm.bias, pn.norm_context.weight, pn.norm_context.bias, a.to_q.weight, a.to_kv.weight,
                       a.to_out[0].weight, a.to_out[0].bias)
                if split[m] and train:
                    x = _ops.ContextSplitAttentionFn.apply(lambda part, st: gather(part.reshape(b, -1), st), reduce, rank == 0, a.heads,
                                                           ctx[m], x, *wts)
                elif split[m]:
                    o, st = hip.attention_partial(x, ctx[m], None, *wts, a.heads)
                    o_all, st_all = gather(o, st)
                    x, _ = hip.attention_merge(x, o_all, st_all, a.to_q.weight, a.to_out[0].weight, a.to_out[0].bias, a.heads, True)
                else:
                    x = hip.attention(x, ctx[m], None, *wts, a.heads, True)
                f = ff.fn
                x = hip.feed_forward(x, ff.norm.weight, ff.norm.bias, f.net[0].weight, f.net[0].bias, f.net[2].weight, f.net[2].bias,
                                     not f.snn, True)
                if model.self_per_cross_attn > 0:
                    x = hm.latent_block(layer[2 * M][0], layer[2 * M][1], x)
        if return_embeddings or not model.final_classifier_head:
            return x
        ln, lin = model.to_logits[1], model.to_logits[2]
        return hip.head(x, ln.weight, ln.bias, lin.weight, lin.bias)


def _capi_lib():
    from . import _capi
    return _capi.lib()


def _capi_mod():
    from . import _capi
    return _capi
