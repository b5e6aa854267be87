"""Training-step tail on the GPU: what the reference's loop does right after
``logits = model.forward(features)`` (healnet/main.py:432-467), as two fused HIP entry points.

    flat  = healnet_amd.train.flatten_parameters(model)                  # parameters / gradients as ONE buffer each
    opt   = healnet_amd.train.FusedL1Adam(flat, lr=..., l1=...)          # torch.optim.Optimizer: schedulers work on it
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=..., epochs=..., steps_per_epoch=...)   # main.py:391-394
    for features, censorship, event_time, y_disc in loader:
        opt.zero_grad()                                                  # one memset
        logits = model(features)
        out = healnet_amd.train.surv_nll_loss(logits, y_disc, censorship, weights=class_weights)    # main.py:439-447
        (out.loss / gc).backward()                                       # gradients land in flat.grads
        healnet_amd.dist.allreduce_mean_([flat.grads])                   # ONE RCCL message
        opt.step()                                                       # L1 sign term + Adam + reg_loss in one pass
        sched.step()
        total = out.loss.item() + float(opt.reg_loss)                    # main.py:458-460

No CPU fallback: host tensors raise.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _capi


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _need_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"healnet_amd.train: {what} must live on a HIP device (got {t.device}); there is no CPU fallback")


# ------------------------------------------------------------------------------------------------
# survival loss
# ------------------------------------------------------------------------------------------------
@dataclass
class SurvivalOutput:
    loss: torch.Tensor          # scalar, differentiable w.r.t. the logits
    hazards: torch.Tensor       # (b, n_bins)  sigmoid(logits)                     main.py:439
    survival: torch.Tensor      # (b, n_bins)  cumprod(1 - hazards)                main.py:440
    risk: torch.Tensor          # (b,)         -sum(survival)                      main.py:441


class _SurvNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y, c, weights, alpha, eps):
        _need_gpu(logits, "logits")
        lib = _capi.lib()
        dev = logits.device
        lg = logits.detach().float().contiguous()
        b, k = lg.shape
        yy = y.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
        cc = c.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        if yy.numel() != b or cc.numel() != b:
            raise ValueError(f"y / c must have {b} entries")
        ww = None if weights is None else weights.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        if ww is not None and ww.numel() != k:
            raise ValueError(f"weights must have {k} entries")
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dl = torch.empty_like(lg)
        hz, sv, rk = torch.empty_like(lg), torch.empty_like(lg), torch.empty(b, dtype=torch.float32, device=dev)
        _capi.check(lib.hn_surv_nll(lg.data_ptr(), yy.data_ptr(), cc.data_ptr(), None if ww is None else ww.data_ptr(), b, k,
                                    float(alpha), float(eps), 1.0, loss.data_ptr(), dl.data_ptr(), hz.data_ptr(), sv.data_ptr(),
                                    rk.data_ptr(), _stream(dev)), "hn_surv_nll")
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(hz, sv, rk)
        ctx.set_materialize_grads(False)      # (otherwise autograd fills a zero gradient for each of hz / sv / rk: three launches per step)
        return loss, hz, sv, rk

    @staticmethod
    def backward(ctx, dloss, *_):
        if dloss is None:
            return None, None, None, None, None, None
        (dl,) = ctx.saved_tensors
        return dl * dloss, None, None, None, None, None


def surv_nll_loss(logits: torch.Tensor, y_disc: torch.Tensor, censorship: torch.Tensor, weights: Optional[torch.Tensor] = None,
                  alpha: float = 0.4, eps: float = 1e-7) -> SurvivalOutput:
    """``nll_loss(hazards=sigmoid(logits), S=cumprod(1 - hazards), Y=y_disc, c=censorship, weights=weights)``
    (healnet/models/survival_loss.py:9-43 as called at healnet/main.py:439-447) in one launch, with the hazards /
    survival / risk tensors the loop logs for the concordance index."""
    loss, hz, sv, rk = _SurvNLL.apply(logits, y_disc, censorship, weights, alpha, eps)
    return SurvivalOutput(loss, hz, sv, rk)


# ------------------------------------------------------------------------------------------------
# flat parameters + fused L1 / Adam step
# ------------------------------------------------------------------------------------------------
class FlatParameters:
    """Re-homes every parameter of ``model`` (shared ones once) as a view into one fp32 buffer and gives each a ``.grad``
    view into a second one: the optimizer step is one kernel over ``params``/``grads``, the data-parallel exchange one
    all-reduce of ``grads`` (38 MB at cfg2: a single message per step on the xGMI mesh)."""

    def __init__(self, model: torch.nn.Module):
        ps = [p for p in model.parameters() if p.requires_grad]
        if not ps:
            raise ValueError("model has no trainable parameters")
        dev = ps[0].device
        _need_gpu(ps[0], "parameters")
        sizes = [(p.numel() + 3) // 4 * 4 for p in ps]            # keep every view 16-byte aligned
        total = sum(sizes)
        self.params = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views: List[torch.nn.Parameter] = ps
        self.offsets: List[int] = []
        off = 0
        for p, n in zip(ps, sizes):
            self.offsets.append(off)
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("all parameters must be fp32 on one device")
            flat = self.params[off:off + p.numel()].view_as(p)
            flat.copy_(p.data)
            p.data = flat
            p.grad = self.grads[off:off + p.numel()].view_as(p)
            off += n
        self.numel = total
        model._hn_flat = self          # the fused backward accumulates straight into p.grad (see healnet.py)

    def zero_grad(self) -> None:
        self.grads.zero_()

    def direct_offsets(self, params) -> Optional[List[int]]:
        """For the fused backward (torch.ops.healnet_hip.fusion_backward): the float offset of every tensor of ``params`` in
        ``grads`` (-1: not trainable), or None when some trainable parameter's ``.grad`` is not its slot of the flat buffer
        (set_to_none zeroing, a foreign ``.grad``) -- the backward then returns ordinary gradients to autograd and
        ``relink()`` repairs the layout before the next optimizer step."""
        # fast path (a training loop calls this with the same parameter list every step: 146 parameters x three tensor queries were
        # 0.13 ms of the kirp step's host time): the previous answer holds while every parameter is the same object and still has the
        # very gradient view this buffer handed out (the objects are kept alive here, so identity cannot be recycled)
        fast = self.__dict__.get("_direct_fast")
        if fast is not None and len(params) == len(fast[0]) and fast[3] == self.grads.data_ptr():
            ok = True
            for p, q, g, off in zip(params, fast[0], fast[1], fast[2]):
                # identity of the parameter and of its gradient view, the view's address (p.grad.data = ... / set_() keep the object
                # and move the storage: one integer compare), and requires_grad for EVERY entry -- a parameter cached as
                # not trainable that has been switched on since must leave the fast path too
                if p is not q or (g is None) != (not p.requires_grad) or (g is not None and (p.grad is not g or g.data_ptr() != fast[3] + 4 * off)):
                    ok = False
                    break
            if ok:
                return fast[2]
        slot = self.__dict__.get("_slot_of")
        if slot is None:
            slot = self._slot_of = {id(p): off for p, off in zip(self.views, self.offsets)}
        base = self.grads.data_ptr()
        out = []
        for p in params:
            off = slot.get(id(p), -1)
            if off < 0:
                if p.requires_grad:
                    return None
                out.append(-1)
                continue
            g = p.grad
            if not p.requires_grad:               # a frozen parameter of the flat buffer: no gradient is written for it
                out.append(-1)
                continue
            if g is None or not g.is_contiguous() or g.data_ptr() != base + 4 * off:
                return None
            out.append(off)
        self._direct_fast = (list(params), [p.grad if o >= 0 else None for p, o in zip(params, out)], out, base)
        return out

    def relink(self) -> None:
        """Called by the optimizer before every step: the flat layout only works while every parameter and gradient is still
        a view of the two buffers.  ``model.zero_grad()`` (set_to_none) or an autograd fallback leaves ``.grad`` elsewhere:
        such gradients are copied into their slot and re-pointed; a re-allocated parameter (``model.to(...)``,
        ``load_state_dict(assign=True)``) cannot be repaired silently and raises."""
        base_p, base_g = self.params.data_ptr(), self.grads.data_ptr()
        for p, off in zip(self.views, self.offsets):
            if p.data_ptr() != base_p + 4 * off:
                raise RuntimeError("healnet_amd: a parameter no longer lives in the flat buffer (model.to(...) / assign-loading after "
                                   "flatten_parameters?) -- call flatten_parameters(model) again and rebuild the optimizer")
            slot = self.grads[off:off + p.numel()].view_as(p)
            if p.grad is None:                    # set_to_none zeroing and no gradient arrived since: the slot may be stale
                slot.zero_()
                p.grad = slot
            elif p.grad.data_ptr() != base_g + 4 * off:
                slot.copy_(p.grad)
                p.grad = slot


def flatten_parameters(model: torch.nn.Module) -> FlatParameters:
    return FlatParameters(model)


class FusedL1Adam(torch.optim.Optimizer):
    """``torch.optim.Adam(model.parameters(), lr)`` (healnet/main.py:390) with the L1 regulariser of
    ``calc_reg_loss`` (healnet/utils/train_utils.py:5-14) folded in: ``step()`` is ONE pass over the flat buffers --
    gradient of l1 * sum |p|, Adam moments, update, and the reg_loss value itself (``.reg_loss``, a 0-dim device tensor
    holding l1 * sum |p| of the parameters the forward just used).  ``param_groups[0]['lr']`` / ``['betas']`` are read
    at every step, so torch's OneCycleLR (which cycles both) drives it unchanged."""

    def __init__(self, flat: FlatParameters, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, l1: float = 0.0,
                 grad_scale: float = 1.0):
        self.flat = flat
        super().__init__([{"params": flat.views}], dict(lr=lr, betas=betas, eps=eps, l1=l1, grad_scale=grad_scale))
        dev = flat.params.device
        # Adam moments and the update count live in Optimizer.state (keyed by the first parameter, as flat tensors over the
        # whole buffer) so that optimizer.state_dict() / load_state_dict() checkpoint and resume them like torch.optim.Adam's
        self.state[flat.views[0]] = {"step": 0, "exp_avg": torch.zeros_like(flat.params), "exp_avg_sq": torch.zeros_like(flat.params)}
        self.reg_loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._ws = torch.empty(_capi.lib().hn_l1_adam_workspace_bytes(), dtype=torch.uint8, device=dev)

    def _flat_state(self) -> dict:
        st = self.state[self.flat.views[0]]
        n = self.flat.numel
        for k in ("exp_avg", "exp_avg_sq"):      # load_state_dict casts / moves state like the parameter it is keyed by
            t = st[k]
            if t.numel() != n or t.dtype != torch.float32 or t.device != self.flat.params.device or not t.is_contiguous():
                if t.numel() != n:
                    raise RuntimeError(f"FusedL1Adam: checkpointed {k} has {t.numel()} elements, the flat buffer {n}")
                st[k] = t.to(device=self.flat.params.device, dtype=torch.float32).contiguous().reshape(-1)
        return st

    @property
    def exp_avg(self) -> torch.Tensor:
        return self._flat_state()["exp_avg"]

    @property
    def exp_avg_sq(self) -> torch.Tensor:
        return self._flat_state()["exp_avg_sq"]

    @property
    def _steps(self) -> int:
        return int(self._flat_state()["step"])

    def zero_grad(self, set_to_none: bool = False) -> None:   # gradients stay views of the flat buffer
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        st = self._flat_state()
        st["step"] = int(st["step"]) + 1
        f = self.flat
        f.relink()
        try:
            _capi.check(_capi.lib().hn_l1_adam_step(f.params.data_ptr(), f.grads.data_ptr(), st["exp_avg"].data_ptr(),
                                                    st["exp_avg_sq"].data_ptr(), f.numel, float(g["l1"]), float(g["grad_scale"]),
                                                    float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                                    st["step"], self.reg_loss.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                                    _stream(f.params.device)), "hn_l1_adam_step")
        except _capi.CoresidencyLost as err:
            # a cluster-mode chain of THIS step (or of one the host had run ahead of) lost an exchange: the gradients may carry NaN.
            # The update is skipped -- on the device an already enqueued hn_l1_adam_step does the same by itself -- and the count
            # taken back; the caller's next step runs without clusters (include/healnet_hip.h "failure signal").
            dev = f.params.device
            _capi.note_coresidency(err, dev.index if dev.index is not None else torch.cuda.current_device(),
                                   lambda: torch.cuda.synchronize(dev))
            st["step"] = int(st["step"]) - 1
            self.skipped_steps = getattr(self, "skipped_steps", 0) + 1
        return loss


def retry_step(step_fn: Callable[[], object], retries: int = 1):
    """Run ``step_fn`` (forward + loss + backward of one batch, gradients zeroed INSIDE it) and, when a fused entry point reports
    HN_E_CORESIDENCY -- a cluster-mode latent chain lost an exchange, the tape / gradients of this step may hold NaN -- run it
    again: cluster mode is off by then, parameters have not been touched (the report reaches the host before ``opt.step()`` or
    the device-side skip of hn_l1_adam_step covers it).  The loop body of healnet/main.py:425-467 wrapped once."""
    for attempt in range(retries + 1):
        try:
            return step_fn()
        except _capi.CoresidencyLost:
            if attempt == retries:
                raise


class GraphedStep:
    """The gradient half of a training step -- zero the flat gradients, tape-recording forward, loss, fused backward -- captured
    into a HIP graph per input signature and replayed with one launch (SURVEY.md 8 f; serves the loop body of
    healnet/main.py:425-467).

    Why: the step of a patch-bag model is ~130 kernel launches behind two operator calls (450 before the chains and batched products of rounds 3-5).  At BASELINE configs[3] the host needs
    3.8-4.9 ms to enqueue what the GPU runs in 5.5 ms, so on a slow or busy host the step stretches to 6.2-6.8 ms; at the reference's
    tuned TCGA shapes (~90 launches of 5-50 us) the eager step is host-bound outright (driver box, round 4: +60 % for blca).  A
    replay costs ~0.06 ms of host time.

        flat = healnet_amd.train.flatten_parameters(model)
        opt = healnet_amd.train.FusedL1Adam(flat, ...)
        step = healnet_amd.train.GraphedStep(model, lambda logits, y, c: surv_nll_loss(logits, y, c).loss)
        for omic, wsi, y, c in loader:
            loss, logits = step([omic, wsi], (y, c))      # gradients are in flat.grads; loss / logits are the graph's static outputs
            opt.step()

    * A graph is captured the first time an input SIGNATURE is seen -- which modalities are present, every tensor's shape and dtype,
      the loss arguments' shapes and dtypes, whether a mask is passed -- after ``warmup`` eager runs on a side stream, and replayed
      from then on: the short last batch of an epoch, or a loader that alternates bag sizes, just adds a graph (``max_graphs``
      least-recently-used ones are kept; all share one memory pool).  ``example_inputs`` / ``loss_args`` / ``mask`` given to the
      constructor capture the first signature up front.
    * Inputs and loss arguments are copied into static device buffers; parameters are read in place, so optimizer updates are
      seen.  The model must have been through ``flatten_parameters`` (the backward then accumulates into one static buffer and
      autograd allocates nothing).  The returned loss / output tensors are the graph's static outputs: overwritten by the next
      replay of the same signature.
    * Dropout: a captured launch bakes the Philox offset into its kernel arguments, so every replay would draw the SAME masks.  The
      model therefore hands its kernels a device word (``hn_rng.offset_dev``) that is added to the offset on the device, and every
      graph's first node increments it: every replay draws fresh masks, forward and backward of one replay the same ones.  The word
      belongs to the MODEL and is shared (reference-counted) by all GraphedStep objects on it.
    * ``loss_fn(output, *loss_args)`` must return a scalar tensor and may only use capture-safe operations (no ``.item()``, no
      host synchronisation).
    * No entry point of the library runs during a replay, so the cluster-mode failure signal (include/healnet_hip.h) is polled
      here: a pending report drops every captured graph (they hold cluster launches), and the step is captured again without
      clusters and run."""

    def __init__(self, model: torch.nn.Module, loss_fn, example_inputs: Optional[Sequence[Optional[torch.Tensor]]] = None,
                 loss_args: Sequence[torch.Tensor] = (), mask: Optional[torch.Tensor] = None, warmup: int = 3, max_graphs: int = 8):
        flat = model.__dict__.get("_hn_flat")
        if flat is None:
            raise ValueError("GraphedStep needs healnet_amd.train.flatten_parameters(model) first (static gradient buffer)")
        self.device = flat.grads.device
        self.model, self.flat, self.loss_fn = model, flat, loss_fn
        self.warmup, self.max_graphs = max(1, int(warmup)), max(1, int(max_graphs))
        # the device word of the dropout generator (harmless without dropout); kept on the model so that its forward passes it on
        word = model.__dict__.get("_hn_rng_word")
        if word is None:
            word = torch.zeros(1, dtype=torch.int32, device=self.device)
            model.__dict__["_hn_rng_word"] = word
            model.__dict__["_hn_rng_word_users"] = 0
        model.__dict__["_hn_rng_word_users"] = model.__dict__.get("_hn_rng_word_users", 0) + 1
        self.word = word
        self._closed = False
        self._graphs: Dict[tuple, "_CapturedStep"] = {}
        self._pool = None
        self.captures = 0
        with torch.cuda.device(self.device):             # (hn_cluster_status creates the word only for the CURRENT device)
            st = _capi.cluster_status(self._dev_index())     # creates the device's status word before anything is captured
        self._cluster_epoch = (st["lost"], st["enabled"])    # what the captured launches were decided under (see __call__)
        if example_inputs is not None:
            self._get(self._signature(example_inputs, loss_args, mask), example_inputs, loss_args, mask)

    # -- compatibility with the single-graph object of round 4 (tests / tools read these)
    @property
    def graph(self):
        return next(reversed(self._graphs.values())).graph

    def _dev_index(self) -> int:
        return self.device.index if self.device.index is not None else torch.cuda.current_device()

    @staticmethod
    def _signature(inputs, loss_args, mask) -> tuple:
        def sig(t):
            return None if t is None else (tuple(t.shape), t.dtype)
        return (tuple(sig(t) for t in inputs), tuple(sig(t) for t in loss_args), sig(mask))

    def _get(self, key, inputs, loss_args, mask) -> "_CapturedStep":
        g = self._graphs.pop(key, None)
        if g is None:
            while len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            if not self._graphs:
                self._pool = None                  # (the shared pool dies with its last graph)
            g = _CapturedStep(self, inputs, loss_args, mask)
            self.captures += 1
        self._graphs[key] = g                      # most recently used last
        return g

    def __call__(self, inputs: Sequence[Optional[torch.Tensor]], loss_args: Sequence[torch.Tensor] = (), mask: Optional[torch.Tensor] = None):
        if self._closed:
            raise RuntimeError("GraphedStep: called after close()")
        for t in list(inputs) + list(loss_args) + [mask]:
            if t is not None and not isinstance(t, torch.Tensor):
                raise TypeError(f"GraphedStep: expected tensors, got {type(t).__name__}")
        st = _capi.cluster_status(self._dev_index())
        if st["pending"] or self._cluster_epoch != (st["lost"], st["enabled"]):
            # a replayed cluster launch lost an exchange (the device-side skip kept the optimizer from applying that step).  The
            # report may already have been consumed elsewhere -- FusedL1Adam.step's hn_l1_adam_step polls too, and is as likely to
            # see it first -- so the test is not `pending` alone but the device's loss counter / enabled flag against
            # what they were when the graphs were captured: the cluster decision is baked into a captured launch, every graph
            # holds such launches -- drop them all, consume a pending report (cluster mode goes off), capture afresh below
            torch.cuda.synchronize(self.device)
            self._graphs.clear()
            self._pool = None                      # (the shared pool dies with its last graph)
            if st["pending"]:
                try:
                    raise _capi.CoresidencyLost(_capi.HN_E_CORESIDENCY, "GraphedStep", "a replayed cluster-mode latent chain lost an exchange; "
                                                "the captured graphs were dropped and the step is captured again without cluster mode")
                except _capi.CoresidencyLost as err:
                    _capi.note_coresidency(err, self._dev_index(), lambda: None)
            st = _capi.cluster_status(self._dev_index())
            self._cluster_epoch = (st["lost"], st["enabled"])
        g = self._get(self._signature(inputs, loss_args, mask), inputs, loss_args, mask)
        g.load(inputs, loss_args, mask)
        g.graph.replay()
        return g.loss, g.output

    def close(self) -> None:
        """Detach from the model; when the last GraphedStep of the model closes, its eager forwards go back to host-advanced
        offsets only."""
        if self._closed:
            return
        self._closed = True
        self._graphs.clear()
        users = self.model.__dict__.get("_hn_rng_word_users", 1) - 1
        if users <= 0:
            self.model.__dict__.pop("_hn_rng_word", None)
            self.model.__dict__.pop("_hn_rng_word_users", None)
        else:
            self.model.__dict__["_hn_rng_word_users"] = users


class _CapturedStep:
    """One captured signature of a GraphedStep: static input buffers, the graph, its static outputs."""

    def __init__(self, owner: GraphedStep, inputs, loss_args, mask):
        dev = owner.device
        self.inputs = [None if t is None else t.detach().to(dev).clone() for t in inputs]
        self.loss_args = [t.detach().to(dev).clone() for t in loss_args]
        self.mask = None if mask is None else mask.detach().to(dev).clone()
        model, flat, loss_fn, word = owner.model, owner.flat, owner.loss_fn, owner.word

        def body():
            word.add_(1)
            flat.zero_grad()
            out = model(list(self.inputs), mask=self.mask)
            loss = loss_fn(out, *self.loss_args)
            loss.backward()
            return loss.detach(), out.detach()

        # A replay writes gradients where the CAPTURE wrote them: every parameter's .grad has to be its slot of the flat buffer before
        # the warm-up runs (model.zero_grad() -- set_to_none by default -- leaves them None: the backward would then hand ordinary
        # gradients to autograd, the capture would record accumulations into tensors of its own that flat.zero_grad() never clears and
        # the fused optimizer never reads) ...
        flat.relink()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        import warnings
        with torch.cuda.stream(side), warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for _ in range(owner.warmup):            # allocator, workspaces, descriptor caches and the autograd thread's state settle
                retry_step(body)
        torch.cuda.current_stream(dev).wait_stream(side)
        # An autograd graph of an EARLIER eager step that is still referenced (a kept `loss` / output tensor) keeps the parameters'
        # AccumulateGrad nodes alive on the stream that step ran on; the engine then joins that stream at the end of every backward,
        # which inside a capture on the side stream invalidates the capture -- hipStreamEndCapture has been seen to take the process
        # down with it.  torch reports the mismatch as a warning in the warm-up runs: refuse to capture instead.
        # ... and still after them (a loss_fn that re-points a .grad, a parameter outside the flat buffer)
        if flat.direct_offsets(list(flat.views)) is None:
            raise RuntimeError("GraphedStep: a parameter's .grad is not its slot of the flat gradient buffer after the warm-up runs "
                               "(the loss function or a hook replaces .grad?) -- a captured step could not deliver its gradients")
        stale = [w for w in caught if "AccumulateGrad node's stream does not match" in str(w.message)]
        for w in caught:
            if w not in stale:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if stale:
            raise RuntimeError("GraphedStep: an autograd graph of an earlier eager step is still alive (a loss or output tensor of it is "
                               "referenced: keep `.detach()`-ed copies / `.item()` values instead, or `del` it) -- its AccumulateGrad "
                               "nodes belong to another stream and a capture with them cannot end cleanly")
        # nothing may be pending when the capture starts (an entry point that finds a report returns HN_E_CORESIDENCY, which would
        # abort the capture): drain, and consume a report of the warm-up runs -- the capture then holds no cluster launches
        torch.cuda.synchronize(dev)
        if _capi.cluster_status(owner._dev_index())["pending"]:
            _capi.note_coresidency(_capi.CoresidencyLost(_capi.HN_E_CORESIDENCY, "GraphedStep", "a warm-up run lost a cluster exchange; "
                                                         "capturing without cluster mode"), owner._dev_index(), lambda: None)
        self.graph = torch.cuda.CUDAGraph()
        if owner._pool is None:
            owner._pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(self.graph, pool=owner._pool):
            self.loss, self.output = body()
        if not owner._graphs:                        # first graph of a (new) set: the state its cluster decisions were taken under
            st = _capi.cluster_status(owner._dev_index())
            owner._cluster_epoch = (st["lost"], st["enabled"])

    def load(self, inputs, loss_args, mask) -> None:
        # (presence, shapes and dtypes are the signature this object was looked up by: they match)
        for dst, src in zip(self.inputs, inputs):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        for dst, src in zip(self.loss_args, loss_args):
            dst.copy_(src, non_blocking=True)
        if self.mask is not None:
            self.mask.copy_(mask, non_blocking=True)
