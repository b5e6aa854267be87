#!/usr/bin/env python3
"""Headline benchmark: fusion-forward samples/sec, 2-modality (tab 1x2000 + img 224x224x3), b=32 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one ``model([tab, img])`` forward (logits out) over a batch of 32 synthetic samples that
already reside in HBM, eval mode, no autograd (BASELINE.json configs[1]).  The forward has
no cross-sample coupling, so N GPUs run N independent replicas on their own batch shard with no
collective in the data path ("weak" scaling); timing = barrier + synchronize on both sides, max over ranks.

`python bench.py --gpus N` with N > 1 and no torchrun environment re-executes itself under `python -m torch.distributed.run`
(one rank per GPU, 127.0.0.1 rendezvous), so the plain command works as well as the explicit launcher line above.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  roofline     : the dominant kernel (split-KV attention core of the image cross-attention), timed with HIP
                 events recorded on the launch stream inside hn_fusion_forward in an instrumented replay of the
                 same K steps right after the timed region (recording events between kernels costs ~1 ms per
                 forward on this runtime, so it stays out of the region `value` comes from);
  cpu_baseline : oracle/healnet_cpu.py (the op-for-op CPU restatement of the reference) timed on this
                 box's host cores on a bounded sample (the same workload at the config's own b=32 when the host has the RAM: one timed run
                 of ~20 s after a calibration of the thread count and a warm-up), rank 0 at N=1 only; train_step carries the oracle's
                 cfg4 forward (b=8) the same way;
  build_id     : hn_build_id() of the loaded library (sha256 of the sources + flags it was built from);
  staged_models: forward + backward of the reference's four tuned TCGA configurations (config/best_hyperparams.yml) at b=8, N=1 only
  patch_bag_precisions: the inference forward of BASELINE configs[3]'s shape (b=8) in fp32 and with core_precision="bf16", N=1 only
  configs      : BASELINE configs[0] / [2] / [3] / [4] at HEAD, N=1 only: forward ms, samples/s, the dominant kernel with its executed work
                 per launch, its HIP-event average and the fraction of the roof that bounds it
  train_step   : SURVEY.md 8(d)'s second figure -- the training step of BASELINE configs[3] (TCGA-BRCA shape: omic 1x2000 + WSI bag
                 4096x768, b=8 per GPU): forward with tape, survival NLL, fused backward, gradient all-reduce over RCCL
                 (overlapped with the backward through hn_grad_ready, healnet_amd.dist.GradReadyAllReduce) and the fused
                 L1 + Adam step, timed after the forward region with the same barrier / max-over-ranks rule; its `roofline` object
                 prices the step's executed matrix FLOPs against the fp32 MFMA peak and carries the two dominant kernels (the patch
                 bag's K/V projection and weight gradient) timed with HIP events in an instrumented replay of the same steps.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KW = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
BATCH = 32
IMG = (224, 224, 3)
TAB = (1, 2000)

# algorithmic FLOPs per sample of the whole forward and of one image cross-attention's
# QK^T + PV (reference formulation), and what the rank-D reassociated kernel actually executes.
FLOPS_FORWARD_PER_SAMPLE = 45.68e9
L_C, HEADS, DIM_HEAD, N_IMG, DP, QK_DIM = 128, 8, 64, 224 * 224, 16, 12
ALGO_FLOPS_CORE_PER_SAMPLE = 4.0 * L_C * N_IMG * DIM_HEAD * HEADS        # 13.15 GF
# executed: QK^T over the 12 kept channels of the packed context (3 MFMA k-steps), P V over 16 columns (12 channels,
# the softmax-denominator ones column, 3 idle) -> 2.88 GF (contraction dims 12 / 16, not 64 / 64)
EXEC_FLOPS_CORE_PER_SAMPLE = 2.0 * L_C * N_IMG * (QK_DIM + DP) * HEADS
PEAK_FP32_MFMA_TFLOPS = 157.3                                            # MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0                                           # dense bf16 MFMA, same table (16 x the fp32 pipe)


def exec_flops_forward_per_sample(depth=3, l_d=128):
    """EXECUTED matrix FLOPs of one cfg2 forward per sample (what the kernels run, not the reference formulation), by the
    schedule of healnet.py:225-245: per layer the image cross block (Q projection, query fold, core, folded value projection,
    out-projection) + its feed-forward block, the one-token tabular block + its feed-forward block, and behind EACH of the two
    the latent self-attention block (Q|K|V, core, out) + feed-forward block.  317.3 GF per 32 samples."""
    inner = HEADS * DIM_HEAD
    ff = 2.0 * L_C * l_d * 8 * l_d + 2.0 * L_C * 4 * l_d * l_d
    # image block: the chain in front projects LN(x) straight onto the 8 x 16 FOLDED query columns (W_q W_k staged once per forward:
    # 128 columns, not inner = 512 followed by a fold -- round 4 moved the fold into the chain), the core, the
    # folded value projection (16 -> inner per head) in the chain behind it, the out-projection
    img = 2.0 * L_C * l_d * HEADS * DP + 2.0 * L_C * inner * DP + EXEC_FLOPS_CORE_PER_SAMPLE + 2.0 * L_C * inner * l_d + ff
    tab = 2.0 * 2005 * inner + 2.0 * inner * l_d + ff
    self_blk = 2.0 * L_C * l_d * 3 * inner + 4.0 * L_C * L_C * DIM_HEAD * HEADS + 2.0 * L_C * inner * l_d + ff
    return depth * (img + tab + 2 * self_blk)


class HipEvents:
    """Raw hipEvent_t pairs (the C ABI records them on the launch stream around the dominant kernel)."""

    def __init__(self, n):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.n = n
        self.start = (ctypes.c_void_p * n)()
        self.stop = (ctypes.c_void_p * n)()
        for arr in (self.start, self.stop):
            for i in range(n):
                ev = ctypes.c_void_p()
                assert self.hip.hipEventCreate(ctypes.byref(ev)) == 0
                arr[i] = ev

    def elapsed_ms(self, count):
        out = []
        for i in range(count):
            ms = ctypes.c_float()
            rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), ctypes.c_void_p(self.start[i]), ctypes.c_void_p(self.stop[i]))
            if rc == 0:
                out.append(ms.value)
        return out


def _sha256(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the newest committed PMC passes (profiles/rNN_*_pmc_cfg2_b32.json,
    tools/pmc_collect.py): counters cannot be collected from inside this process.  (2*FETCH_SIZE + WRITE_SIZE) KiB, per
    MI355X_MICROARCH.md's gfx950 note.  The profile records the SHA-256 of the kernel source it was taken from; when
    attention.hip has changed since, the figure is stale and is dropped (traffic = null) instead of being reported."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_cfg2_b32.json")) if "bf16core" not in f)
    if not files:
        return None, {"source": None, "reason": "no PMC profile committed"}
    now = _sha256(os.path.join(ROOT, "healnet_amd", "csrc", "attention.hip"))
    stale = None
    # newest first by name (rNN_<call>); file names of one round do not sort by time (r04_zz sorts behind r04_ao), so the passes
    # that were taken from THIS attention.hip are looked for among all of them -- the newest round's first
    for path in reversed(files):
        try:
            with open(path) as f:
                doc = json.load(f)
            then = (doc.get("source_sha256") or {}).get("attention.hip")
            label = {"source": os.path.relpath(path, ROOT), "attention_hip_sha256": then, "collected_at_commit": doc.get("git_head")}
            if then == now:
                return float(doc["dominant_kernel_traffic_bytes_per_launch"]["fetch_doubled"]), label
            if stale is None:
                label["reason"] = "attention.hip changed since the PMC passes were collected (sha256 mismatch): stale, dropped"
                stale = label
        except Exception as e:      # noqa: BLE001
            if stale is None:
                stale = {"source": os.path.relpath(path, ROOT), "reason": f"unreadable: {e}"}
    return None, stale


def _host_ram_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


def _calibrate_threads(run_one, t_all, budget_s):
    """Fastest thread count of {8, 16, 32, 64} on ONE sample (every core of a 256-core box runs this small-GEMM / elementwise mix 50x
    slower than 32 threads -- 31.7 s per sample measured -- so counts above 64 are not tried)."""
    ncpu = os.cpu_count() or 1
    calib = {}
    for threads in sorted({t for t in (8, 16, 32, 64, min(ncpu, 64)) if t <= ncpu}):
        if time.time() - t_all > budget_s:
            break
        torch.set_num_threads(threads)
        run_one()                                              # warm-up at this thread count
        t0 = time.time()
        run_one()
        calib[threads] = time.time() - t0
    best = min(calib, key=calib.get)
    torch.set_num_threads(best)
    return best, calib


def cpu_baseline(budget_s=45.0):
    """Oracle forward on the host cores (BASELINE.md §3 protocol): thread count calibrated on one sample, one untimed warm-up, then the
    workload AT THE CONFIG'S OWN BATCH (cfg2: b = 32; the oracle materialises K/V, scores and probabilities: ~0.6 GB per sample and
    block, so b = 32 wants ~25 GB of free host RAM -- with less, b = 4 is timed and the reason recorded).  Median of 3 timed runs
    (~20 s of CPU work each at b = 32; throughput is flat in b: 1.78-1.89 samples/s at b = 1..4 in the survey container)."""
    from oracle import healnet_cpu as O
    import healnet_amd
    torch.manual_seed(0)
    model = healnet_amd.HealNet(**KW)                      # parameters only (seed-0 default init), stays on the host
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    cfg = O.FusionConfig(**KW)
    gen = torch.Generator().manual_seed(1234)
    ram = _host_ram_gb()
    b = BATCH if ram >= 64.0 else 4
    tab, img = torch.rand(b, *TAB, generator=gen), torch.rand(b, *IMG, generator=gen)
    ncpu = os.cpu_count() or 1
    t_all = time.time()
    with torch.no_grad():
        best_threads, calib = _calibrate_threads(lambda: O.fusion_forward(sd, cfg, [tab[:1], img[:1]]), t_all, budget_s / 4)
        O.fusion_forward(sd, cfg, [tab[:2], img[:2]])          # warm-up (allocator, thread pool) at the chosen count
        times = []
        for _ in range(3):                                     # median of >= 3 (3 x ~20 s at the config batch)
            t0 = time.time()
            O.fusion_forward(sd, cfg, [tab, img])
            times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(b / med, 3), "unit": "samples/s", "cores": best_threads, "kind": "port", "host_cores": ncpu,
            "batch": b, "host_ram_available_gb": round(ram, 1),
            "runs_s": [round(t, 3) for t in times], "calibration_s_per_sample": {str(k): round(v, 3) for k, v in calib.items()},
            "sample": f"oracle/healnet_cpu.py fusion_forward, b={b} of the same 2-modality workload"
                      f"{' (the config batch)' if b == BATCH else ' (host RAM below 64 GB: not the config batch of 32)'}, fp32, torch "
                      f"{torch.__version__} CPU, 1 warm-up + {len(times)} timed run(s); {best_threads} threads (fastest of {sorted(calib)} "
                      f"on one sample) out of {ncpu} host cores"}


def cpu_baseline_cfg4(budget_s=20.0):
    """The oracle's forward of BASELINE configs[3] (omic 1x2000 + WSI bag 4096x768) at the config's b = 8 on the host cores, beside
    train_step / patch_bag_precisions: same thread calibration, 1 warm-up + median of 3."""
    from oracle import healnet_cpu as O
    import healnet_amd
    torch.manual_seed(0)
    sd = {k: v.detach() for k, v in healnet_amd.HealNet(**TRAIN_KW).state_dict().items()}
    cfg = O.FusionConfig(**TRAIN_KW)
    gen = torch.Generator().manual_seed(4321)
    ins = [torch.rand(TRAIN_BATCH, *s, generator=gen) for s in TRAIN_SHAPES]
    t_all = time.time()
    with torch.no_grad():
        best_threads, calib = _calibrate_threads(lambda: O.fusion_forward(sd, cfg, [t[:1].clone() for t in ins]), t_all, budget_s / 3)
        O.fusion_forward(sd, cfg, [t.clone() for t in ins])
        times = []
        for _ in range(3):
            t0 = time.time()
            O.fusion_forward(sd, cfg, [t.clone() for t in ins])
            times.append(time.time() - t0)
    times.sort()
    med = times[1]
    return {"value": round(TRAIN_BATCH / med, 3), "unit": "samples/s (forward)", "cores": best_threads, "kind": "port", "batch": TRAIN_BATCH,
            "runs_s": [round(t, 3) for t in times], "calibration_s_per_sample": {str(k): round(v, 3) for k, v in calib.items()},
            "sample": f"oracle/healnet_cpu.py fusion_forward of cfg4 at b={TRAIN_BATCH}, fp32, no_grad, 1 warm-up + median of 3; {best_threads} threads"}


# BASELINE.json configs[3]: TCGA-BRCA-shaped training step, per-GPU batch 8
TRAIN_KW = dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4)
TRAIN_SHAPES = [(1, 2000), (4096, 768)]
TRAIN_BATCH = 8


def exec_flops_train_step_per_sample(depth=3, l_c=128, l_d=128, heads=8, dh=64, n_bag=4096, d_bag=773, d_omic=2005):
    """EXECUTED matrix FLOPs of one cfg4 training step per sample (forward with tape + backward; real dimensions, no tile padding).
    Per layer (healnet.py:225-245): omic one-token block + FF, latent self block + FF, bag cross block + FF, latent self block + FF.
    Backward of a Linear = 2x its forward (dX and dW) except the bag K/V projection (the context carries no gradient: dW only);
    attention cores: dQ kernel 6 l_c N dh h (recomputes S and dP), dK/dV kernel 8 l_c N dh h; the fused feed-forward backward
    recomputes its first layer.  58.0 GF per sample; the two patch-bag GEMMs (K/V projection, G = dKV^T z) are 38.9 GF of it."""
    inner = heads * dh
    ff1, ff2 = 2.0 * l_c * l_d * 8 * l_d, 2.0 * l_c * 4 * l_d * l_d
    ff_f, ff_b = ff1 + ff2, 2 * (ff1 + ff2) + ff1
    q_proj, out_proj = 2.0 * l_c * l_d * inner, 2.0 * l_c * inner * l_d
    kv_bag = 2.0 * n_bag * d_bag * 2 * inner
    core_f = lambda n: 4.0 * l_c * n * dh * heads           # noqa: E731
    core_b = lambda n: 14.0 * l_c * n * dh * heads          # noqa: E731
    bag_f = q_proj + kv_bag + core_f(n_bag) + out_proj + ff_f
    bag_b = 2 * q_proj + kv_bag + core_b(n_bag) + 2 * out_proj + ff_b
    self_f = 3 * q_proj + core_f(l_c) + out_proj + ff_f
    self_b = 6 * q_proj + core_b(l_c) + 2 * out_proj + ff_b
    omic_f = 2.0 * d_omic * inner + 2.0 * inner * l_d + ff_f          # one-token shortcut: V projection + out-projection on one row
    omic_b = 2 * (2.0 * d_omic * inner + 2.0 * inner * l_d) + ff_b
    fwd = depth * (bag_f + omic_f + 2 * self_f)
    bwd = depth * (bag_b + omic_b + 2 * self_b)
    return fwd, bwd, depth * 2 * kv_bag


class KernelTimers:
    """hn_set_kernel_timers (include/healnet_hip.h): hipEvent pairs around every launch of the named kernel classes, recorded on the
    launch stream inside the training forward / backward."""

    def __init__(self, names, per_name):
        from healnet_amd import _capi
        self.lib = _capi.lib()
        self.names = names
        self.events = [HipEvents(per_name) for _ in names]
        self.table = (_capi.KernelTimer * len(names))()
        for i, (name, ev) in enumerate(zip(names, self.events)):
            self.table[i].kernel = name.encode()
            self.table[i].ev_start = ctypes.cast(ev.start, ctypes.POINTER(ctypes.c_void_p))
            self.table[i].ev_stop = ctypes.cast(ev.stop, ctypes.POINTER(ctypes.c_void_p))
            self.table[i].n_events = per_name
            self.table[i].n_recorded = 0

    def __enter__(self):
        assert self.lib.hn_set_kernel_timers(self.table, len(self.names)) == 0
        return self

    def __exit__(self, *exc):
        self.lib.hn_set_kernel_timers(None, 0)

    def averages_ms(self):
        out = {}
        for i, name in enumerate(self.names):
            n = min(int(self.table[i].n_recorded), int(self.table[i].n_events))
            ms = self.events[i].elapsed_ms(n)
            out[name] = (sum(ms) / len(ms) if ms else None, len(ms))
        return out


def train_step_record(dev, rank, world, distributed, barrier, steps, warmup):
    """fwd + bwd + gradient all-reduce + fused L1/Adam step at cfg4, b=8 per GPU (the reference's loop body, main.py:425-467)."""
    import healnet_amd as hn
    from healnet_amd import dist as hdist
    import torch.distributed as dist
    inject = os.environ.get("HN_BENCH_INJECT", "")       # tests only ("raise:<rank>" / "hang:<rank>"): a rank that dies / never
    if inject == f"raise:{rank}":                         # returns here must not cost the driver the headline line (main())
        raise RuntimeError("injected failure (HN_BENCH_INJECT)")
    if inject == f"hang:{rank}":
        time.sleep(1e6)
    torch.manual_seed(0)
    model = hn.HealNet(**TRAIN_KW).train().to(dev)
    gen = torch.Generator().manual_seed(4321 + rank)
    b = TRAIN_BATCH
    ins = [torch.rand(b, *s, generator=gen).to(dev) for s in TRAIN_SHAPES]
    y = torch.randint(0, TRAIN_KW["out_dims"], (b,), generator=gen).to(dev)
    c = torch.randint(0, 2, (b,), generator=gen).to(dev)
    flat = hn.train.flatten_parameters(model)
    opt = hn.train.FusedL1Adam(flat, lr=1e-4, l1=1e-4)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=5 * (steps + warmup) + 8)
    sync = hdist.GradReadyAllReduce(model, flat)

    def step(overlap):
        opt.zero_grad()
        out = hn.train.surv_nll_loss(model(list(ins)), y, c)
        if overlap:
            out.loss.backward()                 # the all-reduces are released from inside hn_fusion_backward
            sync.wait()
        else:
            sync.close()
            out.loss.backward()
            hdist.allreduce_mean_([flat.grads])
            hn.ops.register_backward_hook(flat.grads, sync)
        opt.step()
        sched.step()
        return out.loss

    def timed(overlap):
        for _ in range(warmup):
            step(overlap)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(overlap)
        barrier()
        dt = time.perf_counter() - t0
        return (hdist.max_over_ranks(dt, dev) if distributed else dt), float(loss)

    dt, loss = timed(True)
    rec = {"config": "cfg4 (BASELINE configs[3]): HealNet(2,[2000,768],[1,1],4) default hyper-parameters, omic (b,1,2000) + WSI bag "
                     "(b,4096,768), fp32, train mode, survival NLL + L1 + Adam under OneCycleLR",
           "batch_per_gpu": b, "global_batch": b * world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 4), "value": round(b * world * steps / dt, 2), "unit": "samples/s",
           "world_size": world, "backend": (dist.get_backend() if distributed else None),
           "allreduce": "healnet_amd.dist.GradReadyAllReduce: flat gradient buffer, one bucket per layer released by hn_grad_ready "
                        "signals on a side stream while the lower layers run their backward",
           "allreduce_buckets_floats": [hi - lo for _, lo, hi in sync.launched], "gradient_floats": int(flat.numel),
           "final_loss": loss}
    # cluster-mode latent chains beside the side-stream all-reduce (include/healnet_hip.h "failure signal"): what this rank's
    # device reported during the timed steps -- `fallbacks` = calls re-run without clusters, `lost` = losses consumed, `enabled`
    # = the mode is still on (it is switched off for good by the first loss), `optimizer_steps_skipped` = FusedL1Adam host skips
    torch.cuda.synchronize(dev)
    from healnet_amd import _capi as _hc
    cst = _hc.cluster_status(dev.index if dev.index is not None else 0)
    rec["cluster"] = {"fallbacks": cst["fallbacks"], "lost": cst["lost"], "pending": cst["pending"], "enabled": cst["enabled"],
                      "optimizer_steps_skipped": int(getattr(opt, "skipped_steps", 0))}
    if distributed:
        dt2, _ = timed(False)
        rec["ms_per_step_blocking_allreduce"] = round(dt2 / steps * 1e3, 4)
        barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            hdist.allreduce_mean_([flat.grads])
        barrier()
        rec["allreduce_alone_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
    # ---- roofline of the step: instrumented replay of the same K steps (hipEvent pairs around every launch of the two patch-bag
    # GEMMs -- half of the step's executed FLOPs -- recorded on the launch stream inside the training forward / backward)
    fwd_f, bwd_f, gemm_f = exec_flops_train_step_per_sample()
    per_launch = 2.0 * (b * TRAIN_SHAPES[1][0]) * (TRAIN_SHAPES[1][1] + 5) * 1024        # 2 M K N of either GEMM (D = 768 + 5)
    timed_names = ["gemm_nt_x6", "gemm_tn_x6", "x6_split", "x6_split_t", "x6_tn_reduce", "gemm_nt_glds", "gemm_tn_glds"]
    with KernelTimers(timed_names, 4 * steps + 8) as kt:
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        barrier()
        dt_instr = time.perf_counter() - t0
    avg = kt.averages_ms()
    step_s = dt / steps
    x6 = avg["gemm_nt_x6"][1] > 0
    roof = {"bound": "mfma", "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "flops_executed_per_step": (fwd_f + bwd_f) * b,
            "achieved": round((fwd_f + bwd_f) * b / step_s / 1e12, 2),
            "frac": round((fwd_f + bwd_f) * b / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            "forward_flops": fwd_f * b, "backward_flops": bwd_f * b,
            "dominant_kernels": {}, "instrumented_ms_per_step": round(dt_instr / steps * 1e3, 4),
            "patch_bag_gemm_route": "x6 (fp32-exact on the bf16 pipe, gemm_x6.hip)" if x6 else "fp32 MFMA (gemm_nt.hip)",
            "timing": "hipEvent pairs on the launch stream around every launch of the patch-bag GEMM classes (hn_set_kernel_timers), in an "
                      "instrumented replay of the same K steps right after the timed region",
            "note": "frac = fp32 matrix FLOPs the step RETURNS (forward with tape + backward: exec_flops_train_step_per_sample, real "
                    "dimensions) / driver-timed step / fp32 MFMA peak; the survival loss, L1 + Adam sweep, LayerNorm / softmax vector work "
                    "and the all-reduce are time without matrix FLOPs.  On the x6 route the two patch-bag GEMMs (half of those FLOPs) run "
                    "on the bf16 pipe -- each fp32 product as six bf16 products of three-plane operands, error not above the fp32 MFMA's "
                    "(tests/test_gpu_x6.py) -- so this figure is a rate of fp32 results, no longer a utilisation of the fp32 pipe; the "
                    "per-kernel entries price those launches against the bf16 peak with the 6 x FLOPs they execute"}
    if x6:
        kp = (TRAIN_SHAPES[1][1] + 5 + 15) // 16 * 16
        exec_nt = 6 * 2.0 * (b * TRAIN_SHAPES[1][0]) * kp * 1024                         # 6 products over the padded contraction
        exec_tn = 6 * 2.0 * (b * TRAIN_SHAPES[1][0]) * 800 * 1024                        # 773 + ones column -> 25 tiles of 32
        for name, what, ex in (("gemm_nt_x6", "hn::gemm_nt_x6_kernel<4,2,2,4,3,0> (bag K/V projection, forward)", exec_nt),
                               ("gemm_tn_x6", "hn::gemm_nt_x6_kernel<1,5,8,1,3,1> (bag weight gradient G = dKV^T z + colsum, backward, split-k)", exec_tn)):
            ms, n = avg[name]
            roof["dominant_kernels"][name] = {
                "kernel": what, "launches_timed": n, "launches_per_step": 3, "fp32_flops_returned_per_launch": per_launch,
                "bf16_flops_executed_per_launch": ex, "avg_launch_ms": None if ms is None else round(ms, 4),
                "fp32_equivalent_tflops": None if not ms else round(per_launch / (ms * 1e-3) / 1e12, 2),
                "achieved": None if not ms else round(ex / (ms * 1e-3) / 1e12, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s (bf16 MFMA)",
                "frac": None if not ms else round(ex / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}
        roof["x6_image_kernels_us_per_step"] = {
            k: (None if avg[k][0] is None else round(avg[k][0] * 1e3 * avg[k][1] / steps, 1)) for k in ("x6_split", "x6_split_t", "x6_tn_reduce")}
    else:
        for name, what in (("gemm_nt_glds", "hn::gemm_nt_glds_kernel (bag K/V projection, forward)"),
                           ("gemm_tn_glds", "hn::gemm_tn_glds_kernel (bag weight gradient G = dKV^T z, backward)")):
            ms, n = avg[name]
            roof["dominant_kernels"][name] = {
                "kernel": what, "launches_timed": n, "launches_per_step": 3, "flops_per_launch": per_launch,
                "avg_launch_ms": None if ms is None else round(ms, 4),
                "achieved": None if not ms else round(per_launch / (ms * 1e-3) / 1e12, 2),
                "frac": None if not ms else round(per_launch / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
    rec["roofline"] = roof
    # forward (tape-recording) share of the step
    with torch.enable_grad():
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(list(ins))
        barrier()
    rec["forward_train_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
    sync.close()
    # ---- the same step with its gradient half (zero_grad + forward + loss + backward) replayed as ONE HIP graph
    # (healnet_amd.train.GraphedStep): what the step costs when the host enqueues one launch instead of ~130.  N = 1 only -- with more
    # ranks the overlapped all-reduce is released from host callbacks inside the backward, which a capture cannot hold.
    if not distributed:
        gstep = hn.train.GraphedStep(model, lambda logits, yy, cc: hn.train.surv_nll_loss(logits, yy, cc).loss, list(ins), (y, c))

        def graphed():
            gstep(ins, (y, c))
            opt.step()
            sched.step()
        for _ in range(warmup):
            graphed()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            graphed()
        barrier()
        rec["graphed_ms_per_step"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        gstep.close()
    return rec


PEAK_EXP_PER_S = 256 * 4 * 64 / 16 * 2.4e9      # v_exp_f32 at quarter rate: 256 CUs x 4 SIMDs x 64 lanes / 16 cycles x 2.4 GHz (MI355X_MICROARCH.md)
CONFIGS = {     # BASELINE.json configs[0], [2], [3], [4] (configs[1] is the headline above); cfg5 at the per-GPU share b = 32 / 8
    "cfg1": dict(kw=dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4), b=4,
                 shapes=[(1, 2000), (224, 224, 3)], dtype="float32", core="fp32", dom=("img", 224 * 224, 13)),
    "cfg3": dict(kw=dict(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4), b=16,
                 shapes=[(1, 2000), (224, 224, 3), (12, 224, 224, 3)], dtype="bfloat16", core="bf16", dom=("vol", 12 * 224 * 224, 18)),
    "cfg4": dict(kw=dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4), b=8,
                 shapes=[(1, 2000), (4096, 768)], dtype="float32", core="fp32", dom=("bag", 4096, 773)),
    "cfg5": dict(kw=dict(n_modalities=4, channel_dims=[2000, 768, 768, 3], num_spatial_axes=[1, 1, 1, 3], out_dims=4, depth=8), b=4,
                 shapes=[(1, 2000), (4096, 768), (4096, 768), (12, 224, 224, 3)], dtype="float32", core="fp32",
                 dom=("vol", 12 * 224 * 224, 18)),
}


def configs_record(dev, steps=20, warmup=3):
    """The other BASELINE configs at HEAD, one entry each: forward ms and samples/s over `steps`
    un-instrumented forwards, then the dominant kernel timed with HIP events on the launch stream in an instrumented replay --
    the attention core of the modality with the most tokens through hn_profile (as the headline's), the patch-bag K/V projection
    through hn_set_kernel_timers -- with its executed work per launch (formula in the entry) and the fraction of the roof that
    bounds it: fp32 MFMA for the fp32 kernels, the chip's v_exp_f32 rate for the bf16 core (one exponential per attention score:
    DESIGN.md 4.4)."""
    import healnet_amd as hn
    from healnet_amd import _capi
    out = {}
    for name, c in CONFIGS.items():
        torch.manual_seed(0)
        model = hn.HealNet(**c["kw"], core_precision=c["core"]).eval().to(dev)
        gen = torch.Generator().manual_seed(1234)
        ins = [torch.rand(c["b"], *sh, generator=gen).to(dev).to(getattr(torch, c["dtype"])) for sh in c["shapes"]]
        b, depth = c["b"], model.depth
        kind, n_tok, d_ctx = c["dom"]
        n_dom = sum(1 for sh in c["shapes"] if (len(sh) == 2 and sh[0] == n_tok and kind == "bag"))      # bags share one kernel class
        with torch.no_grad():
            for _ in range(warmup):
                model(list(ins))
            torch.cuda.synchronize(dev)
            # settle like the headline does (the first few hundred ms after an idle period run at a lower clock): keep stepping,
            # untimed, until the device has been busy for ~0.4 s -- 3 warm-up + 10 timed forwards of a 0.7 ms configuration are 9 ms
            # of device time in all, and cfg1 read 0.73 ms there against 0.65 ms in any longer loop (round 6)
            t_settle = time.perf_counter()
            while time.perf_counter() - t_settle < 0.4:
                for _ in range(5):
                    model(list(ins))
                torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                res = model(list(ins))
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / steps * 1e3
            assert torch.isfinite(res).all(), name
            # instrumented replay
            if kind == "bag":
                with KernelTimers(["gemm_nt_x6", "gemm_nt_glds"], depth * n_dom * steps + 8) as kt:
                    for _ in range(steps):
                        model(list(ins))
                    torch.cuda.synchronize(dev)
                avgs = kt.averages_ms()
                if avgs["gemm_nt_x6"][1] > 0:      # the default route: fp32-exact on the bf16 pipe (gemm_x6.hip)
                    avg_ms, n_timed = avgs["gemm_nt_x6"]
                    kp = (d_ctx + 15) // 16 * 16
                    work = 6 * 2.0 * b * n_tok * kp * 2 * HEADS * DIM_HEAD
                    entry_k = {"kernel": "hn::gemm_nt_x6_kernel (patch-bag K/V projection, fp32-exact on the bf16 pipe: 3 bf16 planes per "
                                         "operand, 6 products per fp32 product; HN_NO_X6_GEMM=1 for the fp32-MFMA kernel)",
                               "launches_per_forward": depth * n_dom, "work_per_launch": work, "work_unit": "executed bf16 MFMA FLOPs",
                               "formula": "6 x 2 (b N) Dp (2 inner), N = 4096, Dp = 784 (773 padded to 16), inner = 512", "bound": "mfma (bf16 pipe)",
                               "peak": PEAK_BF16_MFMA_TFLOPS * 1e12,
                               "fp32_equivalent_per_s_note": "2 (b N) D (2 inner) / time: the fp32 product this launch returns"}
                else:
                    avg_ms, n_timed = avgs["gemm_nt_glds"]
                    work = 2.0 * b * n_tok * d_ctx * 2 * HEADS * DIM_HEAD
                    entry_k = {"kernel": "hn::gemm_nt_glds_kernel (patch-bag K/V projection, LDS-DMA operands)", "launches_per_forward": depth * n_dom,
                               "work_per_launch": work, "work_unit": "executed fp32 MFMA FLOPs",
                               "formula": "2 (b N) D (2 inner), N = 4096, D = 773, inner = 512", "bound": "mfma", "peak": PEAK_FP32_MFMA_TFLOPS * 1e12}
            else:
                events = HipEvents(depth * steps)
                prof = _capi.Profile(ev_start=events.start, ev_stop=events.stop, n_events=0, n_recorded=0)
                recorded = 0
                for step in range(steps):
                    off = step * depth * ctypes.sizeof(ctypes.c_void_p)
                    prof.ev_start = ctypes.cast(ctypes.addressof(events.start) + off, ctypes.POINTER(ctypes.c_void_p))
                    prof.ev_stop = ctypes.cast(ctypes.addressof(events.stop) + off, ctypes.POINTER(ctypes.c_void_p))
                    prof.n_events = depth
                    prof.n_recorded = 0
                    model(list(ins), _profile=ctypes.byref(prof))
                    recorded += prof.n_recorded
                torch.cuda.synchronize(dev)
                t = events.elapsed_ms(recorded)
                avg_ms, n_timed = (sum(t) / len(t) if t else None), len(t)
                dp = 16 if d_ctx <= 15 else 32
                kq = 4 * ((d_ctx - 1 + 3) // 4)
                if c["core"] == "bf16":
                    work = 1.0 * L_C * n_tok * HEADS * b
                    entry_k = {"kernel": "hn::attn_core_bf16_kernel (split-KV core of the %s cross-attention on bf16 MFMA)" % kind,
                               "launches_per_forward": depth, "work_per_launch": work, "work_unit": "attention scores (one v_exp_f32 each)",
                               "formula": "l_c N h b, N = %d" % n_tok, "bound": "valu (v_exp_f32, quarter rate)", "peak": PEAK_EXP_PER_S}
                else:
                    work = 2.0 * L_C * n_tok * (kq + dp) * HEADS * b
                    entry_k = {"kernel": "hn::attn_core_kernel (split-KV fp32 core of the %s cross-attention, rank-D binding)" % kind,
                               "launches_per_forward": depth, "work_per_launch": work, "work_unit": "executed fp32 MFMA FLOPs",
                               "formula": "2 l_c N (%d + %d) h b, N = %d (packed context: QK^T over %d channels, P V over %d columns)" % (kq, dp, n_tok, kq, dp),
                               "bound": "mfma", "peak": PEAK_FP32_MFMA_TFLOPS * 1e12}
        entry_k["avg_launch_ms"] = None if avg_ms is None else round(avg_ms, 4)
        entry_k["launches_timed"] = n_timed
        entry_k["achieved_per_s"] = None if not avg_ms else entry_k["work_per_launch"] / (avg_ms * 1e-3)
        entry_k["frac"] = None if not avg_ms else round(entry_k["achieved_per_s"] / entry_k["peak"], 4)
        entry_k["share_of_forward"] = None if not avg_ms else round(avg_ms * entry_k["launches_per_forward"] / ms, 4)
        out[name] = {"workload": "HealNet(%s) forward, b=%d, inputs %s %s resident in HBM, core_precision=%s, eval / no_grad" %
                                 (", ".join("%s=%s" % kv for kv in c["kw"].items()), b, c["shapes"], c["dtype"], c["core"]),
                     "batch": b, "ms_per_forward": round(ms, 4), "samples_per_s": round(b / ms * 1e3, 1), "steps": steps, "warmup": warmup,
                     "dominant_kernel": entry_k}
        del model, ins
        torch.cuda.empty_cache()
    return out


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks of this script under torch.distributed.run."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


TUNED = {   # config/best_hyperparams.yml of the reference: depth, num_latents, latent_dim, cross_dim_head, latent_dim_head, dropouts
    "blca": dict(depth=2, l_c=25, l_d=119, cross_dim_head=16, latent_dim_head=127, attn_dropout=0.0830, ff_dropout=0.4733),
    "brca": dict(depth=2, l_c=17, l_d=126, cross_dim_head=63, latent_dim_head=20, attn_dropout=0.4553, ff_dropout=0.3647),
    "kirp": dict(depth=5, l_c=17, l_d=62, cross_dim_head=27, latent_dim_head=113, attn_dropout=0.3179, ff_dropout=0.0474),
    "ucec": dict(depth=2, l_c=16, l_d=65, cross_dim_head=103, latent_dim_head=51, attn_dropout=0.2488, ff_dropout=0.0571),
}


def staged_models_record(dev, steps=15, warmup=3):
    """The reference's four tuned TCGA configurations (odd latent widths, ONE narrow cross head, dropout on), forward + backward at
    b = 8 on the cfg4 input shapes: they run the fused latent kernels as zero-padded images (DESIGN.md 4.10 "staged models")."""
    import healnet_amd as hn
    gen = torch.Generator().manual_seed(1)
    ins = [torch.rand(TRAIN_BATCH, *s, generator=gen).to(dev) for s in TRAIN_SHAPES]
    rec = {"workload": "forward + backward (both dropouts on), b=8, omic (b,1,2000) + WSI bag (b,4096,768), gradients into the flat buffer",
           "unit": "ms", "steps": steps, "warmup": warmup, "configs": {}}
    for name, kw in TUNED.items():
        torch.manual_seed(0)
        model = hn.HealNet(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4, x_heads=1, l_heads=8,
                           self_per_cross_attn=0, num_freq_bands=2, max_freq=2.0, **kw).to(dev).train()
        flat = hn.train.flatten_parameters(model)

        def step():
            flat.zero_grad()
            model(list(ins)).sum().backward()
        for _ in range(warmup):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        rec["configs"][name] = {"fwd_bwd_ms": round((time.perf_counter() - t0) / steps * 1e3, 4), "staged": bool(model.runs_staged())}
        gstep = hn.train.GraphedStep(model, lambda out: out.sum(), list(ins), ())          # the same work as one graph replay
        for _ in range(warmup):
            gstep(ins)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            gstep(ins)
        torch.cuda.synchronize(dev)
        rec["configs"][name]["fwd_bwd_graphed_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        gstep.close()
        del model, flat, gstep
    return rec


def patch_bag_record(dev, steps=30, warmup=5):
    """Inference forward of BASELINE configs[3]'s shape (omic 1x2000 + WSI bag 4096x768, b=8) in fp32 and with
    core_precision="bf16" (the bag's K/V projection and attention core on bf16 MFMA; tolerance of that
    configuration 2e-2 max-norm against the fp32 oracle, tests/test_gpu_bf16proj.py)."""
    import healnet_amd as hn
    gen = torch.Generator().manual_seed(1)
    ins = [torch.rand(TRAIN_BATCH, *s, generator=gen).to(dev) for s in TRAIN_SHAPES]
    rec = {"workload": "inference forward, b=8, omic (b,1,2000) + WSI bag (b,4096,768), fp32 tensors, eval / no_grad", "unit": "ms",
           "steps": steps, "warmup": warmup}
    outs = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(0)
        model = hn.HealNet(**TRAIN_KW, core_precision=prec).eval().to(dev)
        model.keep_attention_stats = False
        with torch.no_grad():
            for _ in range(warmup):
                y = model(list(ins))
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                y = model(list(ins))
            torch.cuda.synchronize(dev)
        rec[f"{prec}_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        outs[prec] = y.float()
        del model
    rec["bf16_maxnorm_diff_vs_fp32"] = float((outs["bf16"] - outs["fp32"]).abs().max() / outs["fp32"].abs().max())
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH, help="samples per GPU per step (headline config: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-staged-models", action="store_true", help="skip the tuned-shape (staged models) record")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config record (cfg1 / cfg3 / cfg4 / cfg5 forwards at N = 1)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the cfg4 training-step record")
    ap.add_argument("--train-steps", type=int, default=30)
    ap.add_argument("--train-watchdog-s", type=float, default=240.0,
                    help="N > 1: print the line without the training-step record if that record has not returned by then")
    ap.add_argument("--core-precision", choices=["fp32", "bf16", "bf16x3"], default="fp32",
                    help="development switch: bf16 MFMA in the image cross-attention core (the headline is fp32)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HN_BENCH_FORCE_DIST=1 (1-GPU box): take the `distributed` branch end to end on a ONE-rank RCCL group -- nccl process group
    # with device_id, dist.barrier() inside the timed regions, the MAX all-reduce of the timing, the overlapped and the blocking
    # gradient all-reduce of the training step (ReduceOp.AVG on the side stream) all really execute
    force_dist = os.environ.get("HN_BENCH_FORCE_DIST", "0") == "1"
    if force_dist:
        os.environ["HN_FORCE_COLLECTIVES"] = "1"
    distributed = world > 1 or force_dist
    # HN_BENCH_SHARED_GPU=1 (tests on a 1-GPU box only): every rank uses cuda:0 and the ranks talk over gloo -- exercises the
    # launcher and the distributed code paths; RCCL needs one device per rank, which is what the real runs use
    shared = os.environ.get("HN_BENCH_SHARED_GPU", "0") == "1"
    visible = torch.cuda.device_count()
    if not shared and args.gpus > visible:               # one clear line instead of a rendezvous hang / an invalid-device trace
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {visible} GPU(s) are visible on this node "
                         f"(one process per GPU over RCCL; nothing was launched)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                                  # does not return
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        from healnet_amd import dist as hdist
        hdist.init_from_env("gloo" if shared else "nccl")  # nccl = RCCL; one process per GPU

    import healnet_amd
    from healnet_amd import _capi
    torch.manual_seed(0)                                   # every replica holds the same seed-0 default-init model
    model = healnet_amd.HealNet(**KW, core_precision=args.core_precision).eval().to(dev)
    # keep_attention_stats stays at its default (True): every attention block's softmax statistics and input are kept so
    # that Attention.attn_weights (healnet.py:420) can be rebuilt on demand -- zero-copy, the blocks write them in place
    gen = torch.Generator().manual_seed(1234 + rank)       # SURVEY.md §8d synthetic inputs, U[0,1)
    b = args.batch
    tab = torch.rand(b, *TAB, generator=gen).to(dev)
    img = torch.rand(b, *IMG, generator=gen).to(dev)

    n_core = model.depth                                   # dominant-kernel launches per forward
    events = HipEvents(n_core * args.steps)
    prof = _capi.Profile(ev_start=events.start, ev_stop=events.stop, n_events=0, n_recorded=0)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            model([tab, img])
        # settle: keep stepping (untimed) until the device has been busy for ~1.5 s -- the first few hundred ms after
        # an idle period run at a lower clock / with cold caches (measured: 4.26 ms vs 3.74 ms per step)
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < 1.5:
            for _ in range(10):
                model([tab, img])
            torch.cuda.synchronize(dev)
        # ---- timed region: exactly K steps, barrier + synchronize on both sides, no instrumentation
        barrier()
        t0 = time.perf_counter()
        for step in range(args.steps):
            out = model([tab, img])
        barrier()
        elapsed = time.perf_counter() - t0

        # ---- instrumented replay of the same K steps: HIP events recorded on the launch stream around every
        # launch of the dominant kernel (recording timing events between kernels costs ~1 ms per forward on
        # this runtime, so it is kept out of the region `value` is computed from)
        recorded = 0
        t1 = time.perf_counter()
        for step in range(args.steps):
            off = step * n_core * ctypes.sizeof(ctypes.c_void_p)
            prof.ev_start = ctypes.cast(ctypes.addressof(events.start) + off, ctypes.POINTER(ctypes.c_void_p))
            prof.ev_stop = ctypes.cast(ctypes.addressof(events.stop) + off, ctypes.POINTER(ctypes.c_void_p))
            prof.n_events = n_core
            prof.n_recorded = 0
            model([tab, img], _profile=ctypes.byref(prof))
            recorded += prof.n_recorded
        barrier()
        elapsed_instrumented = time.perf_counter() - t1
    assert torch.isfinite(out).all()

    if distributed:
        elapsed = hdist.max_over_ranks(elapsed, dev)      # the slowest rank defines the step time

    if rank == 0:
        total_samples = b * args.steps * world
        core_ms = events.elapsed_ms(recorded)
        avg_core_ms = sum(core_ms) / max(1, len(core_ms))
        bf16_core = args.core_precision != "fp32"
        # bf16 development switch: QK^T contracts 32 channel slots, P V 16 columns, on v_mfma_f32_16x16x32_bf16
        exec_flops = (2.0 * L_C * N_IMG * (32 + DP) * HEADS) if bf16_core else EXEC_FLOPS_CORE_PER_SAMPLE
        peak = 2500.0 if bf16_core else PEAK_FP32_MFMA_TFLOPS
        exec_tf = exec_flops * b / (avg_core_ms * 1e-3) / 1e12 if core_ms else None
        algo_tf = ALGO_FLOPS_CORE_PER_SAMPLE * b / (avg_core_ms * 1e-3) / 1e12 if core_ms else None
        useful_tf = 2.0 * L_C * N_IMG * (QK_DIM + 13) * HEADS * b / (avg_core_ms * 1e-3) / 1e12 if core_ms else None
        traffic, traffic_label = (None, {"reason": "bf16 development switch"}) if bf16_core else pmc_traffic()
        result = {
            "metric": "fusion-forward samples/sec (2-modality, b=32)",
            "value": round(total_samples / elapsed, 2),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16 attention-core operands, f32 elsewhere (development switch, not the headline)" if bf16_core else "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: HealNet(2,[2000,3],[1,2],4) default hyper-parameters, forward (logits) on "
                                   "tab (b,1,2000) + img (b,224,224,3) U[0,1) fp32 resident in HBM, eval/no_grad",
                       "batch_per_gpu": b, "global_batch": b * world, "parallelism": f"dp{world} (independent replicas, "
                       "no data-path collective)", "seed": "model torch.manual_seed(0); inputs 1234+rank"},
            "forward_tflops_algorithmic": round(FLOPS_FORWARD_PER_SAMPLE * total_samples / elapsed / 1e12, 2),
            "roofline": {
                "kernel": ("hn::attn_core_bf16_kernel<1,4>" if bf16_core else "hn::attn_core_kernel<1,4,true,3>") +
                          " (split-KV attention core of the image cross-attention, N=50176)",
                "bound": "mfma",
                "achieved": None if exec_tf is None else round(exec_tf, 2),
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": None if exec_tf is None else round(exec_tf / peak, 4),
                "frac_useful": None if (useful_tf is None or bf16_core) else round(useful_tf / peak, 4),
                "traffic": traffic,
                "traffic_source": traffic_label,
                "avg_launch_ms": round(avg_core_ms, 4),
                "timing": "hipEvent pairs on the launch stream around each of the 3 launches per forward, recorded in an "
                          "instrumented replay of the same K steps right after the timed region",
                "instrumented_ms_per_step": round(elapsed_instrumented / args.steps * 1e3, 4),
                "launches_timed": len(core_ms),
                "flops_per_launch_executed": exec_flops * b,
                "flops_per_launch_algorithmic": ALGO_FLOPS_CORE_PER_SAMPLE * b,
                "effective_algorithmic_tflops": None if algo_tf is None else round(algo_tf, 2),
                # the whole forward against the same roof: executed matrix FLOPs of every kernel of one step / step time / peak
                "forward_flops_executed_per_step": None if bf16_core else exec_flops_forward_per_sample() * b,
                "forward_frac_executed": None if bf16_core else round(
                    exec_flops_forward_per_sample() * total_samples / elapsed / 1e12 / world / PEAK_FP32_MFMA_TFLOPS, 4),
                "note": "achieved/frac use EXECUTED fp32-MFMA FLOPs (rank-D reassociation + packed context: QK^T contracts 12 "
                        "channels and P V 16 columns instead of dim_head 64 each, 4.6x fewer than the reference formulation, "
                        "SURVEY.md §8d); frac_useful prices only the useful P V columns (12 channels + the softmax-denominator ones "
                        "column = 13 of the 16 MFMA columns; 3 are idle padding); effective_algorithmic_tflops "
                        "prices the same launch at the reference formulation's 13.15 GF/sample",
            },
        }
    # ---- secondary records.  None of them may cost the driver the headline line measured above: an exception becomes an
    # {"error": ...} entry, and at N > 1 -- where the training-step record runs collectives on a fabric this code has never met --
    # a watchdog prints the line without it and ends every rank if the record has not returned in time (a hung collective cannot
    # be caught as an exception; every rank arms the same timer, so none is left behind holding the launcher).
    def emit():
        if rank == 0:
            print(json.dumps(result), flush=True)

    def guarded(fn, *a):
        try:
            return fn(*a)
        except Exception as e:      # noqa: BLE001
            return {"error": f"{type(e).__name__}: {e}"[:400]}

    if rank == 0:
        result["build_id"] = _capi.lib().hn_build_id().decode()
    if not args.no_train_step:
        del model, tab, img
        torch.cuda.empty_cache()
        watchdog = None
        if world > 1:
            def fire():
                if rank == 0:
                    result["train_step"] = {"error": f"watchdog: the training-step record did not return within {args.train_watchdog_s} s "
                                                     "(hung collective or a dead rank); the headline above was measured before it"}
                emit()
                os._exit(0)
            watchdog = threading.Timer(args.train_watchdog_s, fire)
            watchdog.daemon = True
            watchdog.start()
        train_rec = guarded(train_step_record, dev, rank, world, distributed, barrier, args.train_steps, 5)
        if watchdog is not None:
            watchdog.cancel()
        if world > 1 and "error" in train_rec:      # the peers may be parked in a collective this rank left: no clean shutdown
            if rank == 0:
                result["train_step"] = train_rec
            emit()
            sys.stdout.flush()
            os._exit(0)
        if rank == 0:
            if world == 1 and not args.no_cpu_baseline and "error" not in train_rec:
                train_rec["cpu_baseline_forward"] = guarded(cpu_baseline_cfg4)
            result["train_step"] = train_rec
    if rank == 0:
        if world == 1 and not args.no_staged_models:
            result["staged_models"] = guarded(staged_models_record, dev)
            result["patch_bag_precisions"] = guarded(patch_bag_record, dev)
        if world == 1 and not args.no_configs:
            result["configs"] = guarded(configs_record, dev)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = guarded(cpu_baseline)
    emit()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
