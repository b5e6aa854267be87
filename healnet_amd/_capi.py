"""ctypes binding of libhealnet_hip.so (C ABI declared in include/healnet_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised -- the product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from typing import Optional

HN_MAX_AXES = 4
HN_ABI_VERSION = 5
HN_F32, HN_BF16, HN_U8 = 0, 1, 2
HN_CORE_F32, HN_CORE_BF16, HN_CORE_BF16X3 = 0, 1, 2
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HN_LIB_PATH") or os.path.join(_HERE, "libhealnet_hip.so")   # HN_LIB_PATH: kernel experiments (tools/)
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["api.hip", "gemm.hip", "attention.hip", "attention_bf16.hip", "attention_bwd.hip", "encode.hip", "misc.hip",
           "backward.hip", "train.hip", "chain.hip", "self_attention.hip", "bchain.hip", "gemm_bf16.hip"]

c_float_p = C.POINTER(C.c_float)


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("offset", C.c_uint32), ("stream", C.c_uint32)]


class AttnParams(C.Structure):
    _fields_ = [
        ("heads", C.c_int), ("dim_head", C.c_int), ("query_dim", C.c_int),
        ("norm_w", C.c_void_p), ("norm_b", C.c_void_p),
        ("ctx_gamma", C.c_void_p), ("ctx_beta", C.c_void_p),
        ("w_q", C.c_void_p), ("w_kv", C.c_void_p), ("w_out", C.c_void_p), ("b_out", C.c_void_p),
        ("dropout", C.c_float), ("rng", Rng),
        ("dim_head_valid", C.c_int), ("query_dim_valid", C.c_int),        # staged layout (0 = off; include/healnet_hip.h)
    ]


class FFParams(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("gate", C.c_int),
        ("norm_w", C.c_void_p), ("norm_b", C.c_void_p),
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("dropout", C.c_float), ("rng", Rng),
        ("dim_valid", C.c_int),
    ]


class AttnGrads(C.Structure):
    _fields_ = [("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("ctx_gamma", C.c_void_p), ("ctx_beta", C.c_void_p),
                ("w_q", C.c_void_p), ("w_kv", C.c_void_p), ("w_out", C.c_void_p), ("b_out", C.c_void_p)]


class FFGrads(C.Structure):
    _fields_ = [("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p),
                ("b2", C.c_void_p)]


class ModalityInput(C.Structure):
    _fields_ = [("data", C.c_void_p), ("spatial", C.c_int * HN_MAX_AXES), ("dtype", C.c_int)]


class Model(C.Structure):
    _fields_ = [
        ("n_modalities", C.c_int), ("depth", C.c_int), ("l_c", C.c_int), ("l_d", C.c_int),
        ("self_per_cross_attn", C.c_int), ("final_classifier_head", C.c_int), ("out_dims", C.c_int),
        ("num_freq_bands", C.c_int), ("max_freq", C.c_float), ("fourier_encode_data", C.c_int),
        ("channel_dims", C.POINTER(C.c_int)), ("num_spatial_axes", C.POINTER(C.c_int)),
        ("latents", C.c_void_p),
        ("cross_attn", C.POINTER(AttnParams)), ("cross_ff", C.POINTER(FFParams)),
        ("self_attn", C.POINTER(AttnParams)), ("self_ff", C.POINTER(FFParams)),
        ("head_norm_w", C.c_void_p), ("head_norm_b", C.c_void_p), ("head_w", C.c_void_p), ("head_b", C.c_void_p),
        ("core_precision", C.c_int), ("rng", Rng),
        ("l_d_valid", C.c_int),
    ]


class ModelGrads(C.Structure):
    _fields_ = [("latents", C.c_void_p), ("cross_attn", C.POINTER(AttnGrads)), ("cross_ff", C.POINTER(FFGrads)),
                ("self_attn", C.POINTER(AttnGrads)), ("self_ff", C.POINTER(FFGrads)),
                ("head_norm_w", C.c_void_p), ("head_norm_b", C.c_void_p), ("head_w", C.c_void_p), ("head_b", C.c_void_p)]


class Profile(C.Structure):
    _fields_ = [("ev_start", C.POINTER(C.c_void_p)), ("ev_stop", C.POINTER(C.c_void_p)),
                ("n_events", C.c_int), ("n_recorded", C.c_int)]


READY_FN = C.CFUNCTYPE(None, C.c_int, C.c_void_p)


class GradReady(C.Structure):
    _fields_ = [("events", C.POINTER(C.c_void_p)), ("notify", READY_FN), ("user", C.c_void_p)]


# every symbol include/healnet_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "hn_abi_version": (C.c_int, []),
    "hn_last_error_string": (C.c_char_p, []),
    "hn_fourier_encode_concat": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_float,
                                           C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "hn_encode_norm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_float, C.c_int,
                                 C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "hn_context_pitch": (C.c_int, [C.c_int, C.c_int]),
    "hn_attn_fwd": (C.c_int, [C.POINTER(AttnParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_attn_workspace_bytes": (C.c_size_t, [C.POINTER(AttnParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hn_attn_probs": (C.c_int, [C.POINTER(AttnParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_attn_importance": (C.c_int, [C.POINTER(AttnParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_fourier_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_void_p]),
    "hn_glu_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "hn_temperature_softmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_void_p]),
    "hn_dropout_mask": (C.c_int, [C.c_float, Rng, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "hn_surv_nll": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hn_l1_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_l1_adam_workspace_bytes": (C.c_size_t, []),
    "hn_ff_fwd": (C.c_int, [C.POINTER(FFParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_ff_workspace_bytes": (C.c_size_t, [C.POINTER(FFParams), C.c_int]),
    "hn_attn_saved_floats": (C.c_size_t, [C.POINTER(AttnParams)] + [C.c_int] * 7),
    "hn_attn_fwd_train": (C.c_int, [C.POINTER(AttnParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_attn_bwd": (C.c_int, [C.POINTER(AttnParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AttnGrads),
                              C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_attn_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(AttnParams)] + [C.c_int] * 7),
    "hn_ff_bwd": (C.c_int, [C.POINTER(FFParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(FFGrads),
                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_ff_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(FFParams), C.c_int]),
    "hn_head_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_head_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "hn_head_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                              C.c_void_p, C.c_void_p]),
    "hn_fusion_forward": (C.c_int, [C.POINTER(Model), C.POINTER(ModalityInput), C.c_int, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.POINTER(Profile)]),
    "hn_fusion_is_staged": (C.c_int, [C.POINTER(Model)]),
    "hn_fusion_tape_bytes": (C.c_size_t, [C.POINTER(Model), C.POINTER(ModalityInput), C.c_int, C.c_int, C.c_int]),
    "hn_fusion_tape_layout": (C.c_int, [C.POINTER(Model), C.POINTER(ModalityInput), C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "hn_fusion_forward_train": (C.c_int, [C.POINTER(Model), C.POINTER(ModalityInput), C.c_int, C.c_void_p, C.c_int, C.c_int,
                                          C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_fusion_backward_workspace_bytes": (C.c_size_t, [C.POINTER(Model), C.POINTER(ModalityInput), C.c_int, C.c_int]),
    "hn_fusion_backward": (C.c_int, [C.POINTER(Model), C.POINTER(ModalityInput), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.POINTER(ModelGrads), C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.POINTER(GradReady)]),
    "hn_fusion_workspace_bytes": (C.c_size_t, [C.POINTER(Model), C.POINTER(ModalityInput), C.c_int]),
    "hn_latent_block_fwd": (C.c_int, [C.POINTER(AttnParams), C.POINTER(FFParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hn_latent_block_workspace_bytes": (C.c_size_t, [C.POINTER(AttnParams), C.POINTER(FFParams), C.c_int, C.c_int]),
    "hn_latent_block_bwd": (C.c_int, [C.POINTER(AttnParams), C.POINTER(FFParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AttnGrads), C.POINTER(FFGrads), C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "hn_latent_block_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(AttnParams), C.POINTER(FFParams), C.c_int, C.c_int]),
}

_lib: Optional[C.CDLL] = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 into healnet_amd/libhealnet_hip.so (in-tree).  One hipcc process per
    translation unit, in parallel, objects under healnet_amd/build/ (git-ignored); a unit is recompiled when it, common.h or
    the public header is newer than its object.  ``force`` (or HN_FORCE_REBUILD=1) rebuilds everything from a clean slate."""
    from concurrent.futures import ThreadPoolExecutor
    force = force or os.environ.get("HN_FORCE_REBUILD", "0") == "1"
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "chain_common.h"), os.path.join(_HERE, "..", "include", "healnet_hip.h")]
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: MFMA accumulators live in plain VGPRs (gfx950 has a unified register file), which
    # removes the per-iteration v_accvgpr_read/write shuffles hipcc otherwise emits around the softmax / epilogues.
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + \
        os.environ.get("HN_EXTRA_HIPCC_FLAGS", "").split()
    stamp = os.path.join(objdir, "flags.txt")
    flag_text = " ".join([hipcc] + flags)
    # a rebuild is forced only when a flag stamp EXISTS and differs; a box that received the prebuilt library without the
    # build/ directory (no stamp) keeps it as long as it is newer than every source
    if os.path.exists(stamp) and open(stamp).read() != flag_text:
        force = True
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(
            [hdr_time] + [os.path.getmtime(os.path.join(CSRC, src)) for src in SOURCES]):
        if not os.path.exists(stamp):
            try:
                with open(stamp, "w") as f:
                    f.write(flag_text)
            except OSError:
                pass
        return LIB_PATH          # up to date (also the case on a GPU box that received the prebuilt library without objects)
    jobs = []
    for src in SOURCES:
        path, obj = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), hdr_time):
            jobs.append((path, obj))
    objs = [os.path.join(objdir, src + ".o") for src in SOURCES]
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def compile_one(job):
        cmd = [hipcc] + flags + ["-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        list(pool.map(compile_one, jobs))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.run(link, check=True, cwd=CSRC)
    with open(stamp, "w") as f:
        f.write(flag_text)
    return LIB_PATH


def lib() -> C.CDLL:
    """Load the library once; raise loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"healnet_amd: {LIB_PATH} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.hn_abi_version() != HN_ABI_VERSION:
            raise RuntimeError("healnet_amd: ABI version mismatch between _capi.py and libhealnet_hip.so")
        _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().hn_last_error_string()
        raise RuntimeError(f"healnet_hip {what} failed (status {status}): {msg.decode() if msg else '?'}")
