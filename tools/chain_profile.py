#!/usr/bin/env python3
"""Stage-by-stage cycle profile of latent_chain_kernel (development tool).

    python tools/chain_profile.py [batch]

Builds a private copy of the library with -DCHAIN_PROFILE (chain.hip then stamps s_memtime at its stage boundaries for one
workgroup), runs cfg2 forwards through it and prints, per chain of one forward, the cycles between the stamps."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "healnet_amd", "libhealnet_prof.so")
if not os.path.exists(lib):
    src = os.path.join(ROOT, "healnet_amd", "csrc")
    from healnet_amd import _capi as _c
    objs = [os.path.join(ROOT, "healnet_amd", "build", src + ".o") for src in _c.SOURCES if src != "chain.hip"]      # (the product's own unit list: a stale object of a removed unit must not be linked)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-DCHAIN_PROFILE",
                           "-c", os.path.join(src, "chain.hip"), "-o", "/tmp/chain_prof.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["/tmp/chain_prof.o"])
if len(sys.argv) > 1 and sys.argv[1] == "--build-only":
    sys.exit(0)
os.environ["HN_LIB_PATH"] = lib
import torch
import healnet_amd as hn
from healnet_amd import _capi
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to("cuda:0")
m.keep_attention_stats = False
torch.set_grad_enabled(False)
tab, img = torch.rand(b, 1, 2000, device="cuda:0"), torch.rand(b, 224, 224, 3, device="cuda:0")
for _ in range(5):
    m([tab, img])
torch.cuda.synchronize()
L = ctypes.CDLL(lib)
buf = (ctypes.c_ulonglong * 256)()
seq = ctypes.c_int()
assert L.hn_debug_chain_prof(buf, ctypes.byref(seq)) == 0
names = ["entry->table", "prologue (x, O, params, 5 blocks)", "OUT", "LN ff", "FF1", "FF2", "x_out..LN'", "Q", "KV", ]
print("launches so far", seq.value, "(12 per forward)")
for k in range(12):
    s = (seq.value - 12 + k) & 15
    t = [buf[s * 16 + i] for i in range(16)]
    meta = t[10]
    nb, nko, nq, nkv = meta & 0xffff, (meta >> 16) & 0xffff, (meta >> 32) & 0xffff, (meta >> 48) & 0xffff
    d = [t[i + 1] - t[i] for i in range(9)]
    # stamps 7/8 only exist with projections; 3 only with the out stage etc.: print raw deltas
    print(f"chain {k}: blocks {nb} (out {nko}, ff {48 if nb - nko - 4*(nq+nkv) else 0}, q {4*nq}, kv {4*nkv})  total {t[9]-t[0]} cycles = {(t[15]-t[14]) / 100.0:.2f} us (100 MHz clock)")
    print("    " + "  ".join(f"{n}: {x}" for n, x in zip(names, d)))
