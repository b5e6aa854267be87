#!/usr/bin/env python3
"""Stage-by-stage time profile of latent_layer_kernel (development tool).   python tools/lchain_profile.py [batch]

Builds a private copy of the library with -DLCHAIN_PROFILE (lchain.hip then stamps the 100 MHz clock at its stage boundaries for one
workgroup), runs cfg2 forwards through it and prints, per layer chain of the last forward, the microseconds between the stamps."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
extra = os.environ.get("HN_PROF_EXTRA", "").split()      # extra hipcc flags for an A/B variant of lchain.hip (own library name)
lib = os.path.join(ROOT, "healnet_amd", "libhealnet_prof_l%s.so" % ("_" + "".join(c for c in "".join(extra) if c.isalnum()) if extra else ""))
src = os.path.join(ROOT, "healnet_amd", "csrc")
if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(os.path.join(src, "lchain.hip")):
    from healnet_amd import _capi as _c
    objs = [os.path.join(ROOT, "healnet_amd", "build", src + ".o") for src in _c.SOURCES if src != "lchain.hip"]      # (the product's own unit list: a stale object of a removed unit must not be linked)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-DLCHAIN_PROFILE"] + extra + [
                           "-c", os.path.join(src, "lchain.hip"), "-o", "/tmp/lchain_prof.o"], stderr=subprocess.DEVNULL)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["/tmp/lchain_prof.o"])
if len(sys.argv) > 1 and sys.argv[1] == "--build-only":
    sys.exit(0)
os.environ["HN_LIB_PATH"] = lib
import torch
import healnet_amd as hn
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
buf = (ctypes.c_ulonglong * 512)()
seq = ctypes.c_int()
assert L.hn_debug_lchain_prof(buf, ctypes.byref(seq)) == 0
names = ["PV+O (prev)", "OUT", "LN", "FF1", "FF2", "x_out+LN'", "K|V (or qf)", "-", "drain+flag", "Q01", "wait", "Q23+qread", "S", "softmax"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    m([tab, img])
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 100:.4f} ms per forward ({' '.join(extra) or 'product flags'})")
print("launches so far", seq.value, "(4 per forward)")
for k in range(4):
    s = (seq.value - 4 + k) & 7
    t = [buf[s * 64 + i] for i in range(64)]
    nseg = t[63]
    last = max(x for x in t[:63] if x)
    print(f"layer chain {k}: {nseg} segments, {(last - t[0]) / 100.0:.2f} us;  entry->table {(t[1] - t[0]) / 100:.2f}  prologue {(t[2] - t[1]) / 100:.2f}")
    prev = t[2]
    for g in range(int(nseg)):
        parts = []
        for j in range(14):
            x = t[3 + g * 14 + j]
            if x == 0 or x < prev:
                continue
            if j > 0 or g > 0:
                parts.append(f"{names[j]} {(x - prev) / 100.0:.2f}")
            prev = x
        print(f"   seg {g}: " + "  ".join(parts))
