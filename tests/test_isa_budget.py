"""CPU: register / scratch budgets of the hot kernels, read from hipcc's own metadata (no GPU needed: gfx950 cross-compiles here).

Why this exists: at the end of round 5 a refactor of the image core's token loop left the DROPOUT variants of `attn_core_kernel` with
their fragment / accumulator arrays in scratch (`.private_segment_fixed_size` 832 bytes, the step lambda no longer inlined).  Every
parity test stayed green -- the results were right -- and the dropout training step took 48 ms instead of 10.9.  Numbers that decide
occupancy or put arrays into memory are checked here, where a change that moves them fails a test instead of a benchmark."""
import os
import re
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "healnet_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only"]

pytestmark = pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not available")


def _metadata(unit):
    """{kernel symbol: (scratch bytes, VGPRs, spilled VGPRs)} of one translation unit."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, unit + ".s")
        subprocess.run([HIPCC] + FLAGS + ["-o", out, os.path.join(CSRC, unit)], check=True, cwd=CSRC, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        text = open(out).read()
    table = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        body = m.group(2)

        def field(key):
            return int(re.search(key + r":\s+(\d+)", body).group(1))
        table[m.group(1)] = (field(r"\.private_segment_fixed_size"), field(r"\.vgpr_count"), field(r"\.vgpr_spill_count"))
    return table


@pytest.fixture(scope="module")
def meta():
    from healnet_amd import _capi
    units = list(_capi.SOURCES)                  # every translation unit of the library (about 25 s on eight threads)
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 4)) as pool:
        return dict(zip(units, pool.map(_metadata, units)))


def test_no_kernel_keeps_its_fragment_arrays_in_scratch(meta):
    """A few spilled registers outside the loops are tolerated (the image core trades 11 for its fourth wave per SIMD); an ARRAY in
    scratch is hundreds of bytes per lane."""
    assert sum(len(t) for t in meta.values()) > 100
    for unit, table in meta.items():              # (the api_*.hip units hold launch schedules: some define no kernel)
        for name, (scratch, vgprs, spilled) in table.items():
            assert scratch <= 128, f"{unit}: {name} uses {scratch} bytes of scratch per lane ({vgprs} VGPRs, {spilled} spilled)"


def test_image_core_fits_four_waves_per_simd(meta):
    """attn_core_geometry plans 4096 resident waves for the dp = 16 image core (4 per SIMD): 128 VGPRs, no more."""
    hot = [k for k in meta["attention.hip"] if "attn_core_kernelILi1ELi4ELb1ELi" in k and k.split("ELb1ELi")[1][1:].startswith("ELb0")]
    assert len(hot) == 4, hot                                   # KS = 1 .. 4, no dropout
    for k in hot:
        assert meta["attention.hip"][k][1] <= 128, (k, meta["attention.hip"][k])
    vol = [k for k in meta["attention.hip"] if "attn_core_kernelILi2ELi2ELb1ELi5ELb0" in k or "attn_core_kernelILi2ELi2ELb1ELi6ELb0" in k]
    assert len(vol) == 2
    for k in vol:                                               # the dp = 32 volume core of cfg3 / cfg5 (18 channels: 5 k-steps)
        assert meta["attention.hip"][k][1] <= 128, (k, meta["attention.hip"][k])


def test_chain_and_self_core_keep_two_waves_per_simd(meta):
    """512-thread workgroups: two waves per SIMD need <= 256 VGPRs; the default chain stays well below (149-152 at the time of writing),
    which is what lets its EXT = false instance hold the weight ring and the merge head's landing registers apart."""
    for k, (scratch, vgprs, _) in meta["chain.hip"].items():
        if "latent_chain_kernel" in k:
            assert vgprs <= 160 and scratch == 0, (k, vgprs, scratch)
    for k, (scratch, vgprs, _) in meta["self_attention.hip"].items():
        assert vgprs <= 128 and scratch == 0, (k, vgprs, scratch)


def test_layer_chain_holds_its_attention_phase_in_registers(meta):
    """latent_layer_kernel (lchain.hip): 512-thread workgroups, two waves per SIMD = at most 256 VGPRs, and NOTHING in scratch -- a
    spilled lane address reloaded inside the attention phase is a scratch load that waits for the whole weight ring with vmcnt(0)
    (round 6: four spilled registers cost ~1 us per self-attention block until the hoisted addresses were pinned to their use)."""
    hot = [v for k, v in meta["lchain.hip"].items() if "latent_layer_kernel" in k]
    assert len(hot) == 1
    scratch, vgprs, spilled = hot[0]
    assert scratch == 0 and spilled == 0 and vgprs <= 256, hot[0]


def test_x6_gemms_keep_accumulators_and_fragments_in_registers(meta):
    """gemm_nt_x6_kernel (gemm_x6.hip): 512-thread workgroups at one per CU = two waves per SIMD, at most 256 VGPRs; 128 (NT) / 80
    (TN) accumulator registers and the double-buffered fragment sets must not spill -- a scratch access inside the k-loop would sit
    behind the LDS-DMA ring (the kernel's waits on it are counted by hand)."""
    hot = {k: v for k, v in meta["gemm_x6.hip"].items() if "gemm_nt_x6_kernel" in k}
    assert len(hot) >= 3, sorted(hot)
    for k, (scratch, vgprs, spilled) in hot.items():
        assert scratch == 0 and spilled == 0 and vgprs <= 256, (k, scratch, vgprs, spilled)
