#!/usr/bin/env python3
"""Per-config roofline evidence from the CURRENT build (Next 2): for every BASELINE config that names a roofline
(cfg2 fp32, cfg3 bf16, cfg4 fp32 forward, cfg5 fp32 and bf16 core) run

  1. `rocprofv3 --kernel-trace --stats` over `tools/bench_configs.py --cfg C --core-precision P`  -> <tag>_cfgC_P_kernel_stats.csv
  2. ONE `rocprofv3 --kernel-trace --pmc` pass (SQ / GRBM set; never combined with another trace domain) over the same command
  3. the un-profiled forward time of the same command

and write <tag>_rooflines.json: per config the dominant kernel, its EXECUTED FLOPs per launch with the formula, its average
duration (from the CSV of step 1), the fraction of the right peak, the MFMA / VALU busy shares (step 2), and the END-TO-END
executed fraction: sum of executed matrix FLOPs of one forward / forward time / peak.

    python tools/roofline_configs.py --out gpurun_out/r03_a --tag r03_a [--cfg 2 3 4 5]

Peaks (MI355X_MICROARCH.md chip table): fp32 MFMA 157.3 TF/s.  The bf16 cores are bound by `v_exp_f32`, not by the bf16 MFMA
(DESIGN 4.4): their roof is the chip's exponential rate, 256 CUs x 4 SIMDs x 64 lanes / 16 cycles (quarter rate) x 2.4 GHz =
9.83e12 exp/s, and their work unit is one attention score (= one exponential).
"""
import argparse, collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_FP32 = 157.3e12
PEAK_BF16 = 2.5e15      # dense bf16 MFMA (the patch-bag K/V projection under core_precision = bf16, gemm_bf16.hip)
PEAK_EXP = 256 * 4 * 64 / 16 * 2.4e9
L, d, H, DH = 128, 128, 8, 64
INNER = H * DH

# modality kinds: (name, N tokens, D context channels)
TAB = ("tab", 1, 2005)
IMG = ("img", 224 * 224, 13)
VOL = ("vol", 12 * 224 * 224, 18)
BAG = ("bag", 4096, 773)
CFG = {2: dict(b=32, depth=3, mods=[TAB, IMG]), 3: dict(b=16, depth=3, mods=[TAB, IMG, VOL]),
       4: dict(b=8, depth=3, mods=[TAB, BAG]), 5: dict(b=4, depth=8, mods=[TAB, BAG, BAG, VOL])}


def rank_d_geometry(D):
    """(dp, 4*KS): context pitch and the packed QK^T contraction of the fp32 rank-D core (D-1 kept channels)."""
    dp = 16 if D <= 15 else 32
    return dp, 4 * ((D - 1 + 3) // 4)


def core_flops(mod, b, prec):
    """Executed matrix FLOPs of ONE launch of the modality's attention core, with the formula as text."""
    name, N, D = mod
    if name == "tab":
        return 0.0, "one-token shortcut: no attention core"
    if name == "bag":
        return 2.0 * L * N * (DH + DH) * H * b, "2*l_c*N*(dh + dh)*h*b (explicit K/V binding: QK^T over dim_head, P.V over dim_head)"
    dp, kq = rank_d_geometry(D)
    if prec == "bf16":
        return 2.0 * L * N * (32 + dp) * H * b, f"2*l_c*N*(32 + {dp})*h*b (bf16 core: QK^T contracts one 32-slot MFMA k-step, P.V {dp} columns)"
    return 2.0 * L * N * (kq + dp) * H * b, f"2*l_c*N*({kq} + {dp})*h*b (rank-D reassociation + packed context: QK^T over {kq} channels, P.V over {dp} columns)"


def forward_flops(cfg, prec):
    """Executed matrix FLOPs of one forward, by component (the schedule of healnet.py:225-245)."""
    b, depth = cfg["b"], cfg["depth"]
    rows = b * L
    ff = 2.0 * rows * d * 8 * d + 2.0 * rows * 4 * d * d
    comp = collections.OrderedDict()
    add = lambda k, v: comp.__setitem__(k, comp.get(k, 0.0) + v)      # noqa: E731
    for _ in range(depth):
        for mod in cfg["mods"]:
            name, N, D = mod
            if name == "tab":
                add("one-token cross blocks (W_v c, W_out)", 2.0 * b * D * INNER + 2.0 * b * INNER * d)
            elif name == "bag":
                add("latent-side projections of cross blocks (Q, out)", 2.0 * rows * d * INNER + 2.0 * rows * INNER * d)
                add("patch-bag K/V projection", 2.0 * b * N * D * 2 * INNER)
                add("attention cores (bag)", core_flops(mod, b, prec)[0])
            else:
                dp, _ = rank_d_geometry(D)
                add("latent-side projections of cross blocks (Q, out)", 2.0 * rows * d * INNER + 2.0 * rows * INNER * d)
                add("rank-D folds (query fold, value projection)", 2 * 2.0 * rows * INNER * dp)
                add(f"attention cores ({name})", core_flops(mod, b, prec)[0])
            add("feed-forward blocks", ff)
            # the latent self block runs behind every modality (healnet.py:241-245)
            add("latent self-attention (Q|K|V, core, out)", 2.0 * rows * d * 3 * INNER + 4.0 * L * L * DH * H * b + 2.0 * rows * INNER * d)
            add("feed-forward blocks", ff)
    return comp


def run(cmd, **kw):
    return subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r03_rooflines")
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--cfg", type=int, nargs="+", default=[2, 3, 4, 5])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-pmc", action="store_true")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    cases = [(c, p) for c in args.cfg for p in {2: ["fp32"], 3: ["bf16"], 4: ["fp32", "bf16"], 5: ["fp32", "bf16"]}[c]]
    doc = {"peaks": {"fp32_mfma_flops": PEAK_FP32, "v_exp_f32_per_s": PEAK_EXP,
                     "derivation": "fp32 MFMA: MI355X_MICROARCH.md chip table; exp: 256 CUs x 4 SIMDs x 64 lanes / 16 cycles x 2.4 GHz"},
           "git_head": os.environ.get("HN_GIT_HEAD"), "configs": []}
    for c, prec in cases:
        cfg = CFG[c]
        base = [sys.executable, os.path.join(ROOT, "tools/bench_configs.py"), "--cfg", str(c), "--core-precision", prec, "--steps", str(args.steps)]
        tagc = f"{args.tag}_cfg{c}_b{cfg['b']}_{prec}"
        # 3. un-profiled forward time
        r = run(base, capture_output=True, text=True, timeout=900)
        row = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        ms = row["ms_per_forward"]
        # 1. kernel stats
        dstat = os.path.join(out, tagc + "_stats")
        shutil.rmtree(dstat, ignore_errors=True)
        run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", dstat, "-o", "s", "--"] + base,
            timeout=1200, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = glob.glob(os.path.join(dstat, "**", "*kernel_stats.csv"), recursive=True)
        assert files, f"no kernel_stats.csv under {dstat}"
        csv_path = os.path.join(out, tagc + "_kernel_stats.csv")
        shutil.copy(files[0], csv_path)
        shutil.rmtree(dstat, ignore_errors=True)
        kernels = list(csv.DictReader(open(csv_path)))
        forwards = args.steps + 3                           # bench_configs.py: 3 warm-up forwards + the timed ones
        total_ns = sum(float(k["TotalDurationNs"]) for k in kernels)
        # 2. one PMC pass
        pmc = {}
        if not args.no_pmc:
            dp_ = os.path.join(out, tagc + "_pmc")
            shutil.rmtree(dp_, ignore_errors=True)
            counters = "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
            try:
                run(["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", dp_, "-o", "p", "--"] +
                    base[:-1] + ["3"], timeout=900, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            except subprocess.TimeoutExpired:
                print(f"{tagc}: PMC pass timed out", file=sys.stderr)
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for f in glob.glob(os.path.join(dp_, "**", "*counter_collection.csv"), recursive=True):
                for r_ in csv.DictReader(open(f)):
                    agg[r_["Kernel_Name"]][r_["Counter_Name"]].append(float(r_["Counter_Value"]))
            for n, cs in agg.items():
                v = {k: sum(x) / len(x) for k, x in cs.items()}
                if v.get("GRBM_GUI_ACTIVE"):
                    cyc = v["GRBM_GUI_ACTIVE"] / 8.0          # per-XCD active cycles
                    v["mfma_pipe_busy"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc
                    # SQ_ACTIVE_INST_VALU counts per-wave busy cycles summed over waves: per SIMD share (4 cycles per quad-rate issue)
                    v["valu_active_per_simd"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / 1024.0 / cyc
                pmc[n] = v
            shutil.rmtree(dp_, ignore_errors=True)
        # dominant kernel
        # (among the kernels this tool can price: attention cores and the patch-bag projection; at cfg4 with the bf16 projection the
        # latent chain's 12 launches add up to as much as either, see the "kernels" table of the entry)
        priced = [k for k in kernels if any(t in k["Name"] for t in ("attn_core", "gemm_big", "gemm_nt_glds", "gemm_bf16_kernel"))]
        dom = max(priced or kernels, key=lambda k: float(k["TotalDurationNs"]))
        dom_name = dom["Name"]
        dom_us = float(dom["AverageNs"]) / 1e3
        # which modality the dominant kernel serves: the one with the largest core FLOPs
        mod = max(cfg["mods"], key=lambda m: core_flops(m, cfg["b"], prec)[0])
        if "gemm_big" in dom_name or "gemm_nt_glds" in dom_name or "gemm_bf16_kernel" in dom_name:
            fl, formula = 2.0 * cfg["b"] * 4096 * 773 * 2 * INNER, "2*(b*N)*D*(2*inner) (patch-bag K/V projection, N=4096, D=773, inner=512)"
        else:
            fl, formula = core_flops(mod, cfg["b"], prec)
        comp = forward_flops(cfg, prec)
        total_fl = sum(comp.values())
        entry = {
            "config": f"cfg{c}", "batch": cfg["b"], "core_precision": prec, "tensors": row["tensors"], "depth": cfg["depth"],
            "modalities": [m[0] for m in cfg["mods"]],
            "ms_per_forward_unprofiled": ms, "samples_per_s": row["samples_per_s"],
            "kernel_stats_csv": os.path.basename(csv_path),
            "kernel_time_per_forward_ms_from_csv": round(total_ns / forwards / 1e6, 4),
            "dominant_kernel": {"name": dom_name, "calls": int(dom["Calls"]), "avg_us": round(dom_us, 2),
                                "share_of_kernel_time": round(float(dom["TotalDurationNs"]) / total_ns, 4),
                                "executed_flops_per_launch": fl, "formula": formula,
                                "achieved_tflops": round(fl / (dom_us * 1e-6) / 1e12, 2)},
            "forward_executed_flops": {k: v for k, v in comp.items()},
            "forward_executed_flops_total": total_fl,
            "forward_executed_tflops": round(total_fl / (ms * 1e-3) / 1e12, 2),
        }
        dk = entry["dominant_kernel"]
        n_scores = sum(1.0 * L * m[1] * H * cfg["b"] for m in cfg["mods"] if m[0] in ("img", "vol")) * cfg["depth"]
        if prec == "bf16" and "attn_core_bf16" in dom_name:
            scores = 1.0 * L * mod[1] * H * cfg["b"]
            dk["scores_per_launch"] = scores
            dk["scores_per_s"] = scores / (dom_us * 1e-6)
            dk["bound"] = "valu (v_exp_f32)"
            dk["frac"] = round(dk["scores_per_s"] / PEAK_EXP, 4)
            dk["frac_of"] = "chip v_exp_f32 rate (one exponential per attention score)"
        elif "gemm_bf16_kernel" in dom_name:
            dk["bound"] = "mfma (bf16)"
            dk["frac"] = round(fl / (dom_us * 1e-6) / PEAK_BF16, 4)
            dk["frac_of"] = "dense bf16 MFMA peak 2.5 PF/s"
        else:
            dk["bound"] = "mfma (fp32)"
            dk["frac"] = round(fl / (dom_us * 1e-6) / PEAK_FP32, 4)
            dk["frac_of"] = "fp32 MFMA peak 157.3 TF/s"
        # the bf16 K/V projection of the patch bags (core_precision = bf16), whether or not it dominates: both of its roofs
        for k in kernels:
            if "gemm_bf16_kernel" in k["Name"]:
                us = float(k["AverageNs"]) / 1e3
                pfl = 2.0 * cfg["b"] * 4096 * 773 * 2 * INNER
                img = "<true>" in k["Name"]                               # bf16 K / V images (explicit bf16 core behind it) or fp32 K|V rows
                byts = cfg["b"] * 4096 * (2 * INNER * (2 if img else 4) + 832 * 2)       # K|V written + the bf16 context image read once
                entry_kv = {"name": k["Name"], "calls": int(k["Calls"]), "avg_us": round(us, 2), "executed_flops_per_launch": pfl,
                            "achieved_tflops": round(pfl / (us * 1e-6) / 1e12, 1), "frac_of_bf16_mfma_peak": round(pfl / (us * 1e-6) / PEAK_BF16, 4),
                            "algorithmic_bytes_per_launch": byts, "achieved_tb_s": round(byts / (us * 1e-6) / 1e12, 2),
                            "frac_of_hbm_8tb_s": round(byts / (us * 1e-6) / 8e12, 4),
                            "note": ("bf16 K / V images" if img else "fp32 K|V rows") + " (b*N x 1024) written + bf16 context image (b*N x 832) read per launch"}
                break
        else:
            entry_kv = None
        if entry_kv:
            entry["bf16_kv_projection_kernel"] = entry_kv
        if prec == "bf16":
            # two roofs in one forward: the bf16 cores are priced in exponentials (their bound), everything else in fp32 matrix
            # FLOPs; the end-to-end fraction is the sum of the two ideal times over the measured time
            proj_fl = comp.get("patch-bag K/V projection", 0.0)      # on bf16 MFMA under core_precision = bf16
            fp32_fl = sum(v for k, v in comp.items() if not k.startswith("attention cores (img") and not k.startswith("attention cores (vol")) - proj_fl
            ideal = fp32_fl / PEAK_FP32 + n_scores / PEAK_EXP + proj_fl / PEAK_BF16
            entry["forward_ideal_ms"] = {"fp32_matrix": round(fp32_fl / PEAK_FP32 * 1e3, 4), "bf16_core_exponentials": round(n_scores / PEAK_EXP * 1e3, 4),
                                         "bf16_kv_projection": round(proj_fl / PEAK_BF16 * 1e3, 4)}
            entry["forward_frac_executed"] = round(ideal / (ms * 1e-3), 4)
            entry["forward_frac_of"] = "(fp32 matrix FLOPs outside the bf16 cores and the bf16 K/V projection / fp32 MFMA peak + attention scores of the bf16 cores / chip v_exp_f32 rate + patch-bag K/V projection FLOPs / bf16 MFMA peak) / forward time"
            entry["forward_scores_frac_of_exp_rate"] = round(n_scores / (ms * 1e-3) / PEAK_EXP, 4)
        else:
            entry["forward_frac_executed"] = round(total_fl / (ms * 1e-3) / PEAK_FP32, 4)
            entry["forward_frac_of"] = "sum of executed matrix FLOPs of one forward / forward time / fp32 MFMA peak"
        if pmc:
            keep = ("mfma_pipe_busy", "valu_active_per_simd", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU",
                    "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE")
            entry["pmc"] = {n[:100]: {k: v[k] for k in keep if k in v} for n, v in pmc.items()
                            if n.startswith(("hn::", "void hn::")) and any(float(k["TotalDurationNs"]) / total_ns > 0.02 and k["Name"] == n for k in kernels)}
        # per-kernel table (share > 1 %)
        entry["kernels"] = [{"name": k["Name"][:110], "calls": int(k["Calls"]), "avg_us": round(float(k["AverageNs"]) / 1e3, 2),
                             "share": round(float(k["TotalDurationNs"]) / total_ns, 4)} for k in kernels
                            if float(k["TotalDurationNs"]) / total_ns > 0.01]
        doc["configs"].append(entry)
        print(json.dumps({"cfg": c, "prec": prec, "ms": ms, "dominant": dom_name[:50], "avg_us": dk["avg_us"], "frac": dk["frac"],
                          "forward_frac_executed": entry["forward_frac_executed"]}), flush=True)
        with open(os.path.join(out, args.tag + "_rooflines.json"), "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
