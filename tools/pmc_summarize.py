"""Reduces the rocprofv3 --pmc CSVs that tools/profile_round.sh collected (gpurun_out/prof/) to profiles/pmc_summary.json
and copies the per-kernel rows of the column kernels next to it.  HBM bytes per launch of a column kernel =
2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md: the counter reports half the bytes of wide coalesced reads)
+ WRITE_SIZE, both in KiB, each from its own pass, averaged over the launches of that kernel.  The summary records the
sha256 of the library the counters were collected on: bench.py reports them only for that library.

    python tools/pmc_summarize.py r02
"""
import csv, glob, json, os, sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
PRODUCTS = {"bf16": 1, "fp32": 3}
COLS = 32768                      # columns per batch of the column kernel (the library's COL_BATCH)
BATCHES = 512 * 512 // COLS       # batches per 512^3 sweep
QUERIES = COLS * 512              # queries per batch


def rows(sub, counter):
    out = []
    for f in glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "grid_mlp_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                out.append(r)
    # (round 5: a bf16 / fp16 batch is one launch of grid_mlp_kernel_v12 followed by one of grid_mlp_kernel_v10 in tile mode over the
    #  tiles v12 handed over - nearly always none: the figures are v12's)
    if any("grid_mlp_kernel_v12" in r["Kernel_Name"] for r in out):
        out = [r for r in out if "grid_mlp_kernel_v12" in r["Kernel_Name"]]
    return out


def mean(rs):
    v = [float(r["Counter_Value"]) for r in rs]
    return sum(v) / len(v) if v else None


def dur_ms(rs):
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rs]
    return sum(v) / len(v) if v else None


def keep(sub, counter, name):
    rs = rows(sub, counter)
    if rs:
        with open(os.path.join(ROOT, "profiles", name), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rs[0].keys()))
            w.writeheader()
            w.writerows(rs)
    return rs


def ksteps(prec):
    """(tile, MLP) pairs and residual layer-1 k-steps of one sweep = 16 launches, printed by tools/gpu_grid_once.py."""
    for sub in ("pmc_mfma_", "pmc_fetch_", "pmc_clk_"):
        f = os.path.join(SRC, sub + prec + ".log")
        if os.path.exists(f):
            for line in open(f):
                if line.startswith("done tile_mlps_per_sweep"):
                    w = line.split()
                    return float(w[2]), float(w[4])
    return None, None


def one(prec):
    fetch = keep("pmc_fetch_" + prec, "FETCH_SIZE", "%s_pmc_fetch_size_%s.csv" % (TAG, prec))
    write = keep("pmc_write_" + prec, "WRITE_SIZE", "%s_pmc_write_size_%s.csv" % (TAG, prec))
    hit, miss = rows("pmc_l2_" + prec, "TCC_HIT_sum"), rows("pmc_l2_" + prec, "TCC_MISS_sum")
    gui = rows("pmc_clk_" + prec, "GRBM_GUI_ACTIVE")
    mfma = keep("pmc_mfma_" + prec, "SQ_VALU_MFMA_BUSY_CYCLES", "%s_pmc_mfma_busy_%s.csv" % (TAG, prec))
    # the fp32-grade kernel runs a batch in two passes (lr, then hr): the figures below are per BATCH = the sum of its launches
    # (tools/gpu_grid_once.py: 3 sweeps x 16 batches)
    passes = max(1, round(len(fetch) / (3.0 * BATCHES))) if fetch else 1
    mean_b = lambda rs: (mean(rs) * passes) if rs else None
    dur_b = lambda rs: (dur_ms(rs) * passes) if rs else None
    f_kb, w_kb = mean_b(fetch), mean_b(write)
    n = PRODUCTS[prec]
    weights = 2 * 42 * 32768 * (2 if prec == "fp32" else 1)     # the packed weight stream once (fp32: hi + lo parts)
    tiles, ks = ksteps(prec)
    # MFMAs one wave issues per (tile, MLP): layer 1 = the affine k-step + the residual k-steps, layers 2 and 3 dense.
    # bf16 (v7, 128-voxel tile): 16 per k-step, 256 + 64; fp32-grade (v8, 64-voxel tile): 8 affine, 24 per k-step, 384 + 96
    if tiles:
        per_wave = (16 * tiles + 16 * ks + 320 * tiles) if prec != "fp32" else (8 * tiles + 24 * ks + 480 * tiles)
        expected = per_wave * 4 * 32 / float(BATCHES)      # 4 waves, 32 cycles per MFMA, per batch
    else:
        expected = QUERIES * 2752512 * n / 32768 * 32   # dense layer 1 (v3 / v5)
    return {
        "kernel": fetch[0]["Kernel_Name"] if fetch else None,
        "launch": "%d columns x 512 voxels = %d queries (R=512), tools/gpu_grid_once.py 512 %s" % (COLS, QUERIES, prec),
        "fetch_size_kb_raw_per_launch": f_kb,
        "write_size_kb_per_launch": w_kb,
        "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0 if f_kb is not None and w_kb is not None else None,
        # the same per 16 384 columns = 8 388 608 queries, the batch of rounds 1 - 3a (what VERDICT r2 item 6 measured: 19.3 GB for v8)
        "hbm_bytes_per_16384_columns": (2.0 * f_kb + w_kb) * 1024.0 * 16384.0 / COLS if f_kb is not None and w_kb is not None else None,
        "l2_hit_rate": (mean(hit) / (mean(hit) + mean(miss))) if hit and miss else None,
        "scratch_bytes_per_lane": int(fetch[0]["Scratch_Size"]) if fetch else None,
        "launches_per_batch": passes,
        "queries_per_launch": QUERIES,
        "effective_clock_ghz": (mean(gui) / 8.0 / (dur_ms(gui) * 1e-3) * 1e-9) if gui else None,
        "avg_launch_ms_under_pmc": dur_b(fetch),
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles the matrix pipes were busy, summed over the chip's 1024 SIMDs (32 per
        # v_mfma_f32_32x32x16_*); the denominator is the launch's shader cycles (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs
        "mfma_busy_cycles_per_launch": mean_b(mfma) if mfma else None,
        "mfma_busy_fraction": (mean(mfma) / (mean(gui) / 8.0 * 1024.0)) if mfma and gui else None,
        "mfma_busy_cycles_expected": expected,   # MFMAs issued x 32 cycles
        "layer1_residual_ksteps_per_tile_mlp": (ks / tiles) if tiles else None,
        # compulsory bytes of one launch: the two output fields (2 x 4 B per voxel) and the weight stream once; the
        # per-column constants (CC_PAD floats per column) and masks are the sweep's own intermediate, listed separately
        "algorithmic_bytes_per_launch": QUERIES * 8 + weights,
        "column_constant_bytes_per_launch": COLS * 2944 * 4 + COLS * 4,
        # restated kernels: the per-column vectors R (5 x 512 fp32), read by the column kernel, which builds the affine A fragments
        # itself (until round 3 a kernel of its own wrote them: 32 KiB per column = 1.07 GB per launch, read back once)
        "r_vector_bytes_per_launch": COLS * 5 * 512 * 4 if tiles else 0,
    }


def sq(prec):
    """Averages per launch of every SQ counter collected for the column kernel of `prec` (tools/profile_round.sh)."""
    out = {}
    for sub in ("pmc_sq1_" + prec, "pmc_sq2_" + prec):
        acc = {}
        for f in glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                # (bf16: the launches of grid_mlp_kernel_v12; the tile-mode launches of v10 behind them are left out)
                if "grid_mlp_kernel" in r["Kernel_Name"] and not (prec != "fp32" and "grid_mlp_kernel_v10" in r["Kernel_Name"]):
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    out["kernel"] = r["Kernel_Name"]
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
    if "SQ_WAVE_CYCLES" in out:
        w = out["SQ_WAVE_CYCLES"]
        out["fraction_of_wave_cycles"] = {k: out[k] / w for k in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                                                   "SQ_WAIT_INST_LDS") if k in out}
    out["note"] = ("rocprofv3 --pmc over tools/gpu_grid_once.py 512 %s, averages per launch of 32768 columns x 512 voxels; SQ_* cycle "
                   "counters are in units of 4 clocks, summed over the resident waves" % prec)
    return out


for prec in ("bf16", "fp32"):
    d = sq(prec)
    if len(d) > 1:
        json.dump(d, open(os.path.join(ROOT, "profiles", "%s_sq_counters_%s.json" % (TAG, prec)), "w"), indent=1)

s = {"lib_sha256": open(os.path.join(SRC, "lib_sha256.txt")).read().split()[0] if os.path.exists(os.path.join(SRC, "lib_sha256.txt")) else None,
     "round": TAG, "kernels": {p: one(p) for p in ("bf16", "fp32")},
     "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); separate --pmc passes; GRBM_GUI_ACTIVE is summed over the 8 XCDs"}
json.dump(s, open(os.path.join(ROOT, "profiles", "pmc_summary.json"), "w"), indent=1)
print(json.dumps(s, indent=1))
