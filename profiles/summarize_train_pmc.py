#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench_train.py (or of a microbenchmark) per kernel.

Usage: python profiles/summarize_train_pmc.py <out stem> <dir with <prefix>fetch/ <prefix>write/ <prefix>sq/ <prefix>l2/> [--prefix train_pmc_]
       [--kernels name1,name2,...]
Each pass is its own run (`rocprofv3 --pmc ... --kernel-trace --output-format csv`), as the MI355X guide prescribes.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide (16 B per lane) streaming reads by 2x, so
both readings are given. Per kernel: mean launch time, clock held (GRBM_GUI_ACTIVE / 8 XCDs / time), MFMA busy as a
fraction of the cycles actually clocked and of 2.4 GHz, HBM bytes per launch and the rate they amount to.
"""
import glob
import json
import sys

import pandas as pd

KERNELS = ["nerf_mlp_h2_kernel", "nerf_mlp_bwd_h2_kernel", "grad_batch_pair_dma_kernel", "grad_batch_pair_kernel", "grad_batch_kernel",
           "row_exponents_kernel", "embed_train_kernel", "refresh_gather_kernel", "refresh_convert_kernel"]
PREFIX = "train_pmc_"


def load(root, name):
    files = glob.glob(f"{root}/{PREFIX}{name}/*/*_counter_collection.csv")
    if not files:
        return None
    df = pd.read_csv(files[0])
    df["dur_ns"] = df.End_Timestamp - df.Start_Timestamp
    return df


def main(stem, root):
    passes = {n: load(root, n) for n in ("fetch", "write", "sq", "l2")}
    out = {}
    for k in KERNELS:
        rec = {}
        per = {}
        for name, df in passes.items():
            if df is None:
                continue
            d = df[df.Kernel_Name.str.contains(k)]
            if d.empty:
                continue
            for c, v in d.groupby("Counter_Name").Counter_Value.mean().items():
                per[c] = float(v)
            rec[name + "_avg_launch_ms"] = float(d.drop_duplicates("Dispatch_Id").dur_ns.mean() / 1e6)
            rec[name + "_launches"] = int(d.Dispatch_Id.nunique())
        if not per:
            continue
        if "GRBM_GUI_ACTIVE" in per:
            d = passes["l2"][passes["l2"].Kernel_Name.str.contains(k)]
            g = d[d.Counter_Name == "GRBM_GUI_ACTIVE"]
            rec["effective_clock_ghz"] = float((g.Counter_Value / 8 / g.dur_ns).mean())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            dur = rec["sq_avg_launch_ms"] * 1e6
            rec["mfma_busy_frac_of_2.4GHz_x_1024_simd"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * 2.4 * 1024)
            if "effective_clock_ghz" in rec:
                rec["mfma_busy_frac_of_clocked_cycles"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * rec["effective_clock_ghz"] * 1024)
            if "SQ_WAIT_ANY" in per and "SQ_WAVE_CYCLES" in per:
                rec["wait_any_frac_of_wave_cycles"] = per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            f, w = per["FETCH_SIZE"] * 1024, per["WRITE_SIZE"] * 1024
            t = rec.get("fetch_avg_launch_ms", 0) * 1e-3
            rec["hbm_read_bytes_per_launch"], rec["hbm_read_bytes_per_launch_x2"], rec["hbm_write_bytes_per_launch"] = f, 2 * f, w
            if t:
                rec["hbm_gb_per_s_uncorrected"], rec["hbm_gb_per_s_fetch_x2"] = (f + w) / t / 1e9, (2 * f + w) / t / 1e9
        if "SQ_LDS_BANK_CONFLICT" in per and per.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_cycles_frac_of_lds_active"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
        if "TCC_HIT_sum" in per:
            rec["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
        rec["counters_per_launch_mean"] = per
        out[k] = rec
    json.dump(out, open(stem + ".json", "w"), indent=1)
    with open(stem + ".md", "w") as f:
        f.write(f"# {stem}: PMC summary per kernel (separate --pmc passes; means per launch)\n\n")
        for k, rec in out.items():
            f.write(f"## `{k}`\n\n| quantity | value |\n|---|---|\n")
            for q, v in rec.items():
                if q != "counters_per_launch_mean":
                    f.write(f"| {q} | {v:.6g} |\n" if isinstance(v, float) else f"| {q} | {v} |\n")
            f.write("\n")
    print(json.dumps({k: {q: v for q, v in r.items() if q != "counters_per_launch_mean"} for k, r in out.items()}, indent=1))


if __name__ == "__main__":
    if "--prefix" in sys.argv:
        i = sys.argv.index("--prefix")
        PREFIX = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    if "--kernels" in sys.argv:
        i = sys.argv.index("--kernels")
        KERNELS = sys.argv[i + 1].split(",")
        del sys.argv[i:i + 2]
    main(sys.argv[1], sys.argv[2])
