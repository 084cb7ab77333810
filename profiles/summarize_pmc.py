#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into profiles/<tag>_pmc_summary.{json,md}.

Usage: python profiles/summarize_pmc.py <tag> <dir with pmc_fetch/ pmc_write/ pmc_sq/ pmc_l2/> [--kernel NAME]
(<tag> like r01 for the default fp16-pair kernel, r01_f32 with --kernel nerf_mlp_kernel for the fp32 one)
Each pass is its own run (`rocprofv3 --pmc ... --kernel-trace --output-format csv`), as the MI355X
guide prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass). FETCH_SIZE/WRITE_SIZE are in KiB.
On gfx950 FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x; the kernel's dominant read
stream is 4 B/lane (z_vals), which the guide calls uncalibrated, so both figures are given.
"""
import glob
import json
import sys

import pandas as pd

KERNEL = "nerf_mlp_h2_kernel"   # pass --kernel nerf_mlp_kernel for the fp32 kernel
SUFFIX = ""                     # --suffix _s16: read pmc_sq_s16/, pmc_l2_s16/ ... (passes of another arithmetic mode)


def load(root, name):
    f = glob.glob(f"{root}/{name}/*/*_counter_collection.csv")[0]
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains(KERNEL)].copy()
    df["dur_ns"] = df.End_Timestamp - df.Start_Timestamp
    return df


def main(tag, root):
    out = {}
    per = {}
    have = [n for n in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_l2") if glob.glob(f"{root}/{n}{SUFFIX}/*/*_counter_collection.csv")]
    for name in have:
        df = load(root, name + SUFFIX)
        for c, v in df.groupby("Counter_Name").Counter_Value.mean().items():
            per[c] = float(v)
        out[name + "_launches"] = int(df.Dispatch_Id.nunique())
        out[name + "_avg_launch_ms"] = float(df.drop_duplicates("Dispatch_Id").dur_ns.mean() / 1e6)
    out["counters_per_launch_mean"] = per
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        fetch, write = per["FETCH_SIZE"] * 1024, per["WRITE_SIZE"] * 1024
        out["hbm_bytes_per_launch_uncorrected"] = fetch + write
        out["hbm_bytes_per_launch_fetch_x2"] = 2 * fetch + write
    out["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
    l2 = load(root, "pmc_l2" + SUFFIX)
    g = l2[l2.Counter_Name == "GRBM_GUI_ACTIVE"]
    out["effective_clock_ghz"] = float((g.Counter_Value / 8 / g.dur_ns).mean())
    sq = load(root, "pmc_sq" + SUFFIX)
    dur = sq.drop_duplicates("Dispatch_Id").dur_ns.mean()
    out["mfma_busy_frac_of_2.4GHz_x_1024_simd"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * 2.4 * 1024)
    out["wait_any_frac_of_wave_cycles"] = per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"]
    out["lds_bank_conflict_cycles"] = per["SQ_LDS_BANK_CONFLICT"]
    rnd, _, suffix = tag.partition("_")
    stem = f"profiles/{rnd}_pmc_summary" + (f"_{suffix}" if suffix else "")
    json.dump(out, open(stem + ".json", "w"), indent=1)
    with open(stem + ".md", "w") as f:
        f.write(f"# {tag}: PMC summary for `{KERNEL}` (bench.py --steps 1 --warmup 0 --no-cpu-baseline)\n\n")
        f.write("Mean per launch over the 40 launches of one 800x800 64+128 frame (20 coarse + 20 fine).\n\n")
        f.write("| quantity | value |\n|---|---|\n")
        for k, v in out.items():
            if k != "counters_per_launch_mean":
                f.write(f"| {k} | {v:.6g} |\n" if isinstance(v, float) else f"| {k} | {v} |\n")
        f.write("\n| counter | mean per launch |\n|---|---|\n")
        for k, v in per.items():
            f.write(f"| {k} | {v:.6g} |\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if "--kernel" in sys.argv:
        i = sys.argv.index("--kernel")
        KERNEL = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    if "--suffix" in sys.argv:
        i = sys.argv.index("--suffix")
        SUFFIX = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    main(sys.argv[1], sys.argv[2])
