#!/bin/bash
# copies what profiles/measure.sh left under gpurun_out/measure into profiles/<round>_* and regenerates the PMC summaries
# usage: tools/collect_profiles.sh r03 [bench] [pmc] [train]   (default: every part present)
set -e
R=${1:-r03}; shift || true
M=gpurun_out/measure; P=profiles
parts=${*:-bench pmc train}
for part in $parts; do
  case $part in
  bench)
    for f in bench_n1 bench_n1_f32 bench_n1_rccl_group; do grep "^{" $M/$f.json > $P/${R}_$f.json; done
    grep "^{" $M/bench_c1.json > $P/${R}_bench_c1_lego400_coarse.json
    grep "^{" $M/bench_c4.json > $P/${R}_bench_c4_fern_ndc.json
    cp $M/stats/*/*kernel_stats.csv $P/${R}_bench_kernel_stats.csv
    tail -1 $M/shard_projection.log > $P/${R}_shard_projection.json ;;
  pmc)
    python $P/summarize_pmc.py $R $M > /dev/null ;;
  train)
    cp $M/train_stats/*/*kernel_stats.csv $P/${R}_train_kernel_stats.csv
    for f in bench_train bench_train_rowmajor bench_train_legacy_glue bench_train_noviewdirs bench_train_bwd_f32 bench_train_all_f32; do
      [ -f $M/$f.json ] && grep "^{" $M/$f.json > $P/${R}_$f.json; done
    python $P/summarize_train_pmc.py $P/${R}_train_pmc_summary $M > /dev/null ;;
  esac
done
python - "$R" <<'PY'
import json, sys
R = sys.argv[1]
def last(f): return json.loads(open(f).read().strip().splitlines()[-1])
d = last(f"profiles/{R}_bench_n1.json"); r = d["roofline"]
print("frame ms", round(d["ms_per_step"], 1), "samples/s %.3g" % d["value"], "frac", round(r["frac"], 4), "kernel ms", round(r["avg_launch_ms"], 3),
      "TFLOP/s", round(r["achieved"], 1), "executed", round(r["mfma_executed"]), "cpu %.3g" % d["cpu_baseline"]["value"])
t = d["train"]; print("train leg", round(t["value"], 1), {k: (round(v["ms_per_iter"], 3), round(v.get("frac_of_pipe_peak", 0), 3), v.get("hbm_gb_per_s")) for k, v in t["kernels"].items()})
for f in ("bench_n1_f32", "bench_c1_lego400_coarse", "bench_c4_fern_ndc", "bench_n1_rccl_group"):
    e = last(f"profiles/{R}_{f}.json"); print(f, round(e["ms_per_step"], 1), "%.3g" % e["value"], round(e["roofline"]["frac"], 4), round(e["roofline"]["avg_launch_ms"], 3), (e.get("other_precision") or {}).get("ms_per_step"))
import os
for f in ("bench_train", "bench_train_rowmajor", "bench_train_legacy_glue", "bench_train_noviewdirs", "bench_train_bwd_f32", "bench_train_all_f32"):
    if os.path.exists(f"profiles/{R}_{f}.json"):
        e = last(f"profiles/{R}_{f}.json"); print(f, round(e["value"], 1), round(e["ms_per_iter"], 3))
PY
