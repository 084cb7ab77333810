#!/bin/bash
# A/B of kernel variants on one box: bash gpurun_bin/ab.sh "base pk acc ..." [rounds]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
variants="$1"; rounds=${2:-2}
for v in $variants; do
  NERF_MI355X_LIB=$R/gpurun_bin/lib_$v.so timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "f16x2 and not s16 and (mlp or bench_scale or stagewise or end_to_end)" > $O/test_$v.log 2>&1; echo "$v tests rc=$? $(tail -1 $O/test_$v.log)"
done
for r in $(seq $rounds); do for v in $variants; do
  NERF_MI355X_LIB=$R/gpurun_bin/lib_$v.so timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-precision --no-train > $O/bench_${v}_$r.json 2>$O/bench_${v}_$r.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_${v}_$r.json")); print("$v round $r: %.3f ms/launch  %.1f ms/frame frac %.4f"%(d["roofline"]["avg_launch_ms"], d["ms_per_step"], d["roofline"]["frac"]))
except Exception as e: print("$v round $r: FAILED", e)
PY
done; done
