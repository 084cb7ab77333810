#!/bin/bash
# tools/gpu/train_stats.sh [ENV=..]: rocprofv3 kernel stats of bench_train.py (33 iterations), per-iteration table of every kernel
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
[ -n "$1" ] && export "$@"
rm -rf $R/gpurun_out/ts
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ts -- python3 $R/bench_train.py --iters 20 --warmup 3 > $R/gpurun_out/ts.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/ts/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
it = 33
tot = n = 0
for r in rows:
    per = int(r["Calls"]) / it
    if per >= 0.9:
        print(f"{per:6.2f} {float(r['TotalDurationNs'])/it/1e3:8.1f} us  {r['Name'][:70]}")
        tot += float(r['TotalDurationNs'])/it/1e3; n += per
print("launches per iteration %.1f, kernel time %.1f us" % (n, tot))
PY
grep -o '"value": [0-9.]*' $R/gpurun_out/ts.log
