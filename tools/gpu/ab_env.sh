#!/bin/bash
# training iterations per second and per-kernel times under environment switches: tools/gpu/ab_env.sh "" "NERF_TRAIN_DW_EXTRA=0" ...
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
k=0
for v in "$@"; do
  k=$((k+1))
  ( if [ -n "$v" ]; then export $v; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/e$k -- python3 $R/bench_train.py --iters 20 --warmup 3 > $R/gpurun_out/e$k.log 2>&1 || exit 1
  echo "== [$v] (under the profiler): $(grep -o '"value": [0-9.]*' $R/gpurun_out/e$k.log)"
  head -9 $R/gpurun_out/e$k/*/*kernel_stats.csv | cut -d, -f1-4 | sed 's/"//g' | awk -F, '{printf "   %-60.60s %6s %12s %12s\n", $1, $(NF-2), $(NF-1), $NF}'
  timeout -k 10 200 python3 $R/bench_train.py --iters 60 2>/dev/null | tail -1 | cut -c1-100 ) || exit 1
done
