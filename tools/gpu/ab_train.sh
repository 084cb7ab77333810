#!/bin/bash
# training iterations per second and per-kernel times of library variants: tools/gpu/ab_train.sh A D8 D16 (A = the shipped library,
# others nerf-projects_amd/variants/lib<name>.so)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = A ]; then L=$R/nerf-projects_amd/libnerf_mi355x.so; else L=$R/nerf-projects_amd/variants/lib$v.so; fi
  export NERF_MI355X_LIB=$L
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v$v -- python3 $R/bench_train.py --iters 20 --warmup 3 > $R/gpurun_out/v$v.log 2>&1 || exit 1
  echo "== $v (under the profiler): $(grep -o '"value": [0-9.]*' $R/gpurun_out/v$v.log)"
  head -9 $R/gpurun_out/v$v/*/*kernel_stats.csv | cut -d, -f1-4 | sed 's/"//g' | awk -F, '{printf "   %-60.60s %6s %12s %12s\n", $1, $(NF-2), $(NF-1), $NF}'
  timeout -k 10 200 python3 $R/bench_train.py --iters 60 2>/dev/null | tail -1 | cut -c1-100
done
