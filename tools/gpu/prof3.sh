#!/bin/bash
# per-kernel times of the training iteration for library variants (A = shipped)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in A B C; do
  if [ $v = A ]; then L=$R/nerf-projects_amd/libnerf_mi355x.so; else L=$R/nerf-projects_amd/variants/lib$v.so; fi
  export NERF_MI355X_LIB=$L
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v$v -- python3 $R/bench_train.py --iters 20 --warmup 3 > $R/gpurun_out/v$v.log 2>&1 || exit 1
  grep -h "train_iterations" $R/gpurun_out/v$v.log | cut -c1-110
done
