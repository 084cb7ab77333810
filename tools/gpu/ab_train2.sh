#!/bin/bash
# tools/gpu/ab_train2.sh A NT BLK ...: training it/s (no profiler, 80 iterations, twice) and bench.py's per-kernel spans of library variants
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = A ]; then L=$R/nerf-projects_amd/libnerf_mi355x.so; else L=$R/nerf-projects_amd/variants/lib$v.so; fi
  export NERF_MI355X_LIB=$L
  for rep in 1 2; do
    timeout -k 10 200 python3 $R/bench_train.py --iters 80 --only 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels',{})
print('$v', 'it/s %.1f' % d['value'], ' '.join('%s %.3f' % (n[:12], k[n]['ms_per_iter']) for n in k))
" || exit 1
  done
done
