#!/bin/bash
# bench_train.py --no-viewdirs under the given environment switches ("-" = as shipped), e.g.
#   bash tools/gpu/ab_noviews.sh - NERF_TRAIN_BLOCKED=0 NERF_TRAIN_NOVIEWS=f32
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
[ $# -eq 0 ] && set -- - NERF_TRAIN_NOVIEWS=f32
for e in "$@"; do
  ( [ "$e" != "-" ] && export $e; timeout -k 10 200 python3 $R/bench_train.py --iters 60 --no-viewdirs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels',{})
print('[$e] no-viewdirs it/s %.1f' % d['value'], ' '.join('%s %.3f' % (n[:12], k[n]['ms_per_iter']) for n in k), 'loss %.5f' % d['final_loss'])" ) || exit 1
done
