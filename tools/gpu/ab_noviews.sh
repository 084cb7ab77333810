#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for e in "-" "NERF_TRAIN_NOVIEWS=f32"; do
  ( [ "$e" != "-" ] && export $e; timeout -k 10 200 python3 $R/bench_train.py --iters 60 --no-viewdirs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels',{})
print('[$e] no-viewdirs it/s %.1f' % d['value'], ' '.join('%s %.3f' % (n[:12], k[n]['ms_per_iter']) for n in k), 'loss %.5f' % d['final_loss'])" ) || exit 1
done
