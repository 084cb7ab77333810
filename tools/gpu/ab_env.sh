#!/bin/bash
# tools/gpu/ab_env.sh "VAR=value ..." ["VAR2=value ..."] ...: training it/s and per-kernel spans of the shipped library under each
# environment (quote a set; "-" = none), twice each
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for envset in "$@"; do
  for rep in 1 2; do
    ( [ "$envset" != "-" ] && export $envset; timeout -k 10 200 python3 $R/bench_train.py --iters 80 --only 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels',{})
print('[$envset]', 'it/s %.1f' % d['value'], ' '.join('%s %.3f' % (n[:12], k[n]['ms_per_iter']) for n in k), 'loss %.5f' % d['final_loss'])
" ) || exit 1
  done
done
