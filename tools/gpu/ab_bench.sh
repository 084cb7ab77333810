#!/bin/bash
# same-box A/B of library variants on the headline frame: ms per frame and the dominant kernel's launch time, both arithmetics
R=$GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  if [ $v = A ]; then L=$R/nerf-projects_amd/libnerf_mi355x.so; else L=$R/nerf-projects_amd/variants/lib$v.so; fi
  NERF_MI355X_LIB=$L timeout -k 10 300 python $R/bench.py --no-cpu-baseline --no-train --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'],1), round(d['roofline']['avg_launch_ms'],3), round(d['roofline']['frac'],4), '| f32', round(d['other_precision']['ms_per_step'],1), round(d['other_precision']['roofline']['frac'],4))" || exit 1
done
done
