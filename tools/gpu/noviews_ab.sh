#!/bin/bash
R=$GRAFT_REPO_ROOT
for e in "-" "NERF_TRAIN_BWD=f32" "NERF_TRAIN_DW=f32" "NERF_TRAIN_GLUE=legacy" "NERF_TRAIN_NOVIEWS=f32"; do
  ( [ "$e" != "-" ] && export $e; echo "== $e"; python3 $R/tools/gpu/noviews_diag.py 2>&1 | grep "f.pts_linears.0\|f.output_linear\|f.pts_linears.1.w\|losses" )
done
