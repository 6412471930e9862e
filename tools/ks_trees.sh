#!/bin/bash
# usage: tools/ks_trees.sh "<bench args>" dirA dirB ... -> per-kernel medians (rocprofv3) for several checked-out trees
R=$GRAFT_REPO_ROOT
args="$1"; shift
for d in "$@"; do
  (cd $R/$d && mkdir -p gpurun_out && GRAFT_REPO_ROOT=$PWD bash tools/kstats_cfg.sh "$args" "T=$d")
done
