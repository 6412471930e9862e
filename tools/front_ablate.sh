#!/bin/bash
# usage (GPU box): tools/front_ablate.sh  -> k_front duration with one role switched off (diagnostic build)
# RATSDF_DEBUG: 3 = no visible-list role, 11 = no candidate-consume role, 12 = no pool-release role.
# (--cpu-frames 0: no parity check -- an ablated frame is wrong on purpose)
export RATSDF_LIB=$GRAFT_REPO_ROOT/ra-slam_amd/csrc/build/libratsdf_stamps.so
for cfg in "vga --steps 3 --warmup 1 --reps 1" "bigmap --config bigmap"; do
  set -- $cfg; name=$1; shift
  for dbg in 0 3 11 12; do
    bash tools/kd.sh ${name}_d$dbg "$*" RATSDF_DEBUG=$dbg 2>&1 | grep -E "^==|k_front"
  done
done
