#!/bin/bash
# GPU box: interleaved A/B of variant libraries on the bench configs
# usage: tools/ab_bench.sh "<bench args>" lib_a.so lib_b.so ...   (paths relative to ra-slam_amd/csrc/build)
cd $GRAFT_REPO_ROOT
args="$1"; shift
for i in 1 2 3; do for lib in "$@"; do
  v=$(RATSDF_LIB=$GRAFT_REPO_ROOT/ra-slam_amd/csrc/build/$lib python3 bench.py --cpu-frames 0 --host-frames 0 --no-secondary --streams 0 $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; f=d.get('flythrough') or {}; print(d['value'], 'frames/s  k_integrate', r.get('avg_launch_us'), 'us', (f.get('k_integrate_us') or ''))")
  echo "round $i [$lib] $v"
done; done
