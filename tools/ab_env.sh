#!/bin/bash
# GPU box: interleaved A/B of environment settings on a bench config
# usage: tools/ab_env.sh "<bench args>" "ENV=a" "ENV=b" ...   (use X=1 for the default)
cd $GRAFT_REPO_ROOT
args="$1"; shift
for i in 1 2 3; do for kv in "$@"; do
  v=$(env $kv python3 bench.py --cpu-frames 0 --host-frames 0 --no-secondary --streams 0 $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], 'frames/s  k_integrate', r.get('avg_launch_us'), 'us')")
  echo "round $i [$kv] $v"
done; done
