#!/bin/bash
# usage: tools/ab_env.sh N "<bench args>" "ENV=.." "ENV=.." ... -> interleaved bench values (no profile, no CPU legs)
cd $GRAFT_REPO_ROOT
n=$1; args="$2"; shift 2
for i in $(seq 1 $n); do
  for kv in "$@"; do
    v=$(env $kv python3 bench.py --no-profile --cpu-frames 0 --host-frames 0 --streams 0 --no-secondary --steps 30 $args 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
    echo "round $i [$kv] $v"
  done
done
