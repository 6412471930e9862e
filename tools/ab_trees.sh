#!/bin/bash
# usage: tools/ab_trees.sh N "<bench args>" dirA dirB ... -> interleaved bench values of several checked-out trees
R=$GRAFT_REPO_ROOT
n=$1; args="$2"; shift 2
for i in $(seq 1 $n); do
  for d in "$@"; do
    v=$(cd $R/$d && GRAFT_REPO_ROOT=$PWD python3 bench.py --no-profile --cpu-frames 0 --host-frames 0 --steps 30 $args 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
    echo "round $i $d $v"
  done
done
