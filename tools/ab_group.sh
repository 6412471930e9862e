#!/bin/bash
# interleaved A/B of the S=4 group path (frames/s, k_integrate_g us): tools/ab_group.sh "ENV=.. " "ENV=.." ...
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for kv in "$@"; do
  v=$(env $kv python3 tools/streams_probe.py --only-group --streams ${S:-4} --steps 20 2>/dev/null | grep '"group"' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['frames_per_s'], 'k_integrate_g', d['k_integrate_us'])")
  echo "round $i [$kv] $v"
done; done
