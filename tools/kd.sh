#!/bin/bash
# usage: tools/kd.sh <name> "<bench args>" [ENV=..]  -> kernel duration distribution under rocprofv3 (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; args="$2"; shift; shift
rm -rf gpurun_out/kd_$name
env "$@" timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kd_$name -- python3 bench.py --cpu-frames 0 --host-frames 0 --no-profile --no-secondary --streams 0 $args > gpurun_out/kd_$name.log 2>&1
echo "== $name ($args $*): $(tail -1 gpurun_out/kd_$name.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "frames/s")' 2>/dev/null)"
python3 tools/kdist.py gpurun_out/kd_$name 8
