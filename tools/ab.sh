#!/bin/bash
# usage: tools/ab.sh "ENV=.. ENV2=.." ... -> bench value (no profile, no CPU legs) per setting, vga5mm and hd2mm
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  for cfg in vga5mm hd2mm; do
    env $kv timeout -k 10 300 python3 bench.py --config $cfg --cpu-frames 0 --host-frames 0 --steps 20 > gpurun_out/ab.log 2>&1
    python3 - "$kv" $cfg <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/ab.log').read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    print(sys.argv[1], sys.argv[2], "fps", d['value'], "k_integrate_us", r.get('avg_launch_us'), "frac", r.get('frac'))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e); print(open('gpurun_out/ab.log').read()[-800:])
PY
  done
done
