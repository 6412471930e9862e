#!/bin/bash
# usage: tools/sweep_cfg.sh "<bench args>" "VAR=v1" ... -> one short bench line per setting (GPU box)
args="$1"; shift
for kv in "$@"; do
  out=$(env $kv timeout -k 10 300 python bench.py --steps 4 --warmup 2 --cpu-frames 0 --host-frames 0 $args 2>/dev/null | tail -1)
  python3 - "$kv" "$out" <<'PY'
import json,sys
kv,out=sys.argv[1],sys.argv[2]
try:
    j=json.loads(out); r=j.get("roofline") or {}
    print(f"{kv:40s} fps={j['value']:9.1f} ms/step={j['ms_per_step']:8.3f} k_integrate_us={r.get('avg_launch_us')} frac={r.get('frac')}")
except Exception as e:
    print(kv, "FAILED", out[-200:])
PY
done
