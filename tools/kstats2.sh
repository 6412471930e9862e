#!/bin/bash
# usage: tools/kstats2.sh [-c config] "ENV=..." ... -> median kernel durations under rocprofv3 for each setting,
# single-stream leg only (no multi-stream / secondary / CPU legs: ablation switches break their parity asserts)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=vga5mm
if [ "$1" = "-c" ]; then CFG=$2; shift 2; fi
i=0
for kv in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/ks_$i
  env $kv timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ks_$i -- python3 bench.py --config $CFG --steps 3 --warmup 1 --reps 2 --cpu-frames 0 --host-frames 0 --streams 0 --no-secondary --no-profile > gpurun_out/ks_$i.log 2>&1
  python3 - "$kv" gpurun_out/ks_$i <<'PY'
import csv,glob,statistics,collections,sys
kv,d=sys.argv[1],sys.argv[2]
fs=sorted(glob.glob(d+'/*/*kernel_trace.csv'))
if not fs:
    print(kv, "FAILED"); print(open(d+'.log').read()[-600:]); sys.exit(0)
dur=collections.defaultdict(list); st=[]
for r in csv.DictReader(open(fs[-1])):
    n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
    a,b=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    dur[n].append((b-a)/1e3)
    if n.startswith('k_front'): st.append(a)
st.sort(); gaps=[(st[i+1]-st[i])/1e3 for i in range(len(st)-1)]
print(kv, "| period", round(statistics.median(gaps),1), "|", " ".join(f"{n[2:]}={statistics.median(v):.1f}" for n,v in dur.items() if n.startswith('k_') and len(v) > 50))
PY
done
