#!/bin/bash
# usage: tools/kstats_cfg.sh "<bench args>" "ENV=..." ... -> median kernel durations under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="$1"; shift
i=0
for kv in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/kc_$i
  env $kv timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kc_$i -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --no-profile $args > gpurun_out/kc_$i.log 2>&1
  python3 - "$kv" gpurun_out/kc_$i <<'PY'
import csv,glob,statistics,collections,sys
kv,d=sys.argv[1],sys.argv[2]
f=sorted(glob.glob(d+'/*/*kernel_trace.csv'))[-1]
dur=collections.defaultdict(list); st=[]
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
    a,b=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    dur[n].append((b-a)/1e3)
    if n=='k_front': st.append(a)
st.sort(); gaps=[(st[i+1]-st[i])/1e3 for i in range(len(st)-1)]
print(kv, "| period", round(statistics.median(gaps),1), "|", " ".join(f"{n[2:]}={statistics.median(v):.1f}" for n,v in dur.items() if n.startswith('k_')))
PY
done
