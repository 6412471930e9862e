#!/bin/bash
# usage: tools/ks_group.sh S "ENV=..." ... -> per-kernel medians (rocprofv3) of the group run with S members
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=$1; shift
i=0
for kv in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/kg_$i
  env $kv timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kg_$i -- python3 tools/streams_probe.py --only-group --streams $S --steps 4 > gpurun_out/kg_$i.log 2>&1
  python3 - "$kv S=$S" gpurun_out/kg_$i <<'PY'
import csv,glob,statistics,collections,sys
kv,d=sys.argv[1],sys.argv[2]
f=sorted(glob.glob(d+'/*/*kernel_trace.csv'))[-1]
dur=collections.defaultdict(list); st=[]
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
    a,b=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    dur[n].append((b-a)/1e3)
    if n=='k_front_g': st.append(a)
st.sort(); gaps=[(st[i+1]-st[i])/1e3 for i in range(len(st)-1)]
print(kv, "| period", round(statistics.median(gaps),1), "|", " ".join(f"{n[2:]}={statistics.median(v):.1f}" for n,v in dur.items() if n.endswith('_g') or '_g<' in n))
PY
done
