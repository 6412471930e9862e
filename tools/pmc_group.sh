#!/bin/bash
# usage: tools/pmc_group.sh S "COUNTERS..." ["COUNTERS..." ...] -> per-kernel mean of each counter for the group run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=$1; shift
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pg_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pg_$i -- python3 tools/streams_probe.py --only-group --streams $S --steps 3 --frames 30 > gpurun_out/pg_$i.log 2>&1
  python3 - "$set" gpurun_out/pg_$i <<'PY'
import csv,glob,collections,sys,os
d=sys.argv[2]
f=sorted(glob.glob(d+'/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
    if not n.startswith('k_'): continue
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in agg.items():
    print(n, " ".join(f"{k}={sum(v[len(v)//2:])/max(len(v[len(v)//2:]),1):.4g}" for k,v in c.items()))
PY
done
