#!/bin/bash
# usage: tools/pmc.sh "<counters>" [ENV=..]  -> per-kernel average of the counters (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_tmp
env $2 timeout -k 10 400 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d gpurun_out/pmc_tmp -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile > gpurun_out/pmc_tmp.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/pmc_tmp/*/*counter_collection.csv'))[-1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in agg.items():
    if not n.startswith('k_'): continue
    print(n, " ".join(f"{k}={sum(v[len(v)//2:])/max(1,len(v)-len(v)//2):.0f}" for k,v in sorted(c.items())))
PY
