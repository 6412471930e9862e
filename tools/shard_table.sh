#!/bin/bash
# Per-shard kernel times of BASELINE config 4's split, measured one rank at a time on ONE GPU:
#   bash tools/shard_table.sh   -> gpurun_out/profiles_out/r04_shard_table_raw.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_out
mkdir -p $O
OUT=$O/r04_shard_table_raw.txt
: > $OUT
export RATSDF_GRAPH=0
for cfg in vga hd; do
  for N in 1 2 4 8; do
    for r in $(seq 0 $((N-1))); do
      rm -rf gpurun_out/sh_run
      timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sh_run -- python3 tools/shard_probe.py $cfg $N $r > gpurun_out/sh_run.log 2>&1
      python3 - gpurun_out/sh_run gpurun_out/sh_run.log >> $OUT <<'PY'
import csv,glob,statistics,collections,sys
d,log=sys.argv[1],sys.argv[2]
line=[l for l in open(log) if l.startswith("SHARD")]
fs=sorted(glob.glob(d+'/*/*kernel_trace.csv'))
if not line or not fs:
    print("FAILED", open(log).read()[-300:]); sys.exit(0)
dur=collections.defaultdict(list)
for r in csv.DictReader(open(fs[-1])):
    n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
    dur[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
pick=lambda p: next((statistics.median(v[len(v)//2:]) for n,v in dur.items() if n.startswith(p) and len(v)>50), float('nan'))
print(line[-1].strip(), f"k_front={pick('k_front'):.1f} k_integrate={pick('k_integrate'):.1f}")
PY
    done
  done
done
cat $OUT
