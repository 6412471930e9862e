#!/bin/bash
# SQ counters of k_raycast on the bench map (GPU box): bash tools/raycast_counters.sh  -> gpurun_out/raycast_counters.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/raycast_counters.txt
: > $OUT
SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM"
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  if ! timeout -k 5 60 rocprofv3-avail pmc-check $set > gpurun_out/pmc_check.log 2>&1; then echo "pass $i skipped: $set" >> $OUT; continue; fi
  rm -rf gpurun_out/rc_$i
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/rc_$i -- python3 tools/raycast_probe.py > gpurun_out/rc_$i.log 2>&1 || { echo "pass $i failed" >> $OUT; continue; }
  python3 - gpurun_out/rc_$i >> $OUT <<'PY'
import csv,glob,collections,sys,os
d=sys.argv[1]
f=sorted(glob.glob(d+'/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_raycast' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print(' '.join(f"{k}={sum(v)/len(v):.4g}" for k,v in sorted(agg.items())), f"launches={len(next(iter(agg.values()))) if agg else 0}")
PY
done
cat $OUT
