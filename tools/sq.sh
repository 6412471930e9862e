#!/bin/bash
# SQ / TCC counters of the frame's kernels at the current code state (GPU box).
#   bash tools/sq.sh <tag>       -> gpurun_out/profiles_out/<tag>_sq_counters.txt
# Three workloads: 640x480 / 5 mm single stream (k_integrate), 1280x720 / 2 mm (k_integrate) and the
# S = 4 group (k_integrate_g).  One rocprofv3 pass per counter set (8 SQ slots; FETCH_SIZE and
# WRITE_SIZE do not fit one pass).  Counter passes carry --kernel-trace only (no other trace domain).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
O=gpurun_out/profiles_out
mkdir -p $O
OUT=$O/${TAG}_sq_counters.txt
: > $OUT
SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32"
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
      "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS"
      "FETCH_SIZE"
      "WRITE_SIZE")
run() {  # name, program + args
  local name=$1; shift
  local i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    [ -n "$SQ_NSETS" ] && [ $i -gt $SQ_NSETS ] && break   # (SQ_NSETS=3: without the FETCH_SIZE / WRITE_SIZE passes)
    # A set that does not fit one pass makes rocprofv3 abort inside the program's hipInit (error 38, "Request exceeds
    # the capabilities of the hardware to collect") and the aborted process then lingers until the guard kills it
    # (round 4: four times five minutes).  Ask first -- rocprofv3-avail pmc-check needs no program -- and skip.
    if ! timeout -k 5 60 rocprofv3-avail pmc-check $set > gpurun_out/pmc_check.log 2>&1; then
      echo "$name pass $i SKIPPED (pmc-check: the set does not fit one pass): $set" >> $OUT; continue
    fi
    rm -rf gpurun_out/sq_${name}_$i
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/sq_${name}_$i -- "$@" > gpurun_out/sq_${name}_$i.log 2>&1 || { echo "pass $i of $name failed" >> $OUT; continue; }
    python3 - "$name" gpurun_out/sq_${name}_$i >> $OUT <<'PY'
import csv,glob,collections,sys,os
name,d=sys.argv[1],sys.argv[2]
f=sorted(glob.glob(d+'/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
    if not n.startswith('k_'): continue
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in sorted(agg.items()):
    # second half of the dispatches = steady-state frames
    print(name, n, " ".join(f"{k}={sum(v[len(v)//2:])/max(len(v[len(v)//2:]),1):.5g}" for k,v in sorted(c.items())), f"launches={len(next(iter(c.values())))}")
PY
  done
}
B="--steps 2 --warmup 1 --reps 1 --cpu-frames 0 --host-frames 0 --no-profile --no-secondary --streams 0"
run vga python3 bench.py $B
run hd2mm python3 bench.py --config hd2mm $B
run groupS4 python3 tools/streams_probe.py --only-group --streams 4 --steps 3 --frames 30
[ -n "$SQ_BIGMAP" ] && run bigmap python3 bench.py --config bigmap --cpu-frames 0 --no-profile
cat $OUT
