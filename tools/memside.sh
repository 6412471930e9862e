#!/bin/bash
# Memory-side counters of the frame's kernels (GPU box): vector L1 (TCP), its address / data units (TA, TD), L2 (TCC),
# address translation (UTCL1) and the workgroup dispatcher's resource stalls (SPI), for the questions of VERDICT r3
# item 4: why a 5 mm block costs more than a 2 mm block in the same kernel, and why k_integrate_g (4 streams) takes
# more cycles than the 1280x720 launch with the same waves, instructions and bytes.
#   bash tools/memside.sh <tag>       -> gpurun_out/profiles_out/<tag>_memside_counters_raw.txt
# One rocprofv3 pass per counter set, --kernel-trace only beside --pmc (no other trace domain); the program
# itself after `--`.  A pass whose counters do not fit one pass is reported and skipped.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
O=gpurun_out/profiles_out
mkdir -p $O
OUT=$O/${TAG}_memside_counters_raw.txt
[ -n "$MEMSIDE_APPEND" ] || : > $OUT
SETS=("GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum"
      "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
      "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"
      "GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"
      "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum"
      "GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
      "GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"
      "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum"
      "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
      "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_BUSY_avr"
      "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_LEVEL_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum"
      "GRBM_GUI_ACTIVE SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN"
      "GRBM_GUI_ACTIVE SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_BUSY SPI_CSN_WAVE SPI_CSN_NUM_THREADGROUPS"
      "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU")
run() {  # name, program + args
  local name=$1; shift
  local i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    # A set that does not fit one pass makes rocprofv3 abort inside the program's hipInit (error 38, "Request exceeds
    # the capabilities of the hardware to collect") and the aborted process then lingers until the guard kills it
    # (round 4: four times five minutes).  Ask first -- rocprofv3-avail pmc-check needs no program -- and skip.
    if ! timeout -k 5 60 rocprofv3-avail pmc-check $set > gpurun_out/pmc_check.log 2>&1; then
      echo "$name pass $i SKIPPED (pmc-check: the set does not fit one pass): $set" >> $OUT; continue
    fi
    rm -rf gpurun_out/ms_${name}_$i
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/ms_${name}_$i -- "$@" > gpurun_out/ms_${name}_$i.log 2>&1 || { echo "$name pass $i FAILED: $set" >> $OUT; continue; }
    python3 - "$name" gpurun_out/ms_${name}_$i >> $OUT <<'PY'
import csv,glob,collections,sys,os
name,d=sys.argv[1],sys.argv[2]
fs=sorted(glob.glob(d+'/*/*counter_collection.csv'), key=os.path.getmtime)
if not fs:
    print(name, "no counter file"); sys.exit(0)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[-1])):
    n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
    if not n.startswith('k_'): continue
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in sorted(agg.items()):
    if not (n.startswith('k_integrate') or n.startswith('k_front') or n.startswith('k_cand')): continue
    # second half of the dispatches = steady-state frames
    print(name, n, " ".join(f"{k}={sum(v[len(v)//2:])/max(len(v[len(v)//2:]),1):.5g}" for k,v in sorted(c.items())), f"launches={len(next(iter(c.values())))}")
PY
  done
}
B="--steps 2 --warmup 1 --reps 1 --cpu-frames 0 --host-frames 0 --no-profile --no-secondary --streams 0"
export RATSDF_GRAPH=0   # (the frame-by-frame launches: k_integrate<2> with by-value operands, the kernel the roofline is about)
# workloads: all by default, or the ones named after the tag
W=${@:2}
[ -z "$W" ] && W="vga5mm vga2mm hd2mm groupS4 pure5mm pure2mm"
for w in $W; do
  case $w in
    vga5mm) run vga5mm python3 bench.py $B ;;
    vga2mm) run vga2mm python3 bench.py --voxel 0.002 --frames-per-step 40 $B ;;
    hd2mm) run hd2mm python3 bench.py --config hd2mm $B ;;
    groupS4) run groupS4 python3 tools/group_probe.py 4 ;;
    # the PURE voxel update: serial role at the tail of k_front, the whole look-ahead pass in k_front
    pure5mm) RATSDF_FRONT_TAIL=1 RATSDF_CAND_SPLIT=100,0 run pure5mm python3 bench.py $B ;;
    pure2mm) RATSDF_FRONT_TAIL=1 RATSDF_CAND_SPLIT=100,0 run pure2mm python3 bench.py --voxel 0.002 --frames-per-step 40 $B ;;
  esac
done
cat $OUT
