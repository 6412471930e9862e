# k_front per-launch counters, round-4 visible-list role (build variant libratsdf_oldvis.so = commit d937e3e, tools/build_variant.sh) vs the current build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=$GRAFT_REPO_ROOT/ra-slam_amd/csrc/build
OUT=gpurun_out/r5_front_counters.log
: > $OUT
for set in "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_TAG_STALL_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  if ! timeout -k 5 60 rocprofv3-avail pmc-check $set > gpurun_out/pmc_check.log 2>&1; then echo "SKIPPED (pmc-check): $set" >> $OUT; continue; fi
  for lib in libratsdf_oldvis.so libratsdf.so; do
    rm -rf gpurun_out/fc_tmp
    RATSDF_LIB=$B/$lib RATSDF_LIB_VARIANT=1 timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/fc_tmp -- python3 bench.py --steps 2 --warmup 1 --reps 1 --cpu-frames 0 --host-frames 0 --no-profile --no-secondary --streams 0 > gpurun_out/fc_tmp.log 2>&1 || { echo "$lib FAILED: $set" >> $OUT; continue; }
    python3 - "$lib" >> $OUT <<'PY'
import csv,glob,collections,sys,os
lib=sys.argv[1]
f=sorted(glob.glob('gpurun_out/fc_tmp/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
    if n.startswith('k_front'): agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in sorted(agg.items()):
    print(lib, n, " ".join(f"{k}={sum(v[len(v)//2:])/max(len(v[len(v)//2:]),1):.5g}" for k,v in sorted(c.items())), f"launches={len(next(iter(c.values())))}")
PY
  done
done
cat $OUT
