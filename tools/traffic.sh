#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters (separate passes, as the microarch guide
# prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Writes profiles/traffic_latest.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="$1"            # extra bench.py arguments, e.g. "--config hd2mm"
OUT="${2:-traffic_latest.json}"
export RATSDF_GRAPH=0
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/traffic_$c -- python3 bench.py --steps 2 --warmup 1 --reps 1 --cpu-frames 0 --host-frames 0 --no-profile $ARGS > gpurun_out/traffic_$c.log 2>&1
done
python3 - "$OUT" "$ARGS" <<'PY'
import csv,glob,json,collections,os,sys
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=sorted(glob.glob(f'gpurun_out/traffic_{c}/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
        if r['Counter_Name']==c: agg[n].append(float(r['Counter_Value']))
    for n,v in agg.items():
        v=v[len(v)//3:]   # skip the map-building frames
        out.setdefault(n,{})[c+"_KB_per_launch"]=sum(v)/len(v)
ki=[k for k in out if k.startswith('k_integrate')][0]
f_kb=out[ki]["FETCH_SIZE_KB_per_launch"]; w_kb=out[ki]["WRITE_SIZE_KB_per_launch"]
res={"kernel":ki,"fetch_size_kb":round(f_kb,1),"write_size_kb":round(w_kb,1),
     "k_integrate_bytes_per_launch": round((2*f_kb+w_kb)*1024),
     "raw_bytes_per_launch_uncorrected": round((f_kb+w_kb)*1024),
     "note":"FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B read request); WRITE_SIZE as is; bench workload, steady-state frames",
     "code_state": os.environ.get("RATSDF_CODE_STATE", "round 4: 2 launches per frame (frame-by-frame launches, RATSDF_GRAPH=0, so that the kernel is k_integrate<2>), serial role inside k_integrate"),
     "bench_args": "--steps 2 --warmup 1 --reps 1 --cpu-frames 0 --host-frames 0 --no-profile " + (sys.argv[2] if len(sys.argv) > 2 else ""),
     "all_kernels_kb":{k:{kk:round(vv,1) for kk,vv in v.items()} for k,v in out.items()}}
os.makedirs('gpurun_out/profiles_out',exist_ok=True)
json.dump(res,open('gpurun_out/profiles_out/'+sys.argv[1],'w'),indent=1)
print(json.dumps(res))
PY
