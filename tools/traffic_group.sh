#!/bin/bash
# HBM traffic of k_integrate_g (S streams in one launch) from PMC counters, separate passes as the microarch guide
# prescribes.  Writes gpurun_out/profiles_out/traffic_group<S>.json (copy to profiles/).   usage: tools/traffic_group.sh [S]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${1:-4}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/trg_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/trg_$c -- python3 tools/group_probe.py $S > gpurun_out/trg_$c.log 2>&1
done
python3 - $S <<'PY'
import csv,glob,json,collections,os,sys
S=int(sys.argv[1]); out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=sorted(glob.glob(f'gpurun_out/trg_{c}/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('ratsdf::','').replace('void ','')
        if r['Counter_Name']==c: agg[n].append(float(r['Counter_Value']))
    for n,v in agg.items():
        v=v[len(v)//3:]
        out.setdefault(n,{})[c+"_KB_per_launch"]=sum(v)/len(v)
ki=[k for k in out if k.startswith('k_integrate_g')][0]
f_kb=out[ki]["FETCH_SIZE_KB_per_launch"]; w_kb=out[ki]["WRITE_SIZE_KB_per_launch"]
res={"kernel":ki,"streams":S,"fetch_size_kb":round(f_kb,1),"write_size_kb":round(w_kb,1),
     "k_integrate_bytes_per_launch": round((2*f_kb+w_kb)*1024),
     "note":"one launch = S frames; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B read request); WRITE_SIZE as is; tools/group_probe.py workload (S x 640x480 / 5 mm 'room' streams, 60-frame ping-pong), steady-state launches",
     "all_kernels_kb":{k:{kk:round(vv,1) for kk,vv in v.items()} for k,v in out.items() if k.startswith('k_')}}
os.makedirs('gpurun_out/profiles_out',exist_ok=True)
json.dump(res,open(f'gpurun_out/profiles_out/traffic_group{S}.json','w'),indent=1)
print(json.dumps(res))
PY
