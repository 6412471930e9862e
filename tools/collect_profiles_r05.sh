#!/bin/bash
# Round-5 collection (GPU box): everything tools/collect_profiles.sh gathers, plus the S = 4 group's own kernel
# statistics (VERDICT r4 missing #4), the SQ pass behind roofline.issue (item 3a), the config-4 rehearsal line
# (item 1), the zero-update-wave count (item 5a), the host-link probe and the page-locked path's copy timeline
# (item 2), and the counter-set check (item 7).  -> gpurun_out/profiles_out/r05_*
set -o pipefail
TAG=r05
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_out
mkdir -p $O
stats() {  # name, program + args
  local name=$1; shift
  rm -rf gpurun_out/kstats_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_$name -- "$@" > gpurun_out/kstats_$name.log 2>&1 &&
  cp "$(ls -t gpurun_out/kstats_$name/*/*kernel_stats.csv | head -1)" $O/${TAG}_kernel_stats_$name.csv &&
  { grep -E '^\{|^S [0-9]' gpurun_out/kstats_$name.log | tail -1 > $O/${TAG}_bench_line_under_rocprof_$name.json; }
  echo "stats $name rc=$?"
}
B="--cpu-frames 0 --host-frames 0 --no-secondary"
stats vga5mm python3 bench.py $B --steps 5 --warmup 2 --streams 0
stats hd2mm python3 bench.py $B --config hd2mm --steps 5 --warmup 2 --streams 0
stats bigmap python3 bench.py $B --config bigmap
stats group4 python3 tools/group_probe.py 4
bash tools/traffic.sh "--streams 0 --no-secondary" traffic_latest.json > gpurun_out/traffic.log 2>&1
bash tools/traffic.sh "--config hd2mm --streams 0" traffic_hd2mm.json > gpurun_out/traffic_hd.log 2>&1
bash tools/traffic.sh "--config bigmap" traffic_bigmap.json > gpurun_out/traffic_big.log 2>&1
bash tools/traffic_group.sh 4 > gpurun_out/traffic_grp.log 2>&1
echo "traffic done"
SQ_NSETS=3 SQ_BIGMAP=1 bash tools/sq.sh $TAG > gpurun_out/sq_r05.log 2>&1; echo "sq rc=$?"
timeout -k 10 800 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err && tail -1 gpurun_out/bench_default.log > $O/${TAG}_bench_line.json; echo "bench rc=$?"
timeout -k 10 500 python bench.py --config hd2mm --host-frames 0 --streams 0 > gpurun_out/bench_hd2mm.log 2>&1 && tail -1 gpurun_out/bench_hd2mm.log > $O/${TAG}_bench_line_hd2mm.json
timeout -k 10 500 python bench.py --config bigmap > gpurun_out/bench_bigmap.log 2>&1 && tail -1 gpurun_out/bench_bigmap.log > $O/${TAG}_bench_line_bigmap.json
timeout -k 10 500 python bench.py --config flythrough --host-frames 0 --streams 0 > gpurun_out/bench_fly.log 2>&1 && tail -1 gpurun_out/bench_fly.log > $O/${TAG}_bench_line_flythrough_hd2mm.json
echo "bench lines done"
# BASELINE config 4 through the plain entry: 2 subvolume ranks on the ONE device, gloo instead of RCCL (a rehearsal
# of the data path -- frame broadcast inside the timed region --, not a scaling measurement)
RATSDF_BENCH_DEVICE=0 RATSDF_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --shard --config hd2mm --steps 3 --warmup 1 --reps 3 > gpurun_out/bench_shard2.log 2>&1 && grep '^{' gpurun_out/bench_shard2.log | tail -1 > $O/${TAG}_bench_line_shard2_rehearsal_hd2mm.json
RATSDF_BENCH_DEVICE=0 RATSDF_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --shard --steps 3 --warmup 1 --reps 3 > gpurun_out/bench_shard2v.log 2>&1 && grep '^{' gpurun_out/bench_shard2v.log | tail -1 > $O/${TAG}_bench_line_shard2_rehearsal_vga5mm.json
echo "shard rehearsal rc=$?"
timeout -k 10 300 python tools/zero_waves.py > $O/${TAG}_zero_update_waves.txt 2> gpurun_out/zero_waves.err
timeout -k 5 60 tools/probes/h2d_probe > $O/${TAG}_h2d_probe.txt 2>&1
rm -rf gpurun_out/r5_copytrace
timeout -k 10 200 rocprofv3 --memory-copy-trace --output-format csv -d gpurun_out/r5_copytrace -- python3 tools/pinned_probe.py 768 > gpurun_out/r5_copytrace.log 2>&1
{ grep "pinned path" gpurun_out/r5_copytrace.log; python3 tools/copy_gaps.py "$(ls gpurun_out/r5_copytrace/*/*memory_copy_trace.csv | head -1)" 19660800; } > $O/${TAG}_pinned_path_copy_gaps.txt 2>&1
timeout -k 10 120 python tools/pinned_probe.py >> $O/${TAG}_pinned_path_copy_gaps.txt 2>&1
cp gpurun_out/r5_pmc_check.log $O/${TAG}_pmc_check.txt 2>/dev/null
# (gpurun copies back at most 64 MiB of gpurun_out/: the raw rocprofv3 trees -- kernel traces of thousands of launches --
# stay on the box, the summaries above are what is kept)
rm -rf gpurun_out/kstats_*/ gpurun_out/sq_*/ gpurun_out/traffic*/ gpurun_out/r5_copytrace/ gpurun_out/tr_*/ 2>/dev/null
du -sh gpurun_out | tail -1
ls -la $O | tail -40
echo "collect done"
