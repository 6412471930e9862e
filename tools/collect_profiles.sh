#!/bin/bash
# Collects the artefacts kept under profiles/ (GPU box): kernel stats, PMC traffic, bench lines.
#   bash tools/collect_profiles.sh [tag]     -> gpurun_out/profiles_out/<tag>_*
set -o pipefail
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_out
mkdir -p $O
stats() {  # name, bench args
  rm -rf gpurun_out/kstats_$1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_$1 -- python3 bench.py --cpu-frames 0 --host-frames 0 --no-secondary $2 > gpurun_out/kstats_$1.log 2>&1 &&
  cp "$(ls -t gpurun_out/kstats_$1/*/*kernel_stats.csv | head -1)" $O/${TAG}_kernel_stats_$1.csv &&
  tail -1 gpurun_out/kstats_$1.log > $O/${TAG}_bench_line_under_rocprof_$1.json
}
stats vga5mm "--steps 5 --warmup 2 --streams 0" &&
stats hd2mm "--config hd2mm --steps 5 --warmup 2 --streams 0" &&
stats bigmap "--config bigmap" &&
bash tools/traffic.sh "--streams 0 --no-secondary" traffic_latest.json > gpurun_out/traffic.log 2>&1 &&
bash tools/traffic.sh "--config hd2mm --streams 0" traffic_hd2mm.json > gpurun_out/traffic_hd.log 2>&1 &&
bash tools/traffic.sh "--config bigmap" traffic_bigmap.json > gpurun_out/traffic_big.log 2>&1 &&
bash tools/traffic_group.sh 4 > gpurun_out/traffic_grp.log 2>&1 &&
cp $O/traffic_latest.json $O/traffic_hd2mm.json $O/traffic_bigmap.json $O/traffic_group4.json profiles/ &&
timeout -k 10 800 python bench.py > gpurun_out/bench_default.log 2>&1 && tail -1 gpurun_out/bench_default.log > $O/${TAG}_bench_line.json &&
timeout -k 10 500 python bench.py --config hd2mm --host-frames 0 --streams 0 > gpurun_out/bench_hd2mm.log 2>&1 && tail -1 gpurun_out/bench_hd2mm.log > $O/${TAG}_bench_line_hd2mm.json &&
timeout -k 10 500 python bench.py --config bigmap > gpurun_out/bench_bigmap.log 2>&1 && tail -1 gpurun_out/bench_bigmap.log > $O/${TAG}_bench_line_bigmap.json &&
timeout -k 10 500 python bench.py --config flythrough --host-frames 0 --streams 0 > gpurun_out/bench_fly.log 2>&1 && tail -1 gpurun_out/bench_fly.log > $O/${TAG}_bench_line_flythrough_hd2mm.json
echo "collect rc=$?"
