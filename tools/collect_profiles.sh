#!/bin/bash
# Collects the artefacts kept under profiles/ (GPU box): kernel stats, PMC traffic, bench lines.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_out
rm -rf gpurun_out/kstats_final
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_final -- python3 bench.py --steps 5 --warmup 2 --cpu-frames 0 --host-frames 0 > gpurun_out/kstats_final.log 2>&1 &&
cp "$(ls -t gpurun_out/kstats_final/*/*kernel_stats.csv | head -1)" gpurun_out/profiles_out/kernel_stats.csv &&
bash tools/traffic.sh > gpurun_out/traffic.log 2>&1 &&
cp gpurun_out/profiles_out/traffic_latest.json profiles/traffic_latest.json &&
timeout -k 10 800 python bench.py > gpurun_out/bench_default.log 2>&1 && tail -1 gpurun_out/bench_default.log > gpurun_out/profiles_out/bench_line.json &&
timeout -k 10 500 python bench.py --config hd2mm --host-frames 0 > gpurun_out/bench_hd2mm.log 2>&1 && tail -1 gpurun_out/bench_hd2mm.log > gpurun_out/profiles_out/bench_line_hd2mm.json
echo "collect rc=$?"
