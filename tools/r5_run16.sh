cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SECONDS=0
python bench.py > gpurun_out/r5_bench_notorch.json 2> gpurun_out/r5_bench_notorch.err; echo "bench rc=$? seconds=$SECONDS"; tail -c 300 gpurun_out/r5_bench_notorch.err
python bench.py --config hd2mm --host-frames 0 --streams 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hd2mm', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
python bench.py --config bigmap 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bigmap', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
timeout -k 10 300 python -m pytest tests/test_bench_launch.py -m gpu -q 2>&1 | tail -3
