cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/r5_tests4.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_tests4.log; tail -6 gpurun_out/r5_tests4.log
timeout -k 10 120 python tools/pinned_probe.py > gpurun_out/r5_pinned_probe.log 2>&1; tail -2 gpurun_out/r5_pinned_probe.log
rm -rf gpurun_out/r5_copytrace
timeout -k 10 200 rocprofv3 --hip-runtime-trace --memory-copy-trace --output-format csv -d gpurun_out/r5_copytrace -- python3 tools/pinned_probe.py 512 > gpurun_out/r5_copytrace.log 2>&1
tail -1 gpurun_out/r5_copytrace.log
timeout -k 10 400 python bench.py > gpurun_out/r5_bench3.json 2> gpurun_out/r5_bench3.err; echo "bench rc=$?"; tail -c 400 gpurun_out/r5_bench3.err
