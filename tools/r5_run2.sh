cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r5_tests2.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_tests2.log; tail -8 gpurun_out/r5_tests2.log
timeout -k 10 300 python tools/zero_waves.py > gpurun_out/r5_zero_waves.log 2>&1; tail -5 gpurun_out/r5_zero_waves.log
timeout -k 10 120 python tools/pinned_probe.py > gpurun_out/r5_pinned_probe.log 2>&1; tail -2 gpurun_out/r5_pinned_probe.log
rm -rf gpurun_out/r5_copytrace
timeout -k 10 200 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d gpurun_out/r5_copytrace -- python3 tools/pinned_probe.py 768 > gpurun_out/r5_copytrace.log 2>&1
tail -2 gpurun_out/r5_copytrace.log
f=$(ls gpurun_out/r5_copytrace/*/*memory_copy_trace.csv | head -1); head -3 $f; python3 tools/copy_gaps.py $f
rm -f gpurun_out/r5_copytrace/*/*kernel_trace.csv
