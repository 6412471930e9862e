cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/r5_tests3.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_tests3.log; tail -8 gpurun_out/r5_tests3.log
timeout -k 10 300 python tools/zero_waves.py > gpurun_out/r5_zero_waves.log 2>&1; tail -5 gpurun_out/r5_zero_waves.log
rm -rf gpurun_out/r5_copytrace
timeout -k 10 200 rocprofv3 --hip-runtime-trace --memory-copy-trace --output-format csv -d gpurun_out/r5_copytrace -- python3 tools/pinned_probe.py 512 > gpurun_out/r5_copytrace.log 2>&1
tail -2 gpurun_out/r5_copytrace.log; ls -la gpurun_out/r5_copytrace/*/
timeout -k 10 400 python bench.py > gpurun_out/r5_bench2.json 2> gpurun_out/r5_bench2.err; echo "bench rc=$?"; tail -c 400 gpurun_out/r5_bench2.err
