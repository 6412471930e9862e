cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x -k "visible_list_over" > gpurun_out/r5_tests7.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_tests7.log; tail -15 gpurun_out/r5_tests7.log
timeout -k 10 500 python bench.py --config flythrough --host-frames 0 --streams 0 > gpurun_out/r5_fly_hd.log 2>&1; echo "fly rc=$?"; tail -c 1500 gpurun_out/r5_fly_hd.log
timeout -k 10 600 python tools/soak.py > gpurun_out/r5_soak.log 2>&1; echo "soak rc=$?"; tail -5 gpurun_out/r5_soak.log
