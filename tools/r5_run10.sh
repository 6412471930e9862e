cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/r5_tests10.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_tests10.log; tail -4 gpurun_out/r5_tests10.log
B=$GRAFT_REPO_ROOT/ra-slam_amd/csrc/build
{
bash tools/kstats2.sh "RATSDF_LIB=$B/libratsdf_pre.so" "RATSDF_LIB=$B/libratsdf.so" "RATSDF_LIB=$B/libratsdf_pre.so" "RATSDF_LIB=$B/libratsdf.so"
bash tools/kstats2.sh -c hd2mm "RATSDF_LIB=$B/libratsdf_pre.so" "RATSDF_LIB=$B/libratsdf.so" "RATSDF_LIB=$B/libratsdf_pre.so" "RATSDF_LIB=$B/libratsdf.so"
bash tools/kstats2.sh -c bigmap "RATSDF_LIB=$B/libratsdf_pre.so" "RATSDF_LIB=$B/libratsdf.so"
} > gpurun_out/r5_ab_reload.log 2>&1
cat gpurun_out/r5_ab_reload.log
bash tools/ab_bench.sh "" libratsdf_pre.so libratsdf.so > gpurun_out/r5_ab_reload_bench.log 2>&1; cat gpurun_out/r5_ab_reload_bench.log
bash tools/ab_bench.sh "--config hd2mm" libratsdf_pre.so libratsdf.so > gpurun_out/r5_ab_reload_bench_hd.log 2>&1; cat gpurun_out/r5_ab_reload_bench_hd.log
