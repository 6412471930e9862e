cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_bench.sh "" libratsdf_split0.so libratsdf_split5.so libratsdf_split10.so libratsdf_split15.so libratsdf.so > gpurun_out/r5_split_ab2.log 2>&1; cat gpurun_out/r5_split_ab2.log
