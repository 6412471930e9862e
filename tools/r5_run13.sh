cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "## 640x480: k_integrate grid 3072 / 4096 (default) / 6144"
bash tools/ab_bench.sh "" libratsdf_g3072.so libratsdf.so libratsdf_g6144.so
echo "## 1280x720: look-ahead split 80 / 90 / 100 (default) % in k_front; grid 6144 / 8192 (default) / 12288"
bash tools/ab_bench.sh "--config hd2mm" libratsdf_hs80.so libratsdf_hs90.so libratsdf.so libratsdf_hg6144.so libratsdf_hg12288.so
} > gpurun_out/r5_sweeps.log 2>&1; cat gpurun_out/r5_sweeps.log
