# k_front role ablations and look-ahead split on the diagnostic build (RATSDF_DEBUG 3 / 11 / 12, RATSDF_CAND_SPLIT)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=$GRAFT_REPO_ROOT/ra-slam_amd/csrc/build/libratsdf_stamps.so
{
echo "## role ablations of k_front, diagnostic build (RATSDF_DEBUG 0 = all roles, 3 = no visible list, 11 = no consumers, 12 = no pool releases), look-ahead pass hosted as by default (20 %) and not at all (CAND_SPLIT=0,0)"
bash tools/kstats2.sh "RATSDF_LIB=$S RATSDF_DEBUG=0" "RATSDF_LIB=$S RATSDF_DEBUG=3" "RATSDF_LIB=$S RATSDF_DEBUG=11" "RATSDF_LIB=$S RATSDF_DEBUG=12"
bash tools/kstats2.sh "RATSDF_LIB=$S RATSDF_CAND_SPLIT=0,0 RATSDF_DEBUG=0" "RATSDF_LIB=$S RATSDF_CAND_SPLIT=0,0 RATSDF_DEBUG=3" "RATSDF_LIB=$S RATSDF_CAND_SPLIT=0,0 RATSDF_DEBUG=11"
echo "## look-ahead split (percent of the next frame's candidate pass in k_front; the rest in k_integrate)"
bash tools/kstats2.sh "RATSDF_LIB=$S RATSDF_CAND_SPLIT=0" "RATSDF_LIB=$S RATSDF_CAND_SPLIT=10" "RATSDF_LIB=$S RATSDF_CAND_SPLIT=20" "RATSDF_LIB=$S RATSDF_CAND_SPLIT=30" "RATSDF_LIB=$S RATSDF_CAND_SPLIT=40" "RATSDF_LIB=$S RATSDF_CAND_SPLIT=60"
} > gpurun_out/r5_front_roles.log 2>&1
cat gpurun_out/r5_front_roles.log
