cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for set in "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES"; do
  timeout -k 5 60 rocprofv3-avail pmc-check $set > gpurun_out/pmc_check_one.log 2>&1; rc=$?
  echo "pmc-check rc=$rc :: $set"; tail -3 gpurun_out/pmc_check_one.log
done > gpurun_out/r5_pmc_check.log 2>&1
cat gpurun_out/r5_pmc_check.log
timeout -k 10 120 python tools/pinned_probe.py > gpurun_out/r5_pinned_probe.log 2>&1; tail -1 gpurun_out/r5_pinned_probe.log
