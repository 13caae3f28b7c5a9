#!/bin/bash
S="--workload stress --reads 100000 --e2e-reads 0"
for v in 0 1; do
  export PLO_LANE_STREAM=$v
  K=k_lift_lanes_g; [ $v = 1 ] && K=k_lift_stream
  echo "== PLO_LANE_STREAM=$v ($K)"
  tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "$K" $S
  tools/pmc_pass.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" "$K" $S
  tools/pmc_pass.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU" "$K" $S
  tools/pmc_pass.sh "FETCH_SIZE" "$K" $S
  tools/pmc_pass.sh "WRITE_SIZE" "$K" $S
done
