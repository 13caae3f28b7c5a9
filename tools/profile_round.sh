#!/bin/bash
# usage (GPU box): tools/profile_round.sh <version tag>  -- bench line, rocprofv3 kernel stats of the same command, PMC passes
# (one counter group per pass, --pmc never combined with trace domains other than the kernel trace); then the stress workload.
set -u
export TMPDIR=/tmp
v=${1:-v1}
o=gpurun_out/$v
mkdir -p $o
python bench.py > $o/bench.json 2>/dev/null
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --no-cpu-baseline --e2e-reads 0 --overlap-workers 0 --window-calls 0 > $o/bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $o/kernel_stats.csv
PMC_EXTRA="--e2e-reads 0"
tools/pmc_pass.sh "FETCH_SIZE" "${PLO_PROFILE_KERNEL:-^k_lift_lanes\\(}" $PMC_EXTRA > $o/pmc_fetch.csv 2>&1
tools/pmc_pass.sh "WRITE_SIZE" "${PLO_PROFILE_KERNEL:-^k_lift_lanes\\(}" $PMC_EXTRA > $o/pmc_write.csv 2>&1
tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "${PLO_PROFILE_KERNEL:-^k_lift_lanes\\(}" $PMC_EXTRA > $o/pmc_sq1.csv 2>&1
tools/pmc_pass.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" "${PLO_PROFILE_KERNEL:-^k_lift_lanes\\(}" $PMC_EXTRA > $o/pmc_sq2.csv 2>&1
tools/pmc_pass.sh "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "${PLO_PROFILE_KERNEL:-^k_lift_lanes\\(}" $PMC_EXTRA > $o/pmc_tcc.csv 2>&1
# what FETCH_SIZE counts for scattered 16-byte loads (tools/calib_fetch.hip): the factor save_profiles.py applies
tools/calib_fetch.sh $o/fetch_calibration.json > $o/fetch_calibration.log 2>&1
# stress: the lane-per-item kernel over heavy items (100 k heavy items: above its threshold)
S="--workload stress --reads 100000 --steps 3 --warmup 1 --e2e-reads 0"
SK=${PLO_PROFILE_STRESS_KERNEL:-k_lift_stream}
python bench.py $S > $o/stress_bench.json 2>/dev/null
rm -rf /tmp/kt2; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python3 bench.py $S --no-cpu-baseline --overlap-workers 0 --window-calls 0 > $o/stress_bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/kt2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $o/stress_kernel_stats.csv
tools/pmc_pass.sh "FETCH_SIZE" "$SK" $S > $o/stress_pmc_fetch.csv 2>&1
tools/pmc_pass.sh "WRITE_SIZE" "$SK" $S > $o/stress_pmc_write.csv 2>&1
tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "$SK" $S > $o/stress_pmc_sq1.csv 2>&1
tools/pmc_pass.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" "$SK" $S > $o/stress_pmc_sq2.csv 2>&1
# the same workload through the kernel over global regions (PLO_LANE_STREAM=0), for the traffic comparison
PLO_LANE_STREAM=0 tools/pmc_pass.sh "FETCH_SIZE" "k_lift_lanes_g" $S > $o/stress_g_pmc_fetch.csv 2>&1
PLO_LANE_STREAM=0 tools/pmc_pass.sh "WRITE_SIZE" "k_lift_lanes_g" $S > $o/stress_g_pmc_write.csv 2>&1
PLO_LANE_STREAM=0 python bench.py $S --no-cpu-baseline --overlap-workers 0 --window-calls 0 > $o/stress_g_bench.json 2>/dev/null
head -c 400 $o/bench.json; echo; grep "k_lift" $o/kernel_stats.csv | head -5; grep "k_lift" $o/stress_kernel_stats.csv | head -5; cat $o/pmc_*.csv $o/stress_pmc_*.csv
