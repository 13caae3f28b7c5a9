#!/bin/bash
# usage (GPU box): tools/final_round_stress.sh <version tag>  -- the stress half of tools/profile_round.sh (bench line, rocprofv3 kernel stats,
# traffic passes, instruction-count passes of k_lift_stream) for a call of its own when the budget does not hold both halves
set -u
export TMPDIR=/tmp
v=${1:-r06v4}
o=gpurun_out/$v
mkdir -p $o
S="--workload stress --reads 100000 --steps 3 --warmup 1 --e2e-reads 0"
SK=${PLO_PROFILE_STRESS_KERNEL:-k_lift_stream}
timeout 150 python bench.py $S > $o/stress_bench.json 2>/dev/null
head -c 300 $o/stress_bench.json; echo
PMC_TIMEOUT=100 tools/pmc_pass.sh "FETCH_SIZE" "$SK" $S > $o/stress_pmc_fetch.csv 2>&1
PMC_TIMEOUT=100 tools/pmc_pass.sh "WRITE_SIZE" "$SK" $S > $o/stress_pmc_write.csv 2>&1
cat $o/stress_pmc_fetch.csv $o/stress_pmc_write.csv
rm -rf /tmp/kt2; timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python3 bench.py $S --no-cpu-baseline --overlap-workers 0 --window-calls 0 > $o/stress_bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/kt2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $o/stress_kernel_stats.csv
grep "k_lift" $o/stress_kernel_stats.csv | head -3
PMC_TIMEOUT=100 tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "$SK" $S > $o/stress_pmc_sq1.csv 2>&1
PMC_TIMEOUT=100 tools/pmc_pass.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" "$SK" $S > $o/stress_pmc_sq2.csv 2>&1
cat $o/stress_pmc_sq1.csv $o/stress_pmc_sq2.csv
