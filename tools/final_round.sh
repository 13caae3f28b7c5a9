#!/bin/bash
# usage (GPU box): tools/final_round.sh <version tag>  -- the round's last GPU call in order of importance: the whole -m gpu suite, the two
# traffic passes of the dominant kernel (profiles/hbm_traffic.json is keyed by the kernel-source hash), rocprofv3 kernel stats of the bench
# command, a bench line (without the BAM -> BAM leg), the instruction-count passes.  Every step under its own timeout; what the budget cuts
# off is simply missing from gpurun_out/<tag>/ (tools/save_profiles.py takes what is there).
set -u
export TMPDIR=/tmp
v=${1:-r06v4}
o=gpurun_out/$v
mkdir -p $o
timeout 900 python -m pytest tests -q -m gpu > $o/gpu_suite.txt 2>&1
tail -3 $o/gpu_suite.txt
K='^k_lift_lanes\('
PMC_TIMEOUT=120 tools/pmc_pass.sh "FETCH_SIZE" "$K" --e2e-reads 0 > $o/pmc_fetch.csv 2>&1
PMC_TIMEOUT=120 tools/pmc_pass.sh "WRITE_SIZE" "$K" --e2e-reads 0 > $o/pmc_write.csv 2>&1
cat $o/pmc_fetch.csv $o/pmc_write.csv
rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --no-cpu-baseline --e2e-reads 0 --overlap-workers 0 --window-calls 0 > $o/bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $o/kernel_stats.csv
grep "k_lift\|k_scan\|k_seg\|k_item" $o/kernel_stats.csv | head -8
timeout 240 python bench.py --e2e-reads 0 > $o/bench.json 2>/dev/null
head -c 600 $o/bench.json; echo
PMC_TIMEOUT=120 tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "$K" --e2e-reads 0 > $o/pmc_sq1.csv 2>&1
PMC_TIMEOUT=120 tools/pmc_pass.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" "$K" --e2e-reads 0 > $o/pmc_sq2.csv 2>&1
cat $o/pmc_sq1.csv $o/pmc_sq2.csv
