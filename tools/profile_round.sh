set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/v11
python bench.py > gpurun_out/v11/bench.json 2>/dev/null
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --no-cpu-baseline > gpurun_out/v11/bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/v11/kernel_stats.csv
tools/pmc_pass.sh "FETCH_SIZE" "k_lift" > gpurun_out/v11/pmc_fetch.csv 2>&1
tools/pmc_pass.sh "WRITE_SIZE" "k_lift" > gpurun_out/v11/pmc_write.csv 2>&1
tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" > gpurun_out/v11/pmc_sq1.csv 2>&1
tools/pmc_pass.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" > gpurun_out/v11/pmc_sq2.csv 2>&1
head -c 600 gpurun_out/v11/bench.json; echo; head -8 gpurun_out/v11/kernel_stats.csv; cat gpurun_out/v11/pmc_*.csv
