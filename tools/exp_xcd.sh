#!/bin/bash
# usage (GPU box): tools/exp_xcd.sh <lib names...> -- k_lift_stream on stress 100 k: time, FETCH_SIZE, WRITE_SIZE per build / placement
S="--workload stress --reads 100000 --e2e-reads 0"
for lib in "$@"; do
  export PLO_LIB=$PWD/portello_amd/$lib.so
  echo "== $lib"
  python bench.py $S --no-cpu-baseline --window-calls 0 --steps 5 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel'], r['roofline']['kernel_ms'], r['ms_per_step'], 'retry', r['config']['retry_items_per_gpu'])"
  tools/pmc_pass.sh "FETCH_SIZE" k_lift_stream $S
  tools/pmc_pass.sh "WRITE_SIZE" k_lift_stream $S
done
