#!/bin/bash
# usage: tools/exp_ab.sh <lib names...> -- stress 100 k and 50 k / 25 k through k_lift_stream for each build (PLO_LIB)
for lib in "$@"; do
  for r in 100000 50000; do
    PLO_LIB=$PWD/portello_amd/$lib.so PLO_LANE_STREAM=1 PLO_LANE_HEAVY_MIN=0 python bench.py --workload stress --reads $r --no-cpu-baseline --e2e-reads 0 --window-calls 0 --steps 6 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', $r, r['roofline']['kernel'], '%.3f ms'%r['roofline']['kernel_ms'], 'step %.3f'%r['ms_per_step'], 'util %.2f'%r['roofline']['lane_utilisation'])"
  done
done
