#!/bin/bash
# usage (GPU box): tools/exp_scan.sh  -- the wgs30x step and the 50 k-read window with the one-launch scan (k_scan_chain) and with the three scan launches
for m in 1 0 1 0; do
  PLO_SCAN_CHAIN=$m python bench.py --no-cpu-baseline --e2e-reads 0 --overlap-workers 0 --window-calls 200 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PLO_SCAN_CHAIN=$m step %.4f ms, enumerate %.4f ms, lanes %.4f ms, window_50k %.4f ms' % (d['ms_per_step'], d['roofline']['enumerate_ms'], d['roofline']['kernel_ms'], d['window_50k']['ms_per_call']))"
done
