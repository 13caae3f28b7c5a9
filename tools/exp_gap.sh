#!/bin/bash
# usage (GPU box): tools/exp_gap.sh <lib names...> -- k_lift_lanes on wgs30x (2 M reads) with smaller LDS regions (the liftover's allowance per
# block-map key below its bound: overflowing items take the retry list) and the groups cut by LDS budget inside wider sort windows
out=gpurun_out/exp_gap.txt
: > $out
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --e2e-reads 0 --window-calls 0 --overlap-workers 0 --steps 6 "$@" > /tmp/exp.json 2> /tmp/exp.err
  python3 - "$label" <<'PY' >> gpurun_out/exp_gap.txt
import json, sys
try:
    r = json.loads(open('/tmp/exp.json').read().strip().splitlines()[-1])
    ro = r['roofline']
    print(f"{sys.argv[1]:44s} step {r['ms_per_step']:6.3f} ms  lanes {ro['lift_lanes_ms']:6.3f}  enum {ro['enumerate_ms']:5.3f}  retry {ro['lift_retry_ms']:5.3f} ({r['config']['retry_items_per_gpu']})  util {ro['lane_utilisation']:.2f}")
except Exception as e:
    print(sys.argv[1], 'ERR', e, open('/tmp/exp.err').read()[-400:])
PY
}
P=$PWD/portello_amd
for lib in "$@"; do
  run "$lib default" PLO_LIB=$P/$lib.so --
  for w in 256 512 1024; do run "$lib budget window $w" PLO_LIB=$P/$lib.so PLO_LANE_BUDGET=1 PLO_LANE_SORT_WINDOW=$w --; done
done
cat $out
