#!/bin/bash
# usage (GPU box): tools/exp_nomem.sh <lib names...> -- timing-experiment builds (portello_amd/exp_*.so, chosen with PLO_LIB) of the
# streaming kernel on the stress profile, 100 k and 500 k reads
out=gpurun_out/exp_nomem.txt
: > $out
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 900 python bench.py --workload stress --no-cpu-baseline --e2e-reads 0 --window-calls 0 "$@" > /tmp/exp.json 2> /tmp/exp.err
  python3 - "$label" <<'PY' >> gpurun_out/exp_nomem.txt
import json, sys
try:
    r = json.loads(open('/tmp/exp.json').read().strip().splitlines()[-1])
    ro = r['roofline']
    print(f"{sys.argv[1]:40s} step {r['ms_per_step']:7.2f} ms  heavy {ro.get('lift_heavy_ms', ro.get('lift_heavy_lanes_ms', 0.0)):7.2f}  util {ro['lane_utilisation']:.2f}")
except Exception as e:
    print(sys.argv[1], 'ERR', e, open('/tmp/exp.err').read()[-400:])
PY
}
P=$PWD/portello_amd
for lib in "$@"; do
  run "100k stream/64 $lib" PLO_LIB=$P/$lib.so PLO_LANE_STREAM=1 PLO_LANE_HEAVY_PER=64 -- --reads 100000 --steps 5
  run "500k stream/64 $lib" PLO_LIB=$P/$lib.so PLO_LANE_STREAM=1 PLO_LANE_HEAVY_PER=64 -- --reads 500000 --steps 3
  run "500k stream/128 $lib" PLO_LIB=$P/$lib.so PLO_LANE_STREAM=1 PLO_LANE_HEAVY_PER=128 -- --reads 500000 --steps 3
done
cat $out
