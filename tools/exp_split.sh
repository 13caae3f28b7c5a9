#!/bin/bash
out=gpurun_out/exp_split.txt
: > $out
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --e2e-reads 0 --window-calls 0 --overlap-workers 0 --steps 8 "$@" > /tmp/exp.json 2> /tmp/exp.err
  python3 - "$label" <<'PY' >> gpurun_out/exp_split.txt
import json, sys
try:
    r = json.loads(open('/tmp/exp.json').read().strip().splitlines()[-1])
    ro = r['roofline']
    print(f"{sys.argv[1]:44s} step {r['ms_per_step']:6.3f} ms  lanes {ro['lift_lanes_ms']:6.3f}  enum {ro['enumerate_ms']:5.3f}  retry {ro['lift_retry_ms']:5.3f} ({r['config']['retry_items_per_gpu']})  util {ro['lane_utilisation']:.2f}")
except Exception as e:
    print(sys.argv[1], 'ERR', e, open('/tmp/exp.err').read()[-400:])
PY
}
run "one launch (default)" PLO_X=0 --
run "two launches, forward class without shift" PLO_LANE_SPLIT=1 --
run "one launch (default) again" PLO_X=0 --
run "two launches again" PLO_LANE_SPLIT=1 --
cat $out
