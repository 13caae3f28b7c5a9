#!/bin/bash
# usage (GPU box): tools/exp_stream.sh [quick] -- the streaming heavy-item kernel (k_lift_stream) against k_lift_lanes_g on the stress
# profile: items per team at 100 k reads, the 500 k-read batch of the streamed 2 M configuration, a reference-sized 50 k-read window
out=gpurun_out/exp_stream.txt
: > $out
run() {  # label, env..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 900 python bench.py --workload stress --no-cpu-baseline --e2e-reads 0 --window-calls 0 "$@" > /tmp/exp.json 2> /tmp/exp.err
  python3 - "$label" <<'PY' >> gpurun_out/exp_stream.txt
import json, sys
try:
    r = json.loads(open('/tmp/exp.json').read().strip().splitlines()[-1])
    ro = r['roofline']
    print(f"{sys.argv[1]:34s} {r['value']/1e6:7.2f} M reads/s  step {r['ms_per_step']:7.2f} ms  heavy {ro.get('lift_heavy_ms', ro.get('lift_heavy_lanes_ms', 0.0)):7.2f}  mid {ro['lift_mid_ms']:6.2f}  retry {ro['lift_retry_ms']:5.2f} ({r['config']['retry_items_per_gpu']})  util {ro['lane_utilisation']:.2f}")
except Exception as e:
    print(sys.argv[1], 'ERR', e, open('/tmp/exp.err').read()[-400:])
PY
}
run "100k g"            PLO_LANE_STREAM=0 -- --reads 100000
for p in 64 128; do run "100k stream per_team $p" PLO_LANE_STREAM=1 PLO_LANE_HEAVY_PER=$p -- --reads 100000; done
run "50k mid (default)" PLO_LANE_STREAM=0 -- --reads 50000
run "50k stream forced" PLO_LANE_STREAM=1 PLO_LANE_HEAVY_MIN=0 -- --reads 50000
if [ "${1:-}" != quick ]; then
run "25k stream forced" PLO_LANE_STREAM=1 PLO_LANE_HEAVY_MIN=0 -- --reads 25000
run "25k mid"           PLO_LANE_STREAM=0 -- --reads 25000
fi
run "500k g (w3)"       PLO_LANE_STREAM=0 -- --reads 500000 --steps 5
for p in 128 256; do run "500k stream per_team $p" PLO_LANE_STREAM=1 PLO_LANE_HEAVY_PER=$p -- --reads 500000 --steps 5; done
cat $out
