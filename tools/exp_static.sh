#!/bin/bash
# usage (GPU box): tools/exp_static.sh -- k_lift_lanes on wgs30x: rounds of fixed slots in front of the ticket dealing (default: by the launch's rounds)
run() { echo -n "$*: "; env "$@" python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 8 2>&1 | grep -o "lanes [0-9.]* ms" | head -1; }
for rep in 1 2; do
run PLO_X=default
run PLO_LANE_STATIC=1
run PLO_LANE_STATIC=8
done
