#!/bin/bash
run() { echo "== $*"; env "$@" python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 8 2>&1 | grep -o "lanes [0-9.]* ms" | tr '\n' ' '; echo; }
for rep in 1 2; do
run PLO_LANE_H16=0 PLO_LANE_TAIL=0 PLO_LANE_STATIC=-1
run PLO_LANE_H16=0 PLO_LANE_TAIL=0
run PLO_LANE_H16=0 PLO_LANE_TAIL=1
run PLO_LANE_H16=0 PLO_LANE_TAIL=2
run PLO_LANE_H16=0 PLO_LANE_TAIL=3
run PLO_LANE_H16=0 PLO_LANE_TAIL=2 PLO_LANE_STATIC=-1
done
