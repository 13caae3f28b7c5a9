#!/bin/bash
# usage (GPU box): tools/exp_h16.sh -- k_lift_lanes with 16-bit regions: slice, staged block-map entries and sort window (statistics kernel where the utilisation is wanted)
run() { echo "== $*"; env "$@" python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 6 2>&1 | grep -o "lanes [0-9.]* ms ([0-9]* items).*mid [0-9.]* ms ([0-9]* items, [0-9]* retried)\|lane utilisation [0-9.]*" | sed 's/heavy lanes.*mid 0.000 ms//' | tr '\n' ' '; echo; }
run PLO_LANE_H16=0
run PLO_LANE_H16=1
run PLO_LANE_CAPW=2048 PLO_LANE_KVS=512
run PLO_LANE_CAPW=2048 PLO_LANE_KVS=512 PLO_LANE_SORT_WINDOW=256
run PLO_LANE_CAPW=2048 PLO_LANE_KVS=512 PLO_LANE_SORT_WINDOW=512
run PLO_LANE_CAPW=2048 PLO_LANE_KVS=512 PLO_LANE_SORT_WINDOW=512 PLO_LANE_STATS=1
run PLO_LANE_CAPW=2048 PLO_LANE_KVS=512 PLO_LANE_SORT_WINDOW=1024 PLO_LANE_STATS=1
run PLO_LANE_CAPW=2048 PLO_LANE_KVS=256 PLO_LANE_SORT_WINDOW=256 PLO_LANE_STATS=1
