#!/bin/bash
# usage (GPU box): tools/exp_window.sh  -- the lane kernel's phases (timing build) for sort windows of 128 / 512 items, groups cut by LDS budget or not
for cfg in "128 0" "512 0" "512 1" "1024 1"; do set -- $cfg; echo "== window $1 budget $2"; PLO_LANE_SORT_WINDOW=$1 PLO_LANE_BUDGET=$2 PLO_LANE_STATS=1 python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 4 --timing 2>&1 | grep -v "^\[plo\]\|amdgpu.ids\|phase share: desc " | cut -c1-100,230-400; done
