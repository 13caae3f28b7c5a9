#!/bin/bash
o=gpurun_out/budget.log; : > $o
run() { echo "== $*" >> $o; env "$@" python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 6 2>&1 | grep -v "amdgpu.ids\|^\[plo\]" | cut -c1-60,150-330 >> $o; }
run PLO_LANE_BUDGET=0
run PLO_LANE_BUDGET=1
run PLO_LANE_BUDGET=1 PLO_LANE_SORT_WINDOW=256
run PLO_LANE_BUDGET=1 PLO_LANE_SORT_WINDOW=1024
run PLO_LANE_BUDGET=1 PLO_LANE_SORT_WINDOW=2048
run PLO_LANE_BUDGET=1 PLO_LANE_CAPW=2816
cat $o
