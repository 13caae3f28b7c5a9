#!/bin/bash
# usage (GPU box): tools/exp_budget.sh  -- slice size (occupancy) x sort window x groups by LDS budget, lane kernel on wgs30x 2 M reads;
# libplo_lw1.so: a build of engine.hip with LANE_WAVES = 1 (one wave per workgroup: 10 or 11 waves per CU become possible; the constant is not a
# macro in the tree -- the measurement is in DESIGN.md 4.0b)
o=gpurun_out/budget.log; : > $o
run() { lib=$1; shift; echo "== $lib $*" >> $o; env "$@" PLO_X=0 python tools/tune.py --workload wgs30x --reads 2000000 --sorted --settings auto --steps 6 ${lib:+--lib $lib} 2>&1 | grep -v "amdgpu.ids\|^\[plo\]" | cut -c1-80,270-330 >> $o; }
run ""
run libplo_lw1.so
run libplo_lw1.so PLO_LANE_CAPW=3328
run libplo_lw1.so PLO_LANE_CAPW=3328 PLO_LANE_SORT_WINDOW=256
run libplo_lw1.so PLO_LANE_CAPW=3840
run libplo_lw1.so PLO_LANE_CAPW=3840 PLO_LANE_SORT_WINDOW=256
run libplo_lw1.so PLO_LANE_CAPW=3840 PLO_LANE_SORT_WINDOW=512
run libplo_lw1.so PLO_LANE_CAPW=3840 PLO_LANE_SORT_WINDOW=512 PLO_LANE_BUDGET=1
cat $o
