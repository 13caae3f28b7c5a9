#!/bin/bash
# usage (GPU box): tools/exp_sweep.sh <out-name>  -- lane-kernel knob sweep on wgs30x (2 M reads), timing build for trip counts
set -u
o=gpurun_out/${1:-sweep}.log
: > $o
run() { echo "== $*" >> $o; env "$@" python tools/tune.py --workload wgs30x --reads 2000000 --sorted --settings auto --steps 4 --timing >> $o 2>&1; }
run PLO_X=0
for w in 256 512 2048; do run PLO_LANE_SORT_WINDOW=$w; done
for w in 128 256 512; do run PLO_LANE_SORT_WINDOW=$w PLO_LANE_CAPW=3840; done
run PLO_LANE_SORT=0
echo "== plain build" >> $o
for w in 128 256 512; do echo "== window $w" >> $o; PLO_LANE_SORT_WINDOW=$w python tools/tune.py --workload wgs30x --reads 2000000 --sorted --settings auto --steps 6 >> $o 2>&1; done
grep -v "^\[plo\]" $o | tail -60
