#!/bin/bash
# usage (GPU box): tools/exp_small.sh <out-name>  -- small windows: the lane-per-item kernel against the wave-cooperative tile kernel
set -u
o=gpurun_out/${1:-small}.log
: > $o
for n in 12500 25000 50000 100000 200000 400000; do
  for mw in default -1; do
    echo "== reads $n lane_max_w $mw" >> $o
    if [ $mw = default ]; then
      PLO_X=0 python tools/tune.py --workload wgs30x --reads $n --sorted --settings auto --steps 6 >> $o 2>&1
    else
      PLO_LANE_MAX_W=$mw python tools/tune.py --workload wgs30x --reads $n --sorted --settings auto --steps 6 >> $o 2>&1
    fi
  done
done
grep -v "^\[plo\]\|amdgpu.ids" $o | cut -c1-330
