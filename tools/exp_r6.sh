#!/bin/bash
# usage (GPU box): tools/exp_r6.sh <out-name> [lib ...]  -- k_lift_lanes on wgs30x 2 M reads for each build (file names under portello_amd/, default the
# product library) x PLO_LANE_STATIC in -1 (fixed slots only) 1 3 6: lane kernel time by HIP events (tools/tune.py), then the timing build's phase shares
set -u
o=gpurun_out/${1:-exp_r6}.log; shift
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(libportello_liftover.so)
: > $o
for lib in "${libs[@]}"; do
  for s in ${PLO_EXP_STATIC:--1 1 3 6}; do
    echo "== $lib PLO_LANE_STATIC=$s" >> $o
    PLO_LANE_STATIC=$s PLO_LANE_STATS=${PLO_LANE_STATS:-0} python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 6 --lib $lib 2>&1 | grep -v "^\[plo\]\|amdgpu.ids" | cut -c1-260 >> $o
  done
done
if [ -f portello_amd/libportello_liftover_timing.so ]; then
  echo "== timing build" >> $o
  python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 4 --timing 2>&1 | grep -v "^\[plo\]\|amdgpu.ids\|phase share: desc " | cut -c1-330 >> $o
  python tools/wave_timeline.py 2>&1 | grep -v "^\[plo\]\|amdgpu.ids" | head -4 >> $o
fi
cat $o
