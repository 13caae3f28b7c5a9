#!/bin/bash
# usage (GPU box): tools/exp_quick.sh <out-name> [env assignments...]  -- lane kernel on wgs30x 2 M reads: timing build (phase shares, trip
# counts), plain build, stress 100 k
set -u
o=gpurun_out/${1:-quick}.log; shift
: > $o
echo "== timing build $*" >> $o; env "$@" PLO_X=0 python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 4 --timing >> $o 2>&1
echo "== plain build $*" >> $o; env "$@" PLO_X=0 python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 6 >> $o 2>&1
echo "== stress 100k $*" >> $o; env "$@" PLO_X=0 python tools/tune.py --workload stress --reads 100000 --settings auto --steps 3 >> $o 2>&1
grep -v "^\[plo\]\|amdgpu.ids" $o | grep -v "phase share: desc " | cut -c1-330
