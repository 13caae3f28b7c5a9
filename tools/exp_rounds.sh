#!/bin/bash
# usage (GPU box): tools/exp_rounds.sh  -- the static group schedule's rounds: lane kernel time per item at 10.0, 10.5, 10.9 and 11.6 rounds of
# 3 072 resident waves (wgs30x, read counts chosen for the group counts)
o=gpurun_out/quant.log; : > $o
for n in 1903000 2000000 2200000 2082000; do echo "== reads $n" >> $o; python tools/tune.py --workload wgs30x --reads $n --sorted --settings auto --steps 10 2>&1 | grep lanes | cut -c50-90,270-340 >> $o; done
cat $o
