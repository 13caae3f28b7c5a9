#!/bin/bash
# usage (GPU box): tools/exp_ab_libs.sh <lib file names under portello_amd/ ...>  -- k_lift_lanes on wgs30x 2 M reads for each build, twice, interleaved
for rep in 1 2; do for lib in "$@"; do echo -n "$lib: "; python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 8 --lib $lib 2>&1 | grep -o "lanes [0-9.]* ms ([0-9]* items)" ; done; done
