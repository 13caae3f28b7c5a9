#!/bin/bash
# usage: tools/kres.sh <kernel-name-regex> [extra hipcc flags...]  -- registers / spills / occupancy of the engine's kernels as hipcc reports them
# (device code only, no GPU needed); the ISA goes to /tmp/engine.s
rx=${1:-k_lift_lanes}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iportello_amd/csrc --cuda-device-only -S -o /tmp/engine.s -Rpass-analysis=kernel-resource-usage "$@" portello_amd/csrc/engine.hip 2>&1 \
  | grep -A12 "Function Name: .*\($rx\)" | grep "Function Name\|VGPRs:\|Spill\|Occupancy\|SGPRs:\|LDS Size" | sed 's/.*remark: [^ ]* //' | paste - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
