#!/bin/bash
# usage (GPU box): tools/trace_window.sh <out dir>  -- kernel trace of the bench's window_50k calls: the timeline of ONE call (kernels, durations,
# gaps between them), averaged over the last calls
set -u
export TMPDIR=/tmp
o=${1:-gpurun_out/trace_window}
mkdir -p $o
rm -rf /tmp/ktw; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktw -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --overlap-workers 0 --window-calls 40 > $o/bench.json 2>/dev/null
f=$(find /tmp/ktw -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/trace_window.py "$f" > $o/timeline.txt
cat $o/timeline.txt
