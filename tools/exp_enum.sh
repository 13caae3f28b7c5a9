#!/bin/bash
# usage (GPU box): tools/exp_enum.sh <out-name> [lib ...]  -- the enumerate pass: per-kernel times (rocprofv3 --stats) of a loop over one wgs30x batch,
# for the library and for variant builds of it (file names under portello_amd/)
set -u
n=${1:-enum}; shift
o=$GRAFT_REPO_ROOT/gpurun_out/$n.log
: > $o
export TMPDIR=/tmp
for lib in "" "$@"; do
  echo "== lib ${lib:-default}" >> $o
  python tools/tune.py --workload wgs30x --reads 2000000 --sorted --settings auto --steps 8 ${lib:+--lib $lib} >> $o 2>&1
  python tools/tune.py --workload wgs30x --reads 50000 --sorted --settings auto --steps 8 ${lib:+--lib $lib} >> $o 2>&1
  rm -rf /tmp/kt_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -o enum -- python3 tools/tune.py --workload wgs30x --reads 2000000 --sorted --settings auto --steps 10 ${lib:+--lib $lib} > /dev/null 2>&1
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 -c "import csv,sys; [print(r[0][:40], r[1], r[3]) for r in csv.reader(open(sys.argv[1])) if r[0].startswith(\"k_\")]" "$f" >> $o
done
grep -v "^\[plo\]\|amdgpu.ids" $o | cut -c1-330
