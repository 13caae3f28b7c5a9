#!/bin/bash
# usage: tools/pmc_pass.sh "<counter list>" [kernel regex] [bench args...]  -- one rocprofv3 PMC pass over bench.py, prints
# the per-launch mean of every counter for the selected kernel.  GPU box only.
set -u
export TMPDIR=/tmp
ctrs="$1"; rx="${2:-k_lift_tiles}"; shift; shift || true
d=$(mktemp -d /tmp/pmc.XXXXXX)
timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc $ctrs --kernel-include-regex "$rx" --output-format csv -d "$d" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-workers 0 --window-calls 0 "$@" > /dev/null 2>&1
python3 - "$d" <<'PY'
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(f"{k},{acc[k] / n[k]:.1f},{n[k]}")
PY
rm -rf "$d"
