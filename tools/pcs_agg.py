#!/usr/bin/env python3
"""Aggregates rocprofv3 PC-sampling CSV output into a histogram by code-object offset (prints `offset,count`)."""
import collections
import csv
import glob
import sys

csv.field_size_limit(1 << 30)
hist = collections.Counter()
cols = None
for f in glob.glob(sys.argv[1] + "/**/*pc_sampling*.csv", recursive=True):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        key = next((c for c in cols if "offset" in c.lower()), None)
        inst = next((c for c in cols if "instruction" in c.lower() and "comment" not in c.lower()), None)
        for r in rd:
            hist[(r.get(key, ""), r.get(inst, "") if inst else "")] += 1
print("# columns:", cols, file=sys.stderr)
for (off, ins), n in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(f"{off},{n},{ins}")
