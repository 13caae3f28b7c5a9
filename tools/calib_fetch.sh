#!/bin/bash
# usage (GPU box): tools/calib_fetch.sh [out.json]  -- FETCH_SIZE (and the request counters behind it) per touched 128-byte line for a
# streaming read and for three scattered patterns of known line counts (tools/calib_fetch.hip); one counter group per pass.
set -u
export TMPDIR=/tmp
out=${1:-gpurun_out/fetch_calibration.json}
[ -x tools/calib_fetch ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/calib_fetch tools/calib_fetch.hip || exit 1
tools/calib_fetch > /tmp/calib_plain.jsonl || exit 1
rx="k_stream|k_scatter16|k_scatter_2h|k_probe20"
rm -rf /tmp/calib_ok && mkdir -p /tmp/calib_ok  # the directories of THIS run's passes that succeeded (stale CSVs of earlier runs are not read)
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=/tmp/calib_$(echo $grp | tr ' ' '_')
  rm -rf $d
  if timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "$rx" --output-format csv -d $d -- tools/calib_fetch > /dev/null 2> $d.err; then
    ln -s $d /tmp/calib_ok/$(basename $d)
  else
    echo "pass '$grp' failed (see $d.err): its counters are left out" >&2
  fi
done
python3 - "$out" <<'PY'
import csv, glob, json, collections, sys
plain = [json.loads(l) for l in open("/tmp/calib_plain.jsonl") if l.startswith("{")]
lines = {r["kernel"]: r["lines_128B"] for r in plain}
useful = {r["kernel"]: r["useful_bytes"] for r in plain}
best = {}
for r in plain:
    best[r["kernel"]] = min(best.get(r["kernel"], 1e9), r["ms"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
import os
for f in [f_ for d in sorted(glob.glob("/tmp/calib_ok/*")) for f_ in glob.glob(os.path.realpath(d) + "/**/*counter_collection.csv", recursive=True)]:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"_what": "tools/calib_fetch.hip under rocprofv3 --pmc (tools/calib_fetch.sh): counters per dispatch (mean of 3) and per touched 128-byte line",
       "_note": "FETCH_SIZE is reported in KiB by rocprofv3; bytes_per_line = FETCH_SIZE x 1024 / lines"}
for k, ctrs in sorted(acc.items()):
    name = next((n for n in lines if n in k), k)
    e = {"lines_128B": lines.get(name), "useful_bytes": useful.get(name), "best_ms_unprofiled": best.get(name)}
    for c, v in sorted(ctrs.items()):
        m = sum(v) / len(v)
        e[c] = m
        if lines.get(name):
            e[c + "_per_line"] = (m * 1024 if c == "FETCH_SIZE" else m) / lines[name]
    if "FETCH_SIZE" in e and lines.get(name):
        e["fetch_bytes_per_line"] = e["FETCH_SIZE"] * 1024 / lines[name]
    res[name] = e
s, sc, h2, pr = (res.get(n, {}).get("fetch_bytes_per_line") for n in ("k_stream", "k_scatter16", "k_scatter_2h", "k_probe20"))
if s and sc:
    res["_conclusion"] = {"stream_bytes_counted_per_128B_line": s, "scatter16_bytes_counted_per_line": sc, "scatter_both_halves_per_line": h2, "probe20_per_line": pr,
                          "factor_stream": 128.0 / s, "request_granularity_bytes": 128 if (h2 and h2 < 1.5 * sc) else 64,
                          "factor_scattered": (128.0 / sc) if (h2 and h2 < 1.5 * sc) else (64.0 / sc)}
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res.get("_conclusion", res), indent=1))
PY
