#!/usr/bin/env python3
"""Copies the summaries written by tools/profile_round.sh (gpurun_out/<ver>/) into profiles/ under round-prefixed names
and refreshes profiles/hbm_traffic.json.  usage: tools/save_profiles.py v5 [round]"""
import csv
import json
import os
import shutil
import sys

ver = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r05"
src = f"gpurun_out/{ver}"
pre = f"profiles/{rnd}_{ver}_wgs30x"
rows = list(csv.reader(open(f"{src}/kernel_stats.csv")))
hdr, body = rows[0], rows[1:]
tot = sum(int(r[2]) for r in body)
eng = [r for r in body if r[0].startswith("k_")]
with open(f"{pre}_kernel_stats.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-reads 0 --overlap-workers 0 --window-calls 0   (default workload wgs30x, 10 steps + 2 warm-up)\n")
    f.write(f"# rows of the engine kernels only (the other {len(body) - len(eng)} rows are torch kernels of the synthetic generator); total traced kernel time {tot} ns\n")
    w = csv.writer(f)
    w.writerow(hdr)
    for r in eng:
        w.writerow(r)
shutil.copy(f"{src}/bench.json", f"{pre}_bench.json")
shutil.copy(f"{src}/bench_under_rocprof.json", f"{pre}_bench_under_rocprof.json")


def rd(p):
    return {l.split(",")[0]: float(l.split(",")[1]) for l in open(p) if "," in l}


def insts(*paths):
    """wave-instructions per launch of the dominant kernel by class (the SQ passes): bench.py derives roofline.issue_floor_ms from them"""
    d = {}
    for p in paths:
        if os.path.exists(p):
            d.update(rd(p))
    keys = {"valu": "SQ_INSTS_VALU", "salu": "SQ_INSTS_SALU", "lds": "SQ_INSTS_LDS", "vmem_rd": "SQ_INSTS_VMEM_RD", "vmem_wr": "SQ_INSTS_VMEM_WR",
            "branch": "SQ_INSTS_BRANCH", "wave_cycles": "SQ_WAVE_CYCLES", "wait_any": "SQ_WAIT_ANY"}
    return {k: d[v] for k, v in keys.items() if v in d}


fe, wr = rd(f"{src}/pmc_fetch.csv")["FETCH_SIZE"], rd(f"{src}/pmc_write.csv")["WRITE_SIZE"]
with open(f"{pre}_pmc_summary.csv", "w") as f:
    f.write("# rocprofv3 --pmc <counters> --kernel-include-regex <dominant kernel> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-workers 0 --window-calls 0 --e2e-reads 0\n")
    f.write("# one pass per counter group (tools/pmc_pass.sh, tools/profile_round.sh); value = mean over the 3 launches of the dominant kernel (k_lift_lanes for this workload); FETCH_SIZE / WRITE_SIZE in KiB\n")
    f.write("counter,mean_per_launch,launches\n")
    for p in ["pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_tcc"]:
        if os.path.exists(f"{src}/{p}.csv"):
            for l in open(f"{src}/{p}.csv"):
                if "," in l:
                    f.write(l)
# the factor between FETCH_SIZE and bytes moved, MEASURED for scattered 16-byte loads (tools/calib_fetch.hip; 2.0 = the L2 fetches whole
# 128-byte lines and tallies 64 bytes per request, for scattered accesses as for streaming ones)
fetch_factor, calib = 2.0, None
if os.path.exists(f"{src}/fetch_calibration.json"):
    calib = json.load(open(f"{src}/fetch_calibration.json"))
    shutil.copy(f"{src}/fetch_calibration.json", f"profiles/{rnd}_{ver}_fetch_calibration.json")
    if "_conclusion" in calib:
        fetch_factor = float(calib["_conclusion"]["factor_scattered"])
h = json.load(open("profiles/hbm_traffic.json"))
b = json.load(open(f"{src}/bench.json"))
h["_about"] = ("HBM traffic per launch of the dominant kernel from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, --kernel-include-regex "
               "<kernel>; tools/profile_round.sh).  The counters are KiB; traffic_bytes = (_fetch_factor * FETCH_SIZE + WRITE_SIZE) * 1024.  _fetch_factor is MEASURED for this "
               "kernel's access pattern by tools/calib_fetch.hip (scattered 16-byte loads at known distinct 128-byte lines: one request per line, tallied at 64 bytes, also "
               "when both halves of the line are touched => the L2 fetches whole lines and the counter reports half the bytes, as for streaming reads).  bench.py copies the "
               "number for its workload into roofline.traffic when workload, read count, GPU count and kernel-source hash (portello_amd/build.py source_hash()) match.  "
               "Every entry is written by tools/save_profiles.py from the passes' output; none is typed by hand.")
# (the calibration file is the source only when its FETCH_SIZE pass succeeded, i.e. it carries a conclusion; else the factor is the guide's, and says so)
calibrated = bool(calib) and "_conclusion" in calib
h["_fetch_factor"] = {"value": fetch_factor, "calibrated": calibrated,
                      "source": (f"profiles/{rnd}_{ver}_fetch_calibration.json" if calibrated else
                                 "MI355X_MICROARCH.md (streaming reads); NOT calibrated in this run" + (": the calibration's FETCH_SIZE pass failed" if calib else ""))}
h["wgs30x"] = {b["roofline"]["kernel"]: int((fetch_factor * fe + wr) * 1024), "_fetch_size_kib": fe, "_write_size_kib": wr, "_fetch_factor": fetch_factor,
               "_source": os.path.basename(f"{pre}_pmc_summary.csv"), "_algorithmic_bytes": b["roofline"]["algorithmic_bytes_per_launch"],
               "_reads": b["config"]["reads_this_rank"], "_source_hash": b["config"]["kernel_source_hash"], "_n_gpus": b["n_gpus"],
               "_insts": insts(f"{src}/pmc_sq1.csv", f"{src}/pmc_sq2.csv")}
json.dump(h, open("profiles/hbm_traffic.json", "w"), indent=1)
print(h["wgs30x"])
print(json.dumps(b["roofline"]))
print(b["value"], b["ms_per_step"], b["cpu_baseline"]["value"])
for r in eng[:4]:
    print(r[0][:40], r[1], r[3])

# ---- stress workload: its dominant kernel (k_lift_lanes_g at 100 k reads) --------------------------------------------------------------------------
if os.path.exists(f"{src}/stress_kernel_stats.csv"):
    pre = f"profiles/{rnd}_{ver}_stress"
    rows = list(csv.reader(open(f"{src}/stress_kernel_stats.csv")))
    eng = [r for r in rows[1:] if r[0].startswith("k_") or "k_lift" in r[0]]
    with open(f"{pre}_kernel_stats.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload stress --reads 100000 --steps 3 --warmup 1 --e2e-reads 0 --no-cpu-baseline\n")
        f.write("# rows of the engine kernels only\n")
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in eng:
            w.writerow(r)
    shutil.copy(f"{src}/stress_bench.json", f"{pre}_bench.json")
    shutil.copy(f"{src}/stress_bench_under_rocprof.json", f"{pre}_bench_under_rocprof.json")
    sfe, swr = rd(f"{src}/stress_pmc_fetch.csv")["FETCH_SIZE"], rd(f"{src}/stress_pmc_write.csv")["WRITE_SIZE"]
    with open(f"{pre}_pmc_summary.csv", "w") as f:
        sk = json.load(open(f"{src}/stress_bench.json"))["roofline"]["kernel"]
        f.write(f"# rocprofv3 --pmc <counters> --kernel-include-regex {sk} --output-format csv -- python3 bench.py --workload stress --reads 100000 --steps 3 --warmup 1 --e2e-reads 0 --no-cpu-baseline\n")
        f.write(f"# one pass per counter group; value = mean over the launches of {sk}; FETCH_SIZE / WRITE_SIZE in KiB\n")
        f.write("counter,mean_per_launch,launches\n")
        for p in ["stress_pmc_fetch", "stress_pmc_write", "stress_pmc_sq1", "stress_pmc_sq2"]:
            if os.path.exists(f"{src}/{p}.csv"):
                for l in open(f"{src}/{p}.csv"):
                    if "," in l:
                        f.write(l)
        if os.path.exists(f"{src}/stress_g_pmc_fetch.csv"):
            f.write("# the same workload through k_lift_lanes_g (PLO_LANE_STREAM=0): FETCH_SIZE / WRITE_SIZE per launch\n")
            for p in ["stress_g_pmc_fetch", "stress_g_pmc_write"]:
                for l in open(f"{src}/{p}.csv"):
                    if "," in l:
                        f.write("k_lift_lanes_g:" + l)
    if os.path.exists(f"{src}/stress_g_bench.json"):
        shutil.copy(f"{src}/stress_g_bench.json", f"{pre}_bench_lanes_g.json")
    sb = json.load(open(f"{src}/stress_bench.json"))
    h["stress"] = {sb["roofline"]["kernel"]: int((fetch_factor * sfe + swr) * 1024), "_fetch_size_kib": sfe, "_write_size_kib": swr, "_fetch_factor": fetch_factor,
                   "_source": os.path.basename(f"{pre}_pmc_summary.csv"), "_algorithmic_bytes": sb["roofline"]["algorithmic_bytes_per_launch"],
                   "_reads": sb["config"]["reads_this_rank"], "_source_hash": sb["config"]["kernel_source_hash"], "_n_gpus": sb["n_gpus"],
                   "_insts": insts(f"{src}/stress_pmc_sq1.csv", f"{src}/stress_pmc_sq2.csv")}
    json.dump(h, open("profiles/hbm_traffic.json", "w"), indent=1)
    print(h["stress"])
    print(json.dumps(sb["roofline"]))
    print(sb["value"], sb["ms_per_step"])
