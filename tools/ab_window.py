#!/usr/bin/env python3
"""A/B of the per-call fixed cost of plo_liftover_batch_dev on a reference-sized window (50 k reads of wgs30x) and on the full 2 M-read batch:
the environment switches of the one-round-trip path (PLO_FAST_FUSE, PLO_PHASE_EVENTS, PLO_FAST_LAUNCH_BOUND, PLO_LANE_GROUP) are read at
plo_ctx_create, so every variant gets a context of its own in ONE process on ONE box (boxes differ by +-3 %).  Every variant's results are
compared with the first variant's (bit for bit), interleaved rounds level the clock drift.  GPU only.

usage: python tools/ab_window.py [--reads 2000000] [--calls 300] [--rounds 3] > gpurun_out/ab_window.txt"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, devbatch, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=2_000_000)
ap.add_argument("--calls", type=int, default=300)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--big-steps", type=int, default=20)
ap.add_argument("--lib", default="", help="alternative library file name under portello_amd/ (e.g. a -DPLO_LANE_GAP_HALVES=2 build, whose light-item kernel "
                                          "hands half a per cent of wgs30x's items to the retry list: the retry launch of the one-round-trip path with work to do)")
args = ap.parse_args()

def V(fuse, events, bound, group=None):
    e = {"PLO_FAST_FUSE": str(fuse), "PLO_PHASE_EVENTS": str(events), "PLO_FAST_LAUNCH_BOUND": str(bound)}
    if group:
        e["PLO_LANE_GROUP"] = str(group)
    return e


VARIANTS = [
    ("round 5/6 path: k_lift_retry + k_sum_stats, events, grids by capacity", V(0, 1, 0)),
    ("k_lift_retry_sum", V(1, 1, 0)),
    ("launch bound", V(0, 1, 1)),
    ("k_lift_retry_sum + launch bound (default)", V(1, 1, 1)),
    ("default, no phase events", V(1, 0, 1)),
    ("round 5/6 path, no phase events", V(0, 0, 0)),
    ("default, no phase events, groups of 18 (the window's own choice: 32)", V(1, 0, 1, 18)),
    ("default, no phase events, groups of 20", V(1, 0, 1, 20)),
    ("default, no phase events, groups of 24", V(1, 0, 1, 24)),
    ("default, no phase events, groups of 48", V(1, 0, 1, 48)),
]
KEYS = ("PLO_FAST_FUSE", "PLO_PHASE_EVENTS", "PLO_FAST_LAUNCH_BOUND", "PLO_LANE_GROUP")

if args.lib:
    api.load_library(os.path.join(ROOT, "portello_amd", args.lib))
dev = torch.device("cuda", 0)
w = synth.generate(synth.config("wgs30x", n_reads=args.reads), device=dev)
index = api.Index(w.index_data_device(), 0)
lo = w.n_reads // 2
wdb = devbatch.DeviceBatch.from_workload(w, lo, lo + 50_000)
wdesc = wdb.desc()
bdb = devbatch.DeviceBatch.from_workload(w)
bdesc = bdb.desc()


engines = []
for name, env in VARIANTS:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    engines.append(api.Engine(index, stream=torch.cuda.current_stream().cuda_stream))
for k in KEYS:
    os.environ.pop(k, None)


def group_env(env):
    """PLO_LANE_GROUP is read by every call (the other switches at plo_ctx_create)"""
    if "PLO_LANE_GROUP" in env:
        os.environ["PLO_LANE_GROUP"] = env["PLO_LANE_GROUP"]
    else:
        os.environ.pop("PLO_LANE_GROUP", None)


def signature(eng, desc):
    """everything a caller reads of a result (the third call on the context: the one-round-trip path), the CIGARs item by item -- slabs are
    handed out per wave, so offsets differ between runs and are not compared"""
    for _ in range(3):
        out = eng.liftover_batch_dev(desc)
    eng.sync()
    t = eng.timing()
    n = int(out.n_items)
    arrs = {"seg": eng.download(out.item_seg, np.uint32, n), "cseg": eng.download(out.item_cseg, np.uint32, n),
            "status": eng.download(out.item_status, np.uint8, n), "flip": eng.download(out.item_need_flipped, np.uint8, n),
            "mapq": eng.download(out.item_mapq, np.uint8, n), "chrom": eng.download(out.item_chrom_index, np.uint32, n),
            "pos": eng.download(out.item_ref_pos, np.int64, n), "clen": eng.download(out.item_cigar_len, np.uint32, n)}
    coff = eng.download(out.item_cigar_off, np.uint64, n).astype(np.int64)
    cig = eng.download(out.cigar, np.uint32, int(out.n_cigar))
    clen = arrs["clen"].astype(np.int64)
    idx = np.repeat(coff - np.concatenate([[0], np.cumsum(clen)[:-1]]), clen) + np.arange(int(clen.sum()))
    arrs["ops"] = cig[idx]
    counts = (int(t.n_items), int(t.n_in_ops), int(t.n_out_ops), int(t.n_retry_items), int(t.host_syncs))
    return arrs, counts


ref = {}


def check(vi, name, eng, what, desc):
    arrs, counts = signature(eng, desc)
    if what not in ref:
        ref[what] = (arrs, counts)
        print(f"reference variant, {what}: items {counts[0]}, in ops {counts[1]}, out ops {counts[2]}, retry items {counts[3]}, host syncs {counts[4]}", flush=True)
    else:
        same = counts == ref[what][1] and all(np.array_equal(arrs[k], ref[what][0][k]) for k in arrs)
        print(f"variant '{name}', {what}: results {'identical' if same else 'DIFFERENT'} {'' if same else (counts, ref[what][1])}", flush=True)
        if not same:
            sys.exit(3)


def time_windows(key):
    for r in range(args.rounds):
        for (name, env), eng in zip(VARIANTS, engines):
            group_env(env)
            for _ in range(10):
                eng.liftover_batch_dev(wdesc)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(args.calls):
                eng.liftover_batch_dev(wdesc)
            eng.sync()
            res[name][key].append((time.perf_counter() - t0) / args.calls * 1e3)
            res[name][key + "_dev"].append(float(eng.timing().total_ms))


res = {name: {"win": [], "win_dev": [], "big": [], "after": [], "after_dev": []} for name, _ in VARIANTS}
# phase 1: contexts that have seen windows only (the BAM pipeline's workers)
for vi, ((name, env), eng) in enumerate(zip(VARIANTS, engines)):
    group_env(env)
    check(vi, name, eng, "window", wdesc)
time_windows("win")
# phase 2: the whole batch (results of some variants compared, all timed), then the windows again on contexts whose arrays hold 2 M reads' items
for vi, ((name, env), eng) in enumerate(zip(VARIANTS, engines)):
    if "PLO_LANE_GROUP" in env:
        continue
    group_env(env)
    if vi in (0, 3, 4):
        check(vi, name, eng, "whole batch", bdesc)
for r in range(args.rounds):
    for (name, env), eng in zip(VARIANTS, engines):
        if "PLO_LANE_GROUP" in env:
            continue
        group_env(env)
        for _ in range(3):
            eng.liftover_batch_dev(bdesc)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(args.big_steps):
            eng.liftover_batch_dev(bdesc)
        eng.sync()
        res[name]["big"].append((time.perf_counter() - t0) / args.big_steps * 1e3)
for vi, ((name, env), eng) in enumerate(zip(VARIANTS, engines)):
    group_env(env)
    eng.liftover_batch_dev(bdesc)
    check(vi, name, eng, "window", wdesc)
time_windows("after")
print(f"\n50 k-read window, {args.calls} calls back to back per round, {args.rounds} interleaved rounds (ms per call: wall clock min-max; in brackets the events' total); "
      f"whole batch {args.reads} reads, {args.big_steps} steps per round")
print(f"{'variant':90s} {'contexts that saw windows only':34s} {'after a whole batch':34s} whole batch, ms per step")
for name, _ in VARIANTS:
    r = res[name]
    big = f"{min(r['big']):.3f}-{max(r['big']):.3f}" if r["big"] else "-"
    print(f"{name:90s} {min(r['win']):.4f}-{max(r['win']):.4f} [{np.mean(r['win_dev']):.4f}]          {min(r['after']):.4f}-{max(r['after']):.4f} [{np.mean(r['after_dev']):.4f}]          {big}", flush=True)
