#!/usr/bin/env python3
"""Sweeps the tile-kernel tunables (PLO_WINDOW / PLO_BIG_THRESH / PLO_CAP) on one synthetic workload and prints the
lift-kernel time per setting.  GPU only."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, devbatch, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="wgs30x")
ap.add_argument("--reads", type=int, default=400000)
ap.add_argument("--settings", default="160:176:512,128:144:384,96:112:320,256:256:768,192:208:640")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--stages", default="31")
ap.add_argument("--rev-frac", type=float, default=0.5)
ap.add_argument("--contig-indel", type=float, default=-1.0, help="contig-vs-reference insertion and deletion rate (default: the workload's 1e-4 each)")
ap.add_argument("--sorted", action="store_true", help="reads of every contig in coordinate order")
ap.add_argument("--lib", default="", help="alternative library file name under portello_amd/")
ap.add_argument("--timing", action="store_true", help="use the instrumented build and print per-phase cycles")
args = ap.parse_args()
dev = torch.device("cuda", 0)
if args.lib:
    api.load_library(os.path.join(ROOT, "portello_amd", args.lib))
if args.timing:
    import ctypes as C
    from portello_amd import build
    L = api.load_library(os.path.join(ROOT, "portello_amd", "libportello_liftover_timing.so"))
    L.plo_ctx_phase_cycles.restype = None
    L.plo_ctx_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
over = dict(n_reads=args.reads, rev_contig_frac=args.rev_frac)
if args.sorted:
    over["sorted_reads"] = True
if args.contig_indel >= 0:
    over["contig_rates"] = synth.EditRates(mismatch=1e-3, ins=args.contig_indel, dele=args.contig_indel, hpol_frac=0.3, big_indel_prob=0.02)
w = synth.generate(synth.config(args.workload, **over), device=dev)
index = api.Index(w.index_data_device(), 0)
db = devbatch.DeviceBatch.from_workload(w)
desc = db.desc()
for s in [(a, b) for a in args.settings.split(",") for b in args.stages.split(",")]:
    stages = int(s[1])
    if s[0] == "auto":  # the engine's own per-batch geometry
        for k in ("PLO_TILE_WAVES", "PLO_WINDOW", "PLO_BIG_THRESH", "PLO_CAP"):
            os.environ.pop(k, None)
        win = thr = cap = "auto"
        os.environ["PLO_DEBUG_GEOMETRY"] = "1"
    else:
        parts = s[0].split(":")
        win, thr, cap = parts[:3]
        os.environ["PLO_TILE_WAVES"] = parts[3] if len(parts) > 3 else "4"
        os.environ["PLO_WINDOW"], os.environ["PLO_BIG_THRESH"], os.environ["PLO_CAP"] = win, thr, cap
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    ms, big, en, ln, la, hv = [], [], [], [], [], []
    for i in range(args.steps + 1):
        eng.liftover_batch_dev(desc, stages)
        t = eng.timing()
        if i:
            ms.append(t.lift_ms); big.append(t.big_ms); en.append(t.enumerate_ms); ln.append(t.mid_ms); la.append(t.lanes_ms); hv.append(t.heavy_lanes_ms)
    print(f"tw={os.environ.get('PLO_TILE_WAVES', 'auto')} stages={stages} window={win} thresh={thr} cap={cap}: lanes {np.mean(la):.3f} ms ({t.n_lane_items} items)  heavy lanes {np.mean(hv):.3f} ms ({t.n_heavy_lane_items} items)  mid {np.mean(ln):.3f} ms ({t.n_mid_items} items, {t.n_retry_items} retried)  tiles {np.mean(ms):.3f} ms  big {np.mean(big):.3f} ms ({t.n_big_items} items)  enum {np.mean(en):.3f} ms  "
          f"items {t.n_items}  {t.n_items/(np.mean(ms)+np.mean(ln)+np.mean(big)+np.mean(la))/1e3:.1f} M items/s  lane utilisation {t.lane_utilisation:.3f}", flush=True)
    if args.timing:
        ph = (C.c_ulonglong * 12)()
        L.plo_ctx_phase_cycles(eng.handle, ph)
        tot = sum(ph) or 1
        if t.n_lane_items:
            names = ["desc+alloc", "load", "shift walk", "liftover", "lift finish", "simplify", "output"]
            tot = sum(ph[:7]) or 1
            grp = max(1, ph[11])  # (rounds of groups; wave-uniform counters, flushed by lane 0)
            print("   lane phase share: " + "  ".join(f"{n} {100*ph[i]/tot:.1f}%" for i, n in enumerate(names)) + f"   cycles/group {tot / grp:.0f}", flush=True)
            print(f"   trip counts per group: liftover iterations {ph[7]/grp:.1f}  shift rounds {ph[8]/grp:.1f}  scan iterations {ph[9]/grp:.1f}  load batches {ph[10]/grp:.1f}  (groups {grp})", flush=True)
        names = ["desc", "load+lshiftA", "lshift cc", "lift:stage+passA", "lift:scatter+passB", "lift:cc", "lencheck", "simplify A+H+B", "simplify cc", "output", "lshift H", "lshift B"]
        print("   phase share: " + "  ".join(f"{n} {100*ph[i]/tot:.1f}%" for i, n in enumerate(names)) + f"   cycles/tile-wave {tot/ max(1,(t.n_in_ops//(int(win) if win != 'auto' else 256)+1)):.0f}", flush=True)
    eng.close()
