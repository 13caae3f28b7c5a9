#!/usr/bin/env python3
"""Where do the lane kernel's waves spend the launch?  Runs one workload through the timing build (-DPLO_PHASE_TIMING) and reads every
wave's first / last tick (constant 100 MHz clock) and hardware id from its statistics slot (plo_ctx_wave_clocks): the launch's span, when
the waves start and end, how long the average wave is at work, per XCD and per CU.  GPU only."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, devbatch, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="wgs30x")
ap.add_argument("--reads", type=int, default=2000000)
ap.add_argument("--lib", default="libportello_liftover_timing.so")
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
L = api.load_library(os.path.join(ROOT, "portello_amd", args.lib))
L.plo_ctx_wave_clocks.restype = C.c_uint
L.plo_ctx_wave_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint]
dev = torch.device("cuda", 0)
w = synth.generate(synth.config(args.workload, n_reads=args.reads), device=dev)
index = api.Index(w.index_data_device(), 0)
db = devbatch.DeviceBatch.from_workload(w)
desc = db.desc()
eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
for i in range(args.steps):
    eng.liftover_batch_dev(desc, 31)
t = eng.timing()
n = 8192
SW = 16
buf = (C.c_ulonglong * (SW * n))()
got = L.plo_ctx_wave_clocks(eng.handle, buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, SW)[:got]
a = a[a[:, 6] > 0]
# the lane kernel's waves are the first range of slots; the retry kernel's follow (later start)
beg, end, hw = a[:, 5].astype(np.int64), a[:, 6].astype(np.int64), a[:, 7].astype(np.int64)
order = np.argsort(beg)
# split launches by a gap in start times > 20 us
b_sorted = beg[order]
cut = np.nonzero(np.diff(b_sorted) > 2000)[0]
first = order[: (cut[0] + 1) if len(cut) else len(order)]
beg, end, hw = beg[first], end[first], hw[first]
ag = a[first]
t0 = beg.min()
span = (end.max() - beg.min()) / 100.0
life = (end - beg) / 100.0
print(f"lanes {t.lanes_ms:.3f} ms (events); waves {len(first)}; launch span {span:.1f} us; wave life mean {life.mean():.1f} us  min {life.min():.1f}  p10 {np.percentile(life, 10):.1f}  "
      f"median {np.median(life):.1f}  p90 {np.percentile(life, 90):.1f}  max {life.max():.1f}")
print(f"starts: first {0.0:.1f} us  p50 {(np.median(beg) - beg.min()) / 100:.1f}  p99 {(np.percentile(beg, 99) - beg.min()) / 100:.1f}  last {(beg.max() - beg.min()) / 100:.1f} us")
e = (end - beg.min()) / 100.0
print(f"ends:   first {e.min():.1f} us  p10 {np.percentile(e, 10):.1f}  p50 {np.median(e):.1f}  p90 {np.percentile(e, 90):.1f}  last {e.max():.1f} us;  mean wave-slot use {life.sum() / (len(first) * span):.3f}")
gn, gmax, glast, gprev, glb = (ag[:, k].astype(np.int64) for k in (8, 9, 10, 11, 12))
print(f"groups per wave: min {gn.min()} mean {gn.mean():.2f} max {gn.max()}; mean group {life.sum() / gn.sum():.1f} us; longest group of a wave: median {np.median(gmax) / 100:.1f} p99 {np.percentile(gmax, 99) / 100:.1f} max {gmax.max() / 100:.1f} us")
print(f"last group of a wave: begins p1 {(np.percentile(glb, 1) - t0) / 100:.1f} p50 {(np.median(glb) - t0) / 100:.1f} max {(glb.max() - t0) / 100:.1f} us; lasts median {np.median(glast) / 100:.1f} p90 {np.percentile(glast, 90) / 100:.1f} max {glast.max() / 100:.1f} us; the one before: median {np.median(gprev) / 100:.1f} max {gprev.max() / 100:.1f}")
late = np.argsort(end)[-5:]
for i in late:
    print(f"  late wave: end {(end[i] - t0) / 100:.1f} us  groups {gn[i]}  longest {gmax[i] / 100:.1f}  last {glast[i] / 100:.1f} (began {(glb[i] - t0) / 100:.1f})  previous {gprev[i] / 100:.1f}")
xcc = hw >> 16
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 0x1
simd = (hw >> 4) & 0x3
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  xcc {x}: waves {m.sum():4d}  life mean {life[m].mean():7.1f} us  last end {e[m].max():7.1f} us")
key = (xcc << 12) | (se << 8) | (sh << 7) | (cu << 2)
per_cu = {}
for k_, l_, e_ in zip(key.tolist(), life.tolist(), e.tolist()):
    per_cu.setdefault(k_, []).append((l_, e_))
ends = np.array([max(x[1] for x in v) for v in per_cu.values()])
cnt = np.array([len(v) for v in per_cu.values()])
print(f"  CUs seen {len(per_cu)}; waves per CU min {cnt.min()} max {cnt.max()}; last end per CU: min {ends.min():.1f} p50 {np.median(ends):.1f} max {ends.max():.1f} us")
eng.close()
