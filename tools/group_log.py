#!/usr/bin/env python3
"""Timing build only: one record per group of the lane kernel (ticks, liftover trips, shift rounds, scan iterations, LDS rounds, the wave) ->
what makes a group slow?  GPU only."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, devbatch, synth  # noqa: E402

L = api.load_library(os.path.join(ROOT, "portello_amd", "libportello_liftover_timing.so"))
L.plo_ctx_wave_clocks.restype = C.c_uint
L.plo_ctx_wave_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint]
dev = torch.device("cuda", 0)
w = synth.generate(synth.config("wgs30x", n_reads=int(os.environ.get("READS", "2000000"))), device=dev)
index = api.Index(w.index_data_device(), 0)
db = devbatch.DeviceBatch.from_workload(w)
desc = db.desc()
eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
for i in range(3):
    eng.liftover_batch_dev(desc, 31)
t = eng.timing()
n = 65536
buf = (C.c_ulonglong * (16 * n))()
got = L.plo_ctx_wave_clocks(eng.handle, buf, n)
a = np.frombuffer(buf, dtype=np.uint64)[8192 * 16:].reshape(-1, 4)
ng = (t.n_items + 63) // 64 + 2
a = a[:ng]
a = a[a[:, 0] > 0]
dt = (a[:, 0] & 0xffffffff).astype(np.float64) / 100
beg = (a[:, 0] >> 32).astype(np.float64) / 100
lt = (a[:, 1] & 0xffffffff).astype(np.int64)
sr = (a[:, 1] >> 32).astype(np.int64)
sc = (a[:, 2] & 0xffffffff).astype(np.int64)
rd = (a[:, 2] >> 32).astype(np.int64)
nit = ((a[:, 3] >> 32) & 0xfff).astype(np.int64)
rev = sr > 0
print(f"lanes {t.lanes_ms:.3f} ms; groups {len(a)}; mean {dt.mean():.1f} us; forward {dt[~rev].mean():.1f} us ({(~rev).sum()}), with the shift stage {dt[rev].mean():.1f} us ({rev.sum()})")
for name, m in (("forward", ~rev), ("shift", rev)):
    d = dt[m]
    print(f"  {name}: p1 {np.percentile(d, 1):.0f} p10 {np.percentile(d, 10):.0f} p50 {np.median(d):.0f} p90 {np.percentile(d, 90):.0f} p99 {np.percentile(d, 99):.0f} max {d.max():.0f} us; liftover trips p50 {np.median(lt[m]):.0f} p99 {np.percentile(lt[m], 99):.0f} max {lt[m].max()}; "
          f"shift rounds p50 {np.median(sr[m]):.0f} p99 {np.percentile(sr[m], 99):.0f} max {sr[m].max()}; LDS rounds > 1: {(rd[m] > 1).sum()}")
# regression of duration on the trip counts
X = np.stack([np.ones(len(a)), lt, sr, sc, rd], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(X, dt, rcond=None)
pred = X @ coef
print("  least squares: us = %.1f + %.2f x liftover trips + %.2f x shift rounds + %.2f x scan iterations + %.1f x LDS rounds;  residual sd %.1f us" % (*coef, (dt - pred).std()))
slow = np.argsort(dt)[-8:]
for i in slow:
    print(f"  slow group: {dt[i]:.0f} us (begun at {beg[i]:.0f}), items {nit[i]}, liftover trips {lt[i]}, shift rounds {sr[i]}, scan iterations {sc[i]}, LDS rounds {rd[i]}, predicted {pred[i]:.0f}")
# time dependence: mean duration of groups by the time they began
for lo_ in range(0, 1300, 100):
    m = (beg >= lo_) & (beg < lo_ + 100)
    if m.sum():
        print(f"  begun in [{lo_}, {lo_ + 100}) us: {m.sum():5d} groups, mean {dt[m].mean():6.1f} us, residual {(dt[m] - pred[m]).mean():6.1f}")
eng.close()
