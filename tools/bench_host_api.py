#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (plo_liftover_batch: H2D of the batch, kernels, D2H of the
results).  Not the bench value (bench.py times device-resident inputs); recorded in DESIGN.md."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, synth  # noqa: E402

w = synth.generate(synth.config("chr20"), device="cuda")
index = api.Index(w.index_data_device(), 0)
eng = api.Engine(index)
b = w.batch_data()
eng.liftover_batch(b)
t0 = time.perf_counter()
n = 5
for _ in range(n):
    res = eng.liftover_batch(b)
dt = (time.perf_counter() - t0) / n
nbytes = sum(getattr(b, f).nbytes for f in ("read_is_reverse", "read_seq_len", "read_seq_off", "seq", "seg_read", "seg_contig", "seg_pos",
                                             "seg_is_fwd_strand", "seg_cigar_off", "cigar"))
print(json.dumps({"workload": "chr20", "host_memory": "pageable", "reads": b.n_reads, "ms_per_batch": dt * 1e3, "reads_per_s": b.n_reads / dt,
                  "h2d_bytes": nbytes, "h2d_GBps_equiv": nbytes / dt / 1e9, "items": res.n_items}))

# the same batch in page-locked buffers from plo_host_alloc (what a worker thread would fill in place)
import dataclasses  # noqa: E402

import numpy as np  # noqa: E402

pins = {}
fields = {}
for f in dataclasses.fields(b):
    v = getattr(b, f.name)
    if isinstance(v, np.ndarray) and v.nbytes >= 4096:
        pins[f.name] = api.PinnedArray(v)
        fields[f.name] = pins[f.name].array
    else:
        fields[f.name] = v
bp = type(b)(**fields)
eng.liftover_batch(bp)
t0 = time.perf_counter()
for _ in range(n):
    res2 = eng.liftover_batch(bp)
dt2 = (time.perf_counter() - t0) / n
assert res2.canonical() == res.canonical()
print(json.dumps({"workload": "chr20", "host_memory": "page-locked (plo_host_alloc)", "reads": b.n_reads, "ms_per_batch": dt2 * 1e3,
                  "reads_per_s": b.n_reads / dt2, "h2d_bytes": nbytes, "h2d_GBps_equiv": nbytes / dt2 / 1e9, "items": res2.n_items}))
t = eng.timing()
print(json.dumps({"device_ms": t.total_ms, "enumerate_ms": t.enumerate_ms, "tiles_ms": t.lift_ms}))
