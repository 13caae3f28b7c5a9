#!/usr/bin/env python3
"""Throughput of W host workers (threads), each with its own context and HIP stream, lifting the same resident batch in turn:
the per-GPU arrangement of INTEGRATION.md (one plo_ctx per rayon worker).  GPU only."""
import argparse
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, devbatch, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="wgs30x")
ap.add_argument("--reads", type=int, default=2000000)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--workers", default="1,2,3,4")
args = ap.parse_args()
dev = torch.device("cuda", 0)
w = synth.generate(synth.config(args.workload, n_reads=args.reads), device=dev)
index = api.Index(w.index_data_device(), 0)
db = devbatch.DeviceBatch.from_workload(w)
desc = db.desc()
for W in [int(x) for x in args.workers.split(",")]:
    streams = [torch.cuda.Stream(device=dev) for _ in range(W)]
    engs = [api.Engine(index, stream=s.cuda_stream) for s in streams]
    for e in engs:
        e.liftover_batch_dev(desc)
    torch.cuda.synchronize()

    def run(k):
        for _ in range(k, args.steps, W):
            engs[k].liftover_batch_dev(desc)

    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(k,)) for k in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = engs[0].timing()
    print(f"workers {W}: {dt / args.steps * 1e3:.3f} ms per batch, {w.n_reads * args.steps / dt / 1e6:.1f} M reads/s  (tile kernel {tm.lift_ms:.3f} ms, enumerate {tm.enumerate_ms:.3f} ms)", flush=True)
    for e in engs:
        e.close()
