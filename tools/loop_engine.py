#!/usr/bin/env python3
"""Runs the tile kernel repeatedly on one resident batch (for rocprofv3 counter passes; PC sampling is not supported on this ROCm 7.2 / gfx950 stack).  GPU only."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, devbatch, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="wgs30x")
ap.add_argument("--reads", type=int, default=400000)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--lib", default="")
args = ap.parse_args()
if args.lib:
    api.load_library(os.path.join(ROOT, "portello_amd", args.lib))
dev = torch.device("cuda", 0)
w = synth.generate(synth.config(args.workload, n_reads=args.reads), device=dev)
index = api.Index(w.index_data_device(), 0)
db = devbatch.DeviceBatch.from_workload(w)
desc = db.desc()
eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
for _ in range(args.steps):
    eng.liftover_batch_dev(desc, 31)
t = eng.timing()
print(f"tiles {t.lift_ms:.3f} ms, items {t.n_items}")
