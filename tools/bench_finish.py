#!/usr/bin/env python3
"""Times the record-finishing stage (plo_finish_batch_dev) on a synthetic workload: the scalar kernels and the
HBM-streaming reverse-complement kernel (k_revcomp), with its achieved fraction of the HBM roofline."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import abi, api, devbatch, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="wgs30x")
ap.add_argument("--reads", type=int, default=400000)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
w = synth.generate(synth.config(args.workload, n_reads=args.reads), device=dev)
index = api.Index(w.index_data_device(), 0)
eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
db = devbatch.DeviceBatch.from_workload(w)
desc = db.desc()
fin, keep = devbatch.finish_inputs(w, db)
torch.cuda.synchronize()
out = eng.liftover_batch_dev(desc)
fm, rm = [], []
for i in range(args.steps + 1):
    fo = eng.finish_batch_dev(desc, fin)
    if i:
        fm.append(fo.finish_ms)
        rm.append(fo.revcomp_ms)
# algorithmic bytes of the reversal: every flipped record reads and writes its packed bases and its qualities once
iso = eng.download(fo.item_seq_off, np.uint64, int(out.n_items))
rso = eng.download(fo.read_seq_off, np.uint64, db.n_reads)
item_seg = eng.download(out.item_seg, np.uint32, int(out.n_items))
seg_read = db.seg_read.cpu().numpy()
lens = db.read_seq_len.cpu().numpy().astype(np.int64)
L = np.concatenate([lens[seg_read[item_seg[iso != abi.NO_FLIP]]], lens[rso != abi.NO_FLIP]])
nbytes = int((2 * ((L + 1) // 2 + L)).sum())
ms = float(np.mean(rm))
print(json.dumps({"kernel": "k_revcomp", "flipped_records": int(len(L)), "algorithmic_bytes": nbytes, "kernel_ms": ms,
                  "achieved_GBps": nbytes / (ms * 1e-3) / 1e9, "frac_of_8TBps": nbytes / (ms * 1e-3) / 8e12,
                  "finish_scalar_ms": float(np.mean(fm)), "items": int(out.n_items), "reads": db.n_reads}))
