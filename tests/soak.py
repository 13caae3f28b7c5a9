#!/usr/bin/env python3
"""Full-result parity soak on the GPU: every item of several larger workloads (different seeds, strand mixes and contig
block-map densities, hence different tile geometries) against the oracle.  GPU only; prints one line per case."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from portello_amd import abi, api, devbatch, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=200000)
ap.add_argument("--cases", default="1e-4:0.5:1,1e-3:0.5:2,2e-3:0.3:3,5e-4:1.0:4,3e-3:0.5:5")
ap.add_argument("--workload", default="chr20", help="chr20 (default) or stress_small: the heavy items of the latter all take the "
                                                     "workgroup-per-item kernel (vary PLO_MID_WAVES / PLO_MID_CAP in the environment)")
args = ap.parse_args()
pyoracle.build()
dev = torch.device("cuda", 0)
os.environ["PLO_DEBUG_GEOMETRY"] = "1"
bad = 0
for case in args.cases.split(","):
    rate, rev, seed = case.split(":")
    cr = synth.EditRates(mismatch=1e-3, ins=float(rate), dele=float(rate), hpol_frac=0.3, big_indel_prob=0.02)
    cfg = synth.config(args.workload, n_reads=args.reads, rev_contig_frac=float(rev), contig_rates=cr, seed=synth.config(args.workload).seed + int(seed))
    w = synth.generate(cfg, device=dev)
    index = api.Index(w.index_data_device(), 0)
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    db = devbatch.DeviceBatch.from_workload(w)
    torch.cuda.synchronize()  # torch's default stream is the NULL handle, for which the engine creates a private stream: the batch's
    # tensors must be complete before the engine's stream reads them
    out = eng.liftover_batch_dev(db.desc())
    t = eng.timing()
    got = devbatch.download(eng, out)
    t0 = time.perf_counter()
    ref = pyoracle.liftover_batch(w.index_data(), w.batch_data(), abi.STAGES_ALL, os.cpu_count() or 8)
    dt = time.perf_counter() - t0
    a, b = got.canonical(), ref.canonical()
    same = a == b
    bad += 0 if same else 1
    print(f"case contig-indel {rate} rev {rev} seed {seed}: {t.n_items} items, {t.n_mid_items} workgroup-per-item ({t.mid_ms:.2f} ms), {t.n_big_items} large, "
          f"{t.n_retry_items} retried, tiles {t.lift_ms:.3f} ms, oracle {dt:.1f} s -> {'IDENTICAL' if same else 'MISMATCH'}", flush=True)
    eng.close()
    index.close()
sys.exit(1 if bad else 0)
