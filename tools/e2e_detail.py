#!/usr/bin/env python3
"""BAM -> BAM with the records finished on the device: where the lift stage's host time goes (upload / liftover / compact / finish / SA text /
download), on samples of two sizes (a pipeline of five stages over 12 windows is partly ramp).  GPU only.
usage: tools/e2e_detail.py [reads ...]"""
import os
import shutil
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, bamsynth, pipeline, synth  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [60000, 240000]
dev = torch.device("cuda", 0)
w = synth.generate(synth.config("wgs30x", n_reads=400000), device=dev)
index = api.Index(w.index_data_device(), 0)
ixd = w.index_data()
for n in sizes:
    d = tempfile.mkdtemp(prefix="plo_e2e_")
    try:
        inp = os.path.join(d, "reads.bam")
        lo = (w.n_reads - n) // 2
        meta = bamsynth.write_read_bam(w, inp, lo, lo + n, level=1, n_threads=16)
        cn, rn = meta["contig_names"], bamsynth.ref_names(w)
        rl = [int(s.numel()) for s in w.chrom_seq]
        pipeline.run_bam_to_bam(inp, os.path.join(d, "o.bam"), index, ixd, cn, rn, rl, window_reads=2000, n_workers=1, device_finish=True)
        for kw in (dict(), dict(n_workers=3), dict(window_reads=15000), dict(n_workers=1)):
            a = dict(window_reads=7500, n_workers=2, io_threads=16)
            a.update(kw)
            best = None
            for _ in range(3):
                st = pipeline.run_bam_to_bam(inp, os.path.join(d, "o.bam"), index, ixd, cn, rn, rl, device_finish=True, **a)
                if best is None or st.seconds < best.seconds:
                    best = st
            det = ", ".join(f"{k} {v:.3f}" for k, v in best.lift_detail_s.items())
            print(f"reads={n} {kw}: {best.reads / best.seconds / 1e3:.1f} k reads/s ({best.seconds:.3f} s; busy: read {best.read_s:.2f} batch {best.batch_s:.2f} "
                  f"lift {best.lift_s:.2f} build {best.build_s:.2f} write {best.write_s:.2f}; device lift {best.device_ms / 1e3:.3f} finish {best.finish_device_ms / 1e3:.3f})\n"
                  f"    lift stage: {det}", flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
