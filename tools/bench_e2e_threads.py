#!/usr/bin/env python3
"""BAM -> BAM on a 60 k-read sample of the wgs30x workload with different thread shares of the pipeline's stages.  GPU only.
usage: tools/bench_e2e_threads.py [reads]"""
import os
import shutil
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, bamsynth, pipeline, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
dev = torch.device("cuda", 0)
w = synth.generate(synth.config("wgs30x", n_reads=400000), device=dev)
index = api.Index(w.index_data_device(), 0)
ixd = w.index_data()
d = tempfile.mkdtemp(prefix="plo_e2e_")
try:
    inp = os.path.join(d, "reads.bam")
    lo = (w.n_reads - n) // 2
    meta = bamsynth.write_read_bam(w, inp, lo, lo + n, level=1, n_threads=16)
    cn, rn = meta["contig_names"], bamsynth.ref_names(w)
    rl = [int(s.numel()) for s in w.chrom_seq]
    pipeline.run_bam_to_bam(inp, os.path.join(d, "o.bam"), index, ixd, cn, rn, rl, window_reads=2000, n_workers=1)
    combos = [dict(), dict(read_threads=8, build_threads=4, write_threads=6), dict(read_threads=8, build_threads=8, write_threads=8),
              dict(window_reads=5000), dict(window_reads=3750), dict(window_reads=2500), dict(window_reads=3750, n_workers=3),
              dict(window_reads=3750, read_threads=8, build_threads=8, write_threads=8)]
    for dfin in (True, False):
        for c in combos:
            kw = dict(window_reads=7500, n_workers=2, io_threads=16)
            kw.update(c)
            best = None
            for _ in range(3):
                st = pipeline.run_bam_to_bam(inp, os.path.join(d, "o.bam"), index, ixd, cn, rn, rl, device_finish=dfin, **kw)
                if best is None or st.seconds < best.seconds:
                    best = st
            print(f"device_finish={dfin} {c}: {best.reads / best.seconds / 1e3:.1f} k reads/s  ({best.seconds:.3f} s; busy: read {best.read_s:.2f} batch {best.batch_s:.2f} "
                  f"lift {best.lift_s:.2f} build {best.build_s:.2f} write {best.write_s:.2f})", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
