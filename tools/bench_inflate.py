#!/usr/bin/env python3
"""Reader throughput on a synthetic read->contig BAM: host inflate vs device inflate (PLO_BGZF_DEVICE).  GPU only."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from portello_amd import bam, bamsynth, pipeline, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=40000)
ap.add_argument("--level", type=int, default=1)
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--lib", default="", help="alternative library file name under portello_amd/")
args = ap.parse_args()
if args.lib:
    from portello_amd import api
    api.load_library(os.path.join(ROOT, "portello_amd", args.lib))
w = synth.generate(synth.config("wgs30x", n_reads=args.reads), device=torch.device("cuda", 0))
p = os.path.join(args.dir, "plo_inflate_bench.bam")
th = max(2, min(64, pipeline.effective_cpus()))
bamsynth.write_read_bam(w, p, 0, w.n_reads, level=args.level, n_threads=th)
size = os.path.getsize(p)
try:
    for dev in ("0", "1", "0", "1"):
        os.environ["PLO_BGZF_DEVICE"] = dev
        os.environ["PLO_DEBUG_READER"] = "1"
        rd = bam.BamReader(p, th)
        t0 = time.perf_counter()
        n = 0
        while True:
            win = rd.read_window(20000)
            if win is None:
                break
            n += win.n_records
            win.close()
        dt = time.perf_counter() - t0
        rd.close()
        print(f"PLO_BGZF_DEVICE={dev}: {n} records, {size / 1e6:.0f} MB compressed in {dt:.3f} s = {size / dt / 1e9:.2f} GB/s compressed, {n / dt / 1e3:.0f} k reads/s ({th} host threads)", flush=True)
finally:
    os.remove(p)
