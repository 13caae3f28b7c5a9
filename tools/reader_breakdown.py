#!/usr/bin/env python3
"""The BAM reader alone over one sample with PLO_DEBUG_READER=1: inflate (fill) / record walk / window copy, summed over the windows.  GPU only.
usage: tools/reader_breakdown.py [reads]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    from portello_amd import bam
    inp, threads = sys.argv[2], int(sys.argv[3])
    for _ in range(2):
        rd = bam.BamReader(inp, threads, device_inflate=0)
        t0 = time.perf_counter()
        k = 0
        while True:
            win = rd.read_window(7500)
            if win is None:
                break
            k += win.n_records
            win.close()
        t = time.perf_counter() - t0
        rd.close()
        print(f"RUN {k} reads in {t:.3f} s", file=sys.stderr, flush=True)
    sys.exit(0)

import torch  # noqa: E402

from portello_amd import bamsynth, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 240000
w = synth.generate(synth.config("wgs30x", n_reads=400000), device=torch.device("cuda", 0))
d = tempfile.mkdtemp(prefix="plo_rd_")
try:
    inp = os.path.join(d, "reads.bam")
    lo = (w.n_reads - n) // 2
    bamsynth.write_read_bam(w, inp, lo, lo + n, level=1, n_threads=16)
    for threads in (8, 16):
        e = dict(os.environ, PLO_DEBUG_READER="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", inp, str(threads)], env=e, stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=300)
        runs = r.stderr.split("RUN ")
        last = runs[-2] if len(runs) >= 2 else r.stderr  # the lines in front of the last "RUN": the second pass
        f = w_ = c = 0.0
        for m in re.finditer(r"inflate \(fill\) ([0-9.]+) s, record walk ([0-9.]+) s, copy ([0-9.]+) s", last):
            f += float(m.group(1)); w_ += float(m.group(2)); c += float(m.group(3))
        print(f"{threads} threads: {[x.splitlines()[0] for x in runs[1:]]}; second pass: inflate (fill) {f:.3f} s, record walk {w_:.3f} s, window copy {c:.3f} s", flush=True)
        other = [ln for ln in last.splitlines() if "[plo]" in ln and "read_window" not in ln]
        print("\n".join(other[:12]), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
