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
    for threads, chunk in ((16, 0),) if os.environ.get("PLO_DEBUG_INFLATE") else ((8, 0), (16, 0), (16, 1024)):
        e = dict(os.environ, PLO_DEBUG_READER="1")
        if chunk:
            e["PLO_BGZF_CHUNK_MB"] = str(chunk)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", inp, str(threads)], env=e, stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=300)
        runs = r.stderr.split("RUN ")
        last = runs[-2] if len(runs) >= 2 else r.stderr  # the lines in front of the last "RUN": the second pass
        f = w_ = c = 0.0
        for m in re.finditer(r"inflate \(fill\) ([0-9.]+) s, record walk ([0-9.]+) s, copy ([0-9.]+) s", last):
            f += float(m.group(1)); w_ += float(m.group(2)); c += float(m.group(3))
        print(f"{threads} threads, refills of {chunk or 256} MB: {[x.splitlines()[0] for x in runs[1:]]}; second pass: inflate (fill) {f:.3f} s, record walk {w_:.3f} s, window copy {c:.3f} s", flush=True)
        other = [ln for ln in last.splitlines() if "[plo]" in ln and "read_window" not in ln]
        print("\n".join(other[:12]), flush=True)
        acc = [0.0] * 7
        nf = 0
        for m in re.finditer(r"refill: \d+ blocks in \d+ groups, ([0-9.]+) s: tail move ([0-9.]+), header walk ([0-9.]+), buffer ([0-9.]+), staging ([0-9.]+), waits ([0-9.]+), rest (-?[0-9.]+)", last):
            for q in range(7):
                acc[q] += float(m.group(q + 1))
            nf += 1
        if nf:
            print(f"{nf} refills, {acc[0]:.3f} s: tail move {acc[1]:.3f}, header walk {acc[2]:.3f}, buffer {acc[3]:.3f}, staging {acc[4]:.3f}, waits {acc[5]:.3f}, rest {acc[6]:.3f}", flush=True)
        h = k = dd = 0.0
        n = 0
        for m in re.finditer(r"H2D ([0-9.]+) ms, kernel ([0-9.]+) ms, D2H ([0-9.]+) ms", last):
            h += float(m.group(1)); k += float(m.group(2)); dd += float(m.group(3)); n += 1
        if n:
            print(f"device inflate, {n} groups: H2D {h:.1f} ms, kernels {k:.1f} ms, D2H {dd:.1f} ms in all (per group {h / n:.2f} / {k / n:.2f} / {dd / n:.2f})", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
