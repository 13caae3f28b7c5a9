#!/usr/bin/env python3
"""More of tests/test_bam.py's check of the BAM reader's device-path bookkeeping (groups of blocks, three staging buffers, refill preparation) than
the suite runs: random files x random window sizes, part counts, group sizes, refill sizes, thread counts, with zlib standing in for the
device (PLO_BGZF_TEST_HOST_SLOTS), against the plain host inflate.  CPU only; usage: python tests/fuzz_reader.py <seed>   (225 cases, ~2 min).
Round 5: seeds 1-3 clean on the final sources.  TEST INFRASTRUCTURE."""
import hashlib
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from portello_amd import bam, bamsynth, synth
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
d = tempfile.mkdtemp()
def read_all(path, device, part=None, n_parts=1, window=157, threads=3):
    rd = bam.BamReader(path, threads, device_inflate=device, part=part, n_parts=n_parts)
    out = []
    while True:
        win = rd.read_window(window)
        if win is None: break
        b = win.batch_data()
        h = hashlib.sha1()
        for a in (b.seq, b.cigar, b.seg_pos, b.read_seq_len): h.update(a.tobytes())
        out.append((b.n_reads, h.hexdigest(), win.unmapped_bytes()))
        win.close()
    rd.close()
    return out
bad = 0; n = 0
for fi in range(3):
    w = synth.generate(synth.config("tiny", n_reads=random.choice([300, 900, 2500]), seed=500 + fi, split_read_frac=0.2, sorted_reads=True))
    for level in (0, 1, 9):
        path = os.path.join(d, f"f{fi}_{level}.bam")
        bamsynth.write_read_bam(w, path, level=level)
        for it in range(25):
            window = random.choice([1, 7, 50, 157, 1000, 100000])
            n_parts = random.choice([1, 1, 2, 3, 7, 40])
            for k in ("PLO_BGZF_TEST_HOST_SLOTS", "PLO_BGZF_TEST_CHUNK_BYTES", "PLO_BGZF_NO_PREFETCH"): os.environ.pop(k, None)
            want = [read_all(path, -1, k if n_parts > 1 else None, n_parts, window) for k in range(n_parts)]
            os.environ["PLO_BGZF_TEST_HOST_SLOTS"] = str(random.choice([1, 2, 3, 4, 5, 9, 64]))
            os.environ["PLO_BGZF_TEST_CHUNK_BYTES"] = str(random.choice([1, 1000, 65536, 70000, 131072, 300000, 2000000, 1 << 30]))
            if random.random() < 0.25: os.environ["PLO_BGZF_NO_PREFETCH"] = "1"
            got = [read_all(path, 0, k if n_parts > 1 else None, n_parts, window, threads=random.choice([1, 2, 5])) for k in range(n_parts)]
            n += 1
            if got != want:
                bad += 1
                print("MISMATCH", fi, level, window, n_parts, {k: os.environ.get(k) for k in ("PLO_BGZF_TEST_HOST_SLOTS", "PLO_BGZF_TEST_CHUNK_BYTES", "PLO_BGZF_NO_PREFETCH")}, flush=True)
print("cases", n, "bad", bad)
