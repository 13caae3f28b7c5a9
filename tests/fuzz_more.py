#!/usr/bin/env python3
"""More of tests/test_fuzz_parity.py's adversarial batches than the suite runs: seeds [a, b) x three alphabets x seven stage sets x three routings
(tile path, lane-per-item path, heavy-lane path) of the device algorithm under the CPU emulator against the oracle.  CPU only;
usage: python tests/fuzz_more.py 0 340   (about 1.5 s per seed).  TEST INFRASTRUCTURE."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import emu_lib, fuzz_cases
from portello_amd import abi
from oracle import pyoracle
import test_fuzz_parity as T
pyoracle.build()
orc = pyoracle
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    for alpha in (b"ACGT", b"AC", b"A"):
        ix, b = fuzz_cases.make(5000 + seed, alphabet=alpha, explicit=(seed % 3 == 0), seq_fmt=(abi.SEQ_BAM4 if seed % 4 == 1 else abi.SEQ_ASCII))
        for stages in T.STAGE_SETS:
            ref = orc.liftover_batch(ix, b, stages, 1)
            variants = [dict(), dict(lane_max_w=60, lane_capw=1024 if seed % 2 else 160), dict(lane_max_w=12, lane_capw=1024, lane_heavy_per=(64, 5)[seed % 2]),
                        # round 6: groups dealt by tickets behind 1-3 fixed rounds, cheap groups last (order seeds), ...
                        dict(lane_max_w=60, lane_capw=1024, order_seed=17 + seed)]
            if stages & abi.STAGE_LIFTOVER:  # ... 16-bit regions (k_lift_lanes16) ...
                variants.append(dict(lane_max_w=60, lane_capw=512 if seed % 2 else 160, order_seed=(0, 40 + seed)[seed % 2], h16=True))
            if stages == abi.STAGES_ALL:  # ... and the streaming kernel with its PIPE_PAIR markers and copied-through stretches, waves drifting
                variants += [dict(lane_max_w=12, lane_capw=1024, lane_heavy_per=(64, 5)[seed % 2], lane_stream=1, order_seed=(0, 7 + seed)[seed % 2]),
                             dict(lane_max_w=12, lane_capw=1024, lane_heavy_per=64, lane_stream=2, order_seed=3 + seed)]
            for kw in variants:
                kw = dict(kw)
                if kw.pop("h16", False):
                    os.environ["PLO_EMU_H16"] = "1"
                else:
                    os.environ.pop("PLO_EMU_H16", None)
                rc, got, _ = emu_lib.liftover_batch(ix, b, stages=stages, cap=256, window=48, big_thresh=10, big_cap=4096, **kw)
                try:
                    assert rc == 0
                    T._diff(ref, got, b, f"seed {seed} alpha {alpha} stages {stages} {kw}")
                except AssertionError as e:
                    bad += 1
                    print("FAIL", e, flush=True)
    print("seed", seed, "done", flush=True)
print("bad", bad)
