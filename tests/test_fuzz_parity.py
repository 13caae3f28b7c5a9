"""Adversarial random batches (tests/fuzz_cases.py): every op code, zero-length ops, edge indels, tiny alphabets,
inconsistent lengths, out-of-range positions.  CPU: device algorithm under the emulator vs oracle.  GPU: HIP vs oracle."""
import numpy as np
import pytest

import emu_lib
import fuzz_cases
from portello_amd import abi, api
from portello_amd import cigar as cg

STAGE_SETS = (31, 15, 5, 2, 16, 7, 27)


def _diff(ref, got, b, tag):
    a, c = ref.canonical(), got.canonical()
    assert len(a) == len(c), tag
    for x, y in zip(a, c):
        if x != y:
            s = x[0]
            raise AssertionError(f"{tag}: seg {s} in {cg.decode(b.cigar[b.seg_cigar_off[s]:b.seg_cigar_off[s + 1]])} pos {b.seg_pos[s]}\n"
                                 f" ref {x[:7]} {cg.decode(np.frombuffer(x[7], np.uint32))}\n got {y[:7]} {cg.decode(np.frombuffer(y[7], np.uint32))}")


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_emulated_device_algorithm(oracle, seed, monkeypatch):
    for alpha in (b"ACGT", b"AC", b"A"):
        ix, b = fuzz_cases.make(100 + seed, alphabet=alpha, explicit=(seed % 3 == 0),
                                seq_fmt=(abi.SEQ_BAM4 if seed % 4 == 1 else abi.SEQ_ASCII))
        for stages in STAGE_SETS:
            ref = oracle.liftover_batch(ix, b, stages, 1)
            rc, got, _ = emu_lib.liftover_batch(ix, b, stages=stages, cap=256, window=48, big_thresh=10, big_cap=4096)
            assert rc == 0
            _diff(ref, got, b, f"seed {seed} alpha {alpha} stages {stages}")
            # the same through the lane-per-item path (lane_core.hpp); what it hands on takes the path above
            rc, got, cnt = emu_lib.liftover_batch(ix, b, stages=stages, cap=256, window=48, big_thresh=10, big_cap=4096, lane_max_w=60,
                                                  lane_capw=1024 if seed % 2 else 160)
            assert rc == 0 and cnt[23] > 0
            _diff(ref, got, b, f"lane path: seed {seed} alpha {alpha} stages {stages}")
            if stages & abi.STAGE_LIFTOVER:  # ... with 16-bit ops in the regions (the experiment kernel k_lift_lanes16)
                monkeypatch.setenv("PLO_EMU_H16", "1")
                rc, got, cnt = emu_lib.liftover_batch(ix, b, stages=stages, cap=256, window=48, big_thresh=10, big_cap=4096, lane_max_w=60,
                                                      lane_capw=1024 if seed % 2 else 160, order_seed=(0, 21 + seed)[seed % 2])
                monkeypatch.delenv("PLO_EMU_H16")
                assert rc == 0 and cnt[23] > 0
                _diff(ref, got, b, f"lane path, 16-bit regions: seed {seed} alpha {alpha} stages {stages}")
            # ... and with nearly every item in the heavy classes: regions in global scratch behind LDS windows (k_lift_lanes_g)
            rc, got, cnt = emu_lib.liftover_batch(ix, b, stages=stages, cap=256, window=48, big_thresh=10, big_cap=4096, lane_max_w=12,
                                                  lane_capw=1024, lane_heavy_per=(64, 5)[seed % 2])
            assert rc == 0 and cnt[23] < got.n_items
            _diff(ref, got, b, f"heavy lane path: seed {seed} alpha {alpha} stages {stages}")
            if stages == abi.STAGES_ALL:
                # ... and through the streaming kernel (lane_stream.hpp): teams of three emulated waves, the stages chained through LDS rings
                # (2: with the smallest Q2 the code allows -- unreleased tails outgrow it and the items take the retry list), waves drifting
                for mode, oseed in ((1, 0), (2, 0), (1, 7 + seed)):
                    rc, got, cnt = emu_lib.liftover_batch(ix, b, stages=stages, cap=256, window=48, big_thresh=10, big_cap=4096, lane_max_w=12,
                                                          lane_capw=1024, lane_heavy_per=(64, 5)[seed % 2], lane_stream=mode, order_seed=oseed)
                    assert rc == 0 and cnt[23] < got.n_items
                    _diff(ref, got, b, f"streaming kernel (rings {mode}, order seed {oseed}): seed {seed} alpha {alpha}")


@pytest.mark.gpu
@pytest.mark.parametrize("h16", ["0", "1"])
def test_fuzz_hip(oracle, h16, monkeypatch):
    monkeypatch.setenv("PLO_LANE_H16", h16)  # (1: the light items of stage sets with the liftover through k_lift_lanes16)
    for seed in range(40):
        alpha = (b"ACGT", b"AC", b"A")[seed % 3]
        ix, b = fuzz_cases.make(1000 + seed, alphabet=alpha, n_reads=120, explicit=(seed % 3 == 0),
                                seq_fmt=(abi.SEQ_BAM4 if seed % 4 == 1 else abi.SEQ_ASCII))
        index = api.Index(ix)
        eng = api.Engine(index)
        for stages in STAGE_SETS:
            _diff(oracle.liftover_batch(ix, b, stages, 1), eng.liftover_batch(b, stages), b, f"seed {seed} stages {stages}")
        eng.close()
        index.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["g", "g_w3", "stream"])
@pytest.mark.parametrize("per", ["64", "5"])
def test_fuzz_hip_heavy_lane_kernel(oracle, monkeypatch, per, variant):
    """VERDICT r3 (weak #1), r4 (next #6): the 40 adversarial seeds x 7 stage sets through EVERY instantiation of the heavy-item lane kernel
    ON THE GPU (k_lift_lanes_g at two and at three waves per SIMD; the streaming kernel k_lift_stream, which takes the batches whose stage
    set it covers and leaves the others to k_lift_lanes_g) -- every item above a tiny weight is routed to the heavy classes
    (PLO_LANE_MAX_W), the lane-per-item code takes them whatever their number (PLO_LANE_HEAVY_MIN=0), 64 or 5 items per wave"""
    from variants import HEAVY_VARIANTS

    for k, v in HEAVY_VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PLO_LANE_HEAVY_MIN", "0")
    monkeypatch.setenv("PLO_LANE_MAX_W", "12")
    monkeypatch.setenv("PLO_LANE_HEAVY_PER", per)
    heavy = 0
    for seed in range(40):
        alpha = (b"ACGT", b"AC", b"A")[seed % 3]
        ix, b = fuzz_cases.make(1000 + seed, alphabet=alpha, n_reads=120, explicit=(seed % 3 == 0),
                                seq_fmt=(abi.SEQ_BAM4 if seed % 4 == 1 else abi.SEQ_ASCII))
        index = api.Index(ix)
        eng = api.Engine(index)
        for stages in STAGE_SETS:
            got = eng.liftover_batch(b, stages)
            heavy += int(eng.timing().n_heavy_lane_items)
            _diff(oracle.liftover_batch(ix, b, stages, 1), got, b, f"heavy lanes {variant} ({per} per wave): seed {seed} stages {stages}")
        eng.close()
        index.close()
    assert heavy > 40 * len(STAGE_SETS) * 20  # (most items of every batch went through the heavy-lane kernel)
