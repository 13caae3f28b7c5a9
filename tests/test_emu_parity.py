"""The DEVICE algorithm (portello_amd/csrc/lift_core.hpp) executed on the CPU under the wave64 emulator
(tests/emu) and diffed against the oracle: golden vectors of the reference + seeded synthetic workloads.
This validates the wave-level decomposition (scans, segmented sums, min-plus carry, compaction, tiling, the
large-item path) bit for bit without a GPU; the GPU tests then only have to confirm the hardware mechanics."""
import numpy as np
import pytest

import emu_lib
from portello_amd import abi, api, synth
from portello_amd import cigar as cg


def emu_backend(**kw):
    def run(index, batch, stages):
        rc, res, _ = emu_lib.liftover_batch(index, batch, stages=stages, **kw)
        assert rc == 0
        return res
    return run


def oracle_backend(oracle):
    return lambda index, batch, stages: oracle.liftover_batch(index, batch, stages, 1)


def golden_cases(golden):
    lift, simp, shift = api.CaseSet(), api.CaseSet(), api.CaseSet()
    for v in golden["liftover"]:
        m = v["map"] or {"pos": 0, "cigar": ""}
        lift.add_liftover(m["pos"], cg.encode(m["cigar"]), v["start"], cg.encode(v["cigar"]))
    for v in golden["simplify"]:
        simp.add_simplify(v["pos"], cg.encode(v["cigar"]), v["ref"].encode(), v["read"].encode())
    left = [v for v in golden["shift"] if v["dir"] == "left"]
    for v in left:
        shift.add_left_shift(v["pos"], cg.encode(v["cigar"]), v["ref"].encode(), v["read"].encode())
    return lift, simp, shift, left


def check_golden(golden, backend):
    lift, simp, shift, left = golden_cases(golden)
    for v, r in zip(golden["liftover"], lift.run(abi.STAGE_LIFTOVER, backend)):
        if v["expect"] is None:
            assert r is None, v["id"]
        else:
            assert r is not None and r[0] == v["expect"]["pos"] and cg.decode(r[1]) == v["expect"]["cigar"], (v["id"], r)
    for v, r in zip(golden["simplify"], simp.run(abi.STAGE_SIMPLIFY, backend)):
        assert r[0] == v["expect"]["pos"] and cg.decode(r[1]) == v["expect"]["cigar"], (v["id"], r)
    for v, r in zip(left, shift.run(abi.STAGE_LSHIFT, backend)):
        assert r[0] == v["expect"]["pos"] and cg.decode(r[1]) == v["expect"]["cigar"], (v["id"], r)


def test_golden_vectors_oracle_batch_path(golden, oracle):
    """the batch/ABI form of the oracle agrees with the reference's vectors too (pins orc_liftover_batch's staging)"""
    check_golden(golden, oracle_backend(oracle))


def test_golden_vectors_emulated_device_algorithm(golden):
    check_golden(golden, emu_backend())


def _assert_same(ref: abi.BatchResult, got: abi.BatchResult):
    a, b = ref.canonical(), got.canonical()
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x == y, (x[:7], cg.decode(np.frombuffer(x[7], dtype=np.uint32)), y[:7], cg.decode(np.frombuffer(y[7], dtype=np.uint32)))


@pytest.mark.parametrize("stages", [abi.STAGES_ALL, abi.STAGES_ALL & ~abi.STAGE_SIMPLIFY, abi.STAGE_STRAND | abi.STAGE_LIFTOVER,
                                    abi.STAGE_LSHIFT, abi.STAGE_SIMPLIFY])
def test_synthetic_tiny(oracle, stages):
    w = synth.generate(synth.config("tiny", n_reads=60, split_read_frac=0.2, seed=101))
    ix, b = w.index_data(), w.batch_data()
    rc, res, _ = emu_lib.liftover_batch(ix, b, stages=stages)
    assert rc == 0
    _assert_same(oracle.liftover_batch(ix, b, stages, 1), res)


def test_synthetic_indel_dense_and_large_item_path(oracle):
    cfg = synth.config("tiny", n_reads=24, seed=102, read_len_mean=3000, read_len_sd=500,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=3e-3, dele=3e-3, hpol_frac=0.3, big_indel_prob=0.02))
    w = synth.generate(cfg)
    ix, b = w.index_data(), w.batch_data()
    rc, res, cnt = emu_lib.liftover_batch(ix, b)
    assert rc == 0 and cnt[2] > 0  # large-item kernel path exercised
    _assert_same(oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1), res)


def test_synthetic_small_capacity_chunk_boundaries(oracle):
    """tiny windows/capacities: multi-chunk carries, tile overflow -> re-queue, >64 items per window"""
    w = synth.generate(synth.config("tiny", n_reads=50, seed=103, read_len_mean=4000, read_len_sd=1000))
    ix, b = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1)
    for cap, window, thresh in ((200, 64, 6), (70, 200, 64), (4096, 4096, 4096)):
        rc, res, _ = emu_lib.liftover_batch(ix, b, cap=cap, window=window, big_thresh=thresh, big_cap=8192)
        assert rc == 0
        _assert_same(ref, res)


def test_lane_order_independence(oracle):
    """lanes executed in a shuffled order every round: a missing wave-level sync shows up as a diff"""
    w = synth.generate(synth.config("tiny", n_reads=40, seed=104, split_read_frac=0.2))
    ix, b = w.index_data(), w.batch_data()
    rc, res, _ = emu_lib.liftover_batch(ix, b, order_seed=12345)
    assert rc == 0
    _assert_same(oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1), res)


def test_comp_base_all_bytes(oracle):
    """the device's branch-free comp_base against the oracle's restatement of seq_util.rs:1-15, all 256 byte values"""
    import ctypes as C
    L = emu_lib.lib()
    L.emu_comp_base.restype = C.c_int
    L.emu_comp_base.argtypes = [C.c_int]
    O = oracle.lib()
    for b in range(256):
        assert L.emu_comp_base(b) == O.orc_comp_base(b), b


@pytest.mark.parametrize("mid_waves", [2, 4, 8, 16])
def test_workgroup_per_item_path(oracle, mid_waves):
    """items beyond the routing threshold run through lift_tile<NW> (several waves of one workgroup on one item, scan carries
    crossing the waves through the exchange words) under the multi-wave emulator, lane order shuffled every round; a small
    mid_cap sends some of them on to the one-wave path (LEVEL_MID -> huge list)"""
    cfg = synth.config("tiny", n_reads=16, seed=112, read_len_mean=3000, read_len_sd=800,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=3e-3, dele=3e-3, hpol_frac=0.3, big_indel_prob=0.02))
    w = synth.generate(cfg)
    ix, b = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1)
    for mid_cap, seed in ((2048, 0), (512, 991)):
        rc, res, cnt = emu_lib.liftover_batch(ix, b, big_thresh=100, mid_waves=mid_waves, mid_cap=mid_cap, order_seed=seed)
        assert rc == 0 and cnt[2] > 0
        if mid_cap == 2048:
            assert cnt[20] < cnt[2]  # most items finish in the workgroup path
        else:
            assert cnt[20] > 0  # some are handed on
        _assert_same(ref, res)


def test_workgroup_per_item_stage_subsets(oracle):
    w = synth.generate(synth.config("tiny", n_reads=12, seed=113, split_read_frac=0.3, read_len_mean=2500))
    ix, b = w.index_data(), w.batch_data()
    for stages in (abi.STAGES_ALL & ~abi.STAGE_SIMPLIFY, abi.STAGE_STRAND | abi.STAGE_LIFTOVER, abi.STAGE_LSHIFT, abi.STAGE_SIMPLIFY):
        rc, res, cnt = emu_lib.liftover_batch(ix, b, stages=stages, big_thresh=8, mid_waves=4, mid_cap=1024)
        assert rc == 0 and cnt[2] > 0
        _assert_same(oracle.liftover_batch(ix, b, stages, 1), res)


# ---- lane-per-item path (portello_amd/csrc/lane_core.hpp) ----------------------------------------------------------------------

def test_lane_path_golden_vectors(golden):
    check_golden(golden, emu_backend(lane_max_w=4096, lane_capw=8192))


@pytest.mark.parametrize("h16", ["0", "1"])
@pytest.mark.parametrize("stages", [abi.STAGES_ALL, abi.STAGES_ALL & ~abi.STAGE_SIMPLIFY, abi.STAGE_STRAND | abi.STAGE_LIFTOVER,
                                    abi.STAGE_LSHIFT, abi.STAGE_SIMPLIFY, abi.STAGE_STRAND | abi.STAGE_LSHIFT])
def test_lane_path_synthetic_tiny(oracle, stages, h16, monkeypatch):
    if h16 == "1" and not (stages & abi.STAGE_LIFTOVER):
        pytest.skip("16-bit regions: stage sets with the liftover only")
    monkeypatch.setenv("PLO_EMU_H16", h16)
    w = synth.generate(synth.config("tiny", n_reads=150, split_read_frac=0.2, seed=131))
    ix, b = w.index_data(), w.batch_data()
    rc, res, cnt = emu_lib.liftover_batch(ix, b, stages=stages, lane_max_w=400, lane_capw=3072)
    assert rc == 0 and cnt[7] == 0  # (no item handed on: a lane path that overflows everything would still give right results)
    _assert_same(oracle.liftover_batch(ix, b, stages, 1), res)


@pytest.mark.parametrize("h16", ["0", "1"])
def test_lane_path_small_slices_rounds_and_overflow(oracle, h16, monkeypatch):
    """slices too small for 64 regions (several rounds per group), for some items (-> retry list), shuffled lane order; a weight
    limit that splits the items between the lane path and the wave-cooperative one"""
    monkeypatch.setenv("PLO_EMU_H16", h16)
    w = synth.generate(synth.config("tiny", n_reads=120, seed=132, split_read_frac=0.2, read_len_mean=2500, read_len_sd=900))
    ix, b = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1)
    for max_w, capw, seed in ((4096, 600, 0), (4096, 96, 77), (40, 3072, 0), (4096, 3072, 4711), (4096, 3072, 6)):
        rc, res, cnt = emu_lib.liftover_batch(ix, b, lane_max_w=max_w, lane_capw=capw, order_seed=seed)
        assert rc == 0
        _assert_same(ref, res)


def test_lane_path_groups_cut_by_lds_budget(oracle):
    """the lane kernel's groups as k_chunk_sort lists them when it cuts a sorted window by LDS budget (lane_groups_cut): consecutive items
    while fewer than 64 and their regions fit the slice -- slices of 3072, 600 and 96 dwords (groups of 64, of a dozen, of one or two
    items; an item no slice holds still goes to the retry list), fixed order and shuffled lanes"""
    w = synth.generate(synth.config("tiny", n_reads=200, seed=135, split_read_frac=0.2, read_len_mean=2500, read_len_sd=900))
    ix, b = w.index_data(), w.batch_data()
    for stages in (abi.STAGES_ALL, abi.STAGE_STRAND | abi.STAGE_LIFTOVER):
        ref = oracle.liftover_batch(ix, b, stages, 1)
        for capw, seed in ((3072, 0), (600, 0), (96, 0), (600, 4712)):
            rc, res, cnt = emu_lib.liftover_batch(ix, b, stages=stages, lane_max_w=4096, lane_capw=capw, order_seed=seed, lane_budget=1)
            assert rc == 0
            _assert_same(ref, res)


def test_lane_path_indel_dense(oracle):
    cfg = synth.config("tiny", n_reads=48, seed=133, read_len_mean=1500, read_len_sd=300,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=3e-3, dele=3e-3, hpol_frac=0.3, big_indel_prob=0.02))
    w = synth.generate(cfg)
    ix, b = w.index_data(), w.batch_data()
    for stages in (abi.STAGES_ALL, abi.STAGE_SIMPLIFY, abi.STAGE_LSHIFT):
        rc, res, cnt = emu_lib.liftover_batch(ix, b, stages=stages, lane_max_w=100000, lane_capw=60000)
        assert rc == 0
        _assert_same(oracle.liftover_batch(ix, b, stages, 1), res)


@pytest.mark.parametrize("per", [64, 8, 3])
def test_lane_path_heavy_items_in_fixed_regions(oracle, per):
    """indel-dense items too heavy for an LDS region run through the same lane-per-item code with one fixed region per lane
    (k_lift_lanes_g: wave-private global scratch), `per` items per wave; light ones through the LDS path in the same batch"""
    cfg = synth.config("tiny", n_reads=40, seed=134, read_len_mean=2500, read_len_sd=900, split_read_frac=0.2,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=3e-3, dele=3e-3, hpol_frac=0.3, big_indel_prob=0.02))
    w = synth.generate(cfg)
    ix, b = w.index_data(), w.batch_data()
    for stages in (abi.STAGES_ALL, abi.STAGE_LSHIFT, abi.STAGE_STRAND | abi.STAGE_LIFTOVER):
        rc, res, cnt = emu_lib.liftover_batch(ix, b, stages=stages, lane_max_w=150, lane_capw=3072, lane_heavy_per=per, order_seed=per)
        assert rc == 0 and 0 < cnt[23] < res.n_items and cnt[2] == 0  # some light, some heavy, nothing through the tile kernels
        _assert_same(oracle.liftover_batch(ix, b, stages, 1), res)


def test_lane_path_heavy_items_on_dense_block_maps(oracle):
    """contigs with an indel every fifty bases: the liftover of a heavy item crosses a hundred blocks (two ops of room per block in its
    region, a cursor step in nearly every iteration), through the heavy-item lane path at 16 items per wave"""
    cfg = synth.config("tiny", n_reads=30, seed=135, read_len_mean=2500, read_len_sd=600, split_read_frac=0.2,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=1e-2, dele=1e-2, hpol_frac=0.3, big_indel_prob=0.02))
    w = synth.generate(cfg)
    ix, b = w.index_data(), w.batch_data()
    for stages in (abi.STAGES_ALL, abi.STAGE_STRAND | abi.STAGE_LIFTOVER):
        rc, res, cnt = emu_lib.liftover_batch(ix, b, stages=stages, lane_max_w=150, lane_capw=3072, lane_heavy_per=16, order_seed=3)
        assert rc == 0 and cnt[2] == 0 and cnt[23] < res.n_items  # nothing through the tile kernels, heavy items present
        _assert_same(oracle.liftover_batch(ix, b, stages, 1), res)


def test_lane_path_heavy_items_longer_than_their_region(oracle, monkeypatch):
    """the heavy-item lane path with regions sized below the longer items (the engine caps the regions when one outlier would make
    them huge): those items are handed to the retry list -> wave-cooperative code"""
    monkeypatch.setenv("PLO_EMU_HEAVY_STRIDE", "288")
    cfg = synth.config("tiny", n_reads=40, seed=136, read_len_mean=2500, read_len_sd=900, split_read_frac=0.2,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=3e-3, dele=3e-3, hpol_frac=0.3, big_indel_prob=0.02))
    w = synth.generate(cfg)
    ix, b = w.index_data(), w.batch_data()
    rc, res, cnt = emu_lib.liftover_batch(ix, b, lane_max_w=150, lane_capw=3072, lane_heavy_per=8, order_seed=9, big_thresh=100, cap=512)
    assert rc == 0 and cnt[7] > 0  # items handed on
    _assert_same(oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1), res)



def test_block_map_searches_against_searchsorted():
    """kv_upper_bound (eight-way, unconditional last level), kv_lower_bound and kv_lower_bound_near (enumerate.hpp) on random sorted key
    arrays: empty ranges, ranges inside a larger array, probes below / above / equal to keys"""
    import ctypes as C

    import emu_lib
    L = emu_lib.lib()
    L.emu_kv_search.restype = None
    L.emu_kv_search.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(7)
    out = (C.c_int * 3)()
    for n in [0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 200, 1000, 5000]:
        keys = np.unique(rng.integers(0, max(4, 6 * n), size=n).astype(np.int32)) if n else np.zeros(0, np.int32)
        n = len(keys)
        kp = np.ascontiguousarray(keys if n else np.zeros(1, np.int32)).ctypes.data_as(C.POINTER(C.c_int))
        for _ in range(60):
            lo = int(rng.integers(0, n + 1))
            hi = int(rng.integers(lo, n + 1))
            pool = [int(rng.integers(-3, max(4, 6 * n) + 3))]
            if hi > lo:
                k = int(keys[int(rng.integers(lo, hi))])
                pool += [k, k - 1, k + 1]
            for x in pool:
                near = int(rng.integers(lo, hi + 1))
                L.emu_kv_search(kp, n, lo, hi, x, near, out)
                sub = keys[lo:hi]
                assert out[0] == lo + int(np.searchsorted(sub, x, side="right")), (n, lo, hi, x)
                assert out[1] == lo + int(np.searchsorted(sub, x, side="left")), (n, lo, hi, x)
                assert out[2] == near + int(np.searchsorted(keys[near:hi], x, side="left")), (n, lo, hi, x, near)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_streaming_kernel_indel_dense_items(oracle, case):
    """lane_stream.hpp under the emulator (teams of three waves with random drift between them): indel-dense reads of ~3 kb -- hundreds of
    ops per item, so that every ring wraps many times, headers and end markers of successive items share a ring, lanes take second and
    third items while their neighbours are mid-way -- on clean and on indel-rich contigs (block-map crossings, gap deletions, items
    that do not lift), forward and reverse classes; rings as on the GPU and with the smallest Q2 (retry list)"""
    R = synth.EditRates
    rr, cr, sp = [
        (R(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1), None, 0.1),
        (R(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.9, min_gap=1), R(mismatch=1e-3, ins=2e-3, dele=2e-3, hpol_frac=0.3, big_indel_prob=0.01), 0.3),
        (R(mismatch=2e-2, ins=5e-2, dele=5e-2, hpol_frac=0.7, min_gap=1), R(mismatch=1e-3, ins=1e-2, dele=1e-2, hpol_frac=0.3, big_indel_prob=0.05), 0.3),
    ][case]
    over = dict(n_reads=200, seed=40 + case, split_read_frac=sp, read_len_mean=3000, read_len_sd=1200, read_rates=rr)
    if cr is not None:
        over["contig_rates"] = cr
    w = synth.generate(synth.config("tiny", **over))
    ix, b = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1).canonical()
    for mode, per, oseed in ((1, 64, 0), (1, 150, 99), (2, 64, 5)):
        rc, got, cnt = emu_lib.liftover_batch(ix, b, lane_max_w=12, lane_capw=1024, lane_heavy_per=per, lane_stream=mode, order_seed=oseed)
        assert rc == 0 and cnt[23] == 0  # (every item is heavy: all of them took the streaming kernel)
        assert got.canonical() == ref, f"rings {mode}, {per} items per team, order seed {oseed}"
        assert cnt[7] <= (0 if mode == 1 and case < 2 else got.n_items // 10)  # items handed to the retry list


def test_block_map_built_by_a_wave_equals_the_sequential_builder():
    """k_map_build's wave per contig segment (build_segment_map_wave, enumerate.hpp: scans over 64 ops per step) against the sequential
    build_segment_map -- the restatement of get_read_segment_to_ref_pos_tree_map (lib/rust-vc-utils/src/bam_utils/read_to_ref_map.rs:101-137)
    that tests/test_gpu_parity.py pins to the oracle's builder: random contig->reference CIGARs with every op code, zero-length ops,
    deletions right behind a block (the overwritten None, :111-119), runs that straddle the 64-op steps, empty and all-match CIGARs,
    invalid op codes and positions beyond the 31-bit range (both must refuse)."""
    import ctypes as C

    L = emu_lib.lib()

    class KV(C.Structure):
        _fields_ = [("key", C.c_int), ("val", C.c_int)]

    L.emu_map_build.restype = C.c_int
    L.emu_map_build.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_longlong, C.c_int, C.POINTER(KV), C.c_uint]
    rng = np.random.default_rng(77)

    def both(cig, pos, seed=0):
        a = np.ascontiguousarray(cig, dtype=np.uint32)
        p = a.ctypes.data_as(C.POINTER(C.c_uint32))
        res = []
        for wave in (0, 1):
            out = (KV * (2 * len(a) + 4))()
            n = L.emu_map_build(p, len(a), pos, wave, out, seed)
            res.append((n, [(out[i].key, out[i].val) for i in range(max(0, n))]))
        return res

    cases = [([], 5), ([(100 << 4) | 0], 7), ([(10 << 4) | 4, (50 << 4) | 7, (3 << 4) | 2, (20 << 4) | 8, (5 << 4) | 1, (9 << 4) | 0], 1000),
             ([(5 << 4) | 0, (0 << 4) | 2, (5 << 4) | 0], 3), ([(5 << 4) | 0, (4 << 4) | 2, (0 << 4) | 1, (5 << 4) | 7], 3),
             ([(7 << 4) | 9], 1), ([(0x7ffffff << 4) | 0] * 20 + [(1 << 4) | 2], 0), ([(5 << 4) | 2, (6 << 4) | 0], -3)]
    for n_ops in (1, 2, 63, 64, 65, 127, 128, 129, 500, 3000):
        for _ in range(6):
            mode = rng.integers(0, 3)
            types = rng.choice([0, 7, 8, 1, 2, 3, 4, 5, 6] if mode == 0 else ([7, 8, 2] if mode == 1 else [0, 2, 1]), size=n_ops,
                               p=None if mode else [.3, .2, .1, .1, .1, .05, .05, .05, .05])
            lens = rng.integers(0, 4 if mode == 2 else 60, size=n_ops)
            cases.append((list((lens.astype(np.uint64) << 4 | types.astype(np.uint64)).astype(np.uint32)), int(rng.integers(0, 10_000))))
    n_merge = 0
    for k, (cig, pos) in enumerate(cases):
        seq, wave = both(cig, pos, seed=(k * 13 if k % 3 == 0 else 0))
        assert seq == wave, f"case {k}: {cig[:20]} at {pos}"
        keys = [e[0] for e in seq[1]]
        n_merge += sum(1 for a, b in zip(seq[1], seq[1][1:]) if a[1] != -(2 ** 31) and b[1] != -(2 ** 31))
    assert n_merge > 20  # (deletions right behind a block: consecutive Some entries -- the merged form -- were exercised)


@pytest.mark.parametrize("h16", ["1", "0"])
def test_lane_path_ops_longer_than_a_halfword_holds(oracle, monkeypatch, h16):
    """16-bit regions (lane_core.hpp, H16): ops longer than 8 191 bases are kept as chunks -- by LOAD, by every stage's writer, by the
    trailing-edge rule -- and summed up again on the way out; an op beyond eight chunks, or more extra chunks than a region has room for, sends its item to the retry list.  Reads of 30 kb
    with few edits: match runs of 5 .. 40 kb, clips of several kb; block maps with few, long blocks.  The same through the 32-bit regions."""
    monkeypatch.setenv("PLO_EMU_H16", h16)
    cfg = synth.config("tiny", n_reads=90, seed=136, chrom_lens=(400_000,), read_len_mean=30_000, read_len_sd=9_000, split_read_frac=0.3, clip_read_frac=0.5,
                       read_rates=synth.EditRates(mismatch=4e-5, ins=4e-5, dele=4e-5, hpol_frac=0.7),
                       contig_rates=synth.EditRates(mismatch=1e-4, ins=3e-5, dele=3e-5, hpol_frac=0.3, big_indel_prob=0.3))
    w = synth.generate(cfg)
    ix, b = w.index_data(), w.batch_data()
    lens = b.cigar >> 4
    assert (lens > 8191).sum() > 20 and (lens > 4 * 8191).sum() > 0
    seg_of_op = np.repeat(np.arange(len(b.seg_cigar_off) - 1), np.diff(b.seg_cigar_off))
    huge_segs = set(seg_of_op[lens > 8 * 8191].tolist())
    for stages in (abi.STAGES_ALL, abi.STAGE_STRAND | abi.STAGE_LIFTOVER, abi.STAGES_ALL & ~abi.STAGE_SIMPLIFY):
        ref = oracle.liftover_batch(ix, b, stages, 1)
        n_huge = int(np.isin(ref.item_seg, list(huge_segs)).sum())
        for capw, seed in ((3072, 0), (3072, 4713), (160, 5)):
            rc, res, cnt = emu_lib.liftover_batch(ix, b, stages=stages, lane_max_w=4096, lane_capw=capw, order_seed=seed)
            assert rc == 0
            _assert_same(ref, res)
            if capw == 3072:  # handed on: the items with an op (or a merged run) of more than three chunks or more extra chunks than a region allows
                assert (n_huge <= cnt[7] < res.n_items // 3) if h16 == "1" else cnt[7] == 0, (cnt[7], n_huge, res.n_items)
    assert (ref.item_cigar_len > 0).sum() > 40
