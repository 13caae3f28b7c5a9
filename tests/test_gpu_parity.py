"""GPU parity tests: the HIP path, called through the C ABI, against the oracle and the reference's golden vectors.
Bit-exact bar: (status, need_flipped, mapq, chrom, pos, CIGAR ops) of every item."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from portello_amd import abi, api, synth
from portello_amd import cigar as cg
from variants import HEAVY_VARIANTS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "gpurun_out")


def _dump(name, obj):
    os.makedirs(DUMP, exist_ok=True)
    with open(os.path.join(DUMP, name), "w") as fh:
        json.dump(obj, fh, indent=1)


def _assert_same(ref: abi.BatchResult, got: abi.BatchResult, tag: str):
    a, b = ref.canonical(), got.canonical()
    bad = []
    if len(a) != len(b):
        bad.append({"n_ref": len(a), "n_got": len(b)})
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            bad.append({"i": i, "ref": list(x[:7]) + [cg.decode(np.frombuffer(x[7], dtype=np.uint32))],
                        "got": list(y[:7]) + [cg.decode(np.frombuffer(y[7], dtype=np.uint32))]})
    if bad:
        _dump(f"mismatch_{tag}.json", bad[:50])
    assert not bad, f"{len(bad)} mismatching items (first: {bad[0]})"


def test_wave_primitives_selftest():
    L = api.load_library()
    L.plo_selftest.restype = C.c_int
    L.plo_selftest.argtypes = [C.c_int]
    assert L.plo_selftest(0) == 0


def test_golden_vectors_through_the_c_abi(golden):
    from test_emu_parity import check_golden

    check_golden(golden, api.hip_backend())


def test_reference_named_entry_points(golden):
    v = golden["liftover"][2]
    r = api.liftover_read_alignment((v["map"]["pos"], cg.encode(v["map"]["cigar"])), v["start"], cg.encode(v["cigar"]))
    assert r[0] == v["expect"]["pos"] and cg.decode(r[1]) == v["expect"]["cigar"]
    v = golden["simplify"][5]
    r = api.simplify_alignment_indels(v["pos"], cg.encode(v["cigar"]), v["ref"].encode(), v["read"].encode())
    assert r[0] == v["expect"]["pos"] and cg.decode(r[1]) == v["expect"]["cigar"]
    v = [x for x in golden["shift"] if x["id"] == "SH9-left"][0]
    r = api.left_shift_indels(v["pos"], cg.encode(v["cigar"]), v["ref"].encode(), v["read"].encode())
    assert r[0] == v["expect"]["pos"] and cg.decode(r[1]) == v["expect"]["cigar"]
    assert api.liftover_read_alignment((0, cg.encode("")), 10, cg.encode("10M")) is None


def test_index_block_maps_match_oracle_builder(oracle):
    w = synth.generate(synth.config("tiny", seed=201))
    ixd = w.index_data()
    ix = api.Index(ixd)
    for g in range(ixd.n_segments):
        c = ixd.seg_cigar[ixd.seg_cigar_off[g]: ixd.seg_cigar_off[g + 1]]
        keys, vals = oracle.map_build(int(ixd.seg_pos[g]), c, False)
        k2, v2 = ix.segment_map(g)
        assert list(keys.astype(np.int64)) == list(k2) and list(vals) == list(v2)
    ix.close()


@pytest.mark.parametrize("stages", [abi.STAGES_ALL, abi.STAGES_ALL & ~abi.STAGE_SIMPLIFY, abi.STAGE_STRAND | abi.STAGE_LIFTOVER,
                                    abi.STAGE_LSHIFT, abi.STAGE_SIMPLIFY])
def test_synthetic_tiny_stage_subsets(oracle, stages):
    w = synth.generate(synth.config("tiny", n_reads=300, split_read_frac=0.2, seed=202))
    ix, b = w.index_data(), w.batch_data()
    got = api.hip_backend()(ix, b, stages)
    _assert_same(oracle.liftover_batch(ix, b, stages, 1), got, f"tiny_{stages}")


def test_synthetic_plumbing_config(oracle):
    """BASELINE configs[0]: 1 k reads x 10 kb, one contig -> one chromosome"""
    w = synth.generate(synth.config("plumbing"))
    ix, b = w.index_data(), w.batch_data()
    eng_ix = api.Index(ix)
    eng = api.Engine(eng_ix).set_stats()  # (algo_bytes: the light-item kernel's counting instantiation)
    got = eng.liftover_batch(b)
    _assert_same(oracle.liftover_batch(ix, b, abi.STAGES_ALL, 4), got, "plumbing")
    t = eng.timing()
    n_in = (b.seg_cigar_off[1:] - b.seg_cigar_off[:-1])[got.item_seg]
    assert t.n_items == got.n_items and t.n_in_ops == int(n_in.sum()) and t.n_out_ops == int(got.item_cigar_len.sum())
    assert t.algo_bytes >= 64 * got.n_items
    # run it twice on the same context: buffers are reused, results identical
    _assert_same(got, eng.liftover_batch(b), "plumbing_rerun")
    eng.close()
    eng_ix.close()


def _indel_dense_workload():
    cfg = synth.config("tiny", n_reads=200, seed=203, read_len_mean=6000, read_len_sd=1500,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=3e-3, dele=3e-3, hpol_frac=0.3, big_indel_prob=0.02))
    return synth.generate(cfg)


@pytest.mark.parametrize("mid", ["16", "8", "8:512", "0"])
def test_synthetic_indel_dense_large_item_kernels(oracle, monkeypatch, mid):
    """fixed geometry: items heavier than the routing threshold go to the workgroup-per-item kernel (k_lift_mid, 16 or 8 waves
    per item); with a small capacity the heaviest of them are handed on to the one-wave-per-item kernel in global scratch
    (k_lift_big), with PLO_MID_WAVES=0 all of them; tiles that overflow their LDS slice are re-run one item per wave
    (k_lift_retry)"""
    monkeypatch.setenv("PLO_WINDOW", "256")
    monkeypatch.setenv("PLO_BIG_THRESH", "176")
    monkeypatch.setenv("PLO_CAP", "320")
    waves, _, cap = mid.partition(":")
    monkeypatch.setenv("PLO_MID_WAVES", waves)
    if cap:
        monkeypatch.setenv("PLO_MID_CAP", cap)
    w = _indel_dense_workload()
    ix, b = w.index_data(), w.batch_data()
    eng_ix = api.Index(ix)
    eng = api.Engine(eng_ix)
    got = eng.liftover_batch(b)
    t = eng.timing()
    if waves == "0":
        assert t.n_big_items > 0 and t.n_mid_items == 0
    elif cap:
        assert t.n_big_items > 0 and t.n_mid_items > 0
    else:
        assert t.n_mid_items > 0 and t.n_big_items == 0
    _assert_same(oracle.liftover_batch(ix, b, abi.STAGES_ALL, 4), got, "indel_dense")
    eng.close()
    eng_ix.close()


@pytest.mark.parametrize("variant", list(HEAVY_VARIANTS))
@pytest.mark.parametrize("per", ["64", "7"])
def test_synthetic_indel_dense_heavy_items_lane_per_item(oracle, monkeypatch, per, variant):
    """heavy items through the lane-per-item code, every instantiation (k_lift_lanes_g: regions in wave-private global scratch behind
    per-lane LDS windows, two and three waves per SIMD; k_lift_stream: teams of waves, the all-stages set only); every stage subset that
    changes what the windows carry"""
    for k, v in HEAVY_VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PLO_LANE_HEAVY_MIN", "0")
    monkeypatch.setenv("PLO_LANE_HEAVY_PER", per)
    monkeypatch.setenv("PLO_LANE_MAX_W", "150")
    w = _indel_dense_workload()
    ix, b = w.index_data(), w.batch_data()
    eng_ix = api.Index(ix)
    eng = api.Engine(eng_ix)
    for stages in (abi.STAGES_ALL, abi.STAGE_LSHIFT, abi.STAGE_STRAND | abi.STAGE_LIFTOVER, abi.STAGE_STRAND):
        got = eng.liftover_batch(b, stages)
        t = eng.timing()
        assert t.n_mid_items == 0 and t.n_big_items <= 2
        _assert_same(oracle.liftover_batch(ix, b, stages, 4), got, f"heavy lanes, stages {stages}")
    eng.close()
    eng_ix.close()


def test_heavy_items_longer_than_their_region_are_handed_on(oracle, monkeypatch):
    """k_lift_lanes_g with regions smaller than the longer items of the batch (what the engine does when an outlier would make the
    regions of all resident lanes exceed 8 GB): those items go to the retry list and through the wave-cooperative kernels"""
    monkeypatch.setenv("PLO_LANE_STREAM", "0")  # (the streaming kernel has no regions)
    monkeypatch.setenv("PLO_LANE_HEAVY_MIN", "0")
    monkeypatch.setenv("PLO_LANE_HEAVY_STRIDE", "640")
    monkeypatch.setenv("PLO_LANE_MAX_W", "150")
    w = _indel_dense_workload()
    ix, b = w.index_data(), w.batch_data()
    eng_ix = api.Index(ix)
    eng = api.Engine(eng_ix)
    got = eng.liftover_batch(b)
    t = eng.timing()
    assert t.n_heavy_lane_items > 0 and t.n_retry_items > 0
    _assert_same(oracle.liftover_batch(ix, b, abi.STAGES_ALL, 4), got, "heavy lanes, short regions")
    eng.close()
    eng_ix.close()


def test_synthetic_indel_dense_adaptive_geometry(oracle):
    """default: the routing threshold / LDS slice follow the batch's weight distribution, the same items stay in tiles"""
    w = _indel_dense_workload()
    ix, b = w.index_data(), w.batch_data()
    eng_ix = api.Index(ix)
    eng = api.Engine(eng_ix)
    got = eng.liftover_batch(b)
    t = eng.timing()
    assert t.n_big_items < t.n_items // 4
    _assert_same(oracle.liftover_batch(ix, b, abi.STAGES_ALL, 4), got, "indel_dense_adaptive")
    # a sparse batch after a dense one on the same context goes back to the small geometry
    w2 = synth.generate(synth.config("tiny", n_reads=100, seed=209))
    eng_ix2 = api.Index(w2.index_data())
    eng2 = api.Engine(eng_ix2)
    _assert_same(oracle.liftover_batch(w2.index_data(), w2.batch_data(), abi.STAGES_ALL, 2), eng2.liftover_batch(w2.batch_data()), "sparse")
    eng2.close()
    eng_ix2.close()
    eng.close()
    eng_ix.close()


def test_empty_and_ragged_inputs(oracle):
    w = synth.generate(synth.config("tiny", n_reads=50, seed=204))
    ix, b = w.index_data(), w.batch_data()
    eng_ix = api.Index(ix)
    eng = api.Engine(eng_ix)
    # empty batch
    empty = abi.BatchData(read_is_reverse=[], read_seq_len=[], read_seq_off=[], seq=[], seq_fmt=abi.SEQ_BAM4, seg_read=[],
                          seg_contig=[], seg_pos=[], seg_is_fwd_strand=[], seg_cigar_off=[0], cigar=[])
    assert eng.liftover_batch(empty).n_items == 0
    # ragged: a read with an empty CIGAR, a read on the contig without any segment, a one-op CIGAR
    orphan = len(ix.contig_len) - 1
    rag = abi.BatchData(
        read_is_reverse=[0, 1, 0], read_seq_len=[0, 100, 50], read_seq_off=[0, 0, 50], seq=np.full(100, 0x11, np.uint8),
        seq_fmt=abi.SEQ_BAM4, seg_read=[0, 1, 2], seg_contig=[0, orphan, 0], seg_pos=[int(ix.seg_seq_order_start[0]) + 5, 10,
                                                                                     int(ix.seg_seq_order_start[0]) + 7],
        seg_is_fwd_strand=[1, 0, 1], seg_cigar_off=[0, 0, 1, 2], cigar=cg.encode("100M50M"))
    _assert_same(oracle.liftover_batch(ix, rag, abi.STAGES_ALL, 1), eng.liftover_batch(rag), "ragged")
    eng.close()
    eng_ix.close()


def test_device_resident_api_matches_host_api(oracle):
    import torch

    w = synth.generate(synth.config("tiny", n_reads=400, seed=205, split_read_frac=0.1), device="cuda")
    ix_host, b_host = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ix_host, b_host, abi.STAGES_ALL, 4)
    from portello_amd import devbatch

    eng_ix = api.Index(w.index_data_device())
    eng = api.Engine(eng_ix, stream=torch.cuda.current_stream().cuda_stream)
    db = devbatch.DeviceBatch.from_workload(w)
    res = devbatch.run_and_download(eng, db)
    _assert_same(ref, res, "device_api")
    eng.close()
    eng_ix.close()


def test_full_size_properties_chr20(oracle):
    """BASELINE configs[1] at full size: size-independent properties of every lifted record + oracle parity on a sample"""
    import torch

    from portello_amd import devbatch

    w = synth.generate(synth.config("chr20"), device="cuda")
    eng_ix = api.Index(w.index_data_device())
    eng = api.Engine(eng_ix, stream=torch.cuda.current_stream().cuda_stream)
    db = devbatch.DeviceBatch.from_workload(w)
    res = devbatch.run_and_download(eng, db)
    assert res.n_items >= w.n_reads * 0.9
    lifted = res.item_status == abi.ITEM_LIFTED
    assert lifted.mean() > 0.95 and (res.item_status <= abi.ITEM_NO_LIFTOVER).all()
    # flattened view of all lifted CIGARs
    lens = res.item_cigar_len.astype(np.int64)
    idx = np.repeat(res.item_cigar_off.astype(np.int64), lens) + (np.arange(lens.sum()) - np.repeat(np.cumsum(lens) - lens, lens))
    ops = res.cigar[idx]
    item = np.repeat(np.arange(res.n_items), lens)
    t, L = ops & 15, (ops >> 4).astype(np.int64)
    # (1) read length of the lifted CIGAR == seq_len (the reference's own sanity check, read_alignment_scanner.rs:206-207)
    rl = np.bincount(item, weights=L * np.isin(t, [0, 1, 4, 5, 7, 8]), minlength=res.n_items).astype(np.int64)
    seq_len = w.read_seq_len.cpu().numpy()[w.seg_read.cpu().numpy()[res.item_seg]]
    assert (rl[lifted] == seq_len[lifted]).all()
    # (2) canonical form: no zero-length op, no equal neighbours, only M I D N S H (=/X become M)
    assert (L > 0).all() and np.isin(t, [0, 1, 2, 3, 4, 5]).all()
    same_item = item[1:] == item[:-1]
    assert not (same_item & (t[1:] == t[:-1])).any()
    # (3) no indel at the alignment edges: the first and the last non-clip op of every record is a match
    nonclip = ~np.isin(t, [4, 5])
    first = np.full(res.n_items, -1)
    last = np.full(res.n_items, -1)
    pos = np.nonzero(nonclip)[0]
    first[item[pos][::-1]] = t[pos][::-1]
    last[item[pos]] = t[pos]
    assert (first[lifted] == 0).all() and (last[lifted] == 0).all()
    # (4) the lifted alignment stays inside its chromosome
    ref_span = np.bincount(item, weights=L * np.isin(t, [0, 2, 3]), minlength=res.n_items).astype(np.int64)
    clen = np.array([s.numel() for s in w.chrom_seq])[res.item_chrom_index]
    assert (res.item_ref_pos[lifted] >= 0).all() and ((res.item_ref_pos + ref_span)[lifted] <= clen[lifted]).all()
    # (5) idempotence: the pipeline is a pure function of its inputs
    res2 = devbatch.run_and_download(eng, db)
    assert res2.canonical()[:2000] == res.canonical()[:2000] and (res2.item_ref_pos == res.item_ref_pos).all()
    # (6) oracle parity on 30 blocks of 200 reads spread over the whole coordinate-sorted read set
    import fullsize

    n_cmp, n_flip, n_contigs = fullsize.check_strided_parity(w, res, oracle, 30, 200)
    assert n_cmp > 5000 and n_flip > 0 and n_contigs > 3
    eng.close()
    eng_ix.close()


def _full_size(name, oracle, n_blocks, block, min_lifted=0.95, threads=8, **over):
    import torch

    import fullsize
    from portello_amd import devbatch

    w = synth.generate(synth.config(name, **over), device="cuda")
    eng_ix = api.Index(w.index_data_device())
    eng = api.Engine(eng_ix, stream=torch.cuda.current_stream().cuda_stream)
    db = devbatch.DeviceBatch.from_workload(w)
    res = devbatch.run_and_download(eng, db)
    t = eng.timing()
    fullsize.check_properties(w, res, min_lifted)
    # idempotence: the pipeline is a pure function of its inputs (positions, lengths and every 97th CIGAR)
    res2 = devbatch.run_and_download(eng, db)
    assert (res2.item_ref_pos == res.item_ref_pos).all() and (res2.item_cigar_len == res.item_cigar_len).all()
    assert (res2.item_status == res.item_status).all()
    for i in range(0, res.n_items, 97):
        assert np.array_equal(res.item_cigar(i), res2.item_cigar(i))
    # (the oracle sample is taken from the SECOND call's results: on wgs30x that is the one-round-trip path -- one-launch scan, fused class
    # kernels -- at 1 006 scan tiles and 1 012 class blocks)
    n_cmp, n_flip, n_contigs = fullsize.check_strided_parity(w, res2, oracle, n_blocks, block, threads=threads)
    _dump(f"full_size_{name}.json", {"reads": w.n_reads, "items": int(res.n_items), "items_compared_with_oracle": n_cmp,
                                     "of_them_flipped": n_flip, "contigs_in_sample": n_contigs, "large_items": int(t.n_big_items),
                                     "mid_items": int(getattr(t, "n_mid_items", 0)), "retry_items": int(t.n_retry_items),
                                     "lift_ms": t.lift_ms, "mid_ms": float(getattr(t, "mid_ms", 0.0)), "big_ms": t.big_ms})
    assert n_cmp > 0 and n_flip > 0 and n_contigs > 1
    eng.close()
    eng_ix.close()
    return t


def test_full_size_wgs30x(oracle):
    """BASELINE configs[2] (the bench workload) at full size: properties of all ~2 M records + oracle parity on 250 blocks of
    1 000 reads spread over the whole coordinate-sorted read set (an eighth of all items, as in the bench's own sample)"""
    _full_size("wgs30x", oracle, 250, 1000, threads=os.cpu_count() or 8)


@pytest.mark.parametrize("h16", ["0", "1"])
def test_every_item_of_wgs30x(oracle, h16, monkeypatch):
    """the headline configuration once in full -- every one of the ~2.07 M items of BASELINE configs[2] compared with the
    oracle (blocks of 10 000 reads tiling the whole read set, all host cores; the other full-size test and the bench keep their eighth).
    h16 = 1: the same through the 16-bit-region experiment kernel (k_lift_lanes16)."""
    import torch

    monkeypatch.setenv("PLO_LANE_H16", h16)

    import fullsize
    from portello_amd import devbatch

    w = synth.generate(synth.config("wgs30x"), device="cuda")
    eng_ix = api.Index(w.index_data_device())
    eng = api.Engine(eng_ix, stream=torch.cuda.current_stream().cuda_stream)
    db = devbatch.DeviceBatch.from_workload(w)
    devbatch.run_and_download(eng, db)
    res = devbatch.run_and_download(eng, db)  # (the second call: the one-round-trip path)
    assert int(eng.timing().host_syncs) == 1
    blocks = [(lo, min(lo + 10_000, w.n_reads)) for lo in range(0, w.n_reads, 10_000)]
    n_cmp, n_flip, n_contigs = fullsize.check_strided_parity(w, res, oracle, threads=os.cpu_count() or 8, blocks=blocks)
    _dump("every_item_wgs30x.json" if h16 == "0" else "every_item_wgs30x_16bit_regions.json", {"reads": w.n_reads, "retry_items": int(eng.timing().n_retry_items), "items": int(res.n_items), "items_compared_with_oracle": n_cmp, "of_them_flipped": n_flip,
                                     "contigs_in_sample": n_contigs, "blocks": len(blocks)})
    assert n_cmp == res.n_items and n_flip > 0
    eng.close()
    eng_ix.close()


def test_full_size_stress(oracle):
    """BASELINE configs[4] read profile (20 kb, 5 % indel-dense, ~2 000 ops per read) on the wgs30x contigs: every item is far heavier than a
    shared tile holds; 60 k reads are a reference-sized window task (src/read_alignment_scanner.rs:508-534) -- since round 5 routed to the
    streaming kernel (k_lift_stream: 40 k - 200 k heavy items), before that to the workgroup-per-item kernel, which the second pass
    (PLO_LANE_STREAM=0) still checks"""
    t = _full_size("stress", oracle, 40, 50, min_lifted=0.9, n_reads=60_000)
    assert int(t.heavy_kernel) == 3 and t.n_heavy_lane_items > 50_000  # (k_lift_stream took the window)
    os.environ["PLO_LANE_STREAM"] = "0"
    try:
        t = _full_size("stress", oracle, 40, 50, min_lifted=0.9, n_reads=60_000)
        assert int(t.heavy_kernel) == 0 and t.n_mid_items > 50_000  # (the workgroup-per-item kernel)
    finally:
        del os.environ["PLO_LANE_STREAM"]


def test_full_size_stress_2m_reads_streamed(oracle):
    """BASELINE configs[4] at SURVEY.md 8(d)'s default size, 2 M reads (4 G input ops): more than one batch's 31-bit op indices hold,
    so the read set is lifted as a stream of eight 250 k-read batches on one index, two contexts taking turns
    (portello_amd/stream.py, the way the reference walks its windows, read_alignment_scanner.rs:495-535).  Every batch: properties
    of ALL its records checked on the device, oracle parity on blocks of reads spread over the batch."""
    import torch

    import fullsize
    from portello_amd import devbatch, stream

    dev = torch.device("cuda", 0)
    n_total, chunk = 2_000_000, 250_000
    base = synth.config("stress")
    first = synth.generate(synth.config("stress", n_reads=chunk), device=dev, keep_contigs=True)
    index = api.Index(first.index_data_device())
    ixd = first.index_data()
    runner = stream.StreamRunner(index, dev, 2)
    done = {"reads": 0, "items": 0, "lifted": 0, "cmp": 0, "mid": 0, "heavy_lanes": 0}
    lock = __import__("threading").Lock()
    for pair in range(n_total // chunk // 2):  # two chunks resident at a time, one per context
        ws = [first if k == 0 else synth.generate(synth.config("stress", n_reads=chunk, seed=base.seed + 7919 * k), device=dev, reuse=first)
              for k in (2 * pair, 2 * pair + 1)]
        dbs = [devbatch.DeviceBatch.from_workload(w) for w in ws]
        torch.cuda.synchronize()

        def consume(k, eng, out):
            eng.sync()
            with lock:  # (the checks allocate tens of GB of temporaries on the device: one batch at a time)
                n, nl = fullsize.check_properties_device(ws[k], out, dev)
                res = devbatch.download(eng, out)
                n_cmp, _, _ = fullsize.check_strided_parity(ws[k], res, oracle, 10, 200, threads=os.cpu_count() or 8, ix=ixd)  # 2 000 reads per batch
                done["items"] += n
                done["lifted"] += nl
                done["cmp"] += n_cmp
                done["mid"] += int(eng.timing().n_mid_items)
                done["heavy_lanes"] += int(eng.timing().n_heavy_lane_items)
                done["reads"] += ws[k].n_reads

        runner.run([d.desc() for d in dbs], consume=consume)
        del dbs, ws
    runner.close()
    index.close()
    _dump("full_size_stress_2m_streamed.json", done)
    assert done["reads"] == n_total and done["items"] > n_total and done["cmp"] >= 8 * 2000
    # (250 k heavy items a batch: the lane-per-item kernel over global regions, k_lift_lanes_g, takes them -- test_full_size_stress,
    # 60 k reads, is below its threshold and runs the workgroup-per-item kernel)
    assert done["mid"] == 0 and done["heavy_lanes"] > 0.9 * n_total


def test_every_item_of_100k_stress_reads_heavy_lane_kernel(oracle, monkeypatch):
    """VERDICT r3 (weak #1) / r4 (weak #1, next #6): EVERY item of a 100 k-read batch of the stress profile (BASELINE configs[4]: 20 kb
    reads, 5 % indel-dense, ~2 000 ops per item) against the oracle, through EVERY instantiation of the heavy-item lane kernel
    (HEAVY_VARIANTS: different register budgets, different spills, different code); above the routing threshold, so that kernel --
    the dominant one of that config -- lifts all heavy items.  The oracle runs once per block of reads."""
    import torch

    from portello_amd import devbatch

    w = synth.generate(synth.config("stress", n_reads=100_000), device="cuda")
    index = api.Index(w.index_data_device())
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    db = devbatch.DeviceBatch.from_workload(w)
    gots = {}
    for name, env in HEAVY_VARIANTS.items():
        for k, v in env.items():
            monkeypatch.setenv(k, v)  # (read by every batch call)
        gots[name] = devbatch.run_and_download(eng, db)
        t = eng.timing()
        # (the streaming kernel hands the few items whose unreleased tail outgrows a ring to the wave-cooperative code)
        assert t.n_heavy_lane_items > 0.9 * gots[name].n_items and t.n_mid_items <= (0 if name.startswith("g") else gots[name].n_items // 200), name
    ix = w.index_data()
    n_cmp = 0
    n_items = gots["g"].n_items
    for lo in range(0, w.n_reads, 10_000):  # (blocks: the oracle's result of 10 k reads is ~0.2 GB of Python tuples)
        hi = min(w.n_reads, lo + 10_000)
        ref = oracle.liftover_batch(ix, w.batch_data(lo, hi), abi.STAGES_ALL, os.cpu_count() or 8)
        refc = ref.canonical()
        seg_lo = int(torch.searchsorted(w.seg_read, torch.tensor(lo, device=w.device)).item())
        seg_hi = int(torch.searchsorted(w.seg_read, torch.tensor(hi, device=w.device)).item())
        for name, got in gots.items():
            sub = __import__("fullsize").sub_result(got, seg_lo, seg_hi)
            assert sub.n_items == ref.n_items, name
            assert sub.canonical() == refc, f"{name}: reads [{lo}, {hi})"
        n_cmp += ref.n_items
    assert n_cmp == n_items
    _dump("every_item_stress_100k.json", {"items": int(n_items), "heavy_lane_items": int(t.n_heavy_lane_items), "compared": n_cmp, "variants": list(gots)})
    eng.close()
    index.close()


@pytest.mark.parametrize("window", ["256", "512", "2048"])
def test_lane_groups_cut_by_lds_budget_on_the_gpu(oracle, monkeypatch, capfd, window):
    """PLO_LANE_BUDGET=1 (off by default): k_chunk_sort sorts wider windows and cuts them into the lane kernel's groups by LDS
    budget on the device (prefix sums + a binary search per position + the chain of group starts), the persistent waves walk the list -- every
    item of a 60 k-read workload against the oracle, with the default slice and with one so small that groups hold a dozen items"""
    import torch

    from portello_amd import devbatch

    monkeypatch.setenv("PLO_LANE_BUDGET", "1")
    monkeypatch.setenv("PLO_LANE_SORT_WINDOW", window)
    monkeypatch.setenv("PLO_LANE_GROUP", "64")  # (a batch this small would get groups of 32, which are not cut by budget)
    monkeypatch.setenv("PLO_DEBUG_GEOMETRY", "1")
    w = synth.generate(synth.config("chr20", n_reads=60_000, seed=synth.config("chr20").seed + 31), device="cuda")
    index = api.Index(w.index_data_device())
    ref = oracle.liftover_batch(w.index_data(), w.batch_data(), abi.STAGES_ALL, os.cpu_count() or 8).canonical()
    for capw in ("3072", "768"):
        monkeypatch.setenv("PLO_LANE_CAPW", capw)
        eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
        got = devbatch.run_and_download(eng, devbatch.DeviceBatch.from_workload(w))
        t = eng.timing()
        assert t.n_lane_items > 0.9 * got.n_items
        assert got.canonical() == ref, f"window {window} capw {capw}"
        assert "lane groups cut by LDS budget" in capfd.readouterr().err  # (the path under test was the one that ran)
        eng.close()
    index.close()


def test_geometry_sweep(oracle):
    """every item of five workloads with different contig block-map densities / strand mixes (hence different per-batch tile
    geometries, retry and large-item traffic) against the oracle (tests/soak.py at a size that finishes in a minute)"""
    import torch

    from portello_amd import devbatch

    for rate, rev, seed in ((1e-4, 0.5, 1), (1e-3, 0.5, 2), (2e-3, 0.3, 3), (5e-4, 1.0, 4), (3e-3, 0.5, 5)):
        cr = synth.EditRates(mismatch=1e-3, ins=rate, dele=rate, hpol_frac=0.3, big_indel_prob=0.02)
        cfg = synth.config("chr20", n_reads=12_000, rev_contig_frac=rev, contig_rates=cr, seed=synth.config("chr20").seed + seed)
        w = synth.generate(cfg, device="cuda")
        index = api.Index(w.index_data_device())
        eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
        got = devbatch.run_and_download(eng, devbatch.DeviceBatch.from_workload(w))
        ref = oracle.liftover_batch(w.index_data(), w.batch_data(), abi.STAGES_ALL, 8)
        _assert_same(ref, got, f"sweep_{seed}")
        eng.close()
        index.close()


def test_every_item_of_200k_reads(oracle):
    """tests/soak.py's full-result comparison inside the suite: ALL items of a 200 k-read chr20 workload (207 k items: forward and
    reverse contigs, split reads, every kernel of the default geometry) against the oracle"""
    import torch

    from portello_amd import devbatch

    w = synth.generate(synth.config("chr20", n_reads=200_000, seed=synth.config("chr20").seed + 17), device="cuda")
    index = api.Index(w.index_data_device())
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    got = devbatch.run_and_download(eng, devbatch.DeviceBatch.from_workload(w))
    t = eng.timing()
    ref = oracle.liftover_batch(w.index_data(), w.batch_data(), abi.STAGES_ALL, os.cpu_count() or 8)
    assert got.n_items == ref.n_items and t.n_lane_items > 0.9 * got.n_items
    assert got.canonical() == ref.canonical()
    eng.close()
    index.close()


def test_rccl_gather_path_with_one_rank():
    """bench.py's N > 1 path (window deal, per-rank batch from read ranges, RCCL size all-gather + send/recv of the result arrays,
    verification against the single-GPU result) driven with world size 1 on this GPU -- what the pool's one-GPU boxes allow
    (two ranks on one device are refused by RCCL, profiles/r02_rccl_2proc_1gpu_refused.txt)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLO_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               PLO_BENCH_GATHER_TIMEOUT="120")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "chr20", "--reads", "30000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--e2e-reads", "0", "--overlap-workers", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, (p.stdout[-1500:], p.stderr[-1500:])
    line = json.loads(lines[0])
    assert line["value"] and not line.get("gather_failed") and line["config"]["gather"].startswith("rccl")
    assert line["verify"]["gathered_equals_single_gpu_result"] is True and line["verify"]["reads"] == 30000
    assert set(line["gather_modes"]) >= {"after_every_step", "c_abi"}  # (c_abi: the same exchange through plo_gather_* of the C ABI)
    assert line["verify"]["c_abi_gather_equals_single_gpu_result"] is True


def test_empty_device_batch_through_the_device_path_and_the_gather():
    """a rank whose share of a window-pipeline batch is empty (fewer windows than batches) still takes part in the exchange: an empty device
    batch goes through plo_liftover_batch_dev / plo_compact_output_dev and gathers as zero records"""
    import torch

    from portello_amd import devbatch, gather

    w = synth.generate(synth.config("tiny", n_reads=50, seed=208), device="cuda")
    eng_ix = api.Index(w.index_data_device())
    eng = api.Engine(eng_ix, stream=torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    full = eng.liftover_batch_dev(devbatch.DeviceBatch.from_workload(w).desc())
    assert int(full.n_items) > 0
    db = devbatch.DeviceBatch.from_read_ranges(w, [])
    assert db.n_reads == 0
    out = eng.liftover_batch_dev(db.desc())
    eng.compact_output_dev(out)
    eng.sync()
    assert int(out.n_items) == 0 and int(out.n_cigar) == 0
    t = gather.tensors_from_out(out, dev)
    assert all(int(v.numel()) == 0 for v in t.values())
    ag = gather.AbiGather(eng_ix.lib, None, 0, 1, 0)
    parts = ag.gather(eng, out, dev).wait()
    assert len(parts) == 1 and int(parts[0]["item_seg"].numel()) == 0
    ag.close()
    eng.close()
    eng_ix.close()


def test_c_abi_gather_with_one_rank(oracle):
    """plo_gather_unique_id / _create / _records / _wait / _destroy through ctypes (gather.AbiGather), world size 1 -- what this pool's boxes
    allow: the library opens RCCL by name and creates its own communicator, the size all-gather runs, the root's part is its own arrays; the
    gathered set equals the engine's result and the oracle's.  Wrong arguments are refused."""
    import ctypes as C

    import torch

    from portello_amd import devbatch, gather

    w = synth.generate(synth.config("tiny", n_reads=400, seed=207, split_read_frac=0.2), device="cuda")
    eng_ix = api.Index(w.index_data_device())
    eng = api.Engine(eng_ix, stream=torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    lib = eng_ix.lib
    assert lib.plo_ctx_device(eng.handle) == 0 and lib.plo_ctx_stream(eng.handle)
    ag = gather.AbiGather(lib, None, 0, 1, 0)
    for _ in range(2):  # (the gather object is reused from batch to batch)
        out = eng.liftover_batch_dev(devbatch.DeviceBatch.from_workload(w).desc())
        eng.compact_output_dev(out)
        parts = ag.gather(eng, out, dev).wait()
        assert len(parts) == 1
        got = gather.to_result(gather.combine(parts))
        _assert_same(devbatch.download(eng, out), got, "c_abi_gather")
    _assert_same(oracle.liftover_batch(w.index_data(), w.batch_data(), abi.STAGES_ALL, 2), got, "c_abi_gather_vs_oracle")
    # refused: a root outside the communicator, no place for the gathered records on the root
    outs = (abi.PloBatchOut * 1)()
    assert lib.plo_gather_records(ag.handle, eng.handle, C.byref(out), 3, outs) == abi.PLO_ERR_INVALID_ARG
    assert lib.plo_gather_records(ag.handle, eng.handle, C.byref(out), 0, None) == abi.PLO_ERR_INVALID_ARG
    ident = (C.c_uint8 * 128)()
    h = C.c_void_p()
    assert lib.plo_gather_create(ident, 2, 2, 0, C.byref(h)) == abi.PLO_ERR_INVALID_ARG  # rank outside the world
    ag.close()
    eng.close()
    eng_ix.close()


def test_zero_copy_views_of_device_outputs_for_the_gather(oracle):
    """bench.py's N > 1 path wraps the engine's device outputs as torch tensors (no copy) before the RCCL gather"""
    import torch

    from portello_amd import devbatch, gather

    w = synth.generate(synth.config("tiny", n_reads=300, seed=206), device="cuda")
    eng_ix = api.Index(w.index_data_device())
    eng = api.Engine(eng_ix, stream=torch.cuda.current_stream().cuda_stream)
    out = eng.liftover_batch_dev(devbatch.DeviceBatch.from_workload(w).desc())
    eng.sync()
    t = gather.tensors_from_out(out, torch.device("cuda", 0))
    via_views = gather.to_result(gather.unpack(gather.pack(t), int(out.n_items), int(out.n_cigar)))
    _assert_same(devbatch.download(eng, out), via_views, "views")
    _assert_same(oracle.liftover_batch(w.index_data(), w.batch_data(), abi.STAGES_ALL, 2), via_views, "views_vs_oracle")
    eng.close()
    eng_ix.close()


def test_boundary_validation_errors():
    """the host-buffer entry point validates what it is handed and reports a status instead of computing garbage"""
    w = synth.generate(synth.config("tiny", n_reads=20, seed=208))
    ix, b = w.index_data(), w.batch_data()
    index = api.Index(ix)
    eng = api.Engine(index)
    import copy

    bad = copy.deepcopy(b)
    bad.seg_pos = bad.seg_pos.copy()
    bad.seg_pos[0] = 2**31 + 5
    with pytest.raises(api.PortelloError) as e:
        eng.liftover_batch(bad)
    assert e.value.status == abi.PLO_ERR_RANGE
    bad = copy.deepcopy(b)
    bad.cigar = bad.cigar.copy()
    bad.cigar[3] = (5 << 4) | 11
    with pytest.raises(api.PortelloError) as e:
        eng.liftover_batch(bad)
    assert e.value.status == abi.PLO_ERR_RANGE
    bad = copy.deepcopy(b)
    bad.seg_contig = bad.seg_contig.copy()
    bad.seg_contig[0] = 10_000
    with pytest.raises(api.PortelloError) as e:
        eng.liftover_batch(bad)
    assert e.value.status == abi.PLO_ERR_INVALID_ARG
    bad = copy.deepcopy(b)
    bad.item_seg = np.array([0], dtype=np.uint32)
    bad.item_cseg = np.array([99], dtype=np.uint32)
    with pytest.raises(api.PortelloError) as e:
        eng.liftover_batch(bad)
    assert e.value.status == abi.PLO_ERR_INVALID_ARG
    # the context is still usable afterwards
    assert eng.liftover_batch(b).n_items > 0
    eng.close()
    index.close()
    # index descriptors are validated too
    ixb = w.index_data()
    ixb.seg_chrom_index = ixb.seg_chrom_index.copy()
    ixb.seg_chrom_index[0] = 77
    with pytest.raises(api.PortelloError) as e:
        api.Index(ixb)
    assert e.value.status == abi.PLO_ERR_INVALID_ARG


@pytest.mark.gpu
def test_compact_output(oracle):
    """plo_compact_output_dev: dense CIGAR array (no slab gaps), same records; finishing / SA text read the dense layout"""
    import torch

    from portello_amd import devbatch

    w = synth.generate(synth.config("tiny", n_reads=500, seed=210, split_read_frac=0.2), device="cuda")
    index = api.Index(w.index_data_device())
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    db = devbatch.DeviceBatch.from_workload(w)
    desc = db.desc()
    out = eng.liftover_batch_dev(desc)
    slab = devbatch.download(eng, out)
    extent = int(out.n_cigar)
    eng.compact_output_dev(out)
    dense = devbatch.download(eng, out)
    assert int(out.n_cigar) == int(dense.item_cigar_len.sum()) <= extent
    assert np.array_equal(dense.item_cigar_off, np.cumsum(dense.item_cigar_len, dtype=np.uint64) - dense.item_cigar_len)
    assert dense.canonical() == slab.canonical()
    eng.compact_output_dev(out)  # idempotent
    assert devbatch.download(eng, out).canonical() == slab.canonical()
    fin, keep = devbatch.finish_inputs(w, db, seed=3)
    fo = eng.finish_batch_dev(desc, fin)
    got = devbatch.download_finish(eng, fo, dense.n_items, db.n_reads)
    b = w.batch_data()
    ref = oracle.finish_batch(b, keep["flags"].cpu().numpy().view(np.uint16), keep["qual"].cpu().numpy(), keep["qoff"].cpu().numpy(),
                              oracle.liftover_batch(w.index_data(), b, abi.STAGES_ALL, 2))
    lifted = dense.item_status == 0
    assert np.array_equal(got["item_ref_end"][lifted], ref["item_ref_end"][lifted])
    assert np.array_equal(got["item_bin"][lifted], ref["item_bin"][lifted])
    eng.close()
    index.close()


def test_device_entry_point_checks_its_batch():
    """plo_liftover_batch_dev validates on the device what the reference's types rule out: index ranges, 31-bit coordinates,
    op codes -- before any lift kernel indexes with them"""
    import torch

    from portello_amd import devbatch

    w = synth.generate(synth.config("tiny", n_reads=64, seed=209), device="cuda")
    index = api.Index(w.index_data_device())
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    good = devbatch.DeviceBatch.from_workload(w)

    def run(mutate):
        import dataclasses

        db = devbatch.DeviceBatch.from_workload(w)
        # (slices of the whole workload share its storage: mutate copies)
        db = dataclasses.replace(db, **{f.name: getattr(db, f.name).clone() for f in dataclasses.fields(db) if f.name != "seq_fmt"})
        mutate(db)
        torch.cuda.synchronize()
        with pytest.raises(api.PortelloError) as e:
            eng.liftover_batch_dev(db.desc())
        return e.value.status

    def set_(name, idx, val):
        def f(db):
            getattr(db, name)[idx] = val
        return f

    def swap_offsets(db):
        db.seg_cigar_off[4] = db.seg_cigar_off[5] + 3

    for armed in (False, True):
        if armed:
            # the same on a context whose last batch left the one-round-trip path armed: there the lift kernels are launched BEFORE the host sees
            # the flags, so a segment that fails the checks must have no items (k_seg_count)
            for _ in range(2):
                assert int(eng.liftover_batch_dev(good.desc()).n_items) > 0
        assert run(set_("seg_contig", 0, 1_000_000)) == abi.PLO_ERR_INVALID_ARG
        assert run(set_("seg_read", 3, 10_000_000)) == abi.PLO_ERR_INVALID_ARG
        assert run(set_("seg_pos", 1, 2**31 + 7)) == abi.PLO_ERR_RANGE
        assert run(set_("seg_pos", 1, -4)) == abi.PLO_ERR_RANGE
        assert run(set_("cigar", 5, (7 << 4) | 12)) == abi.PLO_ERR_RANGE
        assert run(set_("read_seq_off", 2, 2**40)) == abi.PLO_ERR_INVALID_ARG
        assert run(swap_offsets) == abi.PLO_ERR_INVALID_ARG
    # and the context is usable afterwards
    assert int(eng.liftover_batch_dev(good.desc()).n_items) > 0
    eng.close()
    index.close()


def test_one_process_two_indexes(oracle):
    """ONE process drives several `plo_index` objects -- one per device, a worker thread per context, no
    collective, results by D2H (the reference picks its per-worker state the same way: src/worker_thread_data.rs:8-30,
    src/read_alignment_scanner.rs:516-517).  Two indexes (devices 0 and 1 when the box has two, else ordinal 0 twice), two threads, each
    lifts its half of a read set while the other runs; both halves equal the oracle, and each context can still be driven from the other
    thread afterwards (the entry points set the calling thread's device themselves)."""
    import threading

    import torch

    n_dev = torch.cuda.device_count()
    devs = [0, 1 % max(1, n_dev)]
    w = synth.generate(synth.config("tiny", n_reads=600, seed=23, split_read_frac=0.15))
    ix = w.index_data()
    half = w.n_reads // 2
    parts = [w.batch_data(0, half), w.batch_data(half, w.n_reads)]
    indexes = [api.Index(ix, d) for d in devs]
    engines = [api.Engine(indexes[k]) for k in range(2)]
    got, errs = [None, None], []

    def work(k):
        try:
            for _ in range(3):  # (several calls each: the two contexts' kernels and host syncs interleave)
                got[k] = engines[k].liftover_batch(parts[k])
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        ref = oracle.liftover_batch(ix, parts[k], abi.STAGES_ALL, 1)
        assert got[k].canonical() == ref.canonical(), f"index {k} on device {devs[k]}"
    # a thread may drive contexts of different devices in turn
    assert engines[1].liftover_batch(parts[1]).canonical() == got[1].canonical()
    assert engines[0].liftover_batch(parts[0]).canonical() == got[0].canonical()
    for e in engines:
        e.close()
    for i in indexes:
        i.close()


def test_one_host_round_trip_path(oracle, monkeypatch):
    """liftover_fast: a context whose last batch was light items only lifts the next batch with ONE host round trip
    (counts read from device memory by the kernels) -- same results as the oracle and as the careful path; a batch that does not fit what
    the last one left (more items than the arrays hold; heavy items) is run again on the careful path and is right too"""
    import torch

    from portello_amd import devbatch

    w = synth.generate(synth.config("chr20", n_reads=30_000), device="cuda")
    ix = w.index_data()
    index = api.Index(w.index_data_device())
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    parts = [(0, 8000), (0, 8000), (16000, 30000), (1000, 3000)]  # a window, the same again (fast), a larger one (falls back), a smaller one (fast)
    syncs = []
    for lo, hi in parts:
        db = devbatch.DeviceBatch.from_workload(w, lo, hi)
        torch.cuda.synchronize()
        got = devbatch.run_and_download(eng, db)
        syncs.append(int(eng.timing().host_syncs))
        ref = oracle.liftover_batch(ix, w.batch_data(lo, hi), abi.STAGES_ALL, os.cpu_count() or 8)
        assert got.canonical() == ref.canonical(), f"reads [{lo}, {hi})"
    assert syncs[0] >= 3 and syncs[1] == 1 and syncs[2] >= 3 and syncs[3] == 1, syncs
    # a batch with no more segments than the last one but MORE ITEMS than the item arrays hold: the launches made with the arrays' capacity
    # must touch nothing (k_item_emit leaves stale descriptors behind VERR_CAP; round 5 lifted them once -- reads through the last batch's
    # offsets, a memory fault under the BAM pipeline) and the careful path gives the right answer
    cand = [(lo, lo + 8000) for lo in range(0, 22001, 2000)]
    shapes = {}
    for lo, hi in cand:
        b = w.batch_data(lo, hi)
        shapes[(lo, hi)] = (int(len(b.seg_read)), int(len(oracle.liftover_batch(ix, b, abi.STAGES_ALL, os.cpu_count() or 8).item_seg)))
    pairs = [(a, b) for a in cand for b in cand if shapes[b][0] <= shapes[a][0] and shapes[b][1] > shapes[a][1]]
    assert pairs, shapes  # (windows of 8 000 reads differ by tens of segments and items either way)
    monkeypatch.setenv("PLO_FAST_CAP_EXACT", "1")  # (the arrays' head-room would hold the few items more; the capacity check is what is tested)
    for a, b in pairs[:3]:
        eng3 = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
        syncs3 = []
        for lo, hi in (a, a, b, b):
            db = devbatch.DeviceBatch.from_workload(w, lo, hi)
            torch.cuda.synchronize()
            got = devbatch.run_and_download(eng3, db)
            syncs3.append(int(eng3.timing().host_syncs))
            ref = oracle.liftover_batch(ix, w.batch_data(lo, hi), abi.STAGES_ALL, os.cpu_count() or 8)
            assert int(eng3.timing().n_items) == shapes[(lo, hi)][1]
            assert got.canonical() == ref.canonical(), f"reads [{lo}, {hi})"
        assert syncs3[1] == 1 and syncs3[2] >= 4 and syncs3[3] == 1, (syncs3, shapes[a], shapes[b])  # (4: the refused attempt's round trip counts)
        eng3.close()
    monkeypatch.delenv("PLO_FAST_CAP_EXACT")
    # ... and without the switch the arrays' head-room takes the few items more in one round trip
    a, b = pairs[0]
    eng5 = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    for lo, hi in (a, a, b):
        db = devbatch.DeviceBatch.from_workload(w, lo, hi)
        torch.cuda.synchronize()
        got = devbatch.run_and_download(eng5, db)
        assert got.canonical() == oracle.liftover_batch(ix, w.batch_data(lo, hi), abi.STAGES_ALL, os.cpu_count() or 8).canonical()
    assert int(eng5.timing().host_syncs) == 1
    eng5.close()
    # the one-launch scan of the segments' item offsets (k_scan_chain, an experiment that stays switched off) gives the same results
    monkeypatch.setenv("PLO_SCAN_CHAIN", "1")
    eng4 = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    for lo, hi in [(0, 30000), (0, 30000), (2000, 29000)]:
        db = devbatch.DeviceBatch.from_workload(w, lo, hi)
        torch.cuda.synchronize()
        got4 = devbatch.run_and_download(eng4, db)
        assert got4.canonical() == oracle.liftover_batch(ix, w.batch_data(lo, hi), abi.STAGES_ALL, os.cpu_count() or 8).canonical()
    assert int(eng4.timing().host_syncs) == 1
    eng4.close()
    monkeypatch.delenv("PLO_SCAN_CHAIN")
    # the same batches with the path switched off give the same results (and three round trips)
    monkeypatch.setenv("PLO_FAST_PATH", "0")
    eng2 = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    for lo, hi in parts[:2]:
        db = devbatch.DeviceBatch.from_workload(w, lo, hi)
        torch.cuda.synchronize()
        got2 = devbatch.run_and_download(eng2, db)
        assert int(eng2.timing().host_syncs) >= 3
        assert got2.canonical() == oracle.liftover_batch(ix, w.batch_data(lo, hi), abi.STAGES_ALL, os.cpu_count() or 8).canonical()
    eng2.close()
    monkeypatch.delenv("PLO_FAST_PATH")
    # a batch with heavy items behind a light one: careful path, right results; the light batch after it is fast again only after a careful one
    ws = synth.generate(synth.config("stress_small", n_reads=300), device="cuda")
    index_s = api.Index(ws.index_data_device())
    eng_s = api.Engine(index_s, stream=torch.cuda.current_stream().cuda_stream)
    for k in range(2):
        got = devbatch.run_and_download(eng_s, devbatch.DeviceBatch.from_workload(ws))
        assert int(eng_s.timing().host_syncs) >= 3  # (heavy items: never the fast path)
        assert got.canonical() == oracle.liftover_batch(ws.index_data(), ws.batch_data(), abi.STAGES_ALL, os.cpu_count() or 8).canonical()
    eng_s.close()
    index_s.close()
    eng.close()
    index.close()


def test_one_round_trip_path_with_fewer_commands(oracle, monkeypatch):
    """the one-round-trip path with the counters summed inside the retry launch (k_lift_retry_sum), grids bounded by what
    the batch can need instead of what the arrays hold and -- on request -- no event records between its phases (plo_ctx_set_phase_events)
    gives the same results, the same counts (in / out ops, algorithmic bytes, lane utilisation of the counting kernel) and the same single
    round trip as the path with k_lift_retry + k_sum_stats and capacity-sized grids"""
    import torch

    from portello_amd import devbatch

    w = synth.generate(synth.config("chr20", n_reads=30_000), device="cuda")
    ix = w.index_data()
    index = api.Index(w.index_data_device())
    parts = [(0, 30000), (0, 30000), (0, 8000), (9000, 13000), (9000, 9900)]  # (windows far smaller than the arrays: the launch bound at work)
    refs = [oracle.liftover_batch(ix, w.batch_data(lo, hi), abi.STAGES_ALL, os.cpu_count() or 8).canonical() for lo, hi in parts]
    variants = [("old", {"PLO_FAST_FUSE": "0", "PLO_FAST_LAUNCH_BOUND": "0"}, True), ("new", {}, True), ("new_no_events", {}, False),
                ("bound_only", {"PLO_FAST_FUSE": "0"}, True), ("fuse_only", {"PLO_FAST_LAUNCH_BOUND": "0"}, False)]
    counts = {}
    for name, env, events in variants:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream).set_stats()
        for k in env:
            monkeypatch.delenv(k)
        if not events:
            eng.set_phase_events(False)
        rows = []
        for (lo, hi), ref in zip(parts, refs):
            db = devbatch.DeviceBatch.from_workload(w, lo, hi)
            torch.cuda.synchronize()
            got = devbatch.run_and_download(eng, db)
            t = eng.timing()
            assert got.canonical() == ref, f"{name}: reads [{lo}, {hi})"
            rows.append((int(t.n_items), int(t.n_in_ops), int(t.n_out_ops), int(t.algo_bytes), int(t.n_retry_items), int(t.host_syncs), round(float(t.lane_utilisation), 5)))
            if rows[-1][5] == 1:  # a one-round-trip call: times only with the events
                assert (float(t.total_ms) > 0.0) == events and (float(t.lanes_ms) > 0.0) == events, (name, lo, hi, float(t.total_ms))
            else:
                assert float(t.total_ms) > 0.0
        assert [r[5] for r in rows[1:]] == [1, 1, 1, 1], (name, rows)
        counts[name] = rows
        eng.close()
    for name in counts:
        assert counts[name] == counts["old"], (name, counts[name], counts["old"])
    assert all(r[1] > 0 and r[2] > 0 and r[3] > 0 and r[6] > 0 for r in counts["new"])
    index.close()
