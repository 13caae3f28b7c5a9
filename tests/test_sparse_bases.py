"""PLO_SEQ_BAM4_SPARSE: only the read bases around the indels of the read->contig CIGARs travel to the device.

CPU: the packer's layout against the dense bases, and the device algorithm under the wave emulator on sparse batches -- every item
either equals the oracle's result (computed from the dense bases) or reports PLO_ITEM_NEED_BASES; with every granule present
nothing is missed.  GPU (-m gpu): plo_liftover_batch takes a second look at the missed items from `seq_full`, so the result is the
oracle's whatever the margin."""
import dataclasses

import numpy as np
import pytest

import emu_lib
from portello_amd import abi, bam, synth


def workload(n_reads=150, seed=7, **kw):
    cfg = synth.config("tiny", n_reads=n_reads, seed=seed, read_len_mean=2500, read_len_sd=600, split_read_frac=0.2,
                       read_rates=synth.EditRates(mismatch=2e-3, ins=4e-3, dele=4e-3, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=2e-3, dele=2e-3, hpol_frac=0.3, big_indel_prob=0.02), **kw)
    return synth.generate(cfg)


def granule_view(sp: abi.BatchData, r: int):
    """(mask bits per granule, {granule: 16 bytes}) of read r of a sparse batch"""
    n = int(sp.read_seq_len[r])
    off = int(sp.read_seq_off[r])
    nblk = (n + 1023) >> 10
    hb = (nblk * 8 + 15) & ~15
    hdr = sp.seq[off:off + 8 * nblk].view(np.uint32).reshape(nblk, 2)
    present, rank = {}, 0
    for b in range(nblk):
        assert int(hdr[b, 1]) == rank
        for k in range(32):
            if (int(hdr[b, 0]) >> k) & 1:
                at = off + hb + 16 * rank
                present[32 * b + k] = sp.seq[at:at + 16]
                rank += 1
    return present


@pytest.mark.parametrize("margin", [0, 20, 32])
def test_packer_layout_and_marks(margin):
    w = workload(60, seed=11)
    b = w.batch_data()
    assert b.seq_fmt == abi.SEQ_BAM4
    sp = bam.sparse_pack(b, margin, n_threads=3)
    assert sp.seq_fmt == abi.SEQ_BAM4_SPARSE and sp.seq_full is b.seq
    first = np.searchsorted(b.seg_read, np.arange(b.n_reads + 1))
    total = 0
    for r in range(b.n_reads):
        n = int(b.read_seq_len[r])
        assert int(sp.read_seq_off[r]) % 16 == 0
        dense = b.seq[int(b.read_seq_off[r]):int(b.read_seq_off[r]) + (n + 1) // 2]
        present = granule_view(sp, r)
        for g, bytes16 in present.items():  # a present granule holds the dense bytes, zero-padded at the end of the read
            want = np.zeros(16, np.uint8)
            part = dense[16 * g:16 * g + 16]
            want[:len(part)] = part
            assert (bytes16 == want).all(), (r, g)
        # every indel of every segment, widened by the margin, lies in present granules (stored orientation)
        expect = set()
        for s in range(first[r], first[r + 1]):
            changes = bool(b.read_is_reverse[r]) == bool(b.seg_is_fwd_strand[s])
            q = 0
            for c in b.cigar[int(b.seg_cigar_off[s]):int(b.seg_cigar_off[s + 1])]:
                t, L = int(c) & 15, int(c) >> 4
                if t in (1, 2):
                    a, e = q, q + (L if t == 1 else 0)
                    if changes:
                        a, e = n - e, n - a
                    a, e = max(0, a - margin), min(n, e + margin)
                    if e > a:
                        expect.update(range(a >> 5, ((e - 1) >> 5) + 1))
                if t in (0, 1, 4, 5, 7, 8):
                    q += L
        assert set(present) == expect, r
        total += len(present)
    assert total > 0
    assert sp.seq.nbytes < b.seq.nbytes or margin > 0


def test_packer_rejects_unordered_segments():
    from portello_amd import api
    w = workload(10, seed=3)
    b = w.batch_data()
    bad = dataclasses.replace(b, seg_read=b.seg_read[::-1].copy())
    if b.n_segs > 1 and bad.seg_read[0] != bad.seg_read[-1]:
        with pytest.raises(api.PortelloError):
            bam.sparse_pack(bad, 32)


def test_packer_rejects_bases_outside_the_buffer():
    from portello_amd import api
    w = workload(10, seed=4)
    b = w.batch_data()
    off = b.read_seq_off.copy()
    off[-1] = b.seq.nbytes - 3
    with pytest.raises(api.PortelloError):
        bam.sparse_pack(dataclasses.replace(b, read_seq_off=off), 32)


@pytest.mark.parametrize("margin,allow_miss", [(0, True), (32, True), (1 << 20, False)])
def test_emulated_device_algorithm_on_sparse_bases(oracle, margin, allow_miss):
    w = workload(150, seed=7)
    ix, b = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1).canonical()
    sp = bam.sparse_pack(b, margin)
    rc, res, counters = emu_lib.liftover_batch(ix, sp)
    assert rc == 0
    got = res.canonical()
    assert len(got) == len(ref)
    miss = 0
    for x, y in zip(ref, got):
        if y[2] == abi.ITEM_NEED_BASES:
            miss += 1
            assert y[7] == b""  # no result
            assert x[:2] == y[:2]
        else:
            assert x == y
    assert miss == counters[21]  # CNT_NMISS
    if not allow_miss:
        assert miss == 0
    if margin == 0:
        assert miss > 0  # the probes start 16 bases before the indel: with no margin some of them must leave the granules sent
    if margin == 32:
        assert miss <= len(ref) // 10, (miss, len(ref))


def test_emulated_workgroup_per_item_on_sparse_bases(oracle):
    """the same through the several-waves-per-item formulation (heavy items)"""
    cfg = synth.config("tiny", n_reads=16, seed=102, read_len_mean=3000, read_len_sd=500,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=3e-3, dele=3e-3, hpol_frac=0.3, big_indel_prob=0.02))
    w = synth.generate(cfg)
    ix, b = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1).canonical()
    for margin in (8, 1 << 20):
        sp = bam.sparse_pack(b, margin)
        rc, res, counters = emu_lib.liftover_batch(ix, sp, big_thresh=64, mid_waves=4, mid_cap=2048)
        assert rc == 0
        n_miss = 0
        for x, y in zip(ref, res.canonical()):
            if y[2] == abi.ITEM_NEED_BASES:
                n_miss += 1
            else:
                assert x == y
        assert n_miss == counters[21]
        if margin > 8:
            assert n_miss == 0


@pytest.mark.parametrize("path", ["lanes", "heavy lanes"])
def test_emulated_lane_per_item_paths_on_sparse_bases(oracle, path):
    """the lane-per-item code (`lane_tile<SP = true>`: synchronous probes through the granule look-up) -- regions in LDS, and heavy
    items in global regions behind the LDS windows -- on sparse bases: same results, same items reported as missing bases"""
    cfg = synth.config("tiny", n_reads=24, seed=141, read_len_mean=2500, read_len_sd=700, split_read_frac=0.2,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=3e-3, dele=3e-3, hpol_frac=0.3, big_indel_prob=0.02))
    w = synth.generate(cfg)
    ix, b = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1).canonical()
    kw = dict(lane_max_w=100000, lane_capw=60000) if path == "lanes" else dict(lane_max_w=150, lane_capw=3072, lane_heavy_per=16)
    for margin in (8, 1 << 20):
        sp = bam.sparse_pack(b, margin)
        rc, res, counters = emu_lib.liftover_batch(ix, sp, order_seed=5, **kw)
        assert rc == 0 and counters[2] == 0  # nothing through the wave-cooperative kernels
        n_miss = 0
        for x, y in zip(ref, res.canonical()):
            if y[2] == abi.ITEM_NEED_BASES:
                n_miss += 1
            else:
                assert x == y
        assert n_miss == counters[21]
        if margin > 8:
            assert n_miss == 0


def test_sparse_window_batch_equals_packing_the_dense_window_batch(tmp_path):
    """plo_bam_window_batch_sparse (granules straight from the BAM records) == plo_sparse_seq_pack of the window's dense batch;
    seq_full / read_seq_full_off point at the records' own packed bases"""
    import ctypes as C
    from portello_amd import bamsynth
    w = workload(300, seed=9)
    path = str(tmp_path / "reads.bam")
    bamsynth.write_read_bam(w, path, 0, w.n_reads, level=1, n_threads=2)
    rd = bam.BamReader(path, 2)
    try:
        win = rd.read_window(100000)
        dense = win.batch_data()
        want = bam.sparse_pack(dense, 32, n_threads=2)
        d = win.batch_desc(sparse_margin=32)
        n = int(d.n_reads)
        assert int(d.seq_fmt) == abi.SEQ_BAM4_SPARSE and n == dense.n_reads and int(d.seq_bytes) == want.seq.nbytes
        got_seq = np.ctypeslib.as_array(d.seq, shape=(int(d.seq_bytes),))
        got_off = np.ctypeslib.as_array(d.read_seq_off, shape=(n,))
        assert (got_off == want.read_seq_off).all() and (got_seq == want.seq).all()
        full_off = np.ctypeslib.as_array(d.read_seq_full_off, shape=(n,))
        for r in (0, n // 2, n - 1):
            nb = (int(dense.read_seq_len[r]) + 1) // 2
            full = np.ctypeslib.as_array(C.cast(C.addressof(d.seq_full.contents) + int(full_off[r]), C.POINTER(C.c_uint8)), shape=(nb,))
            assert (full == dense.seq[int(dense.read_seq_off[r]):int(dense.read_seq_off[r]) + nb]).all()
        win.close()
    finally:
        rd.close()


# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_sparse_bases_heavy_items_lane_per_item(oracle, monkeypatch):
    """indel-dense items on sparse bases through k_lift_lanes_g_sp (heavy items, global regions behind the LDS windows), items
    that reached absent bases lifted again from the complete ones"""
    from portello_amd import api
    monkeypatch.setenv("PLO_LANE_HEAVY_MIN", "0")
    monkeypatch.setenv("PLO_LANE_MAX_W", "150")
    cfg = synth.config("tiny", n_reads=300, seed=142, read_len_mean=5000, read_len_sd=1500, split_read_frac=0.2,
                       read_rates=synth.EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1),
                       contig_rates=synth.EditRates(mismatch=1e-3, ins=3e-3, dele=3e-3, hpol_frac=0.3, big_indel_prob=0.02))
    w = synth.generate(cfg)
    ixd, b = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ixd, b, abi.STAGES_ALL, 8).canonical()
    index = api.Index(ixd, 0)
    eng = api.Engine(index)
    try:
        for margin in (4, 32):
            sp = bam.sparse_pack(b, margin)
            got = abi.result_from_out(eng.liftover_batch_host(sp.to_desc())).canonical()
            t = eng.timing()
            assert t.n_heavy_lane_items > 0 and t.n_mid_items == 0
            assert got == ref, margin
    finally:
        eng.close()
        index.close()


@pytest.mark.gpu
@pytest.mark.parametrize("margin", [0, 32])
def test_gpu_sparse_bases_second_look_gives_the_oracle_result(oracle, margin):
    from portello_amd import api
    w = workload(4000, seed=21)
    ixd, b = w.index_data(), w.batch_data()
    ref = oracle.liftover_batch(ixd, b, abi.STAGES_ALL, 8).canonical()
    sp = bam.sparse_pack(b, margin)
    index = api.Index(ixd, 0)
    eng = api.Engine(index)
    try:
        out = eng.liftover_batch_host(sp.to_desc())
        got = abi.result_from_out(out).canonical()
        t = eng.timing()
        assert got == ref
        if margin == 0:
            assert t.n_miss_items > 0
        # without the complete bases the same items come back as NEED_BASES, everything else as the oracle has it
        blind = dataclasses.replace(sp, seq_full=None, read_seq_full_off=None)
        got2 = abi.result_from_out(eng.liftover_batch_host(blind.to_desc())).canonical()
        n_need = sum(1 for y in got2 if y[2] == abi.ITEM_NEED_BASES)
        assert n_need == t.n_miss_items
        for x, y in zip(ref, got2):
            assert y[2] == abi.ITEM_NEED_BASES or x == y
        # device-resident form: NEED_BASES resolved from host seq_full as well
        import torch
        dev = torch.device("cuda", 0)
        keep = {}
        d = sp.to_desc()
        import ctypes as C

        def up(name, arr, ctype):
            t_ = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).copy()).to(dev)
            keep[name] = t_
            setattr(d, name, C.cast(t_.data_ptr(), C.POINTER(ctype)))
        up("read_is_reverse", sp.read_is_reverse, C.c_uint8)
        up("read_seq_len", sp.read_seq_len, C.c_uint32)
        up("read_seq_off", sp.read_seq_off, C.c_uint64)
        up("seq", sp.seq, C.c_uint8)
        up("seg_read", sp.seg_read, C.c_uint32)
        up("seg_contig", sp.seg_contig, C.c_uint32)
        up("seg_pos", sp.seg_pos, C.c_int64)
        up("seg_is_fwd_strand", sp.seg_is_fwd_strand, C.c_uint8)
        up("seg_cigar_off", sp.seg_cigar_off, C.c_uint32)
        up("cigar", sp.cigar, C.c_uint32)
        torch.cuda.synchronize()
        dout = eng.liftover_batch_dev(d)
        from portello_amd import devbatch
        got3 = devbatch.download(eng, dout).canonical()
        assert got3 == ref
    finally:
        eng.close()
        index.close()


@pytest.mark.gpu
def test_gpu_sparse_window_batch_from_bam(tmp_path, oracle):
    """BAM window -> sparse batch (bases straight from the records) -> plo_liftover_batch == dense batch of the same window"""
    from portello_amd import api, bamsynth
    w = workload(1500, seed=5)
    path = str(tmp_path / "reads.bam")
    bamsynth.write_read_bam(w, path, 0, w.n_reads, level=1, n_threads=2)
    ixd = w.index_data()
    index = api.Index(ixd, 0)
    eng = api.Engine(index)
    rd = bam.BamReader(path, 2)
    try:
        win = rd.read_window(100000)
        dense = abi.result_from_out(eng.liftover_batch_host(win.batch_desc())).canonical()
        d_bytes = int(win._batch.seq_bytes)
        desc = win.batch_desc(sparse_margin=32)
        assert int(desc.seq_fmt) == abi.SEQ_BAM4_SPARSE and int(desc.seq_bytes) < d_bytes
        sparse = abi.result_from_out(eng.liftover_batch_host(desc)).canonical()
        assert sparse == dense
        s_bytes = int(desc.seq_bytes)
        # strand-aware: forward-only contigs send their insertions' bases only; the result is the same (second look included)
        desc2 = win.batch_desc(sparse_margin=32, index_desc=ixd.to_desc())
        assert int(desc2.seq_bytes) < s_bytes
        assert abi.result_from_out(eng.liftover_batch_host(desc2)).canonical() == dense
        win.close()
    finally:
        rd.close()
        eng.close()
        index.close()


@pytest.mark.gpu
def test_gpu_garbage_granule_headers_stay_inside_the_buffer():
    """headers are caller data: whatever they say, a probe reads inside `seq` or reports the bases absent -- the call returns
    (statuses may be anything the garbage implies), the engine stays usable and a clean batch afterwards gives the oracle's result"""
    from portello_amd import api
    rng = np.random.default_rng(77)
    w = workload(1500, seed=31)
    ixd, b = w.index_data(), w.batch_data()
    sp = bam.sparse_pack(b, 32)
    index = api.Index(ixd, 0)
    eng = api.Engine(index)
    try:
        for trial in range(4):
            seq = sp.seq.copy()
            for r in range(sp.n_reads):  # overwrite every header with random masks / ranks (the last trial: huge ranks)
                n = int(sp.read_seq_len[r])
                off = int(sp.read_seq_off[r])
                nw = (n + 1023) >> 10
                g = rng.integers(0, 2**32, 2 * nw, dtype=np.uint64).astype(np.uint32)
                if trial == 3:
                    g[1::2] = 0xFFFFFFF0
                seq[off:off + 8 * nw] = g.view(np.uint8)
            bad = dataclasses.replace(sp, seq=seq, seq_full=None, read_seq_full_off=None)
            out = abi.result_from_out(eng.liftover_batch_host(bad.to_desc()))
            assert out.n_items > 0
        got = abi.result_from_out(eng.liftover_batch_host(sp.to_desc())).canonical()
        from oracle import pyoracle
        pyoracle.build()
        assert got == pyoracle.liftover_batch(ixd, b, abi.STAGES_ALL, 8).canonical()
    finally:
        eng.close()
        index.close()
