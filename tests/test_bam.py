"""BAM/BGZF input, batch construction from records, output record bytes (include/portello_bam.h) against
 - the reference's own vectors for SA / split-segment parsing (split_read.rs:198-232, sa_tag_parser.rs:66-77),
 - oracle/pyrecords.py, the pure-Python restatement of the record logic (byte for byte),
 - tests/bamcheck.py, an independent reader of the files the BGZF writer produces.
CPU tests use the C oracle for the lifted alignments; the GPU test runs the same window through the HIP engine."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import bamcheck
from oracle import pyrecords as pr
from portello_amd import abi, api, bam, bamsynth, synth
from oracle.expect import expected_records  # (shared with bench.py's verification of its end-to-end sample)
from portello_amd import cigar as cg


def _label_to_index(names):
    return {n: i for i, n in enumerate(names)}


def test_sa_parser_reference_vector():
    """sa_tag_parser.rs:66-77"""
    val = ("chr3,10001,+,5535S10=1D39=2X11438S,60,192;chr3,10001,+,3073S15=2D20=2X11=1X5=1I23=1X5=14798S,22,44;"
           "chr4,106872270,-,23=1I226=1I195=1X147=1D1021=7362S,60,19;")
    r = pr.parse_sa_aux_val(val)
    assert len(r) == 3 and r[2]["rname"] == "chr4" and r[1]["pos"] == 10_000 and not r[2]["is_fwd_strand"]


def _sam_record(tid, pos1, cigar_text, seq, qual, sa=None, flag=0):
    cig = np.array(cg.encode(cigar_text), dtype=np.uint32)
    lut = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    n4 = [lut[c] for c in seq] + ([0] if len(seq) & 1 else [])
    sp = bytes((n4[i] << 4) | n4[i + 1] for i in range(0, len(n4), 2))
    aux = (b"SAZ" + sa.encode() + b"\0") if sa else b""
    return bamsynth.encode_record(tid, pos1 - 1, 60, flag, b"qname", cig, sp, len(seq), bytes(ord(c) - 33 for c in qual), aux)


def test_split_segments_reference_vectors(tmp_path):
    """split_read.rs:198-232 through both restatements: pyrecords and the engine's window builder (via a real BAM file)"""
    names = ["chr0", "chr1", "chr2"]
    seq, qual = "ACGCCGTATCGTCTCGAGGA", "DDDDDEEEEEDDDDDEEEEE"
    r1 = _sam_record(2, 10, "10S5M5S", seq, qual)
    r2 = _sam_record(2, 10, "10S5M5S", seq, qual, sa="chr0,20,-,5M15S,60,0;chr0,100,+,5S5M10S,60,0;chr1,200,-,15S5M,60,0;")
    exp1 = [(10, 15, 2, 9, True, "10S5M5S", True)]
    exp2 = [(0, 5, 1, 199, False, "15S5M", False), (5, 10, 0, 99, True, "5S5M10S", False), (10, 15, 2, 9, True, "10S5M5S", True),
            (15, 20, 0, 19, False, "5M15S", False)]
    l2i = _label_to_index(names)
    for rb, exp in ((r1, exp1), (r2, exp2)):
        got = pr.get_seq_order_read_split_segments(l2i, pr.record_from_bytes(rb))
        assert [(s.seq_order_read_start, s.seq_order_read_end, s.chrom_index, s.pos, s.is_fwd_strand, pr.cigar_to_text(s.cigar),
                 s.from_primary_bam_record) for s in got] == exp
    path = str(tmp_path / "v.bam")
    w = bam.BamWriter(path, "@HD\tVN:1.6\n", names, [1000] * 3, level=6)
    w.write(r1 + r2)
    w.close()
    rd = bam.BamReader(path, 2)
    assert rd.ref_names == names and rd.ref_lens == [1000] * 3
    win = rd.read_window(10)
    b = win.batch_data()
    assert b.n_reads == 2 and b.n_segs == 5
    exp = exp1 + exp2
    assert list(b.seg_read) == [0, 1, 1, 1, 1]
    assert [int(x) for x in b.seg_contig] == [e[2] for e in exp] and [int(x) for x in b.seg_pos] == [e[3] for e in exp]
    assert [bool(x) for x in b.seg_is_fwd_strand] == [e[4] for e in exp]
    for s, e in enumerate(exp):
        assert cg.decode(b.cigar[int(b.seg_cigar_off[s]):int(b.seg_cigar_off[s + 1])]) == e[5]
    assert rd.read_window(10) is None
    win.close()
    rd.close()


@pytest.fixture(scope="module")
def small_bam(tmp_path_factory):
    d = tmp_path_factory.mktemp("bam")
    w = synth.generate(synth.config("tiny", n_reads=300, seed=411, split_read_frac=0.3, sorted_reads=True))
    path = str(d / "reads.bam")
    meta = bamsynth.write_read_bam(w, path, level=6, n_unmapped=4)
    return w, path, meta


def test_writer_output_is_a_valid_bam_and_reader_round_trips(small_bam, tmp_path):
    w, path, meta = small_bam
    text, refs, recs = bamcheck.read_bam(path)
    assert text == meta["header_text"] and [n for n, _ in refs] == meta["contig_names"]
    blocks = bamcheck.bgzf_blocks(path)
    assert blocks[-1] == (28, 0) and all(isz <= 0xff00 for _, isz in blocks)
    flags = [struct.unpack_from("<H", r, 18)[0] for r in recs]
    prim = [r for r, f in zip(recs, flags) if not (f & 0x804)]
    assert len(prim) == w.n_reads and sum(1 for f in flags if f & 4) == 4 and any(f & 0x800 for f in flags)
    # the engine's reader sees the same records, whatever the window size; level 0 (stored blocks) round-trips too
    for max_rec in (7, 64, 10_000):
        rd = bam.BamReader(path, 3)
        got, unm, n_unm = [], b"", 0
        while True:
            win = rd.read_window(max_rec)
            if win is None:
                break
            b = win.batch_data()
            assert b.n_reads == win.n_records
            got.append(b)
            u, k = win.unmapped_bytes()
            unm += u
            n_unm += k
            win.close()
        rd.close()
        assert sum(b.n_reads for b in got) == w.n_reads and n_unm == 4
        assert unm == b"".join(meta["unmapped_records"])  # pass-through records are byte-identical (scan_unmapped_reads :551-555)
    p0 = str(tmp_path / "stored.bam")
    wr = bam.BamWriter(p0, text, [n for n, _ in refs], [l for _, l in refs], level=0)
    wr.write(b"".join(recs))
    wr.close()
    assert bamcheck.read_bam(p0)[2] == recs


@pytest.mark.parametrize("level", [0, 1])
def test_writer_with_blocks_reserved_ahead_writes_the_same_file(small_bam, tmp_path, monkeypatch, level):
    """The output's blocks reserved ahead of the writes (FALLOC_FL_KEEP_SIZE; the default for files beyond 64 MB): the same bytes, and the file
    ends where its last block ends -- what was reserved behind it is given back when the writer closes"""
    _, path, _ = small_bam
    text, refs, recs = bamcheck.read_bam(path)
    out = {}
    for mode in ("0", "2"):  # (2: reserve from the first byte -- by default files below 64 MB reserve nothing)
        monkeypatch.setenv("PLO_BGZF_FALLOCATE", mode)
        p = str(tmp_path / f"reserved_{mode}.bam")
        wr = bam.BamWriter(p, text, [n for n, _ in refs], [l for _, l in refs], level=level, n_threads=3)
        for k in range(0, len(recs), 50):  # several writes: blocks straddle them
            wr.write(b"".join(recs[k:k + 50]))
        wr.close()
        out[mode] = open(p, "rb").read()
        st = os.stat(p)
        assert st.st_size == len(out[mode]) and st.st_blocks * 512 < st.st_size + (8 << 20)  # (no gigabyte left allocated behind the end)
    assert out["0"] == out["2"] and bamcheck.read_bam(str(tmp_path / "reserved_2.bam"))[2] == recs
    if level == 0:  # the default setting on a file beyond 64 MB (where it starts to reserve): the same bytes as without, nothing left behind the end
        blob = b"".join(recs)
        reps = (80 << 20) // len(blob) + 1
        big = {}
        for mode in ("0", None):
            if mode is None:
                monkeypatch.delenv("PLO_BGZF_FALLOCATE")
            else:
                monkeypatch.setenv("PLO_BGZF_FALLOCATE", mode)
            p = str(tmp_path / f"big_{mode}.bam")
            wr = bam.BamWriter(p, text, [n for n, _ in refs], [l for _, l in refs], level=0, n_threads=3)
            for _ in range(reps):
                wr.write(blob)
            wr.close()
            h = __import__("hashlib").sha1()
            with open(p, "rb") as f:
                for piece in iter(lambda: f.read(1 << 24), b""):
                    h.update(piece)
            st = os.stat(p)
            big[mode] = (st.st_size, h.hexdigest())
            assert st.st_size > (80 << 20) and st.st_blocks * 512 < st.st_size + (8 << 20)
            os.unlink(p)
        assert big["0"] == big[None]


def test_window_batch_matches_python_split_segments(small_bam):
    w, path, meta = small_bam
    _, _, recs = bamcheck.read_bam(path)
    prim = [r for r in recs if not (struct.unpack_from("<H", r, 18)[0] & 0x804)]
    rd = bam.BamReader(path, 2)
    win = rd.read_window(100_000)
    b = win.batch_data()
    l2i = _label_to_index(meta["contig_names"])
    s = 0
    for r, rb in enumerate(prim):
        rec = pr.record_from_bytes(rb)
        assert int(b.read_is_reverse[r]) == int(rec.is_reverse()) and int(b.read_seq_len[r]) == rec.l_seq
        o = int(b.read_seq_off[r])
        assert b.seq[o:o + (rec.l_seq + 1) // 2].tobytes() == rec.seq4
        for seg in pr.get_seq_order_read_split_segments(l2i, rec):
            assert (int(b.seg_read[s]), int(b.seg_contig[s]), int(b.seg_pos[s]), bool(b.seg_is_fwd_strand[s])) == \
                   (r, seg.chrom_index, seg.pos, seg.is_fwd_strand)
            assert [int(x) for x in b.cigar[int(b.seg_cigar_off[s]):int(b.seg_cigar_off[s + 1])]] == seg.cigar
            s += 1
    assert s == b.n_segs
    win.close()
    rd.close()


@pytest.mark.parametrize("is_target_region", [False, True])
def test_record_bytes_match_python_restatement(small_bam, oracle, is_target_region):
    """plo_records_build byte for byte against oracle/pyrecords.py (clone_record / PS / ZM / SA / flags / bin / reversed
    seq+qual / unmapped copy), on lifted alignments from the C oracle"""
    w, path, meta = small_bam
    ix = w.index_data()
    _, _, recs = bamcheck.read_bam(path)
    prim = [r for r in recs if not (struct.unpack_from("<H", r, 18)[0] & 0x804)]
    rd = bam.BamReader(path, 2)
    win = rd.read_window(100_000)
    b = win.batch_data()
    res = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 2)
    assert (res.item_status == abi.ITEM_NO_LIFTOVER).any() or True
    o, keep = abi.out_from_result(res)
    cn, rn = meta["contig_names"], bamsynth.ref_names(w)
    data, off, n_lift, n_unm = win.build_records(o, ix.to_desc(), cn, rn, is_target_region=is_target_region, n_threads=3)
    exp = expected_records(prim, ix, cn, rn, res, is_target_region)
    got = [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    assert len(got) == len(exp)
    for i, (a, e) in enumerate(zip(got, exp)):
        assert a == e, (i, pr.record_from_bytes(a), pr.record_from_bytes(e))
    assert n_lift == int((res.item_status == abi.ITEM_LIFTED).sum())
    if not is_target_region:
        assert n_unm == len(exp) - n_lift
    # the hand-checked ordering of the added tags on one split read: [kept tags] PS ZM SA, removed NM / old PS / old ZM / old SA
    multi = [pr.record_from_bytes(e) for e in exp if pr.record_from_bytes(e).aux_get(b"SA") is not None]
    assert multi, "workload has no read with two lifted records"
    tags = [t for t, _ in multi[0].aux]
    assert tags[-3:] == [b"PS", b"ZM", b"SA"] and b"NM" not in tags and tags.count(b"PS") == 1 and tags.count(b"ZM") == 1
    win.close()
    rd.close()


@pytest.mark.parametrize("is_target_region", [False, True])
def test_records_from_device_finished_batch_are_the_same_bytes(small_bam, oracle, is_target_region):
    """plo_records_build_finished (flags, bin, reversed bases / qualities and SA text handed in from the device-side finishing
    -- here: finish_core.hpp run on the host by the emulator library) against plo_records_build and the Python restatement"""
    import ctypes as C

    import emu_lib
    w, path, meta = small_bam
    ix = w.index_data()
    _, _, recs = bamcheck.read_bam(path)
    prim = [r for r in recs if not (struct.unpack_from("<H", r, 18)[0] & 0x804)]
    rd = bam.BamReader(path, 2)
    win = rd.read_window(100_000)
    b = win.batch_data()
    desc, fin_in = win.batch_desc(with_finish=True)
    n = win.n_records
    read_flags = np.ctypeslib.as_array(fin_in.read_flags, shape=(n,)).copy()
    qual = np.ctypeslib.as_array(fin_in.qual, shape=(int(fin_in.qual_bytes),)).copy()
    qoff = np.ctypeslib.as_array(fin_in.read_qual_off, shape=(n,)).copy()
    res = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 2)
    o, keep = abi.out_from_result(res)
    cn, rn = meta["contig_names"], bamsynth.ref_names(w)
    data, off, n_lift, n_unm = win.build_records(o, ix.to_desc(), cn, rn, is_target_region=is_target_region, n_threads=3)
    want = [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    assert want == expected_records(prim, ix, cn, rn, res, is_target_region)
    f = emu_lib.finish_batch(b, read_flags, qual, qoff, res)
    sa_off, sa_text, _ = emu_lib.sa_segments(b, res, f["item_flag"], f["read_n_lifted"], rn)
    assert (f["item_seq_off"] != abi.NO_FLIP).any() and int(sa_off[-1]) > 0  # flipped records and split reads are in the sample
    ptr = lambda a, t: np.ascontiguousarray(a).ctypes.data_as(C.POINTER(t))
    arrs = {k: np.ascontiguousarray(v) for k, v in f.items()}
    fo = abi.PloFinishOut()
    ct = {np.dtype(np.uint16): C.c_uint16, np.dtype(np.int64): C.c_int64, np.dtype(np.uint8): C.c_uint8, np.dtype(np.uint64): C.c_uint64,
          np.dtype(np.uint32): C.c_uint32}
    for name, dt in abi.FINISH_ITEM_FIELDS + abi.FINISH_READ_FIELDS:
        setattr(fo, name, arrs[name].ctypes.data_as(C.POINTER(ct[np.dtype(dt)])))
    pad = lambda a: a if len(a) else np.zeros(1, a.dtype)
    arrs["rev_seq"], arrs["rev_qual"] = pad(arrs["rev_seq"]), pad(arrs["rev_qual"])
    fo.rev_seq, fo.rev_qual = ptr(arrs["rev_seq"], C.c_uint8), ptr(arrs["rev_qual"], C.c_uint8)
    fo.rev_seq_bytes, fo.rev_qual_bytes = len(f["rev_seq"]), len(f["rev_qual"])
    fo.n_items, fo.n_reads = res.n_items, b.n_reads
    sa_off_c, sa_text_c = np.ascontiguousarray(sa_off, np.uint32), pad(np.ascontiguousarray(sa_text, np.uint8))
    for sa in (abi.PloSaOut(res.n_items, ptr(sa_off_c, C.c_uint32), ptr(sa_text_c, C.c_uint8), int(sa_off[-1]), 0.0), None):
        rb = win.build_records_finished_raw(o, fo, sa, ix.to_desc(), cn, rn, is_target_region=is_target_region, n_threads=3)
        got_bytes = C.string_at(rb.bytes, rb.n_bytes) if rb.n_bytes else b""
        roff = np.ctypeslib.as_array(rb.record_off, shape=(int(rb.n_records) + 1,))
        got = [got_bytes[int(roff[i]):int(roff[i + 1])] for i in range(int(rb.n_records))]
        assert len(got) == len(want)
        for i, (a, e) in enumerate(zip(got, want)):
            assert a == e, (i, sa is None, pr.record_from_bytes(a), pr.record_from_bytes(e))
        assert (int(rb.n_lifted), int(rb.n_unmapped_copies)) == (n_lift, n_unm)
    # an incomplete plo_finish_out is refused
    bad = abi.PloFinishOut()
    with pytest.raises(Exception):
        win.build_records_finished_raw(o, bad, None, ix.to_desc(), cn, rn)
    # ... and so are finished arrays that do not fit the lift result: a wrong record count, an offset beyond the reversed bases
    good = arrs["read_n_lifted"].copy()
    arrs["read_n_lifted"][int(np.argmax(good > 0))] += 1
    with pytest.raises(Exception, match="do not belong"):
        win.build_records_finished_raw(o, fo, None, ix.to_desc(), cn, rn)
    arrs["read_n_lifted"][:] = good
    flipped = int(np.argmax(arrs["item_seq_off"] != abi.NO_FLIP))
    keep_off = int(arrs["item_seq_off"][flipped])
    arrs["item_seq_off"][flipped] = fo.rev_seq_bytes - 1
    with pytest.raises(Exception, match="do not belong"):
        win.build_records_finished_raw(o, fo, None, ix.to_desc(), cn, rn)
    arrs["item_seq_off"][flipped] = keep_off
    # arrays of another batch are refused by their extents before anything is indexed ...
    fo.n_items = res.n_items - 1
    with pytest.raises(Exception, match="do not belong"):
        win.build_records_finished_raw(o, fo, None, ix.to_desc(), cn, rn)
    fo.n_items, fo.n_reads = res.n_items, b.n_reads + 1
    with pytest.raises(Exception, match="do not belong"):
        win.build_records_finished_raw(o, fo, None, ix.to_desc(), cn, rn)
    fo.n_reads = b.n_reads
    # ... a record marked as keeping its bases although the lift says it is flipped (a stale array) is refused ...
    arrs["item_seq_off"][flipped] = abi.NO_FLIP
    with pytest.raises(Exception, match="do not belong"):
        win.build_records_finished_raw(o, fo, None, ix.to_desc(), cn, rn)
    arrs["item_seq_off"][flipped] = keep_off
    win.build_records_finished_raw(o, fo, None, ix.to_desc(), cn, rn)  # (restored: fine again)
    # ... and so is a window whose batch was last built with sparse bases
    win.batch_desc(sparse_margin=32)
    with pytest.raises(Exception, match="sparse"):
        win.build_records_finished_raw(o, fo, None, ix.to_desc(), cn, rn)
    win.close()
    rd.close()


def test_hand_checked_record(tmp_path):
    """one read, two lifted records, every byte of the second written out by hand from the reference's statements"""
    seq, qual = "ACGTTGCAAC", "ABCDEFGHIJ"
    rb = bamsynth.encode_record(0, 99, 37, 0x10, b"r1", np.array(cg.encode("4S6M"), np.uint32), bytes([0x12, 0x48, 0x84, 0x21, 0x12]), 10,
                                bytes(ord(c) - 33 for c in qual), b"NMC\x05" + b"XXZkeep\0" + b"SAZctg0,1,+,4M6S,60,0;\0")
    path = str(tmp_path / "one.bam")
    wr = bam.BamWriter(path, "@HD\tVN:1.6\n", ["ctg0"], [5000], level=0)
    wr.write(rb)
    wr.close()
    rd = bam.BamReader(path, 1)
    win = rd.read_window(10)
    b = win.batch_data()
    assert b.n_segs == 2
    # a hand-made lift result: segment 0 (seq order) -> chr1:1001 6M4S mapq 20, not flipped; segment 1 -> chr2:501 flipped, mapq 50
    res = abi.BatchResult(item_seg=np.array([0, 1], np.uint32), item_cseg=np.array([1, 0], np.uint32), item_status=np.zeros(2, np.uint8),
                          item_need_flipped=np.array([0, 1], np.uint8), item_mapq=np.array([20, 50], np.uint8),
                          item_chrom_index=np.array([0, 1], np.uint32), item_ref_pos=np.array([1000, 500], np.int64),
                          item_cigar_off=np.array([0, 2], np.uint64), item_cigar_len=np.array([2, 4], np.uint32),
                          cigar=np.array(list(cg.encode("6M4S")) + list(cg.encode("4S5M1D1M")), np.uint32))
    ix = abi.IndexData(contig_len=np.array([5000]), contig_seg_off=np.array([0, 2], np.uint32), seg_chrom_index=np.array([0, 1], np.uint32),
                       seg_pos=np.array([0, 0]), seg_is_fwd_strand=np.array([1, 0], np.uint8), seg_mapq=np.array([50, 20], np.uint8),
                       seg_seq_order_start=np.array([0, 2500]), seg_seq_order_end=np.array([2500, 5000]), seg_cigar_off=np.array([0, 1, 2], np.uint32),
                       seg_cigar=np.array(list(cg.encode("2500M")) * 2, np.uint32), chrom_seq=[np.zeros(4000, np.uint8)] * 2, rev_contig_seq=[None])
    o, keep = abi.out_from_result(res)
    data, off, n_lift, n_unm = win.build_records(o, ix.to_desc(), ["ctg0"], ["chr1", "chr2"])
    assert (n_lift, n_unm, len(off) - 1) == (2, 0, 2)
    second = data[int(off[1]):int(off[2])]
    # record 2: flipped -> flag 0x10 ^ 0x10 = 0, primary (mapq 50 > 20) -> no 0x800; pos 500; end 500 + 6 = 506 -> bin 4681 + (500 >> 14)
    # seq = revcomp("ACGTTGCAAC") = "GTTGCAACGT" -> nibbles 4 8 8 4 2 1 1 2 4 8; qual reversed
    body = struct.pack("<iiBBHHHIiii", 1, 500, 3, 50, 4681, 4, 0, 10, -1, -1, 0) + b"r1\0" + \
        np.array(cg.encode("4S5M1D1M"), "<u4").tobytes() + bytes([0x48, 0x84, 0x21, 0x12, 0x48]) + bytes(ord(c) - 33 for c in reversed(qual)) + \
        b"XXZkeep\0" + b"PSZctg0_split0+\0" + b"ZMC\x25" + b"SAZchr1,1001,-,6M4S,20,0;\0"
    assert second == struct.pack("<I", len(body)) + body
    first = pr.record_from_bytes(data[:int(off[1])])
    assert first.flag == 0x10 | 0x800 and first.aux[-1] == (b"SA", b"Zchr2,501,+,4S5M1D1M,50,0;\0") and first.aux[-3][1] == b"Zctg0_split1-\0"
    win.close()
    rd.close()


def test_long_cigar_uses_the_cg_tag(tmp_path):
    """more than 65535 CIGAR ops: bam_write1's <l_seq>S<ref_len>N placeholder + CG:B,I (both directions)"""
    n = 70_000
    l_seq = n  # alternating 1M 1I ... : read bases = n
    cig = np.empty(n, np.uint32)
    cig[0::2] = (1 << 4) | 0
    cig[1::2] = (1 << 4) | 1
    cig[-1] = (1 << 4) | 0
    src = pr.Record(0, 10, 60, 0, 0, -1, -1, 0, b"long", [int(x) for x in cig], bytes((l_seq + 1) // 2), l_seq, bytes(l_seq), [(b"rq", b"f" + struct.pack("<f", 1.0))])
    rb = src.to_bytes()
    assert struct.unpack_from("<H", rb, 16)[0] == 2  # placeholder on disk
    path = str(tmp_path / "long.bam")
    wr = bam.BamWriter(path, "@HD\tVN:1.6\n", ["ctg0"], [500000], level=1)
    wr.write(rb)
    wr.close()
    rd = bam.BamReader(path, 1)
    win = rd.read_window(10)
    b = win.batch_data()
    assert int(b.seg_cigar_off[1]) == n and np.array_equal(b.cigar, cig)  # the real CIGAR reaches the batch
    res = abi.BatchResult(item_seg=np.zeros(1, np.uint32), item_cseg=np.zeros(1, np.uint32), item_status=np.zeros(1, np.uint8),
                          item_need_flipped=np.zeros(1, np.uint8), item_mapq=np.array([33], np.uint8), item_chrom_index=np.zeros(1, np.uint32),
                          item_ref_pos=np.array([77], np.int64), item_cigar_off=np.zeros(1, np.uint64), item_cigar_len=np.array([n], np.uint32),
                          cigar=cig)
    ix = abi.IndexData(contig_len=np.array([500000]), contig_seg_off=np.array([0, 1], np.uint32), seg_chrom_index=np.zeros(1, np.uint32),
                       seg_pos=np.zeros(1), seg_is_fwd_strand=np.ones(1, np.uint8), seg_mapq=np.array([33], np.uint8),
                       seg_seq_order_start=np.zeros(1), seg_seq_order_end=np.array([500000]), seg_cigar_off=np.array([0, 1], np.uint32),
                       seg_cigar=np.array(cg.encode("500000M"), np.uint32), chrom_seq=[np.zeros(10, np.uint8)], rev_contig_seq=[None])
    o, keep = abi.out_from_result(res)
    data, off, _, _ = win.build_records(o, ix.to_desc(), ["ctg0"], ["chr1"])
    exp = pr.finish_remapped_alignment_set(["chr1"], src, [pr.lifted_record(pr.record_from_bytes(rb), "ctg0", 0, True, 0, 33, False, 77, [int(x) for x in cig])], False)
    assert data == exp[0].to_bytes()
    win.close()
    rd.close()


def test_corrupt_and_malformed_inputs(small_bam, tmp_path):
    w, path, meta = small_bam
    raw = bytearray(open(path, "rb").read())
    bad = str(tmp_path / "crc.bam")
    raw2 = bytearray(raw)
    raw2[len(raw2) // 2] ^= 0x5A
    open(bad, "wb").write(raw2)
    with pytest.raises(api.PortelloError) as e:
        rd = bam.BamReader(bad, 2)
        while rd.read_window(50) is not None:
            pass
    assert e.value.status == bam.ERR_IO
    trunc = str(tmp_path / "trunc.bam")
    open(trunc, "wb").write(raw[: len(raw) // 3])
    with pytest.raises(api.PortelloError) as e:
        rd = bam.BamReader(trunc, 2)
        while rd.read_window(50) is not None:
            pass
    assert e.value.status == bam.ERR_IO
    with pytest.raises(api.PortelloError) as e:
        bam.BamReader(str(tmp_path / "missing.bam"))
    assert e.value.status == bam.ERR_IO
    # records the reference panics on (split_read.rs:112-151, sa_tag_parser.rs:27-31) are reported as PLO_ERR_DATA
    for sa in ("ctg0,1,+,10S,60,0;", "ctg0,1,+,5M,60,0;", "nosuch,1,+,5M5S,60,0;", "ctg0,1,+,5M5S,60;", "ctg0,x,+,5M5S,60,0;"):
        p = str(tmp_path / "sa.bam")
        wr = bam.BamWriter(p, "@HD\tVN:1.6\n", ["ctg0"], [1000], level=0)
        wr.write(_sam_record(0, 5, "5S5M", "ACGTACGTAC", "IIIIIIIIII", sa=sa))
        wr.close()
        rd = bam.BamReader(p, 1)
        win = rd.read_window(5)
        with pytest.raises(api.PortelloError) as e:
            win.batch_data()
        assert e.value.status == bam.ERR_DATA, sa
        with pytest.raises((AssertionError, KeyError, ValueError)):
            pr.get_seq_order_read_split_segments({"ctg0": 0}, pr.record_from_bytes(_sam_record(0, 5, "5S5M", "ACGTACGTAC", "IIIIIIIIII", sa=sa)))
        win.close()
        rd.close()


def test_output_header():
    t = bam.output_header(["chr1", "chr2"], [1000, 2000], "portello", "0.6.1", "portello --ref x.fa")
    assert t == "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000\n@SQ\tSN:chr2\tLN:2000\n@PG\tPN:portello\tID:portello-0.6.1\tVN:0.6.1\tCL:portello --ref x.fa\n"


@pytest.mark.gpu
def test_bam_to_bam_through_the_hip_engine(oracle, tmp_path):
    """BAM file -> windows -> plo_liftover_batch (HIP) -> plo_records_build -> BGZF file; the output records are the
    pure-Python expectation computed from the oracle's alignments, byte for byte, and the output file parses"""
    w = synth.generate(synth.config("tiny", n_reads=2500, seed=412, split_read_frac=0.2, sorted_reads=True))
    path = str(tmp_path / "reads.bam")
    meta = bamsynth.write_read_bam(w, path, level=1)
    ix = w.index_data()
    index = api.Index(ix)
    eng = api.Engine(index)
    cn, rn = meta["contig_names"], bamsynth.ref_names(w)
    outp = str(tmp_path / "lifted.bam")
    wr = bam.BamWriter(outp, bam.output_header(rn, [len(s) for s in ix.chrom_seq]), rn, [len(s) for s in ix.chrom_seq], level=0)
    _, _, recs = bamcheck.read_bam(path)
    prim = [r for r in recs if not (struct.unpack_from("<H", r, 18)[0] & 0x804)]
    rd = bam.BamReader(path, 4)
    ixd = ix.to_desc()
    all_out, done = [], 0
    while True:
        win = rd.read_window(700)
        if win is None:
            break
        if win.n_records:
            desc = win.batch_desc()
            lift = eng.liftover_batch_host(desc)
            data, off, _, _ = win.build_records(lift, ixd, cn, rn, n_threads=4)
            res = oracle.liftover_batch(ix, win.batch_data(), abi.STAGES_ALL, 4)
            exp = expected_records(prim[done:done + win.n_records], ix, cn, rn, res)
            assert [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)] == exp
            all_out += exp
            wr.write(data)
            done += win.n_records
        win.close()
    wr.close()
    assert done == w.n_reads
    text, refs, out_recs = bamcheck.read_bam(outp)
    assert out_recs == all_out and text.startswith("@HD\tVN:1.6\tSO:unsorted\n") and [n for n, _ in refs] == rn
    eng.close()
    index.close()


@pytest.mark.gpu
def test_device_inflate_matches_host_inflate(tmp_path, monkeypatch):
    """the BGZF blocks of a chunk inflated on the GPU (one thread per block, portello_amd/csrc/inflate.hpp) give the same
    records as zlib / libdeflate on the host, for stored, fast and well-compressed files"""
    import time

    w = synth.generate(synth.config("tiny", n_reads=1500, seed=413, split_read_frac=0.2, sorted_reads=True))
    for level in (0, 1, 9):
        path = str(tmp_path / f"r{level}.bam")
        bamsynth.write_read_bam(w, path, level=level)
        got = {}
        for dev in ("0", "1"):
            monkeypatch.setenv("PLO_BGZF_DEVICE", dev)
            t0 = time.perf_counter()
            rd = bam.BamReader(path, 4)
            wins = []
            while True:
                win = rd.read_window(400)
                if win is None:
                    break
                b = win.batch_data()
                wins.append((b.n_reads, b.seq.tobytes(), b.cigar.tobytes(), b.seg_pos.tobytes(), win.unmapped_bytes()))
                win.close()
            rd.close()
            got[dev] = wins
            print(f"level {level} device {dev}: {time.perf_counter() - t0:.3f} s")
        assert got["0"] == got["1"] and sum(x[0] for x in got["1"]) == w.n_reads


@pytest.mark.parametrize("level", [0, 1])
def test_device_path_bookkeeping_with_zlib_standing_in_for_the_device(tmp_path, monkeypatch, capfd, level):
    """BgzfIn::fill's device path without a device (PLO_BGZF_TEST_HOST_SLOTS: groups of n blocks, three staging buffers, the next group staged
    while two are "on the device", the next refill prepared during the last waits; zlib inflates a group from its STAGING buffer when its wait
    is called): refills of a few blocks with every group size, whole file and as parts, with and without the preparation, give the records the
    plain host inflate gives.  (The same logic with the real device: test_device_inflate_over_several_refills, -m gpu.)"""
    import hashlib

    w = synth.generate(synth.config("tiny", n_reads=1500, seed=433, split_read_frac=0.2, sorted_reads=True))
    path = str(tmp_path / f"t{level}.bam")
    bamsynth.write_read_bam(w, path, level=level)
    n_blocks = len(bamcheck.bgzf_blocks(path))
    assert n_blocks > 12

    def read_all(device, part=None, n_parts=1, window=157):
        rd = bam.BamReader(path, 3, device_inflate=device, part=part, n_parts=n_parts)
        out = []
        while True:
            win = rd.read_window(window)
            if win is None:
                break
            b = win.batch_data()
            h = hashlib.sha1()
            for a in (b.seq, b.cigar, b.seg_pos, b.read_seq_len):
                h.update(a.tobytes())
            out.append((b.n_reads, h.hexdigest(), win.unmapped_bytes()))
            win.close()
        rd.close()
        return out

    want = read_all(-1)
    want_parts = {n: [read_all(-1, k, n) for k in range(n)] for n in (2, 5)}
    assert sum(x[0] for x in want) == w.n_reads
    capfd.readouterr()
    monkeypatch.setenv("PLO_DEBUG_READER", "1")
    refills = 0
    for slots in (1, 2, 3, 7):
        for chunk in (1, 70_000, 200_000, 1_000_000):  # (1 byte: every refill is one round of `slots` blocks)
            for no_prefetch in (None, "1"):
                monkeypatch.setenv("PLO_BGZF_TEST_HOST_SLOTS", str(slots))
                monkeypatch.setenv("PLO_BGZF_TEST_CHUNK_BYTES", str(chunk))
                if no_prefetch:
                    monkeypatch.setenv("PLO_BGZF_NO_PREFETCH", no_prefetch)
                else:
                    monkeypatch.delenv("PLO_BGZF_NO_PREFETCH", raising=False)
                assert read_all(0) == want, (slots, chunk, no_prefetch)
                assert read_all(0, window=10_000) == read_all(-1, window=10_000), (slots, chunk, no_prefetch)
                for n in (2, 5):
                    assert [read_all(0, k, n) for k in range(n)] == want_parts[n], (slots, chunk, no_prefetch, n)
                err = capfd.readouterr().err
                refills += err.count("[plo] refill:")
                if not no_prefetch and chunk < 1_000_000:
                    assert "groups staged)" in err and any(f"{k} groups staged)" in err for k in (1, 2)), "the refill preparation never ran"
    assert refills > 200  # (the stand-in path ran: a reader that fell back to the host inflate prints no refill lines)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 1])
def test_device_inflate_over_several_refills(tmp_path, monkeypatch, level):
    """A file of several refills through the device inflate: groups of 2 048 blocks on two device slots, the next group staged while they run,
    the NEXT REFILL's headers walked and first groups staged while the last groups are waited for (bam_internal.hpp, BgzfIn::fill).  Stored
    blocks make refills of three groups, compressed ones of two, refills of 64 MB of one; every combination gives the records the host inflate
    gives, whole file and as three parts (a part starts with restart_at, which drops what was prepared)."""
    import hashlib

    w = synth.generate(synth.config("chr20", n_reads=26_000, seed=431), device="cuda")
    path = str(tmp_path / f"big{level}.bam")
    bamsynth.write_read_bam(w, path, level=level, n_threads=8)

    def read_all(device, part=None, n_parts=1):
        rd = bam.BamReader(path, 8, device_inflate=device, part=part, n_parts=n_parts)
        out = []
        while True:
            win = rd.read_window(3000)
            if win is None:
                break
            b = win.batch_data()
            h = hashlib.sha1()
            for a in (b.seq, b.cigar, b.seg_pos, b.read_seq_len):
                h.update(a.tobytes())
            out.append((b.n_reads, h.hexdigest(), win.unmapped_bytes()))
            win.close()
        rd.close()
        return out

    want = read_all(-1)
    assert sum(x[0] for x in want) == w.n_reads
    want_parts = [read_all(-1, k, 3) for k in range(3)]
    assert sum(x[0] for p_ in want_parts for x in p_) == w.n_reads
    for chunk in (None, "64"):
        for no_prefetch in (None, "1"):
            for name, val in (("PLO_BGZF_CHUNK_MB", chunk), ("PLO_BGZF_NO_PREFETCH", no_prefetch)):
                if val is None:
                    monkeypatch.delenv(name, raising=False)
                else:
                    monkeypatch.setenv(name, val)
            assert read_all(0) == want, (level, chunk, no_prefetch)
            assert [read_all(0, k, 3) for k in range(3)] == want_parts, (level, chunk, no_prefetch)


@pytest.mark.gpu
def test_device_crc_rejects_a_block_whose_bytes_or_crc_were_changed(tmp_path, monkeypatch):
    """k_bgzf_crc (a wave per inflated block behind k_bgzf_inflate): a BGZF block whose stored CRC-32 -- or whose data, in a stored block --
    was changed must fail the read with device inflate exactly as on the host, and the untouched file reads the same with the device's check
    as with the host's (PLO_BGZF_HOST_CRC=1)"""
    w = synth.generate(synth.config("tiny", n_reads=600, seed=415, split_read_frac=0.2, sorted_reads=True))
    good = str(tmp_path / "good.bam")
    bamsynth.write_read_bam(w, good, level=0)
    blocks = bamcheck.bgzf_blocks(good)
    assert len(blocks) > 4

    def read_all(path):
        rd = bam.BamReader(path, 4, device_inflate=0)
        n = 0
        while True:
            win = rd.read_window(200)
            if win is None:
                break
            n += win.n_records
            win.close()
        rd.close()
        return n

    n_dev = read_all(good)
    monkeypatch.setenv("PLO_BGZF_HOST_CRC", "1")
    assert read_all(good) == n_dev == w.n_reads
    monkeypatch.delenv("PLO_BGZF_HOST_CRC")
    raw = bytearray(open(good, "rb").read())
    k = len(blocks) // 2
    off, size = sum(b[0] for b in blocks[:k]), blocks[k][0]  # (bgzf_blocks gives sizes: the block's file offset is the sum of those before it)
    for name, at in (("crc", off + size - 8), ("data", off + size - 8 - 100)):  # (stored blocks: a data byte changes the content, not the stream)
        bad = bytearray(raw)
        bad[at] ^= 0x5A
        p = str(tmp_path / f"bad_{name}.bam")
        open(p, "wb").write(bad)
        with pytest.raises(Exception) as ei:
            read_all(p)
        assert "CRC" in str(ei.value) or "crc" in str(ei.value) or "corrupt" in str(ei.value).lower(), str(ei.value)


def test_unmapped_record_placed_on_a_contig_is_a_data_error(tmp_path):
    """flag 0x4 with a reference id: the reference's window loop asserts !record.is_unmapped() (read_alignment_scanner.rs:396);
    only tid = -1 records reach its pass-through copy (:544)"""
    p = str(tmp_path / "placed.bam")
    wr = bam.BamWriter(p, "@HD\tVN:1.6\n", ["ctg0"], [1000], level=0)
    wr.write(_sam_record(0, 5, "10M", "ACGTACGTAC", "IIIIIIIIII"))
    wr.write(bamsynth.encode_record(0, 17, 0, 0x4, b"placed_unmapped", np.zeros(0, np.uint32), bytes(5), 10, b"\x20" * 10, b""))
    wr.close()
    rd = bam.BamReader(p, 1)
    with pytest.raises(api.PortelloError) as e:
        rd.read_window(5)
    assert e.value.status == bam.ERR_DATA
    rd.close()
    # the same record without a reference id is passed through
    p2 = str(tmp_path / "unplaced.bam")
    wr = bam.BamWriter(p2, "@HD\tVN:1.6\n", ["ctg0"], [1000], level=0)
    wr.write(_sam_record(0, 5, "10M", "ACGTACGTAC", "IIIIIIIIII"))
    wr.write(bamsynth.encode_record(-1, -1, 0, 0x4, b"unplaced", np.zeros(0, np.uint32), bytes(5), 10, b"\x20" * 10, b""))
    wr.close()
    rd = bam.BamReader(p2, 1)
    win = rd.read_window(5)
    assert win.n_records == 1 and win.unmapped_bytes()[1] == 1
    win.close()
    rd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("device_finish", [False, True])
def test_bam_to_bam_chr20_size_every_record(tmp_path, device_finish):
    """BASELINE configs[1] size (50 k reads): BAM file in -> pipeline.run_bam_to_bam (reader / two lift workers / writer, the path
    bench.py's end_to_end times) -> BAM file out; the written file, re-read with the independent reader, holds exactly the records
    expected from the oracle's alignments and the Python restatement of the record logic (every window, every record).
    device_finish: flags / bin / primary / reversed bases and qualities / SA text from the device kernels (plo_finish_batch_dev,
    plo_sa_segments_dev), copied into place by plo_records_build_finished"""
    from oracle import expect
    from portello_amd import pipeline

    w = synth.generate(synth.config("chr20", n_reads=50_000), device="cuda")
    inp, outp, unp = str(tmp_path / "reads.bam"), str(tmp_path / "lifted.bam"), str(tmp_path / "unassembled.bam")
    meta = bamsynth.write_read_bam(w, inp, level=1, n_threads=8)
    ixd = w.index_data()
    index = api.Index(w.index_data_device())
    cn, rn = meta["contig_names"], bamsynth.ref_names(w)
    st = pipeline.run_bam_to_bam(inp, outp, index, ixd, cn, rn, [int(s.numel()) for s in w.chrom_seq], window_reads=5000, n_workers=2,
                                 io_threads=8, unassembled_path=unp, device_finish=device_finish)
    assert st.reads == w.n_reads and (st.finish_device_ms > 0) == device_finish
    v = expect.verify_lifted_bam(inp, outp, ixd, cn, rn, window=1000, every=1, threads=8, unassembled_bam=unp)
    assert v["ok"] and v["reads_verified"] == w.n_reads and v["records_verified"] == st.records_out == v["records_in_output"], v
    assert v["unassembled_ok"]
    index.close()


@pytest.mark.gpu
def test_bam_to_bam_into_output_shards(tmp_path):
    """out_shards: the lifted records into three files, one writer thread each (buffered writes into ONE file are serialised by its inode
    lock: the one-file pipeline's bound) -- the shards' union holds exactly the expected records, every shard is a BAM of its own"""
    from oracle import expect
    from portello_amd import pipeline

    w = synth.generate(synth.config("chr20", n_reads=20_000), device="cuda")
    inp, outp, unp = str(tmp_path / "reads.bam"), str(tmp_path / "lifted.bam"), str(tmp_path / "unassembled.bam")
    meta = bamsynth.write_read_bam(w, inp, level=1, n_threads=8)
    ixd = w.index_data()
    index = api.Index(w.index_data_device())
    cn, rn = meta["contig_names"], bamsynth.ref_names(w)
    st = pipeline.run_bam_to_bam(inp, outp, index, ixd, cn, rn, [int(s.numel()) for s in w.chrom_seq], window_reads=1500, n_workers=2,
                                 io_threads=8, unassembled_path=unp, device_finish=True, out_shards=3, n_readers=2)  # (two reader chains: the file cut in two)
    assert st.reads == w.n_reads and len(st.out_paths) == 3 and all(os.path.exists(p_) and os.path.getsize(p_) > 1000 for p_ in st.out_paths)
    assert not os.path.exists(outp)
    v = expect.verify_lifted_bam(inp, st.out_paths, ixd, cn, rn, window=1000, every=1, threads=8, unassembled_bam=unp)
    assert v["ok"] and v["reads_verified"] == w.n_reads and v["records_verified"] == st.records_out == v["records_in_output"], v
    assert v["unassembled_ok"]
    index.close()


def test_long_tail_of_unmapped_reads_comes_in_bounded_windows(tmp_path):
    """a window ends on the count of primary records -- and on 4 x that many (+ 1024) unmapped ones: the unmapped tail of a sorted BAM does
    not end up in one window, and every pass-through record still arrives exactly once, in order"""
    p = str(tmp_path / "tail.bam")
    wr = bam.BamWriter(p, "@HD\tVN:1.6\n", ["ctg0"], [1000], level=1)
    recs = [_sam_record(0, 5, "10M", "ACGTACGTAC", "IIIIIIIIII")]
    tail = [bamsynth.encode_record(-1, -1, 0, 0x4, b"u%06d" % i, np.zeros(0, np.uint32), bytes(5), 10, b"\x20" * 10, b"") for i in range(3000)]
    wr.write(b"".join(recs + tail))
    wr.close()
    rd = bam.BamReader(p, 2)
    got, n_windows, n_prim = b"", 0, 0
    while True:
        win = rd.read_window(100)  # unmapped cap = 4 * 100 + 1024
        if win is None:
            break
        u, k = win.unmapped_bytes()
        assert k <= 4 * 100 + 1024
        got += u
        n_prim += win.n_records
        n_windows += 1
        win.close()
    rd.close()
    assert n_prim == 1 and n_windows == 3 and got == b"".join(tail)


def test_c_caller_sees_windows_without_primaries_before_the_end(tmp_path):
    """through the C ABI alone (no BamReader logic) -- a window with 0 primary records is not the end of the file when the
    unmapped tail is longer than 4 x max_records + 1024; the end is 0 primary AND 0 unmapped records, or plo_bam_window_eof()"""
    p = str(tmp_path / "tail.bam")
    wr = bam.BamWriter(p, "@HD\tVN:1.6\n", ["ctg0"], [1000], level=1)
    recs = [_sam_record(0, 5, "10M", "ACGTACGTAC", "IIIIIIIIII")]
    n_tail = 4 * 50 + 1024 + 700  # > 4 * max_records + 1024 for max_records = 50
    tail = [bamsynth.encode_record(-1, -1, 0, 0x4, b"u%06d" % i, np.zeros(0, np.uint32), bytes(5), 10, b"\x20" * 10, b"") for i in range(n_tail)]
    wr.write(b"".join(recs + tail))
    wr.close()
    L = bam.lib()
    rd = C.c_void_p()
    assert L.plo_bam_open(p.encode(), 2, C.byref(rd)) == 0
    seen, zero_primary_before_end, eof_flags = 0, 0, []
    for _ in range(10):
        h = C.c_void_p()
        assert L.plo_bam_read_window(rd, 50, C.byref(h)) == 0
        nrec = L.plo_bam_window_n_records(h)
        pb, nb, k = C.POINTER(C.c_uint8)(), C.c_uint64(), C.c_uint32()
        L.plo_bam_window_unmapped(h, C.byref(pb), C.byref(nb), C.byref(k))
        eof = L.plo_bam_window_eof(h)
        L.plo_bam_window_free(h)
        eof_flags.append(eof)
        if nrec == 0 and k.value == 0:
            assert eof == 1  # the documented end: 0 primary AND 0 unmapped
            break
        if nrec == 0:
            zero_primary_before_end += 1
        seen += k.value
        if eof:
            break
    L.plo_bam_close(rd)
    assert seen == n_tail and zero_primary_before_end >= 1  # a caller stopping at the first 0-primary window would have lost records
    assert eof_flags[-1] == 1 and not any(eof_flags[:-1])


@pytest.mark.timeout(120)
@pytest.mark.parametrize("device_finish", [False, True])
def test_pipeline_raises_instead_of_hanging_when_a_lift_worker_cannot_start(tmp_path, device_finish):
    """the lift worker's set-up (stream, engine, SA inputs) runs inside its try -- when it fails, the worker records the
    error, sets the abort flag and still posts its sentinel, so run_bam_to_bam raises; before, the writer waited for the sentinel
    forever.  (No GPU needed: an index without a library handle makes the engine's creation fail.)"""
    from portello_amd import pipeline

    w = synth.generate(synth.config("tiny", n_reads=40, seed=3))
    inp, outp = str(tmp_path / "reads.bam"), str(tmp_path / "lifted.bam")
    meta = bamsynth.write_read_bam(w, inp, level=1, n_threads=2)

    class NoIndex:  # what api.Engine / torch.device need is missing
        device = 0
        handle = None

    with pytest.raises(RuntimeError, match="lift worker"):
        pipeline.run_bam_to_bam(inp, outp, NoIndex(), w.index_data(), meta["contig_names"], bamsynth.ref_names(w), [int(s.numel()) for s in w.chrom_seq],
                                window_reads=20, n_workers=2, io_threads=2, device_inflate=False, device_finish=device_finish)


def test_cpu_pipeline_baseline_writes_the_expected_records(tmp_path):
    """oracle/cpu_pipeline.py (bench.py's `cpu_baseline.end_to_end` leg: the host reader / batch / record / writer stages with the oracle
    in place of the engine) writes the same records as the expectation -- the figure it produces is a measurement of a correct run"""
    from oracle import cpu_pipeline, expect

    w = synth.generate(synth.config("tiny", n_reads=150, seed=11, split_read_frac=0.2))
    inp, outp, unp = str(tmp_path / "reads.bam"), str(tmp_path / "lifted.bam"), str(tmp_path / "un.bam")
    meta = bamsynth.write_read_bam(w, inp, level=1, n_threads=2)
    ixd = w.index_data()
    cn, rn = meta["contig_names"], bamsynth.ref_names(w)
    st = cpu_pipeline.run_bam_to_bam_cpu(inp, outp, ixd, cn, rn, [int(s.numel()) for s in w.chrom_seq], window_reads=40, io_threads=4, lift_threads=2,
                                         unassembled_path=unp)
    assert st["reads"] == w.n_reads and st["windows"] >= 3
    v = expect.verify_lifted_bam(inp, outp, ixd, cn, rn, window=50, every=1, threads=2, unassembled_bam=unp)
    assert v["ok"] and v["reads_verified"] == w.n_reads and v["records_verified"] == st["records_out"], v


@pytest.mark.parametrize("level", [1, 0])
def test_parts_of_a_bam_are_disjoint_and_complete(tmp_path, level, odd_names=False):
    """plo_bam_open_range (missing #4): a BAM cut by compressed offset into n parts, every part finding its first BGZF block and its
    first record without an index -- the parts' primary records, in part order, are exactly the file's, each once (records that straddle
    a cut, cuts inside the header, parts without a block of their own, the unmapped tail in the last parts); what the reference does with
    one IndexedReader per worker (src/worker_thread_data.rs:21-30, src/read_alignment_scanner.rs:382)"""
    w = synth.generate(synth.config("tiny", n_reads=400, seed=77, split_read_frac=0.3, sorted_reads=True, read_len_mean=3000, read_len_sd=800))
    path = str(tmp_path / "reads.bam")
    bamsynth.write_read_bam(w, path, level=level, n_unmapped=40, odd_names=odd_names)

    def read_all(**kw):
        rd = bam.BamReader(path, 2, **kw)
        recs, unm = [], b""
        while True:
            win = rd.read_window(37)
            if win is None:
                break
            recs += [win.record_bytes(i) for i in range(win.n_records)]
            unm += win.unmapped_bytes()[0]
            win.close()
        rd.close()
        return recs, unm

    whole, whole_unm = read_all()
    assert len(whole) >= 400 and len(whole_unm) > 0
    size = os.path.getsize(path)
    n_blocks = len(bamcheck.bgzf_blocks(path))
    for n_parts in (1, 2, 3, 7, 16, n_blocks + 5, 4 * n_blocks):
        got, got_unm, per_part = [], b"", []
        for part in range(n_parts):
            r, u = read_all(part=part, n_parts=n_parts)
            got += r
            got_unm += u
            per_part.append(len(r))
        assert got == whole, f"{n_parts} parts: {per_part}"
        assert got_unm == whole_unm
        if n_parts in (2, 3, 7):
            assert sum(1 for k in per_part if k) == n_parts  # (every part of a few has records of its own)
    assert size > 0


def test_parts_of_a_bam_with_names_a_strict_test_would_reject(tmp_path):
    """a part's first record is found with the reader's own checks -- a third of the records carry a blank and a byte beyond
    ASCII in their names (the plain reader lifts them; a record test that wanted printable names broke every chain through them and the
    records in front of the start found behind them were read by no part)"""
    test_parts_of_a_bam_are_disjoint_and_complete(tmp_path, 1, odd_names=True)
