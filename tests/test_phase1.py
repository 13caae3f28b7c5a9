"""Phase 1 (index construction from the assembly->reference BAM, include/portello_bam.h plo_phase1_*): the reference's own
vectors for the clipping / identity helpers and the repeated-match trimmer, then the C++ implementation against the
pure-Python restatement (oracle/pyphase1.py) on synthetic aligner output, and the round trip workload -> taken-apart BAM ->
phase 1 -> the workload's segments."""
import struct

import numpy as np
import pytest

import bamcheck
from oracle import pyphase1 as p1
from oracle import pyrecords as pr
from portello_amd import abi, api, bam, bamsynth, synth
from portello_amd import cigar as cg


def C(text):
    return [int(x) for x in cg.encode(text)]


def test_clip_alignment_reference_vectors():
    """clip_alignment.rs:188-248"""
    assert p1.clip_alignment_ref_edges(C("3S15M"), 5, 2) == (C("8S8M2S"), 5)
    assert p1.clip_alignment_ref_edges(C("3S2M3D13M"), 5, 2) == (C("5S11M2S"), 5)
    assert p1.clip_alignment_read_edges(C("3S15M"), 5, 2) == (C("5S11M2S"), 2)
    assert p1.clip_alignment_read_edges(C("3S2M3D13M"), 5, 2) == (C("5S11M2S"), 5)
    assert p1.clip_alignment_read_edges(C("3S3I12M"), 5, 2) == (C("6S10M2S"), 0)


def test_gap_compressed_identity():
    """score_alignment.rs:68-74,138-165 (the =/X form of the vector at :182-189: 6 matches, 2 mismatches, one insertion)"""
    assert p1.gap_compressed_identity_no_align_match(C("2=1X1=2I1X3=")) == 6.0 / 9.0
    assert p1.gap_compressed_identity_no_align_match(C("10S")) == 1.0
    with pytest.raises(ValueError):
        p1.gap_compressed_identity_no_align_match(C("4M"))


@pytest.mark.parametrize("fwd,cigar,exp", [(True, "20M", (100, "10M10S", 0, 10)), (False, "20M", (110, "10S10M", 0, 10)),
                                           (True, "5M10I5M", (100, "5M15S", 0, 5)), (False, "5M10I5M", (105, "15S5M", 0, 5))])
def test_clip_seg_isec_range_reference_vectors(fwd, cigar, exp):
    """contig_repeated_match_trimmer.rs:311-397"""
    seg = p1.Seg(0, 20, 0, 100, fwd, C(cigar), 60, False)
    p1.clip_seg_isec_range(seg, (10, 20))
    assert (seg.pos, cg.decode(np.array(seg.cigar, np.uint32)), seg.seq_order_read_start, seg.seq_order_read_end) == exp


def _norm_clip(cigar):
    """S and H clips are interchangeable for the index (ignore_hard_clip = false everywhere)"""
    return [(c & ~15) | 4 if (c & 15) == 5 else c for c in cigar]


def _segments_of(ixd: abi.IndexData):
    out = []
    for c in range(len(ixd.contig_len)):
        row = []
        for g in range(int(ixd.contig_seg_off[c]), int(ixd.contig_seg_off[c + 1])):
            row.append((int(ixd.seg_seq_order_start[g]), int(ixd.seg_seq_order_end[g]), int(ixd.seg_chrom_index[g]), int(ixd.seg_pos[g]),
                        bool(ixd.seg_is_fwd_strand[g]), int(ixd.seg_mapq[g]),
                        _norm_clip([int(x) for x in ixd.seg_cigar[int(ixd.seg_cigar_off[g]):int(ixd.seg_cigar_off[g + 1])]])))
        out.append(row)
    return out


@pytest.fixture(scope="module")
def contig_bam(tmp_path_factory):
    d = tmp_path_factory.mktemp("p1")
    w = synth.generate(synth.config("tiny", n_reads=50, seed=431, chrom_lens=(600_000, 400_000), n_contigs_per_hap=6, max_segments=4))
    path = str(d / "asm.bam")
    meta = bamsynth.write_contig_bam(w, path, seed=9, perturb=True)
    return w, path, meta


def test_phase1_matches_python_restatement_and_restores_the_segments(contig_bam):
    w, path, meta = contig_bam
    cn = meta["contig_names"]
    n_pieces = sum(len(p) for p in meta["pieces"])
    n_segs = int(w.contig_seg_off[-1])
    assert n_pieces > n_segs + 3, "the synthetic aligner output should have cut some segments"
    ph = bam.Phase1(path, cn, [int(x) for x in w.contig_len], n_threads=2)
    assert ph.ref_names == meta["ref_names"] and ph.n_records == meta["n_records"]
    got = ph.index_data([s.numpy() for s in w.chrom_seq])
    # (1) the independent restatement, fed with the records as the independent reader sees them
    _, _, recs = bamcheck.read_bam(path)
    exp = p1.scan_contig_bam(recs, meta["ref_names"], cn)
    exp_rows = [[(s.seq_order_read_start, s.seq_order_read_end, s.chrom_index, s.pos, s.is_fwd_strand, s.mapq, _norm_clip(s.cigar)) for s in segs]
                for segs in exp.contigs]
    assert _segments_of(got) == exp_rows
    assert (ph.segments_clipped, ph.segments_joined) == (exp.segments_clipped, exp.segments_joined)
    assert ph.segments_joined >= n_pieces - n_segs and ph.segments_clipped > 0
    for c in range(len(cn)):
        a, b = got.rev_contig_seq[c], exp.rev_contig_seq[c]
        assert (a is None) == (b is None) and (a is None or a.tobytes() == b)
    # (2) taking segments apart (cuts at matches / indels, overlaps) changes nothing: the result equals phase 1 of the
    # workload's segments reported whole (the joiner may also merge neighbouring segments of the workload itself when they
    # happen to be colinear within 1 kb with equal MAPQ -- both sides go through the same rule)
    whole = bamsynth.write_contig_bam(w, path + ".whole.bam", seed=9, perturb=False)
    exp_whole = p1.scan_contig_bam(whole["records"], meta["ref_names"], cn)
    assert _segments_of(got) == [[(s.seq_order_read_start, s.seq_order_read_end, s.chrom_index, s.pos, s.is_fwd_strand, s.mapq, _norm_clip(s.cigar))
                                  for s in segs] for segs in exp_whole.contigs]
    untouched = [c for c in range(len(cn)) if len(exp_whole.contigs[c]) == int(w.contig_seg_off[c + 1] - w.contig_seg_off[c])]
    assert len(untouched) >= len(cn) // 2
    ws = _segments_of(w.index_data())
    gs = _segments_of(got)
    for c in untouched:  # contigs whose own segments are not joinable come back exactly as the workload has them
        assert gs[c] == ws[c]
    for c in range(len(cn)):
        if w.rev_contig_seq[c] is not None:
            assert got.rev_contig_seq[c].tobytes() == w.rev_contig_seq[c].numpy().tobytes()
    ph.close()


def test_phase1_hand_made_overlap_prefers_identity_then_mapq(tmp_path):
    """two alignments of one contig overlapping on contig bases [40, 60): the one with the lower gap-compressed identity over
    the overlap is clipped; equal identity -> the lower MAPQ; all equal -> the later one (trimmer.rs:186-206)"""
    rn, cn = ["chr1"], ["ctg"]

    def run(cig1, mq1, cig2, mq2):
        recs = [bamsynth.encode_record(0, 1000, mq1, 0, b"ctg", np.array(C(cig1), np.uint32), bytes(50), 100, b"\xff" * 100,
                                       b"SAZchr1,5001,+,40S60M,%d,0;\0" % mq2),
                bamsynth.encode_record(0, 5000, mq2, 0x800, b"ctg", np.array(C(cig2), np.uint32), b"", 0, b"", b"")]
        path = str(tmp_path / "o.bam")
        wr = bam.BamWriter(path, "@HD\tVN:1.6\n", rn, [100000], level=0)
        wr.write(b"".join(recs))
        wr.close()
        ph = bam.Phase1(path, cn, [100], n_threads=1)
        got = _segments_of(ph.index_data([np.zeros(100000, np.uint8)]))[0]
        exp = p1.scan_contig_bam(recs, rn, cn)
        assert got == [(s.seq_order_read_start, s.seq_order_read_end, s.chrom_index, s.pos, s.is_fwd_strand, s.mapq, _norm_clip(s.cigar))
                       for s in exp.contigs[0]]
        ph.close()
        return [(g[0], g[1], g[3], cg.decode(np.array(g[6], np.uint32))) for g in got]

    # equal identity and MAPQ: the second (later in sequencing order) loses its prefix
    assert run("60=40S", 60, "40H60=", 60) == [(0, 60, 1000, "60=40S"), (60, 100, 5020, "40S20S40=")]  # i.e. 40H20S40=: compress_cigar does not merge H with S
    # the first one has a mismatch inside the overlap: it loses its suffix
    assert run("50=1X9=40S", 60, "40H60=", 60) == [(0, 40, 1000, "40=60S"), (40, 100, 5000, "40S60=")]
    # equal identity, second has the higher MAPQ: the first is clipped
    assert run("60=40S", 20, "40H60=", 60) == [(0, 40, 1000, "40=60S"), (40, 100, 5000, "40S60=")]


def test_phase1_errors(tmp_path):
    rn, cn = ["chr1"], ["ctg"]

    def scan(recs, names=cn):
        path = str(tmp_path / "e.bam")
        wr = bam.BamWriter(path, "@HD\tVN:1.6\n", rn, [100000], level=0)
        wr.write(b"".join(recs))
        wr.close()
        return bam.Phase1(path, names, [100] * len(names), n_threads=1)

    prim = bamsynth.encode_record(0, 1000, 60, 0, b"ctg", np.array(C("60=40S"), np.uint32), bytes(50), 100, b"\xff" * 100,
                                  b"SAZchr1,5001,+,40S60M,60,0;\0")
    with pytest.raises(api.PortelloError) as e:  # the supplementary record of the SA entry is missing (mod.rs:398-416)
        scan([prim])
    assert e.value.status == abi.PLO_ERR_DATA
    supp = bamsynth.encode_record(0, 5000, 60, 0x800, b"ctg", np.array(C("40H60="), np.uint32), b"", 0, b"", b"")
    with pytest.raises(api.PortelloError) as e:  # two supplementary records with the same key (:161-182)
        scan([prim, supp, supp])
    assert e.value.status == abi.PLO_ERR_DATA
    with pytest.raises(api.PortelloError) as e:  # contig unknown to the read->contig BAM
        scan([prim, supp], names=["other"])
    assert e.value.status == abi.PLO_ERR_DATA
    m_supp = bamsynth.encode_record(0, 5000, 60, 0x800, b"ctg", np.array(C("40H60M"), np.uint32), b"", 0, b"", b"")
    with pytest.raises(api.PortelloError) as e:  # an overlap whose identity needs =/X ops (score_alignment.rs:152-156)
        scan([prim, m_supp])
    assert e.value.status == abi.PLO_ERR_DATA
    # with a target region a missing supplementary record is tolerated and segments starting outside the region are dropped
    path = str(tmp_path / "t.bam")
    wr = bam.BamWriter(path, "@HD\tVN:1.6\n", rn, [100000], level=0)
    wr.write(prim)
    wr.close()
    ph = bam.Phase1(path, cn, [100], target_region=(0, 900, 1100), n_threads=1)
    assert [(g[0], g[1], g[3]) for g in _segments_of(ph.index_data([np.zeros(100000, np.uint8)]))[0]] == [(0, 60, 1000)]
    ph.close()


@pytest.mark.gpu
def test_bams_in_lifted_bam_out(oracle, tmp_path):
    """both phases from files: assembly->reference BAM -> plo_phase1_scan -> plo_index_create; read->contig BAM -> windows ->
    HIP liftover -> record bytes; the records equal the Python expectation built on the workload's own index"""
    import struct as st

    from test_bam import expected_records

    w = synth.generate(synth.config("tiny", n_reads=800, seed=432, split_read_frac=0.2, sorted_reads=True))
    asm, reads = str(tmp_path / "asm.bam"), str(tmp_path / "reads.bam")
    m1 = bamsynth.write_contig_bam(w, asm, seed=3)
    m2 = bamsynth.write_read_bam(w, reads, level=1)
    rd = bam.BamReader(reads, 2)
    ph = bam.Phase1(asm, rd.ref_names, rd.ref_lens, n_threads=2)
    ix = ph.index_data([s.numpy() for s in w.chrom_seq])
    index = api.Index(ix)
    eng = api.Engine(index)
    ixd = ix.to_desc()
    _, _, recs = bamcheck.read_bam(reads)
    prim = [r for r in recs if not (st.unpack_from("<H", r, 18)[0] & 0x804)]
    win = rd.read_window(10_000)
    lift = eng.liftover_batch_host(win.batch_desc())
    data, off, _, _ = win.build_records(lift, ixd, rd.ref_names, ph.ref_names)
    ref_ix = w.index_data()
    res = oracle.liftover_batch(ref_ix, win.batch_data(), abi.STAGES_ALL, 4)
    exp = expected_records(prim, ref_ix, m2["contig_names"], m1["ref_names"], res)
    assert [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)] == exp
    win.close()
    rd.close()
    ph.close()
    eng.close()
    index.close()


def test_phase1_supplementary_record_with_long_cigar_in_cg_tag(tmp_path):
    """a supplementary assembly->reference alignment of more than 65535 ops arrives as <l_seq>S<n>N + CG:B,I; htslib hands the
    reference the real CIGAR for every record, so the record's key matches the SA-derived one and its CIGAR replaces the SA tag's
    approximate one (contig_alignment_scanner/mod.rs:135-183, 371-416)"""
    rn, cn = ["chr1"], ["ctg"]
    n_pairs = 35000
    real = np.array([(40 << 4) | 5] + [(1 << 4) | 7, (1 << 4) | 8] * n_pairs, np.uint32)  # 40H (1= 1X) x 35000: 70 001 ops
    l_seq = 2 * n_pairs
    placeholder = np.array([(l_seq << 4) | 4, (l_seq << 4) | 3], np.uint32)
    cg_tag = b"CGBI" + struct.pack("<I", len(real)) + real.astype("<u4").tobytes()
    recs = [bamsynth.encode_record(0, 1000, 60, 0, b"ctg", np.array(C("40=%dS" % l_seq), np.uint32), bytes((40 + l_seq + 1) // 2), 40 + l_seq,
                                   b"\xff" * (40 + l_seq), b"SAZchr1,5001,+,40S%dM,60,0;\0" % l_seq),
            bamsynth.encode_record(0, 5000, 60, 0x800, b"ctg", placeholder, bytes(l_seq // 2), l_seq, b"\xff" * l_seq, cg_tag)]
    path = str(tmp_path / "long.bam")
    wr = bam.BamWriter(path, "@HD\tVN:1.6\n", rn, [200000], level=1)
    wr.write(b"".join(recs))
    wr.close()
    ph = bam.Phase1(path, cn, [40 + l_seq], n_threads=1)
    got = _segments_of(ph.index_data([np.zeros(200000, np.uint8)]))[0]
    exp = p1.scan_contig_bam(recs, rn, cn)
    assert got == [(s.seq_order_read_start, s.seq_order_read_end, s.chrom_index, s.pos, s.is_fwd_strand, s.mapq, _norm_clip(s.cigar))
                   for s in exp.contigs[0]]
    assert len(got) == 2 and len(got[1][6]) > 65535  # the supplementary segment carries the record's own 70 001-op CIGAR
    ph.close()


def test_phase1_target_region_unmatched_sa_segment_has_no_map(tmp_path):
    """targeted run, SA segment inside the region whose supplementary record is missing: the reference keeps the segment (it still
    takes part in the read -> contig segment selection) but leaves its contig_to_ref_map empty (mod.rs:396-414), so nothing lifts
    through the SA tag's approximate CIGAR; a segment the trimmer clips gets its map rebuilt from that CIGAR (trimmer.rs:130-134)"""
    rn, cn = ["chr1"], ["ctg"]

    def scan(prim_cigar, sa):
        prim = bamsynth.encode_record(0, 1000, 60, 0, b"ctg", np.array(C(prim_cigar), np.uint32), bytes(50), 100, b"\xff" * 100, sa)
        path = str(tmp_path / "t2.bam")
        wr = bam.BamWriter(path, "@HD\tVN:1.6\n", rn, [100000], level=0)
        wr.write(prim)
        wr.close()
        ph = bam.Phase1(path, cn, [100], target_region=(0, 900, 1100), n_threads=1)
        ixd = ph.index_data([np.zeros(100000, np.uint8)])
        got = _segments_of(ixd)[0]
        exp = p1.scan_contig_bam([prim], rn, cn, target_region=(0, 900, 1100))
        ph.close()
        return got, exp.contigs[0]

    # no overlap between the two segments: the SA segment stays as it is, without a map (empty CIGAR in the index description)
    got, exp = scan("60=40S", b"SAZchr1,1051,+,60S40M,60,0;\0")
    assert [(g[0], g[1], g[3], len(g[6])) for g in got] == [(0, 60, 1000, 2), (60, 100, 1050, 0)]
    assert [s.no_map for s in exp] == [False, True]
    # overlapping on [50, 60), equal identity / MAPQ: the later (SA) segment is clipped -> its map is rebuilt from the clipped CIGAR
    # (trimmer.rs:130-134); here the clipped segment then continues the first one exactly and the joiner merges the two
    got, exp = scan("60=40S", b"SAZchr1,1051,+,50S50=,60,0;\0")
    assert [(g[0], g[1], g[3]) for g in got] == [(0, 100, 1000)] and cg.decode(np.array(got[0][6], np.uint32)) == "60=40="
    assert [(s.seq_order_read_start, s.seq_order_read_end, s.pos, s.no_map) for s in exp] == [(0, 100, 1000, False)]
    # the same overlap but on the other strand (no join): the clipped SA segment stays a segment of its own, WITH a map
    got, exp = scan("60=40S", b"SAZchr1,1051,-,50=50S,60,0;\0")
    assert [(g[0], g[1]) for g in got] == [(0, 60), (60, 100)] and len(got[1][6]) > 0
    assert [s.no_map for s in exp] == [False, False]
