"""TEST INFRASTRUCTURE ONLY (checker): the records a lifted BAM must hold, from the oracle's alignments and the pure-Python
restatement of the reference's per-read loop (oracle/pyrecords.py), and the comparison of a written BAM with them.
Used by tests/ and by bench.py's verification of its end-to-end sample (after the timed run); never by the product path."""
import os
import struct
import sys
from collections import Counter

from oracle import pyrecords as pr
from portello_amd import abi


def expected_records(rec_bytes, ix: abi.IndexData, contig_names, ref_names, res: abi.BatchResult, is_target_region=False):
    """the reference's per-read loop (src/read_alignment_scanner.rs:393-488) in pure Python, lifted alignments taken from `res`;
    also checks the item enumeration (a8) and need_flipped (a9) of `res` against the Python glue"""
    l2i = {n: i for i, n in enumerate(contig_names)}
    out, k, seg_global = [], 0, 0
    for rb in rec_bytes:
        rec = pr.record_from_bytes(rb)
        segs = pr.get_seq_order_read_split_segments(l2i, rec)
        remapped = []
        for seg in segs:
            c = seg.chrom_index
            g0, g1 = int(ix.contig_seg_off[c]), int(ix.contig_seg_off[c + 1])
            csegs = [(int(ix.seg_seq_order_start[g]), int(ix.seg_seq_order_end[g])) for g in range(g0, g1)]
            for cseg in pr.get_contig_split_segments_from_read_mapping(seg, csegs):
                g = g0 + cseg
                assert int(res.item_seg[k]) == seg_global and int(res.item_cseg[k]) == cseg, (k, seg_global, cseg)
                cfwd = bool(ix.seg_is_fwd_strand[g])
                need_flipped, _, _ = pr.strand_glue(rec.is_reverse(), seg, cfwd, int(ix.contig_len[c]))
                assert int(res.item_need_flipped[k]) == int(need_flipped)
                if int(res.item_status[k]) == abi.ITEM_LIFTED:
                    remapped.append(pr.lifted_record(rec, contig_names[c], cseg, cfwd, int(ix.seg_chrom_index[g]), int(ix.seg_mapq[g]),
                                                     need_flipped, int(res.item_ref_pos[k]), [int(x) for x in res.item_cigar(k)]))
                k += 1
            seg_global += 1
        out += pr.finish_remapped_alignment_set(ref_names, rec, remapped, is_target_region)
    assert k == res.n_items
    return [r.to_bytes() for r in out]


def verify_lifted_bam(in_bam, out_bam, ix: abi.IndexData, contig_names, ref_names, window=500, every=1, threads=4, unassembled_bam=None):
    """Re-reads `out_bam` with the independent reader (tests/bamcheck.py) and compares it with the expectation for the primary
    reads of `in_bam`, taken `window` reads at a time, every `every`-th window (1 = all of them): each expected record must be
    in the output as often as expected (the output order is unspecified, docs/user_guide.md:227-230).  With every == 1 the
    record counts must agree as well.  Returns a dict for the bench line / the tests' asserts."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import bamcheck
    from oracle import pyoracle
    from portello_amd import bam

    pyoracle.build()
    # (`out_bam`: one file, or the shards several ranks wrote -- their union is the output, plo_bam_open_range / INTEGRATION.md section 6)
    out_recs = []
    for p_ in ([out_bam] if isinstance(out_bam, (str, bytes)) else list(out_bam)):
        out_recs += bamcheck.read_bam(p_)[2]
    have = Counter(out_recs)
    _, _, in_recs = bamcheck.read_bam(in_bam)
    prim = [r for r in in_recs if not (struct.unpack_from("<H", r, 18)[0] & 0x804)]
    n_unmapped_in = sum(1 for r in in_recs if struct.unpack_from("<H", r, 18)[0] & 0x4)
    rd = bam.BamReader(in_bam, threads)
    done = k = 0
    reads_checked = recs_checked = missing = 0
    want_all = Counter()
    while True:
        win = rd.read_window(window)
        if win is None:
            break
        if win.n_records:
            if k % every == 0:
                res = pyoracle.liftover_batch(ix, win.batch_data(), abi.STAGES_ALL, threads)
                exp = expected_records(prim[done:done + win.n_records], ix, contig_names, ref_names, res)
                want = Counter(exp)
                want_all.update(want)
                reads_checked += win.n_records
                recs_checked += len(exp)
            done += win.n_records
            k += 1
        win.close()
    rd.close()
    for rec, cnt in want_all.items():
        if have.get(rec, 0) < cnt:
            missing += cnt - have.get(rec, 0)
    ok = missing == 0 and done == len(prim)
    if every == 1:
        ok = ok and sum(want_all.values()) == len(out_recs)
    out = {"ok": bool(ok), "reads_verified": reads_checked, "records_verified": recs_checked, "records_missing_or_different": missing,
           "records_in_output": len(out_recs), "windows_of": window, "every_nth_window": every}
    un_paths = [] if unassembled_bam is None else ([unassembled_bam] if isinstance(unassembled_bam, (str, bytes)) else list(unassembled_bam))
    if un_paths and all(os.path.exists(p_) for p_ in un_paths):
        un = []
        for p_ in un_paths:
            un += bamcheck.read_bam(p_)[2]
        out["unassembled_records"] = len(un)
        out["unassembled_ok"] = len(un) >= n_unmapped_in
    return out
