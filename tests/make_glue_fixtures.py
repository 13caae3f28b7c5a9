#!/usr/bin/env python3
"""Generates tests/golden/glue_fixtures.json: golden vectors for the parts of the path the reference's own tests do not pin --
segment selection (a8), strand glue (a9), record finishing and SA text -- computed by the PURE-PYTHON restatement
oracle/pyrecords.py (written from the Rust, independently of the C oracle and of the engine).  Both the C oracle
(tests, CPU) and the HIP engine (tests, -m gpu) are then checked against these vectors.  Run in the build container:

    python tests/make_glue_fixtures.py

Inputs are seeded synthetic workloads, stored in the fixture in compact form; the lifted alignments that record finishing
starts from are part of the fixture's INPUT (they come from the C oracle's liftover, whose own parity is pinned by the
reference's vectors)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from oracle import pyrecords as pr  # noqa: E402
from portello_amd import abi, synth  # noqa: E402
from portello_amd import cigar as cg  # noqa: E402


def case(seed, n_reads, **over):
    w = synth.generate(synth.config("tiny", n_reads=n_reads, seed=seed, split_read_frac=0.35, **over))
    ix, b = w.index_data(), w.batch_data()
    ref_names = [f"chr{i + 1}" for i in range(len(ix.chrom_seq))]
    items = []
    for s in range(b.n_segs):
        cig = [int(x) for x in b.cigar[int(b.seg_cigar_off[s]):int(b.seg_cigar_off[s + 1])]]
        r, c = int(b.seg_read[s]), int(b.seg_contig[s])
        seg = pr.SeqOrderSplitReadSegment(0, 0, c, int(b.seg_pos[s]), bool(b.seg_is_fwd_strand[s]), cig, 60, False)
        g0, g1 = int(ix.contig_seg_off[c]), int(ix.contig_seg_off[c + 1])
        cs = [(int(ix.seg_seq_order_start[g]), int(ix.seg_seq_order_end[g])) for g in range(g0, g1)]
        for cseg in pr.get_contig_split_segments_from_read_mapping(seg, cs):
            nf, pos, c2 = pr.strand_glue(bool(b.read_is_reverse[r]), seg, bool(ix.seg_is_fwd_strand[g0 + cseg]), int(ix.contig_len[c]))
            items.append({"seg": s, "cseg": cseg, "need_flipped": int(nf), "pos": pos, "cigar": cg.decode(np.array(c2, np.uint32))})
    # record finishing on top of given lifted alignments
    lift = pyoracle.liftover_batch(ix, b, abi.STAGES_ALL, 1)
    assert lift.n_items == len(items)
    rng = np.random.default_rng(seed)
    read_flags = [(0x10 if b.read_is_reverse[r] else 0) | (0x400 if rng.random() < 0.3 else 0) for r in range(b.n_reads)]
    lifted_in, fin_items, reads = [], [], []
    k = 0
    for r in range(b.n_reads):
        recs, idx = [], []
        while k < lift.n_items and int(b.seg_read[int(lift.item_seg[k])]) == r:
            st = int(lift.item_status[k])
            lifted_in.append({"status": st, "need_flipped": int(lift.item_need_flipped[k]), "mapq": int(lift.item_mapq[k]),
                              "chrom": int(lift.item_chrom_index[k]), "pos": int(lift.item_ref_pos[k]),
                              "cigar": cg.decode(lift.item_cigar(k)) if st == 0 else ""})
            if st == 0:
                rec = pr.Record(0, 0, 0, 0, read_flags[r], -1, -1, 0, b"q", [], b"", 0, b"", [])
                x = rec.clone()  # the statements of :245-284 that touch flags / pos / cigar / bin
                x.tid, x.mapq, x.pos, x.cigar = int(lift.item_chrom_index[k]), int(lift.item_mapq[k]), int(lift.item_ref_pos[k]), [int(c) for c in lift.item_cigar(k)]
                if lift.item_need_flipped[k]:
                    x.flag ^= pr.BAM_FREVERSE
                end = x.pos + pr.cigar_ref_offset(x.cigar)
                x.bin = pr.hts_reg2bin(x.pos, end) & 0xFFFF
                x.flag |= pr.BAM_FSUPPLEMENTARY
                recs.append(x)
                idx.append(k)
            k += 1
        out = pr.finish_remapped_alignment_set(ref_names, pr.Record(-1, -1, 0, 0, read_flags[r], -1, -1, 0, b"q", [], b"", 0, b"", []), recs, False)
        if recs:
            for j, x in zip(idx, out):
                sa = x.aux_get(b"SA")
                fin_items.append({"item": j, "flag": x.flag, "bin": x.bin, "ref_end": x.pos + pr.cigar_ref_offset(x.cigar),
                                  "is_primary": int(not (x.flag & pr.BAM_FSUPPLEMENTARY)), "sa": sa[1:-1].decode() if sa else None})
            reads.append({"n_lifted": len(recs), "primary_item": next(j for j, x in zip(idx, out) if not (x.flag & pr.BAM_FSUPPLEMENTARY)), "unmapped_flag": None})
        else:
            reads.append({"n_lifted": 0, "primary_item": None, "unmapped_flag": out[0].flag})
    return {"config": {"name": "tiny", "seed": seed, "n_reads": n_reads, "split_read_frac": 0.35, **over}, "ref_names": ref_names,
            "read_flags": read_flags, "items": items, "lifted": lifted_in, "finish_items": fin_items, "finish_reads": reads}


if __name__ == "__main__":
    pyoracle.build()
    cases = [case(601, 60), case(602, 60, rev_contig_frac=1.0), case(603, 40, max_segments=5, n_contigs_per_hap=3)]
    out = os.path.join(ROOT, "tests", "golden", "glue_fixtures.json")
    with open(out, "w") as fh:
        json.dump({"generator": "tests/make_glue_fixtures.py (oracle/pyrecords.py, pure Python from the Rust sources)", "cases": cases}, fh, indent=0,
                  separators=(",", ":"))
    print(out, os.path.getsize(out), "bytes;", sum(len(c["items"]) for c in cases), "items")
