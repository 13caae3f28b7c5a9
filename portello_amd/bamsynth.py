"""Synthetic workloads written as real BAM files (bench / test support; takes no part in the computation).

`write_read_bam` serialises reads of a synth.Workload as the read->contig BAM the reference would be given (pbmm2 style:
primary records with SA:Z tags for split reads, their supplementary records, a few unmapped reads at the end),
`write_contig_bam` the contig->reference BAM of phase 1 (minimap2 --eqx style).  Records are assembled here in numpy
and written through the engine's own BGZF writer (portello_amd/bam.py); tests re-read the files with the independent
pure-Python parser of tests/bamcheck.py.
"""
from __future__ import annotations

import struct
from typing import List, Optional, Tuple

import numpy as np

from . import bam

_OPS = "MIDNSHP=X"


def reg2bin(beg: int, end: int) -> int:
    """SAM specification 5.3"""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def cigar_text(ops: np.ndarray) -> str:
    return "".join(f"{int(c) >> 4}{_OPS[int(c) & 15]}" for c in ops)


def encode_record(tid: int, pos: int, mapq: int, flag: int, qname: bytes, cigar: np.ndarray, seq_packed: bytes, l_seq: int, qual: bytes,
                  aux: bytes, mtid: int = -1, mpos: int = -1, tlen: int = 0) -> bytes:
    ref_len = int(sum(int(c) >> 4 for c in cigar if (0x18D >> (int(c) & 15)) & 1))
    b = reg2bin(max(pos, 0), max(pos, 0) + max(ref_len, 1)) if not (flag & 4) else 4680
    qn = qname + b"\0"
    body = struct.pack("<iiBBHHHIiii", tid, pos, len(qn), mapq, b, len(cigar), flag, l_seq, mtid, mpos, tlen) + qn + \
        np.asarray(cigar, dtype="<u4").tobytes() + seq_packed + qual + aux
    return struct.pack("<I", len(body)) + body


def contig_names(w) -> List[str]:
    return [f"h{1 + (c % 2)}tg{c:06d}l" for c in range(len(w.contig_len))]


def ref_names(w) -> List[str]:
    return [f"chr{i + 1}" for i in range(len(w.chrom_seq))]


def write_read_bam(w, path: str, lo: int = 0, hi: Optional[int] = None, level: int = 1, seed: int = 11, n_unmapped: int = 5,
                   n_threads: int = 8, odd_names: bool = False) -> dict:
    """reads [lo, hi) of the workload as a read->contig BAM.  Returns what a checker needs: per-read qnames / quals / aux."""
    hi = w.n_reads if hi is None else hi
    rng = np.random.default_rng(seed)
    cn = contig_names(w)
    clen = [int(x) for x in w.contig_len]
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(cn, clen)) + "@PG\tID:pbmm2\tPN:pbmm2\n"
    wr = bam.BamWriter(path, text, cn, clen, level=level, n_threads=n_threads)
    b = w.batch_data(lo, hi)
    seg_first = np.searchsorted(b.seg_read, np.arange(b.n_reads), side="left")
    seg_last = np.searchsorted(b.seg_read, np.arange(b.n_reads), side="right")
    chunks: List[bytes] = []
    meta = {"qname": [], "qual": [], "aux": [], "flag": []}
    packed = b.seq_fmt == 0
    for r in range(b.n_reads):
        s0, s1 = int(seg_first[r]), int(seg_last[r])
        l_seq = int(b.read_seq_len[r])
        so = int(b.read_seq_off[r])
        if packed:
            sp = b.seq[so: so + (l_seq + 1) // 2].tobytes()
        else:  # ASCII workload: pack here
            a = b.seq[so: so + l_seq]
            lut = np.full(256, 15, dtype=np.uint8)
            for i_, c_ in enumerate("=ACMGRSVTWYHKDBN"):
                lut[ord(c_)] = i_
            n4 = lut[a]
            if l_seq & 1:
                n4 = np.concatenate([n4, np.zeros(1, np.uint8)])
            sp = ((n4[0::2] << 4) | n4[1::2]).astype(np.uint8).tobytes()
        qual = rng.integers(0, 94, l_seq, dtype=np.uint8).tobytes()
        cig = b.cigar[int(b.seg_cigar_off[s0]): int(b.seg_cigar_off[s0 + 1])]
        flag = 0x10 if b.read_is_reverse[r] else 0
        qname = f"m84011_{lo + r:09d}/ccs".encode()
        if odd_names and r % 3 == 0:  # names a strict record test would not take: a blank, a byte beyond ASCII
            qname = b"m84011 " + bytes([0xC3, 0xA9]) + f"{lo + r:09d}".encode()
        # aux: a mix of kept and removed tags in varying order
        aux = b"rqf" + struct.pack("<f", 0.999) + b"npi" + struct.pack("<i", int(rng.integers(3, 40)))
        style = int(rng.integers(0, 5))
        if style == 1:
            aux = b"PSZ" + b"oldphase\0" + aux
        nm = int(rng.integers(0, 200))
        aux += (b"NMC" + struct.pack("<B", nm)) if style % 2 else (b"NMi" + struct.pack("<i", nm))
        if s1 - s0 > 1:
            sa = ""
            for s in range(s0 + 1, s1):
                c2 = b.cigar[int(b.seg_cigar_off[s]): int(b.seg_cigar_off[s + 1])]
                sa += f"{cn[int(b.seg_contig[s])]},{int(b.seg_pos[s]) + 1},{'+' if b.seg_is_fwd_strand[s] else '-'},{cigar_text(c2)},60,{int(rng.integers(0, 50))};"
            aux += b"SAZ" + sa.encode() + b"\0"
        if style >= 3:
            aux += b"ZMC" + struct.pack("<B", 7)
        if style == 4:
            arr = rng.integers(0, 255, int(rng.integers(1, 40)), dtype=np.uint8)
            aux += b"mlBC" + struct.pack("<I", len(arr)) + arr.tobytes()
        chunks.append(encode_record(int(b.seg_contig[s0]), int(b.seg_pos[s0]), 60, flag, qname, cig, sp, l_seq, qual, aux))
        for s in range(s0 + 1, s1):  # the supplementary records themselves (skipped by the reader: src/read_alignment_scanner.rs:404)
            c2 = b.cigar[int(b.seg_cigar_off[s]): int(b.seg_cigar_off[s + 1])]
            fl2 = 0x800 | (0 if b.seg_is_fwd_strand[s] else 0x10)
            chunks.append(encode_record(int(b.seg_contig[s]), int(b.seg_pos[s]), 60, fl2, qname, c2, sp, l_seq, qual, b"NMC\x01"))
        meta["qname"].append(qname)
        meta["qual"].append(qual)
        meta["aux"].append(aux)
        meta["flag"].append(flag)
        if len(chunks) >= 256:
            wr.write(b"".join(chunks))
            chunks = []
    unm = []
    for k in range(n_unmapped):  # unmapped reads sit at the end of a coordinate-sorted file
        l_seq = 40 + k
        sp = rng.integers(0, 255, (l_seq + 1) // 2, dtype=np.uint8)
        sp = (sp & 0x77 | 0x11).astype(np.uint8)
        if l_seq & 1:
            sp[-1] &= 0xF0
        rec = encode_record(-1, -1, 0, 4, f"unmapped_{k}".encode(), np.zeros(0, np.uint32), sp.tobytes(), l_seq,
                            rng.integers(0, 94, l_seq, dtype=np.uint8).tobytes(), b"rqf" + struct.pack("<f", 0.5))
        chunks.append(rec)
        unm.append(rec)
    wr.write(b"".join(chunks))
    wr.close()
    meta["unmapped_records"] = unm
    meta["contig_names"] = cn
    meta["header_text"] = text
    return meta


# ---- assembly -> reference BAM (input of phase 1) ---------------------------------------------------------------------------

def _read_len(ops) -> int:
    return int(sum(int(c) >> 4 for c in ops if (0x1B3 >> (int(c) & 15)) & 1))


def _ref_len(ops) -> int:
    return int(sum(int(c) >> 4 for c in ops if (0x18D >> (int(c) & 15)) & 1))


def raw_contig_pieces(w, seed: int = 5, perturb: bool = True):
    """The workload's post-phase-1 contig segments taken apart again into what an aligner would have reported: per contig a
    list of pieces (chrom, pos, fwd, mapq, lead clip, aligned ops, trail clip) -- some segments cut in two at a match boundary,
    at an insertion or at a deletion (colinear pieces the joiner must re-join), some cut with an overlap (a repeated match
    the trimmer must clip first)."""
    rng = np.random.default_rng(seed)
    out = []
    for c in range(len(w.contig_len)):
        pieces = []
        for g in range(int(w.contig_seg_off[c]), int(w.contig_seg_off[c + 1])):
            ops = [int(x) for x in w.seg_cigar[int(w.seg_cigar_off[g]): int(w.seg_cigar_off[g + 1])]]
            lead = ops[0] >> 4 if (ops[0] & 15) in (4, 5) else 0
            trail = ops[-1] >> 4 if (ops[-1] & 15) in (4, 5) and len(ops) > 1 else 0
            core = ops[(1 if lead else 0): (len(ops) - 1 if trail else len(ops))]
            base = dict(chrom=int(w.seg_chrom_index[g]), pos=int(w.seg_pos[g]), fwd=bool(w.seg_is_fwd[g]), mapq=int(w.seg_mapq[g]))
            mode = int(rng.integers(0, 4)) if (perturb and len(core) > 12) else 0
            eq = [k for k in range(3, len(core) - 3) if (core[k] & 15) == 7 and (core[k - 1] & 15) != 7]
            if mode == 1 and eq:  # cut at a match boundary
                k = int(rng.choice(eq))
                cuts = [(0, k), (k, len(core))]
            elif mode == 2:  # cut at an indel: the op itself belongs to neither piece
                ind = [k for k in range(3, len(core) - 3) if (core[k] & 15) in (1, 2) and (core[k] >> 4) <= 900 and (core[k - 1] & 15) in (7, 8)
                       and (core[k + 1] & 15) in (7, 8)]
                if ind:
                    k = int(rng.choice(ind))
                    cuts = [(0, k), (k + 1, len(core))]
                else:
                    cuts = [(0, len(core))]
            elif mode == 3 and len(eq) >= 2:  # overlap: ops [k1, k2) reported by both pieces (a repeated match)
                k1, k2 = sorted(int(x) for x in rng.choice(eq, 2, replace=False))
                cuts = [(0, k2), (k1, len(core))] if k2 - k1 < 40 else [(0, len(core))]
            else:
                cuts = [(0, len(core))]
            for a, b in cuts:
                p = dict(base)
                p["pos"] = base["pos"] + _ref_len(core[:a])
                p["lead"] = lead + _read_len(core[:a])
                p["trail"] = trail + _read_len(core[b:])
                p["ops"] = core[a:b]
                pieces.append(p)
        out.append(pieces)
    return out


def write_contig_bam(w, path: str, seed: int = 5, perturb: bool = True, level: int = 1) -> dict:
    """assembly->reference BAM of the workload: per contig one primary record (soft clips, full contig sequence, SA tag listing
    the other pieces with approximate CIGARs as minimap2 writes them) and one supplementary record per other piece (hard clips,
    exact CIGAR)."""
    rng = np.random.default_rng(seed + 1)
    rn, cn = ref_names(w), contig_names(w)
    rl = [int(s.numel()) for s in w.chrom_seq]
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(rn, rl)) + "@PG\tID:minimap2\tPN:minimap2\n"
    pieces = raw_contig_pieces(w, seed, perturb)
    lut = np.full(256, 15, dtype=np.uint8)
    for i_, c_ in enumerate("=ACMGRSVTWYHKDBN"):
        lut[ord(c_)] = i_
    recs = []
    for c, ps in enumerate(pieces):
        if not ps:
            continue
        prim = int(rng.integers(0, len(ps)))
        for k, p in enumerate(ps):
            clip = 4 if k == prim else 5
            cig = ([(p["lead"] << 4) | clip] if p["lead"] else []) + p["ops"] + ([(p["trail"] << 4) | clip] if p["trail"] else [])
            flag = (0 if p["fwd"] else 0x10) | (0 if k == prim else 0x800)
            if k == prim:
                # SEQ of the primary record = the contig in the orientation of that alignment.  The workload keeps only the reverse
                # complement, and only for contigs with a reverse-mapped segment; where it is missing nobody reads the bases
                rcs = w.rev_contig_seq[c]
                if rcs is None:
                    assert p["fwd"], "reverse-mapped primary without rev_contig_seq"
                    seq_ascii = np.full(int(w.contig_len[c]), ord("N"), dtype=np.uint8)
                elif p["fwd"]:
                    from . import synth as _synth

                    seq_ascii = _synth.revcomp(rcs).cpu().numpy()
                else:
                    seq_ascii = rcs.cpu().numpy()
                n4 = lut[seq_ascii]
                if len(n4) & 1:
                    n4 = np.concatenate([n4, np.zeros(1, np.uint8)])
                sp = ((n4[0::2] << 4) | n4[1::2]).astype(np.uint8).tobytes()
                l_seq = len(seq_ascii)
                qual = b"\xff" * l_seq
                sa = ""
                for j, q in enumerate(ps):
                    if j == prim:
                        continue
                    approx = (f"{q['lead']}S" if q["lead"] else "") + f"{_read_len(q['ops'])}M" + (f"{q['trail']}S" if q["trail"] else "")
                    sa += f"{rn[q['chrom']]},{q['pos'] + 1},{'+' if q['fwd'] else '-'},{approx},{q['mapq']},{int(rng.integers(0, 99))};"
                aux = b"NMi" + struct.pack("<i", 12) + ((b"SAZ" + sa.encode() + b"\0") if sa else b"")
            else:
                sp, l_seq, qual, aux = b"", 0, b"", b"NMi" + struct.pack("<i", 3)
            recs.append((p["chrom"], p["pos"], encode_record(p["chrom"], p["pos"], p["mapq"], flag, cn[c].encode(), np.array(cig, np.uint32), sp, l_seq,
                                                             qual, aux)))
    recs.sort(key=lambda t: (t[0], t[1]))
    wr = bam.BamWriter(path, text, rn, rl, level=level)
    wr.write(b"".join(r[2] for r in recs))
    wr.close()
    return {"ref_names": rn, "contig_names": cn, "pieces": pieces, "n_records": len(recs), "records": [r[2] for r in recs]}
