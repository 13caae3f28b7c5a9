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
                   n_threads: int = 8) -> dict:
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
