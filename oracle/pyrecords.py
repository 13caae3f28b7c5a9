"""oracle/pyrecords.py -- pure-Python restatement of the reference's per-record logic around the hot path.

TEST INFRASTRUCTURE ONLY (never imported by the product).  Written from the Rust sources, operation by operation, on a
small in-memory model of a BAM record (core fields + an ordered list of aux fields), independently of the C oracle
(oracle/portello_oracle.c) and of the product's C++ (portello_amd/csrc/bam_host.cpp):

  get_read_clip_positions                 lib/rust-vc-utils/src/bam_utils/cigar/mod.rs:85-118
  parse_sa_segment / parse_sa_aux_val     lib/rust-vc-utils/src/bam_utils/aux/sa_tag_parser.rs:25-59
  get_seq_order_read_split_segments       lib/rust-vc-utils/src/bam_utils/split_read.rs:56-155
  get_contig_split_segments_from_read_mapping, the strand glue of get_liftover_alignment_for_read_and_contig_segment
                                          src/read_alignment_scanner.rs:80-103, 149-176
  clone_record, record stamping, reverse_alignment_seq_and_qual, get_sa_tag_segment, finish_remapped_alignment_set
                                          src/read_alignment_scanner.rs:105-133, 245-284, 292-366
  bam_reg2bin / get_alignment_end         lib/rust-vc-utils/src/bam_utils/util.rs:10-35, bam_record_utils.rs:21-27

Third-party behaviour restated from its published semantics (rust-htslib 0.50.0 / htslib, absent from the reference
tree; parity unpinned): Record::aux / remove_aux act on the FIRST field with the tag; push_aux appends; Aux::U8 is
type 'C', Aux::String type 'Z'; Record::set re-encodes bases with A=1 C=2 G=4 T=8 N=15; bam_write1 writes
block_size, the 32 fixed bytes, qname (NUL terminated), CIGAR, packed bases, qualities, aux -- and for more than
65535 CIGAR ops the placeholder <l_seq>S<ref_len>N with the real CIGAR in a trailing CG:B,I field.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

OPS = "MIDNSHP=X"
M, I, D, N, S, H, P, EQ, X = range(9)
BAM_FREVERSE, BAM_FUNMAP, BAM_FSUPPLEMENTARY = 0x10, 0x4, 0x800
_DECODE = "=ACMGRSVTWYHKDBN"
_ENCODE = {c: i for i, c in enumerate(_DECODE)}


# ---- CIGAR helpers ------------------------------------------------------------------------------------------------------
def op(c: int) -> Tuple[int, int]:
    return c & 15, c >> 4


def cigarseg_read_offset(c: int, ignore_hard_clip: bool) -> int:  # cigar/mod.rs:26-39
    t, l = op(c)
    if t in (I, S, X, EQ, M):
        return l
    if t == H:
        return 0 if ignore_hard_clip else l
    return 0


def cigarseg_ref_offset(c: int) -> int:  # :41-47
    t, l = op(c)
    return l if t in (D, N, X, EQ, M) else 0


def cigar_ref_offset(cigar: Sequence[int]) -> int:  # :174-180
    return sum(cigarseg_ref_offset(c) for c in cigar)


def is_alignment_match(c: int) -> bool:  # :22-24
    return (c & 15) in (M, EQ, X)


def get_read_clip_positions(cigar: Sequence[int], ignore_hard_clip: bool) -> Tuple[int, int, int]:  # :85-118
    read_pos = 0
    left_clip_size = right_clip_size = 0
    left_clip = True
    for c in cigar:
        t, l = op(c)
        if t == S:
            if left_clip:
                left_clip_size += l
            else:
                right_clip_size += l
        elif t == H:
            if not ignore_hard_clip:
                if left_clip:
                    left_clip_size += l
                else:
                    right_clip_size += l
        else:
            left_clip = False
        read_pos += cigarseg_read_offset(c, ignore_hard_clip)
    return left_clip_size, read_pos - right_clip_size, read_pos


def cigar_from_text(s: str) -> List[int]:
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            if not num or ch not in OPS:
                raise ValueError(f"bad CIGAR text {s!r}")
            out.append((int(num) << 4) | OPS.index(ch))
            num = ""
    if num:
        raise ValueError(f"bad CIGAR text {s!r}")
    return out


def cigar_to_text(cigar: Sequence[int]) -> str:  # rust-htslib Display of CigarString
    return "".join(f"{c >> 4}{OPS[c & 15]}" for c in cigar)


# ---- record model -------------------------------------------------------------------------------------------------------
@dataclass
class Record:
    tid: int
    pos: int
    mapq: int
    bin: int
    flag: int
    mtid: int
    mpos: int
    tlen: int
    qname: bytes          # without the NUL
    cigar: List[int]
    seq4: bytes           # packed bases
    l_seq: int
    qual: bytes
    aux: List[Tuple[bytes, bytes]] = field(default_factory=list)  # (tag, type + value bytes), in file order

    def is_reverse(self) -> bool:
        return bool(self.flag & BAM_FREVERSE)

    def clone(self) -> "Record":
        return Record(self.tid, self.pos, self.mapq, self.bin, self.flag, self.mtid, self.mpos, self.tlen, self.qname, list(self.cigar),
                      self.seq4, self.l_seq, self.qual, list(self.aux))

    def seq_as_bytes(self) -> bytes:  # record.seq().as_bytes()
        out = bytearray(self.l_seq)
        for j in range(self.l_seq):
            b = self.seq4[j >> 1]
            out[j] = ord(_DECODE[(b & 15) if (j & 1) else (b >> 4)])
        return bytes(out)

    def aux_get(self, tag: bytes) -> Optional[bytes]:
        for t, v in self.aux:
            if t == tag:
                return v
        return None

    def remove_aux_if_found(self, tag: bytes):  # aux/mod.rs:99-103: the first field with the tag
        for k, (t, _) in enumerate(self.aux):
            if t == tag:
                del self.aux[k]
                return

    def push_aux_string(self, tag: bytes, s: str):
        self.aux.append((tag, b"Z" + s.encode() + b"\0"))

    def push_aux_u8(self, tag: bytes, v: int):
        self.aux.append((tag, b"C" + bytes([v])))

    def set_seq_qual(self, seq_ascii: bytes, qual: bytes):  # Record::set: bases re-encoded, odd tail nibble 0
        n = len(seq_ascii)
        out = bytearray((n + 1) // 2)
        for j in range(n):
            code = _ENCODE.get(chr(seq_ascii[j]).upper(), 15)
            out[j >> 1] |= code << (0 if (j & 1) else 4)
        self.seq4, self.l_seq, self.qual = bytes(out), n, bytes(qual)

    def to_bytes(self) -> bytes:  # bam_write1
        aux = b"".join(t + v for t, v in self.aux)
        cig = self.cigar
        trailer = b""
        n_cigar = len(cig)
        if n_cigar > 0xFFFF:
            ref_len = cigar_ref_offset(cig)
            trailer = b"CGBI" + struct.pack("<I", n_cigar) + b"".join(struct.pack("<I", c) for c in cig)
            cig = [(self.l_seq << 4) | S, (ref_len << 4) | N]
        qn = self.qname + b"\0"
        body = struct.pack("<iiBBHHHIiii", self.tid, self.pos, len(qn), self.mapq, self.bin, len(cig), self.flag, self.l_seq, self.mtid,
                           self.mpos, self.tlen) + qn + b"".join(struct.pack("<I", c) for c in cig) + self.seq4 + self.qual + aux + trailer
        return struct.pack("<I", len(body)) + body


def _aux_fields(b: bytes) -> List[Tuple[bytes, bytes]]:
    out, i = [], 0
    fixed = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4, "d": 8}
    while i < len(b):
        tag, ty = b[i:i + 2], chr(b[i + 2])
        if ty in fixed:
            n = fixed[ty]
        elif ty in "ZH":
            n = b.index(b"\0", i + 3) - (i + 3) + 1
        elif ty == "B":
            n = 5 + fixed[chr(b[i + 3])] * struct.unpack_from("<I", b, i + 4)[0]
        else:
            raise ValueError("bad aux type")
        out.append((tag, b[i + 2:i + 3 + n]))
        i += 3 + n
    return out


def record_from_bytes(b: bytes) -> Record:
    """one BAM record (block_size prefixed) as htslib hands it to the reference (CG:B,I long CIGARs restored)"""
    (bs,) = struct.unpack_from("<I", b, 0)
    tid, pos, lq, mapq, bin_, nc, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHIiii", b, 4)
    o = 36
    qname = b[o:o + lq - 1]
    o += lq
    cigar = list(struct.unpack_from(f"<{nc}I", b, o))
    o += 4 * nc
    seq4 = b[o:o + (l_seq + 1) // 2]
    o += (l_seq + 1) // 2
    qual = b[o:o + l_seq]
    o += l_seq
    aux = _aux_fields(b[o:4 + bs])
    if nc == 2 and cigar[0] == ((l_seq << 4) | S) and (cigar[1] & 15) == N:
        for k, (t, v) in enumerate(aux):
            if t == b"CG" and v[:2] == b"BI":
                n = struct.unpack_from("<I", v, 2)[0]
                cigar = list(struct.unpack_from(f"<{n}I", v, 6))
                del aux[k]
                break
    return Record(tid, pos, mapq, bin_, flag, mtid, mpos, tlen, qname, cigar, seq4, l_seq, qual, aux)


def split_records(stream: bytes) -> List[bytes]:
    out, i = [], 0
    while i < len(stream):
        (bs,) = struct.unpack_from("<I", stream, i)
        out.append(stream[i:i + 4 + bs])
        i += 4 + bs
    return out


# ---- SA tag / split segments -----------------------------------------------------------------------------------------------
@dataclass
class SeqOrderSplitReadSegment:  # split_read.rs:15-32
    seq_order_read_start: int
    seq_order_read_end: int
    chrom_index: int
    pos: int
    is_fwd_strand: bool
    cigar: List[int]
    mapq: int
    from_primary_bam_record: bool


def _split_terminator(s: str, sep: str) -> List[str]:  # Rust str::split_terminator
    parts = s.split(sep)
    if parts and parts[-1] == "":
        parts.pop()
    return parts


def parse_sa_segment(seg: str):  # sa_tag_parser.rs:25-46
    f = _split_terminator(seg, ",")
    assert len(f) == 6, f"Unexpected segment in bam SA tag: {seg}"
    return dict(rname=f[0], pos=int(f[1]) - 1, is_fwd_strand=f[2] == "+", cigar=cigar_from_text(f[3]), mapq=int(f[4]), nm=int(f[5]))


def parse_sa_aux_val(val: str):  # :55-59
    return [parse_sa_segment(s) for s in _split_terminator(val, ";")]


def get_seq_order_read_split_segments(label_to_index: Dict[str, int], rec: Record) -> List[SeqOrderSplitReadSegment]:  # split_read.rs:56-155
    def seq_order(read_start, read_end, read_size, is_fwd):  # :78-89
        return (read_start, read_end) if is_fwd else (read_size - read_end, read_size - read_start)

    rs, re_, primary_read_size = get_read_clip_positions(rec.cigar, False)
    a, b = seq_order(rs, re_, primary_read_size, not rec.is_reverse())
    segs = [SeqOrderSplitReadSegment(a, b, rec.tid, rec.pos, not rec.is_reverse(), list(rec.cigar), rec.mapq, True)]
    sa = rec.aux_get(b"SA")
    if sa is not None:
        assert sa[:1] == b"Z"
        for g in parse_sa_aux_val(sa[1:-1].decode()):
            assert any(is_alignment_match(c) for c in g["cigar"]), "split segment unaligned"  # :112-115
            s0, e0, size = get_read_clip_positions(g["cigar"], False)
            assert primary_read_size == size  # :118
            a, b = seq_order(s0, e0, size, g["is_fwd_strand"])
            segs.append(SeqOrderSplitReadSegment(a, b, label_to_index[g["rname"]], g["pos"], g["is_fwd_strand"], g["cigar"], g["mapq"], False))
        segs.sort(key=lambda x: x.seq_order_read_start)  # stable, like sort_by_key (:141)
    for s in segs:
        assert s.seq_order_read_start < s.seq_order_read_end  # :146-152
    return segs


# ---- caller glue (a8 / a9) ---------------------------------------------------------------------------------------------------
def get_contig_split_segments_from_read_mapping(seg: SeqOrderSplitReadSegment, contig_segments) -> List[int]:
    """contig_segments: list of (seq_order_read_start, seq_order_read_end); src/read_alignment_scanner.rs:80-103 with
    IntRange::intersect_range = other.end >= self.start && other.start < self.end (int_range.rs:56-58)"""
    r_start = seg.pos
    r_end = seg.pos + cigar_ref_offset(seg.cigar)
    return [k for k, (cs, ce) in enumerate(contig_segments) if r_end >= cs and r_start < ce]


def strand_glue(record_is_reverse: bool, seg: SeqOrderSplitReadSegment, contig_is_fwd_strand: bool, contig_length: int):
    """need_flipped_read_alignment and, for reverse-mapped contig segments, rev_pos + reversed CIGAR (before the left shift);
    src/read_alignment_scanner.rs:149-167"""
    read_segment_changes_strand_from_primary = record_is_reverse == seg.is_fwd_strand
    need_flipped = (not contig_is_fwd_strand) ^ read_segment_changes_strand_from_primary
    if contig_is_fwd_strand:
        return need_flipped, seg.pos, list(seg.cigar)
    read_segment_end = seg.pos + cigar_ref_offset(seg.cigar)
    return need_flipped, contig_length - read_segment_end, list(reversed(seg.cigar))


# ---- record finishing ----------------------------------------------------------------------------------------------------------
def comp_base(b: int) -> int:  # seq_util.rs:1-15
    return {65: 84, 84: 65, 67: 71, 71: 67, 78: 78, 97: 116, 116: 97, 99: 103, 103: 99, 110: 110}.get(b, 78)


def rev_comp(seq: bytes) -> bytes:
    return bytes(comp_base(b) for b in reversed(seq))


def hts_reg2bin(begin: int, end: int, min_shift: int = 14, depth: int = 5) -> int:  # util.rs:10-27
    end -= 1
    l, s, t = depth, min_shift, ((1 << (depth * 3)) - 1) // 7
    while l > 0:
        if begin >> s == end >> s:
            return t + (begin >> s)
        l -= 1
        s += 3
        t -= 1 << (l * 3)
    return 0


def clone_record(rec: Record) -> Record:  # :105-118
    r = rec.clone()
    for tag in (b"NM", b"SA", b"PS", b"ZM"):
        r.remove_aux_if_found(tag)
    return r


def reverse_alignment_seq_and_qual(r: Record):  # :125-133
    r.flag ^= BAM_FREVERSE
    r.set_seq_qual(rev_comp(r.seq_as_bytes()), bytes(reversed(r.qual)))


def lifted_record(rec: Record, contig_name: str, contig_segment_index: int, contig_is_fwd_strand: bool, chrom_index: int, contig_mapq: int,
                  need_flipped: bool, ref2_pos: int, ref2_cigar: Sequence[int]) -> Record:  # :245-284
    r = clone_record(rec)
    r.tid = chrom_index
    original_read_mapq = r.mapq
    r.mapq = contig_mapq
    r.push_aux_string(b"PS", f"{contig_name}_split{contig_segment_index}{'+' if contig_is_fwd_strand else '-'}")
    r.push_aux_u8(b"ZM", original_read_mapq)
    r.pos = ref2_pos
    r.cigar = list(ref2_cigar)
    if need_flipped:
        reverse_alignment_seq_and_qual(r)
    ref2_end = r.pos + cigar_ref_offset(r.cigar)  # get_alignment_end
    r.bin = hts_reg2bin(r.pos, ref2_end) & 0xFFFF
    r.flag |= BAM_FSUPPLEMENTARY
    return r


def get_sa_tag_segment(ref_names: Sequence[str], r: Record) -> str:  # :292-301
    return f"{ref_names[r.tid]},{r.pos + 1},{'-' if r.is_reverse() else '+'},{cigar_to_text(r.cigar)},{r.mapq},0;"


def finish_remapped_alignment_set(ref_names: Sequence[str], orig: Record, remapped: List[Record], is_target_region: bool) -> List[Record]:  # :310-366
    if not remapped:
        if is_target_region:
            return []
        u = clone_record(orig)
        u.flag |= BAM_FUNMAP
        u.flag &= ~BAM_FSUPPLEMENTARY
        u.cigar = []
        u.mapq = 255
        u.tid = -1
        u.pos = -1
        if u.is_reverse():
            reverse_alignment_seq_and_qual(u)
        return [u]
    primary = 0
    for k in range(1, len(remapped)):
        if remapped[primary].mapq < remapped[k].mapq:
            primary = k
    remapped[primary].flag &= ~BAM_FSUPPLEMENTARY
    for k in range(len(remapped)):
        aux_str = "".join(get_sa_tag_segment(ref_names, remapped[j]) for j in range(len(remapped)) if j != k)
        if aux_str:
            remapped[k].push_aux_string(b"SA", aux_str)
    return remapped
