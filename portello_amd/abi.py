"""ctypes mirror of ``include/portello_liftover.h`` plus numpy containers that build the descriptors.

Nothing in here computes anything: it only lays arrays out the way the C ABI wants them.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

PLO_OK = 0
PLO_ERR_INVALID_ARG = 1
PLO_ERR_NO_DEVICE = 2
PLO_ERR_HIP = 3
PLO_ERR_OUT_OF_MEMORY = 4
PLO_ERR_RANGE = 5
PLO_ERR_INTERNAL = 6
PLO_ERR_IO = 7
PLO_ERR_DATA = 8

ITEM_LIFTED = 0
ITEM_NO_LIFTOVER = 1
ITEM_LEN_MISMATCH = 2
ITEM_PANIC = 3
ITEM_NEED_BASES = 4  # sparse bases only: a comparison reached bases the batch does not carry (and no seq_full was given)

SEQ_BAM4 = 0
SEQ_ASCII = 1
SEQ_BAM4_SPARSE = 2

MEM_HOST = 0
MEM_DEVICE = 1

STAGE_STRAND = 1 << 0
STAGE_LSHIFT = 1 << 1
STAGE_LIFTOVER = 1 << 2
STAGE_LENCHECK = 1 << 3
STAGE_SIMPLIFY = 1 << 4
STAGES_ALL = 0x1F

NONE_VAL = -(2**63)  # INT64_MIN: Option::None for block-map values

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_pp = C.POINTER(C.c_void_p)


class PloIndexDesc(C.Structure):
    _fields_ = [
        ("n_contigs", C.c_uint32),
        ("contig_len", _i64p),
        ("contig_seg_off", _u32p),
        ("n_segments", C.c_uint32),
        ("seg_chrom_index", _u32p),
        ("seg_pos", _i64p),
        ("seg_is_fwd_strand", _u8p),
        ("seg_mapq", _u8p),
        ("seg_seq_order_start", _i64p),
        ("seg_seq_order_end", _i64p),
        ("seg_cigar_off", _u32p),
        ("seg_cigar", _u32p),
        ("n_chroms", C.c_uint32),
        ("chrom_len", _i64p),
        ("chrom_seq", _pp),
        ("rev_contig_seq", _pp),
        ("seq_mem", C.c_int32),
    ]


class PloBatchIn(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint32),
        ("read_is_reverse", _u8p),
        ("read_seq_len", _u32p),
        ("read_seq_off", _u64p),
        ("seq", _u8p),
        ("seq_bytes", C.c_uint64),
        ("seq_fmt", C.c_int32),
        ("n_segs", C.c_uint32),
        ("seg_read", _u32p),
        ("seg_contig", _u32p),
        ("seg_pos", _i64p),
        ("seg_is_fwd_strand", _u8p),
        ("seg_cigar_off", _u32p),
        ("cigar", _u32p),
        ("n_items", C.c_uint32),
        ("item_seg", _u32p),
        ("item_cseg", _u32p),
        ("seq_full", _u8p),
        ("read_seq_full_off", _u64p),
    ]


class PloBatchOut(C.Structure):
    _fields_ = [
        ("n_items", C.c_uint32),
        ("item_seg", _u32p),
        ("item_cseg", _u32p),
        ("item_status", _u8p),
        ("item_need_flipped", _u8p),
        ("item_mapq", _u8p),
        ("item_chrom_index", _u32p),
        ("item_ref_pos", _i64p),
        ("item_cigar_off", _u64p),
        ("item_cigar_len", _u32p),
        ("cigar", _u32p),
        ("n_cigar", C.c_uint64),
    ]


PLO_API_VERSION = 6  # include/portello_liftover.h


class PloTiming(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("total_ms", C.c_float),
        ("enumerate_ms", C.c_float),
        ("lift_ms", C.c_float),
        ("big_ms", C.c_float),
        ("n_items", C.c_uint32),
        ("n_big_items", C.c_uint32),
        ("n_in_ops", C.c_uint64),
        ("n_out_ops", C.c_uint64),
        ("algo_bytes", C.c_uint64),
        ("lanes_ms", C.c_float),
        ("retry_ms", C.c_float),
        ("n_lane_items", C.c_uint32),
        ("n_retry_items", C.c_uint32),
        ("mid_ms", C.c_float),
        ("n_mid_items", C.c_uint32),
        ("n_miss_items", C.c_uint32),
        ("miss_ms", C.c_float),
        ("tile_cap", C.c_uint32),
        ("tile_window", C.c_uint32),
        ("heavy_lanes_ms", C.c_float),
        ("n_heavy_lane_items", C.c_uint32),
        ("lane_utilisation", C.c_float),
        ("heavy_kernel", C.c_uint32),
        ("host_syncs", C.c_uint32),
    ]


_u16p = C.POINTER(C.c_uint16)
NO_FLIP = 2**64 - 1


class PloFinishIn(C.Structure):
    _fields_ = [("read_flags", _u16p), ("qual", _u8p), ("read_qual_off", _u64p), ("qual_bytes", C.c_uint64)]


class PloFinishOut(C.Structure):
    _fields_ = [
        ("item_flag", _u16p), ("item_bin", _u16p), ("item_ref_end", _i64p), ("item_is_primary", _u8p),
        ("item_seq_off", _u64p), ("item_qual_off", _u64p),
        ("read_n_lifted", _u32p), ("read_primary_item", _u32p), ("read_unmapped_flag", _u16p),
        ("read_seq_off", _u64p), ("read_qual_off", _u64p),
        ("rev_seq", _u8p), ("rev_qual", _u8p), ("rev_seq_bytes", C.c_uint64), ("rev_qual_bytes", C.c_uint64),
        ("finish_ms", C.c_float), ("revcomp_ms", C.c_float),
        ("n_items", C.c_uint32), ("n_reads", C.c_uint32),
    ]


class PloSaIn(C.Structure):
    _fields_ = [("n_chroms", C.c_uint32), ("chrom_name_off", _u32p), ("chrom_names", _u8p)]


class PloSaOut(C.Structure):
    _fields_ = [("n_items", C.c_uint32), ("item_sa_off", _u32p), ("sa_text", _u8p), ("sa_bytes", C.c_uint64), ("sa_ms", C.c_float)]


FINISH_ITEM_FIELDS = [("item_flag", np.uint16), ("item_bin", np.uint16), ("item_ref_end", np.int64), ("item_is_primary", np.uint8),
                      ("item_seq_off", np.uint64), ("item_qual_off", np.uint64)]
FINISH_READ_FIELDS = [("read_n_lifted", np.uint32), ("read_primary_item", np.uint32), ("read_unmapped_flag", np.uint16),
                      ("read_seq_off", np.uint64), ("read_qual_off", np.uint64)]


def _np(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _ptr(a: Optional[np.ndarray], ctype):
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    return a.ctypes.data_as(C.POINTER(ctype))


@dataclass
class IndexData:
    """Host-side picture of AllContigMappingInfo + reference (src/contig_alignment_scanner/mod.rs:25-47,76)."""

    contig_len: np.ndarray
    contig_seg_off: np.ndarray
    seg_chrom_index: np.ndarray
    seg_pos: np.ndarray
    seg_is_fwd_strand: np.ndarray
    seg_mapq: np.ndarray
    seg_seq_order_start: np.ndarray
    seg_seq_order_end: np.ndarray
    seg_cigar_off: np.ndarray
    seg_cigar: np.ndarray
    chrom_seq: List[np.ndarray]  # uint8 arrays (host) -- or ints (device pointers) when seq_mem == MEM_DEVICE
    rev_contig_seq: List[Optional[np.ndarray]]
    chrom_len: Optional[np.ndarray] = None
    seq_mem: int = MEM_HOST
    _keep: list = field(default_factory=list, repr=False)

    def __post_init__(self):
        self.contig_len = _np(self.contig_len, np.int64)
        self.contig_seg_off = _np(self.contig_seg_off, np.uint32)
        self.seg_chrom_index = _np(self.seg_chrom_index, np.uint32)
        self.seg_pos = _np(self.seg_pos, np.int64)
        self.seg_is_fwd_strand = _np(self.seg_is_fwd_strand, np.uint8)
        self.seg_mapq = _np(self.seg_mapq, np.uint8)
        self.seg_seq_order_start = _np(self.seg_seq_order_start, np.int64)
        self.seg_seq_order_end = _np(self.seg_seq_order_end, np.int64)
        self.seg_cigar_off = _np(self.seg_cigar_off, np.uint32)
        self.seg_cigar = _np(self.seg_cigar, np.uint32)
        if self.seq_mem == MEM_HOST:
            self.chrom_seq = [_np(s, np.uint8) for s in self.chrom_seq]
            self.rev_contig_seq = [None if s is None else _np(s, np.uint8) for s in self.rev_contig_seq]
            if self.chrom_len is None:
                self.chrom_len = np.array([len(s) for s in self.chrom_seq], dtype=np.int64)
        self.chrom_len = _np(self.chrom_len, np.int64)

    @property
    def n_contigs(self) -> int:
        return len(self.contig_len)

    @property
    def n_segments(self) -> int:
        return len(self.seg_pos)

    def to_desc(self) -> PloIndexDesc:
        d = PloIndexDesc()
        d.n_contigs = self.n_contigs
        d.contig_len = _ptr(self.contig_len, C.c_int64)
        d.contig_seg_off = _ptr(self.contig_seg_off, C.c_uint32)
        d.n_segments = self.n_segments
        d.seg_chrom_index = _ptr(self.seg_chrom_index, C.c_uint32)
        d.seg_pos = _ptr(self.seg_pos, C.c_int64)
        d.seg_is_fwd_strand = _ptr(self.seg_is_fwd_strand, C.c_uint8)
        d.seg_mapq = _ptr(self.seg_mapq, C.c_uint8)
        d.seg_seq_order_start = _ptr(self.seg_seq_order_start, C.c_int64)
        d.seg_seq_order_end = _ptr(self.seg_seq_order_end, C.c_int64)
        d.seg_cigar_off = _ptr(self.seg_cigar_off, C.c_uint32)
        d.seg_cigar = _ptr(self.seg_cigar, C.c_uint32)
        d.n_chroms = len(self.chrom_seq)
        d.chrom_len = _ptr(self.chrom_len, C.c_int64)

        def addr(s):
            if s is None:
                return None
            return int(s) if self.seq_mem == MEM_DEVICE else s.ctypes.data

        chrom_ptrs = (C.c_void_p * max(1, len(self.chrom_seq)))(*[addr(s) for s in self.chrom_seq])
        rev_ptrs = (C.c_void_p * max(1, self.n_contigs))(*[addr(s) for s in self.rev_contig_seq])
        self._keep = [chrom_ptrs, rev_ptrs]
        d.chrom_seq = C.cast(chrom_ptrs, _pp)
        d.rev_contig_seq = C.cast(rev_ptrs, _pp)
        d.seq_mem = self.seq_mem
        return d


@dataclass
class BatchData:
    """Host-side picture of one window of primary reads with their split segments."""

    read_is_reverse: np.ndarray
    read_seq_len: np.ndarray
    read_seq_off: np.ndarray
    seq: np.ndarray
    seq_fmt: int
    seg_read: np.ndarray
    seg_contig: np.ndarray
    seg_pos: np.ndarray
    seg_is_fwd_strand: np.ndarray
    seg_cigar_off: np.ndarray
    cigar: np.ndarray
    item_seg: Optional[np.ndarray] = None
    item_cseg: Optional[np.ndarray] = None
    # seq_fmt SEQ_BAM4_SPARSE: the reads' complete BAM4 bases (host), looked up when a comparison runs past the granules sent
    seq_full: Optional[np.ndarray] = None
    read_seq_full_off: Optional[np.ndarray] = None

    def __post_init__(self):
        self.read_is_reverse = _np(self.read_is_reverse, np.uint8)
        self.read_seq_len = _np(self.read_seq_len, np.uint32)
        self.read_seq_off = _np(self.read_seq_off, np.uint64)
        self.seq = _np(self.seq, np.uint8)
        self.seg_read = _np(self.seg_read, np.uint32)
        self.seg_contig = _np(self.seg_contig, np.uint32)
        self.seg_pos = _np(self.seg_pos, np.int64)
        self.seg_is_fwd_strand = _np(self.seg_is_fwd_strand, np.uint8)
        self.seg_cigar_off = _np(self.seg_cigar_off, np.uint32)
        self.cigar = _np(self.cigar, np.uint32)
        if self.item_seg is not None:
            self.item_seg = _np(self.item_seg, np.uint32)
            self.item_cseg = _np(self.item_cseg, np.uint32)
        if self.seq_full is not None:
            self.seq_full = _np(self.seq_full, np.uint8)
            self.read_seq_full_off = _np(self.read_seq_full_off, np.uint64)

    @property
    def n_reads(self) -> int:
        return len(self.read_seq_len)

    @property
    def n_segs(self) -> int:
        return len(self.seg_read)

    def to_desc(self) -> PloBatchIn:
        b = PloBatchIn()
        b.n_reads = self.n_reads
        b.read_is_reverse = _ptr(self.read_is_reverse, C.c_uint8)
        b.read_seq_len = _ptr(self.read_seq_len, C.c_uint32)
        b.read_seq_off = _ptr(self.read_seq_off, C.c_uint64)
        b.seq = _ptr(self.seq, C.c_uint8)
        b.seq_bytes = self.seq.nbytes
        b.seq_fmt = self.seq_fmt
        b.n_segs = self.n_segs
        b.seg_read = _ptr(self.seg_read, C.c_uint32)
        b.seg_contig = _ptr(self.seg_contig, C.c_uint32)
        b.seg_pos = _ptr(self.seg_pos, C.c_int64)
        b.seg_is_fwd_strand = _ptr(self.seg_is_fwd_strand, C.c_uint8)
        b.seg_cigar_off = _ptr(self.seg_cigar_off, C.c_uint32)
        b.cigar = _ptr(self.cigar, C.c_uint32)
        if self.item_seg is not None:
            b.n_items = len(self.item_seg)
            b.item_seg = _ptr(self.item_seg, C.c_uint32)
            b.item_cseg = _ptr(self.item_cseg, C.c_uint32)
        else:
            b.n_items = 0
            b.item_seg = _ptr(None, C.c_uint32)
            b.item_cseg = _ptr(None, C.c_uint32)
        b.seq_full = _ptr(self.seq_full, C.c_uint8)
        b.read_seq_full_off = _ptr(self.read_seq_full_off, C.c_uint64)
        return b


@dataclass
class BatchResult:
    """Host copy of a plo_batch_out."""

    item_seg: np.ndarray
    item_cseg: np.ndarray
    item_status: np.ndarray
    item_need_flipped: np.ndarray
    item_mapq: np.ndarray
    item_chrom_index: np.ndarray
    item_ref_pos: np.ndarray
    item_cigar_off: np.ndarray
    item_cigar_len: np.ndarray
    cigar: np.ndarray

    @property
    def n_items(self) -> int:
        return len(self.item_seg)

    def item_cigar(self, i: int) -> np.ndarray:
        o = int(self.item_cigar_off[i])
        return self.cigar[o : o + int(self.item_cigar_len[i])]

    def canonical(self):
        """Order-independent, layout-independent view used by the parity tests: per item
        (seg, cseg, status, flip, mapq, chrom, pos, cigar-bytes)."""
        rows = []
        for i in range(self.n_items):
            rows.append(
                (
                    int(self.item_seg[i]),
                    int(self.item_cseg[i]),
                    int(self.item_status[i]),
                    int(self.item_need_flipped[i]),
                    int(self.item_mapq[i]),
                    int(self.item_chrom_index[i]),
                    int(self.item_ref_pos[i]),
                    self.item_cigar(i).tobytes(),
                )
            )
        return rows


def result_from_out(out: PloBatchOut) -> BatchResult:
    """Copy a host-resident plo_batch_out into numpy arrays."""
    n = int(out.n_items)
    nc = int(out.n_cigar)

    def cp(p, dtype, count):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(p, shape=(count,)).astype(dtype, copy=True)

    return BatchResult(
        item_seg=cp(out.item_seg, np.uint32, n),
        item_cseg=cp(out.item_cseg, np.uint32, n),
        item_status=cp(out.item_status, np.uint8, n),
        item_need_flipped=cp(out.item_need_flipped, np.uint8, n),
        item_mapq=cp(out.item_mapq, np.uint8, n),
        item_chrom_index=cp(out.item_chrom_index, np.uint32, n),
        item_ref_pos=cp(out.item_ref_pos, np.int64, n),
        item_cigar_off=cp(out.item_cigar_off, np.uint64, n),
        item_cigar_len=cp(out.item_cigar_len, np.uint32, n),
        cigar=cp(out.cigar, np.uint32, nc),
    )


def out_from_result(res: BatchResult):
    """a plo_batch_out whose pointers view the numpy arrays of a BatchResult; returns (struct, keep-alive list)"""
    o = PloBatchOut()
    keep = []

    def put(name, dt, ct):
        a = np.ascontiguousarray(getattr(res, name), dtype=dt)
        if a.size == 0:
            a = np.zeros(1, dtype=dt)
        keep.append(a)
        setattr(o, name, a.ctypes.data_as(C.POINTER(ct)))

    o.n_items = res.n_items
    put("item_seg", np.uint32, C.c_uint32)
    put("item_cseg", np.uint32, C.c_uint32)
    put("item_status", np.uint8, C.c_uint8)
    put("item_need_flipped", np.uint8, C.c_uint8)
    put("item_mapq", np.uint8, C.c_uint8)
    put("item_chrom_index", np.uint32, C.c_uint32)
    put("item_ref_pos", np.int64, C.c_int64)
    put("item_cigar_off", np.uint64, C.c_uint64)
    put("item_cigar_len", np.uint32, C.c_uint32)
    put("cigar", np.uint32, C.c_uint32)
    o.n_cigar = len(res.cigar)
    return o, keep
