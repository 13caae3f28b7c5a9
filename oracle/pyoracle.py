"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference algorithm.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (portello_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

from portello_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
NONE = abi.NONE_VAL
PANIC = -2

_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "portello_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH)
        for p in (src, os.path.join(_HERE, "portello_oracle.h"), os.path.join(_HERE, "..", "include", "portello_liftover.h"))
        if os.path.exists(p)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u32p, u64p, i64p, u8p = (C.POINTER(t) for t in (C.c_uint32, C.c_uint64, C.c_int64, C.c_uint8))
        sz = C.c_size_t
        L.orc_compress_cigar.restype = sz
        L.orc_compress_cigar.argtypes = [u32p, sz, u32p]
        L.orc_clean_up_cigar_edge_indels.restype = C.c_uint64
        L.orc_clean_up_cigar_edge_indels.argtypes = [u32p, sz]
        L.orc_cigar_read_offset.restype = C.c_uint64
        L.orc_cigar_read_offset.argtypes = [u32p, sz, C.c_int]
        L.orc_cigar_ref_offset.restype = C.c_int64
        L.orc_cigar_ref_offset.argtypes = [u32p, sz]
        L.orc_cigarseg_read_offset.restype = C.c_uint64
        L.orc_cigarseg_read_offset.argtypes = [C.c_uint32, C.c_int]
        L.orc_cigarseg_ref_offset.restype = C.c_int64
        L.orc_cigarseg_ref_offset.argtypes = [C.c_uint32]
        L.orc_read_clip_positions.restype = None
        L.orc_read_clip_positions.argtypes = [u32p, sz, C.c_int, u64p]
        L.orc_comp_base.restype = C.c_uint8
        L.orc_comp_base.argtypes = [C.c_uint8]
        L.orc_rev_comp_in_place.restype = None
        L.orc_rev_comp_in_place.argtypes = [u8p, sz]
        L.orc_decode_bam4.restype = None
        L.orc_decode_bam4.argtypes = [u8p, sz, u8p]
        L.orc_indel_breakend_homology.restype = C.c_int
        L.orc_indel_breakend_homology.argtypes = [u8p, C.c_int64, C.c_int64, C.c_int64, u8p, C.c_int64, C.c_int64, C.c_int64, i64p, i64p]
        L.orc_shift_indels.restype = C.c_int
        L.orc_shift_indels.argtypes = [C.c_int, C.c_int64, u32p, sz, u8p, C.c_int64, u8p, C.c_int64, i64p, u32p, C.POINTER(sz)]
        L.orc_map_build.restype = sz
        L.orc_map_build.argtypes = [C.c_int64, u32p, sz, C.c_int, u64p, i64p]
        L.orc_map_get_ref_pos.restype = C.c_int64
        L.orc_map_get_ref_pos.argtypes = [u64p, i64p, sz, C.c_uint64]
        L.orc_map_get_ref_range.restype = None
        L.orc_map_get_ref_range.argtypes = [u64p, sz, C.c_uint64, C.c_uint64, C.POINTER(sz), C.POINTER(sz)]
        L.orc_liftover_read_alignment.restype = C.c_int
        L.orc_liftover_read_alignment.argtypes = [u64p, i64p, sz, C.c_int64, u32p, sz, i64p, u32p, C.POINTER(sz)]
        L.orc_simplify_alignment_indels.restype = C.c_int
        L.orc_simplify_alignment_indels.argtypes = [C.c_int64, u32p, sz, u8p, C.c_int64, u8p, C.c_int64, i64p, u32p, C.POINTER(sz)]
        L.orc_liftover_batch.restype = C.c_int
        L.orc_liftover_batch.argtypes = [C.POINTER(abi.PloIndexDesc), C.POINTER(abi.PloBatchIn), C.c_uint32, C.c_int, C.POINTER(abi.PloBatchOut)]
        L.orc_batch_free.restype = None
        L.orc_batch_free.argtypes = [C.POINTER(abi.PloBatchOut)]
        L.orc_bam_reg2bin.restype = C.c_uint16
        L.orc_bam_reg2bin.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_finish_batch.restype = C.c_int
        L.orc_finish_batch.argtypes = [C.POINTER(abi.PloBatchIn), C.POINTER(abi.PloFinishIn), C.POINTER(abi.PloBatchOut), C.POINTER(abi.PloFinishOut)]
        L.orc_finish_free.restype = None
        L.orc_finish_free.argtypes = [C.POINTER(abi.PloFinishOut)]
        _lib = L
    return _lib


def _u32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


def _u8(a):
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def compress_cigar(cigar) -> np.ndarray:
    c = _u32(cigar)
    out = np.zeros(len(c) + 1, dtype=np.uint32)
    n = lib().orc_compress_cigar(_p(c, C.c_uint32), len(c), _p(out, C.c_uint32))
    return out[:n].copy()


def clean_up_cigar_edge_indels(cigar) -> Tuple[int, np.ndarray]:
    c = _u32(cigar).copy()
    shift = lib().orc_clean_up_cigar_edge_indels(_p(c, C.c_uint32), len(c))
    return int(shift), c


def read_clip_positions(cigar, ignore_hard_clip: bool):
    c = _u32(cigar)
    out = np.zeros(3, dtype=np.uint64)
    lib().orc_read_clip_positions(_p(c, C.c_uint32), len(c), int(ignore_hard_clip), _p(out, C.c_uint64))
    return tuple(int(x) for x in out)


def rev_comp(seq) -> bytes:
    s = _u8(seq).copy()
    lib().orc_rev_comp_in_place(_p(s, C.c_uint8), len(s))
    return s.tobytes()


def decode_bam4(packed, n_bases: int) -> bytes:
    p = _u8(packed)
    out = np.zeros(max(1, n_bases), dtype=np.uint8)
    lib().orc_decode_bam4(_p(p, C.c_uint8), n_bases, _p(out, C.c_uint8))
    return out[:n_bases].tobytes()


def indel_breakend_homology(ref_seq, ref_range, read_seq, read_range):
    r, q = _u8(ref_seq), _u8(read_seq)
    hs, he = C.c_int64(0), C.c_int64(0)
    rc = lib().orc_indel_breakend_homology(
        _p(r, C.c_uint8), len(r), ref_range[0], ref_range[1], _p(q, C.c_uint8), len(q), read_range[0], read_range[1],
        C.byref(hs), C.byref(he))
    if rc != 0:
        return None
    return (hs.value, he.value)


def shift_indels(direction: str, ref_pos: int, cigar, ref_seq, read_seq):
    c, r, q = _u32(cigar), _u8(ref_seq), _u8(read_seq)
    out = np.zeros(2 * len(c) + 2, dtype=np.uint32)
    n = C.c_size_t(0)
    pos = C.c_int64(0)
    rc = lib().orc_shift_indels(0 if direction == "left" else 1, ref_pos, _p(c, C.c_uint32), len(c), _p(r, C.c_uint8),
                                len(r), _p(q, C.c_uint8), len(q), C.byref(pos), _p(out, C.c_uint32), C.byref(n))
    if rc != 0:
        return None
    return pos.value, out[: n.value].copy()


def left_shift_indels(ref_pos, cigar, ref_seq, read_seq):
    return shift_indels("left", ref_pos, cigar, ref_seq, read_seq)


def right_shift_indels(ref_pos, cigar, ref_seq, read_seq):
    return shift_indels("right", ref_pos, cigar, ref_seq, read_seq)


def map_build(ref_pos: int, cigar, ignore_hard_clip: bool = False):
    c = _u32(cigar)
    keys = np.zeros(2 * len(c) + 2, dtype=np.uint64)
    vals = np.zeros(2 * len(c) + 2, dtype=np.int64)
    n = lib().orc_map_build(ref_pos, _p(c, C.c_uint32), len(c), int(ignore_hard_clip), _p(keys, C.c_uint64), _p(vals, C.c_int64))
    return keys[:n].copy(), vals[:n].copy()


def map_get_ref_pos(keys, vals, read_pos: int) -> Optional[int]:
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    v = np.ascontiguousarray(vals, dtype=np.int64)
    r = lib().orc_map_get_ref_pos(_p(k, C.c_uint64), _p(v, C.c_int64), len(k), read_pos)
    return None if r == NONE else int(r)


def map_get_ref_range(keys, vals, a: int, b: int):
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    i0, i1 = C.c_size_t(0), C.c_size_t(0)
    lib().orc_map_get_ref_range(_p(k, C.c_uint64), len(k), a, b, C.byref(i0), C.byref(i1))
    return [(int(keys[i]), None if int(vals[i]) == NONE else int(vals[i])) for i in range(i0.value, i1.value)]


def liftover_read_alignment(keys, vals, start: int, cigar):
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    v = np.ascontiguousarray(vals, dtype=np.int64)
    c = _u32(cigar)
    out = np.zeros(2 * (len(c) + len(k)) + 2, dtype=np.uint32)
    n = C.c_size_t(0)
    pos = C.c_int64(0)
    some = lib().orc_liftover_read_alignment(_p(k, C.c_uint64), _p(v, C.c_int64), len(k), start, _p(c, C.c_uint32), len(c),
                                             C.byref(pos), _p(out, C.c_uint32), C.byref(n))
    if not some:
        return None
    return pos.value, out[: n.value].copy()


def simplify_alignment_indels(ref_pos: int, cigar, ref_seq, read_seq):
    c, r, q = _u32(cigar), _u8(ref_seq), _u8(read_seq)
    out = np.zeros(2 * len(c) + 2, dtype=np.uint32)
    n = C.c_size_t(0)
    pos = C.c_int64(0)
    rc = lib().orc_simplify_alignment_indels(ref_pos, _p(c, C.c_uint32), len(c), _p(r, C.c_uint8), len(r), _p(q, C.c_uint8),
                                             len(q), C.byref(pos), _p(out, C.c_uint32), C.byref(n))
    if rc != 0:
        return None
    return pos.value, out[: n.value].copy()


def liftover_batch(index: abi.IndexData, batch: abi.BatchData, stages: int = abi.STAGES_ALL, n_threads: int = 1) -> abi.BatchResult:
    assert index.seq_mem == abi.MEM_HOST
    d = index.to_desc()
    b = batch.to_desc()
    out = abi.PloBatchOut()
    rc = lib().orc_liftover_batch(C.byref(d), C.byref(b), stages, n_threads, C.byref(out))
    assert rc == 0
    res = abi.result_from_out(out)
    lib().orc_batch_free(C.byref(out))
    return res


def finish_batch(batch: abi.BatchData, read_flags, qual, read_qual_off, lift: abi.BatchResult) -> dict:
    """Record finishing (orc_finish_batch) on host arrays; returns a dict of numpy arrays."""
    b = batch.to_desc()
    rf = np.ascontiguousarray(read_flags, dtype=np.uint16)
    q = np.ascontiguousarray(qual, dtype=np.uint8)
    qo = np.ascontiguousarray(read_qual_off, dtype=np.uint64)
    fin = abi.PloFinishIn(_p(rf, C.c_uint16), _p(q, C.c_uint8), _p(qo, C.c_uint64), q.nbytes)
    arrs = {n: np.ascontiguousarray(getattr(lift, n)) for n in ("item_seg", "item_cseg", "item_status", "item_need_flipped", "item_mapq",
                                                                 "item_chrom_index", "item_ref_pos", "item_cigar_off", "item_cigar_len", "cigar")}
    lo = abi.PloBatchOut()
    lo.n_items = lift.n_items
    for n, t in (("item_seg", C.c_uint32), ("item_cseg", C.c_uint32), ("item_status", C.c_uint8), ("item_need_flipped", C.c_uint8),
                 ("item_mapq", C.c_uint8), ("item_chrom_index", C.c_uint32), ("item_ref_pos", C.c_int64), ("item_cigar_off", C.c_uint64),
                 ("item_cigar_len", C.c_uint32), ("cigar", C.c_uint32)):
        setattr(lo, n, _p(arrs[n], t))
    lo.n_cigar = len(arrs["cigar"])
    out = abi.PloFinishOut()
    rc = lib().orc_finish_batch(C.byref(b), C.byref(fin), C.byref(lo), C.byref(out))
    assert rc == 0
    n, nr = lift.n_items, batch.n_reads

    def cp(p, dt, cnt):
        return np.ctypeslib.as_array(p, shape=(cnt,)).astype(dt, copy=True) if cnt else np.zeros(0, dt)

    res = {name: cp(getattr(out, name), dt, n) for name, dt in abi.FINISH_ITEM_FIELDS}
    res.update({name: cp(getattr(out, name), dt, nr) for name, dt in abi.FINISH_READ_FIELDS})
    res["rev_seq"] = cp(out.rev_seq, np.uint8, int(out.rev_seq_bytes))
    res["rev_qual"] = cp(out.rev_qual, np.uint8, int(out.rev_qual_bytes))
    lib().orc_finish_free(C.byref(out))
    return res


def sa_values(batch: abi.BatchData, lift: abi.BatchResult, item_flag, chrom_names) -> list:
    """SA:Z values (orc_sa_values): bytes per item, None where the record gets no SA tag."""
    b = batch.to_desc()
    arrs = {n: np.ascontiguousarray(getattr(lift, n)) for n in ("item_seg", "item_cseg", "item_status", "item_need_flipped", "item_mapq",
                                                                 "item_chrom_index", "item_ref_pos", "item_cigar_off", "item_cigar_len", "cigar")}
    lo = abi.PloBatchOut()
    lo.n_items = lift.n_items
    for n, t in (("item_seg", C.c_uint32), ("item_cseg", C.c_uint32), ("item_status", C.c_uint8), ("item_need_flipped", C.c_uint8),
                 ("item_mapq", C.c_uint8), ("item_chrom_index", C.c_uint32), ("item_ref_pos", C.c_int64), ("item_cigar_off", C.c_uint64),
                 ("item_cigar_len", C.c_uint32), ("cigar", C.c_uint32)):
        setattr(lo, n, _p(arrs[n], t))
    lo.n_cigar = len(arrs["cigar"])
    fl = np.ascontiguousarray(item_flag, dtype=np.uint16)
    names = (C.c_char_p * max(1, len(chrom_names)))(*[n.encode() if isinstance(n, str) else n for n in chrom_names])
    n = lift.n_items
    vals = (C.c_void_p * max(1, n))()
    L = lib()
    L.orc_sa_values.restype = C.c_int
    L.orc_sa_values.argtypes = [C.POINTER(abi.PloBatchIn), C.POINTER(abi.PloBatchOut), C.POINTER(C.c_uint16), C.POINTER(C.c_char_p),
                                C.POINTER(C.c_void_p)]
    L.orc_sa_free.restype = None
    L.orc_sa_free.argtypes = [C.POINTER(C.c_void_p), C.c_uint32]
    assert L.orc_sa_values(C.byref(b), C.byref(lo), _p(fl, C.c_uint16), names, vals) == 0
    out = [C.string_at(vals[i]) if vals[i] else None for i in range(n)]
    L.orc_sa_free(vals, n)
    return out
