"""ctypes binding of tests/emu/libplo_emu.so: the device algorithm executed under the CPU wave64 emulator.
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from portello_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
_LIB = os.path.join(_HERE, "emu", "libplo_emu.so")
_lib = None


def build(force=False, sanitize=False):
    srcs = [os.path.join(_HERE, "emu", "emu_harness.cpp"), os.path.join(_HERE, "emu", "plo_wave.hpp")] + [
        os.path.join(ROOT, "portello_amd", "csrc", f) for f in ("lift_core.hpp", "lane_core.hpp", "lane_stream.hpp", "inflate.hpp", "finish_core.hpp", "lift_types.hpp", "index_pack.hpp", "enumerate.hpp")]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-I" + os.path.join(_HERE, "emu"),
               "-o", _LIB, srcs[0]]
        if sanitize:
            cmd[1:1] = ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"]
        subprocess.check_call(cmd)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.emu_liftover_batch.restype = C.c_int
        L.emu_liftover_batch.argtypes = [C.POINTER(abi.PloIndexDesc), C.POINTER(abi.PloBatchIn), C.c_uint32, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(abi.PloBatchOut), C.POINTER(C.c_ulonglong)]
        L.emu_finish_batch.restype = C.c_int
        L.emu_finish_batch.argtypes = [C.POINTER(abi.PloBatchIn), C.POINTER(abi.PloFinishIn), C.POINTER(abi.PloBatchOut), C.c_int,
                                       C.POINTER(abi.PloFinishOut)]
        L.emu_free_last.restype = None
        _lib = L
    return _lib


def liftover_batch(index: abi.IndexData, batch: abi.BatchData, stages=abi.STAGES_ALL, cap=768, window=256, big_thresh=256,
                   big_cap=1 << 16, order_seed=0, mid_waves=0, mid_cap=2048, lane_max_w=-1, lane_capw=3072, lane_heavy_per=0, lane_budget=0, lane_stream=0):
    d = index.to_desc()
    b = batch.to_desc()
    out = abi.PloBatchOut()
    counters = (C.c_ulonglong * 24)()
    rc = lib().emu_liftover_batch(C.byref(d), C.byref(b), stages, cap, window, big_thresh, big_cap, order_seed, mid_waves, mid_cap, lane_max_w, lane_capw, lane_heavy_per, lane_budget, lane_stream, C.byref(out), counters)
    res = abi.result_from_out(out)
    lib().emu_free_last()
    return rc, res, list(counters)


def finish_batch(batch: abi.BatchData, read_flags, qual, read_qual_off, lift: abi.BatchResult, nthreads=7) -> dict:
    """finish_core.hpp executed on the host (plain loops instead of GPU threads)"""
    b = batch.to_desc()
    rf = np.ascontiguousarray(read_flags, dtype=np.uint16)
    q = np.ascontiguousarray(qual, dtype=np.uint8)
    qo = np.ascontiguousarray(read_qual_off, dtype=np.uint64)
    p = lambda arr, t: arr.ctypes.data_as(C.POINTER(t))
    fin = abi.PloFinishIn(p(rf, C.c_uint16), p(q, C.c_uint8), p(qo, C.c_uint64), q.nbytes)
    arrs = {n: np.ascontiguousarray(getattr(lift, n)) for n in ("item_seg", "item_cseg", "item_status", "item_need_flipped", "item_mapq",
                                                                 "item_chrom_index", "item_ref_pos", "item_cigar_off", "item_cigar_len", "cigar")}
    lo = abi.PloBatchOut()
    lo.n_items = lift.n_items
    for n, t in (("item_seg", C.c_uint32), ("item_cseg", C.c_uint32), ("item_status", C.c_uint8), ("item_need_flipped", C.c_uint8),
                 ("item_mapq", C.c_uint8), ("item_chrom_index", C.c_uint32), ("item_ref_pos", C.c_int64), ("item_cigar_off", C.c_uint64),
                 ("item_cigar_len", C.c_uint32), ("cigar", C.c_uint32)):
        setattr(lo, n, p(arrs[n], t))
    lo.n_cigar = len(arrs["cigar"])
    out = abi.PloFinishOut()
    assert lib().emu_finish_batch(C.byref(b), C.byref(fin), C.byref(lo), nthreads, C.byref(out)) == 0
    n, nr = lift.n_items, batch.n_reads
    cp = lambda ptr, dt, cnt: np.ctypeslib.as_array(ptr, shape=(cnt,)).astype(dt, copy=True) if cnt else np.zeros(0, dt)
    res = {name: cp(getattr(out, name), dt, n) for name, dt in abi.FINISH_ITEM_FIELDS}
    res.update({name: cp(getattr(out, name), dt, nr) for name, dt in abi.FINISH_READ_FIELDS})
    res["rev_seq"] = cp(out.rev_seq, np.uint8, int(out.rev_seq_bytes))
    res["rev_qual"] = cp(out.rev_qual, np.uint8, int(out.rev_qual_bytes))
    return res


def sa_segments(batch: abi.BatchData, lift: abi.BatchResult, item_flag, read_n_lifted, chrom_names):
    """finish_core.hpp's SA-segment code executed on the host: (offsets [n_items+1], text bytes)"""
    p = lambda arr, t: arr.ctypes.data_as(C.POINTER(t))
    arrs = {n: np.ascontiguousarray(getattr(lift, n)) for n in ("item_seg", "item_cseg", "item_status", "item_need_flipped", "item_mapq",
                                                                 "item_chrom_index", "item_ref_pos", "item_cigar_off", "item_cigar_len", "cigar")}
    lo = abi.PloBatchOut()
    lo.n_items = lift.n_items
    for n, t in (("item_seg", C.c_uint32), ("item_cseg", C.c_uint32), ("item_status", C.c_uint8), ("item_need_flipped", C.c_uint8),
                 ("item_mapq", C.c_uint8), ("item_chrom_index", C.c_uint32), ("item_ref_pos", C.c_int64), ("item_cigar_off", C.c_uint64),
                 ("item_cigar_len", C.c_uint32), ("cigar", C.c_uint32)):
        setattr(lo, n, p(arrs[n], t))
    lo.n_cigar = len(arrs["cigar"])
    fl = np.ascontiguousarray(item_flag, dtype=np.uint16)
    item_read = np.ascontiguousarray(np.asarray(batch.seg_read, dtype=np.uint32)[arrs["item_seg"]], dtype=np.uint32)
    nl = np.ascontiguousarray(read_n_lifted, dtype=np.uint32)
    enc = [n.encode() if isinstance(n, str) else n for n in chrom_names]
    noff = np.zeros(len(enc) + 1, dtype=np.uint32)
    noff[1:] = np.cumsum([len(e) for e in enc])
    blob = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy()
    L = lib()
    L.emu_sa_segments.restype = C.c_int
    L.emu_sa_segments.argtypes = [C.POINTER(abi.PloBatchOut), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint8))]
    L.emu_sa_free.restype = None
    L.emu_sa_free.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
    off_p, text_p = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint8)()
    assert L.emu_sa_segments(C.byref(lo), p(fl, C.c_uint16), p(item_read, C.c_uint32), p(nl, C.c_uint32), p(noff, C.c_uint32),
                             p(blob, C.c_uint8), C.byref(off_p), C.byref(text_p)) == 0
    n = lift.n_items
    off = np.ctypeslib.as_array(off_p, shape=(n + 1,)).copy()
    text = np.ctypeslib.as_array(text_p, shape=(max(1, int(off[n])),)).copy()[: int(off[n])]
    L.emu_sa_free(off_p, text_p)
    return off, text, item_read
