"""Host-side mirror of the reference's interface for the liftover path.

The three public functions carry the reference's names and argument meaning
(``liftover_read_alignment`` src/liftover_read_alignment.rs:137-141, ``simplify_alignment_indels``
src/simplify_alignment_indels.rs:119-124, ``left_shift_indels``
lib/rust-vc-utils/src/bam_utils/cigar/shift_indels/left_shift_indels.rs:17-22) and are thin conveniences over the
batch entry point: each call becomes a one-item batch for ``plo_liftover_batch`` with the matching stage mask.
All computation happens in the HIP library (``libportello_liftover.so``); if it is missing or no GPU is usable the
calls raise -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import abi
from . import cigar as cg

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLO_LIB") or os.path.join(_HERE, "libportello_liftover.so")  # (PLO_LIB: timing-experiment builds, tools/)

Backend = Callable[[abi.IndexData, abi.BatchData, int], abi.BatchResult]


class PortelloError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"portello_liftover status {status}: {msg}")
        self.status = status


_lib = None


def load_library(path: Optional[str] = None):
    """Load the C-ABI library (built by ``python -m portello_amd.build`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise PortelloError(abi.PLO_ERR_NO_DEVICE, f"{path} not built: run `python -m portello_amd.build` (hipcc, gfx950)")
    try:
        # In a process that also uses torch, torch's own HIP runtime must be the one the library binds to: loaded the other
        # way round, two runtimes share the process and the engine's one does not see the device.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp = C.c_void_p
    if not hasattr(L, "plo_api_version") or L.plo_api_version() != abi.PLO_API_VERSION:
        raise PortelloError(abi.PLO_ERR_INVALID_ARG, f"{path} was built for another PLO_API_VERSION than this binding ({abi.PLO_API_VERSION}): rebuild it")
    L.plo_index_create.restype = C.c_int
    L.plo_index_create.argtypes = [C.POINTER(abi.PloIndexDesc), C.c_int, C.POINTER(vp)]
    L.plo_index_destroy.restype = None
    L.plo_index_destroy.argtypes = [vp]
    L.plo_index_segment_map.restype = C.c_int
    L.plo_index_segment_map.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint32)]
    L.plo_ctx_create.restype = C.c_int
    L.plo_ctx_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.plo_ctx_destroy.restype = None
    L.plo_ctx_destroy.argtypes = [vp]
    L.plo_liftover_batch.restype = C.c_int
    L.plo_liftover_batch.argtypes = [vp, C.POINTER(abi.PloBatchIn), C.c_uint32, C.POINTER(abi.PloBatchOut)]
    L.plo_liftover_batch_dev.restype = C.c_int
    L.plo_liftover_batch_dev.argtypes = [vp, C.POINTER(abi.PloBatchIn), C.c_uint32, C.POINTER(abi.PloBatchOut)]
    L.plo_compact_output_dev.restype = C.c_int
    L.plo_compact_output_dev.argtypes = [vp, C.POINTER(abi.PloBatchOut)]
    L.plo_host_alloc.restype = C.c_int
    L.plo_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.plo_host_free.restype = None
    L.plo_host_free.argtypes = [C.c_void_p]
    L.plo_sa_segments_dev.restype = C.c_int
    L.plo_sa_segments_dev.argtypes = [vp, C.POINTER(abi.PloSaIn), C.POINTER(abi.PloSaOut)]
    L.plo_finish_batch_dev.restype = C.c_int
    L.plo_finish_batch_dev.argtypes = [vp, C.POINTER(abi.PloBatchIn), C.POINTER(abi.PloFinishIn), C.POINTER(abi.PloFinishOut)]
    L.plo_ctx_sync.restype = C.c_int
    L.plo_ctx_sync.argtypes = [vp]
    L.plo_ctx_download.restype = C.c_int
    L.plo_ctx_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.plo_gather_unique_id.restype = C.c_int
    L.plo_gather_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.plo_gather_create.restype = C.c_int
    L.plo_gather_create.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.plo_gather_destroy.restype = None
    L.plo_gather_destroy.argtypes = [vp]
    L.plo_gather_records.restype = C.c_int
    L.plo_gather_records.argtypes = [vp, vp, C.POINTER(abi.PloBatchOut), C.c_int, C.POINTER(abi.PloBatchOut)]
    L.plo_gather_wait.restype = C.c_int
    L.plo_gather_wait.argtypes = [vp]
    L.plo_gather_last_error.restype = C.c_char_p
    L.plo_gather_last_error.argtypes = [vp]
    L.plo_ctx_stream.restype = vp
    L.plo_ctx_stream.argtypes = [vp]
    L.plo_ctx_device.restype = C.c_int
    L.plo_ctx_device.argtypes = [vp]
    L.plo_ctx_set_stats.restype = C.c_int
    L.plo_ctx_set_stats.argtypes = [vp, C.c_int]
    L.plo_ctx_set_phase_events.restype = C.c_int
    L.plo_ctx_set_phase_events.argtypes = [vp, C.c_int]
    L.plo_api_version.restype = C.c_uint32
    L.plo_ctx_timing.restype = C.c_int
    L.plo_ctx_timing.argtypes = [vp, C.POINTER(abi.PloTiming)]
    L.plo_last_error.restype = C.c_char_p
    L.plo_last_error.argtypes = [vp]
    L.plo_version.restype = C.c_char_p
    L.plo_version.argtypes = []
    _lib = L
    return L


class Index:
    """Device-resident contig->reference index (plo_index).  Immutable; shareable between engines."""

    def __init__(self, data: abi.IndexData, device: int = 0):
        self.lib = load_library()
        self.data = data
        self._desc = data.to_desc()
        h = C.c_void_p()
        st = self.lib.plo_index_create(C.byref(self._desc), device, C.byref(h))
        if st != abi.PLO_OK:
            raise PortelloError(st, "plo_index_create failed" + (" (no usable HIP device)" if st == abi.PLO_ERR_NO_DEVICE else ""))
        self.handle = h
        self.device = device

    def segment_map(self, global_seg: int) -> Tuple[np.ndarray, np.ndarray]:
        n = C.c_uint32(0)
        self.lib.plo_index_segment_map(self.handle, global_seg, 0, None, None, C.byref(n))
        keys = np.zeros(max(1, n.value), dtype=np.int64)
        vals = np.zeros(max(1, n.value), dtype=np.int64)
        st = self.lib.plo_index_segment_map(self.handle, global_seg, n.value, keys.ctypes.data_as(C.POINTER(C.c_int64)),
                                            vals.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n))
        if st != abi.PLO_OK:
            raise PortelloError(st, "plo_index_segment_map")
        return keys[: n.value], vals[: n.value]

    def close(self):
        if getattr(self, "handle", None):
            self.lib.plo_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One worker context (plo_ctx): the per-thread home the reference gives BamReaderWorkerThreadData
    (src/worker_thread_data.rs:8-18)."""

    def __init__(self, index: Index, stream: Optional[int] = None):
        """stream: a hipStream_t handle to run on, or None / 0 for a private (non-blocking) stream of the context.  torch's
        *default* stream is the NULL handle, i.e. 0: a context created with `torch.cuda.current_stream().cuda_stream` while the
        default stream is current gets a private stream and is NOT ordered behind torch work -- synchronise before handing it
        tensors torch has just written (devbatch.run_and_download does), or run under a `torch.cuda.Stream` of your own."""
        self.lib = index.lib
        self.index = index
        h = C.c_void_p()
        st = self.lib.plo_ctx_create(index.handle, C.c_void_p(stream) if stream else None, C.byref(h))
        if st != abi.PLO_OK:
            raise PortelloError(st, "plo_ctx_create failed")
        self.handle = h

    def set_stats(self, on: bool = True):
        """plo_ctx_set_stats: the light-item kernel counts timing().algo_bytes / lane_utilisation from the next call on (its production
        instantiation is compiled without the counters)"""
        self._check(self.lib.plo_ctx_set_stats(self.handle, 1 if on else 0), "plo_ctx_set_stats")
        return self

    def set_phase_events(self, on: bool = True):
        """plo_ctx_set_phase_events: off = the one-round-trip calls record no HIP events between their phases (two ~6 us bubbles fewer per
        call on the stream); timing() then carries counts but no times for such calls"""
        self._check(self.lib.plo_ctx_set_phase_events(self.handle, 1 if on else 0), "plo_ctx_set_phase_events")
        return self

    def _check(self, st: int, what: str):
        if st != abi.PLO_OK:
            msg = self.lib.plo_last_error(self.handle)
            raise PortelloError(st, f"{what}: {msg.decode() if msg else ''}")

    def liftover_batch(self, batch: abi.BatchData, stages: int = abi.STAGES_ALL) -> abi.BatchResult:
        """Host arrays in, host arrays out (plo_liftover_batch)."""
        b = batch.to_desc()
        out = abi.PloBatchOut()
        self._check(self.lib.plo_liftover_batch(self.handle, C.byref(b), stages, C.byref(out)), "plo_liftover_batch")
        return abi.result_from_out(out)

    def liftover_batch_host(self, desc: abi.PloBatchIn, stages: int = abi.STAGES_ALL) -> abi.PloBatchOut:
        """plo_liftover_batch on a descriptor of host arrays (e.g. a BAM window's batch); the returned arrays are pinned host
        memory owned by the context, valid until its next call."""
        out = abi.PloBatchOut()
        self._check(self.lib.plo_liftover_batch(self.handle, C.byref(desc), stages, C.byref(out)), "plo_liftover_batch")
        return out

    def liftover_batch_dev(self, desc: abi.PloBatchIn, stages: int = abi.STAGES_ALL) -> abi.PloBatchOut:
        """Device pointers in, device pointers out (plo_liftover_batch_dev); asynchronous on the engine's stream."""
        out = abi.PloBatchOut()
        self._check(self.lib.plo_liftover_batch_dev(self.handle, C.byref(desc), stages, C.byref(out)), "plo_liftover_batch_dev")
        return out

    def finish_batch_dev(self, desc: abi.PloBatchIn, fin: abi.PloFinishIn) -> abi.PloFinishOut:
        """Record finishing for the last liftover_batch_dev result (plo_finish_batch_dev); device pointers."""
        out = abi.PloFinishOut()
        self._check(self.lib.plo_finish_batch_dev(self.handle, C.byref(desc), C.byref(fin), C.byref(out)), "plo_finish_batch_dev")
        return out

    def compact_output_dev(self, out: abi.PloBatchOut) -> abi.PloBatchOut:
        """Pack the output CIGARs of the last liftover_batch_dev result densely (plo_compact_output_dev); updates `out`."""
        self._check(self.lib.plo_compact_output_dev(self.handle, C.byref(out)), "plo_compact_output_dev")
        return out

    def sa_segments_dev(self, sa_in: abi.PloSaIn) -> abi.PloSaOut:
        """SA-tag segments of the last liftover + finish result (plo_sa_segments_dev); device pointers."""
        out = abi.PloSaOut()
        self._check(self.lib.plo_sa_segments_dev(self.handle, C.byref(sa_in), C.byref(out)), "plo_sa_segments_dev")
        return out

    def download(self, dev_ptr, dtype, count: int) -> np.ndarray:
        """copy `count` elements of `dtype` from a device pointer (e.g. a plo_batch_out array) to a numpy array"""
        out = np.zeros(max(1, count), dtype=dtype)
        addr = C.cast(dev_ptr, C.c_void_p).value if not isinstance(dev_ptr, int) else dev_ptr
        if count:
            self._check(self.lib.plo_ctx_download(self.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(addr),
                                                  count * out.itemsize), "plo_ctx_download")
        return out[:count]

    def sync(self):
        self._check(self.lib.plo_ctx_sync(self.handle), "plo_ctx_sync")

    def timing(self) -> abi.PloTiming:
        t = abi.PloTiming()
        t.struct_size = C.sizeof(abi.PloTiming)
        self._check(self.lib.plo_ctx_timing(self.handle, C.byref(t)), "plo_ctx_timing")
        return t

    def close(self):
        if getattr(self, "handle", None):
            self.lib.plo_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hip_backend(device: int = 0) -> Backend:
    """Backend that builds a throw-away index + engine per call (unit-level calls and tests)."""

    def run(index: abi.IndexData, batch: abi.BatchData, stages: int) -> abi.BatchResult:
        ix = Index(index, device)
        try:
            eng = Engine(ix)
            try:
                return eng.liftover_batch(batch, stages)
            finally:
                eng.close()
        finally:
            ix.close()

    return run


# ----------------------------------------------------------------------------------------------------------------
# one-item cases: the reference's function signatures expressed as batch inputs
# ----------------------------------------------------------------------------------------------------------------

class CaseSet:
    """Accumulates independent single-call cases into one batch (one contig + one read per case)."""

    def __init__(self):
        self.contig_len: List[int] = []
        self.seg_chrom: List[int] = []
        self.seg_pos: List[int] = []
        self.seg_cigar: List[np.ndarray] = []
        self.chrom_seq: List[np.ndarray] = []
        self.rev_seq: List[Optional[np.ndarray]] = []
        self.read_seq: List[np.ndarray] = []
        self.read_pos: List[int] = []
        self.read_cigar: List[np.ndarray] = []

    def _add(self, map_pos, map_cigar, chrom_seq, rev_seq, contig_len, read_pos, read_cigar, read_seq):
        self.contig_len.append(int(contig_len))
        self.seg_chrom.append(len(self.chrom_seq))
        self.seg_pos.append(int(map_pos))
        self.seg_cigar.append(np.asarray(map_cigar, dtype=np.uint32))
        self.chrom_seq.append(np.frombuffer(bytes(chrom_seq), dtype=np.uint8))
        self.rev_seq.append(None if rev_seq is None else np.frombuffer(bytes(rev_seq), dtype=np.uint8))
        self.read_pos.append(int(read_pos))
        self.read_cigar.append(np.asarray(read_cigar, dtype=np.uint32))
        self.read_seq.append(np.frombuffer(bytes(read_seq), dtype=np.uint8))
        return len(self.contig_len) - 1

    def add_liftover(self, map_pos: int, map_cigar, start: int, cigar) -> int:
        cigar = np.asarray(cigar, dtype=np.uint32)
        n = cg.read_len(cigar)
        return self._add(map_pos, map_cigar, b"N", None, 1 << 30, start, cigar, b"N" * n)

    def add_simplify(self, ref_pos: int, cigar, ref_seq: bytes, read_seq: bytes) -> int:
        return self._add(0, np.zeros(0, np.uint32), ref_seq, None, 1 << 30, ref_pos, cigar, read_seq)

    def add_left_shift(self, ref_pos: int, cigar, ref_seq: bytes, read_seq: bytes) -> int:
        return self._add(0, np.zeros(0, np.uint32), b"N", ref_seq, len(ref_seq), ref_pos, cigar, read_seq)

    def build(self) -> Tuple[abi.IndexData, abi.BatchData]:
        n = len(self.contig_len)
        seg_cigar_off = np.zeros(n + 1, dtype=np.uint32)
        seg_cigar_off[1:] = np.cumsum([len(c) for c in self.seg_cigar])
        index = abi.IndexData(
            contig_len=np.array(self.contig_len, dtype=np.int64), contig_seg_off=np.arange(n + 1, dtype=np.uint32),
            seg_chrom_index=np.array(self.seg_chrom, dtype=np.uint32), seg_pos=np.array(self.seg_pos, dtype=np.int64),
            seg_is_fwd_strand=np.ones(n, dtype=np.uint8), seg_mapq=np.full(n, 60, dtype=np.uint8),
            seg_seq_order_start=np.zeros(n, dtype=np.int64), seg_seq_order_end=np.array(self.contig_len, dtype=np.int64),
            seg_cigar_off=seg_cigar_off,
            seg_cigar=np.concatenate(self.seg_cigar) if n else np.zeros(0, np.uint32),
            chrom_seq=self.chrom_seq, rev_contig_seq=self.rev_seq)
        seq_len = np.array([len(s) for s in self.read_seq], dtype=np.uint32)
        seq_off = np.zeros(n, dtype=np.uint64)
        if n:
            seq_off[1:] = np.cumsum(seq_len[:-1])
        rc_off = np.zeros(n + 1, dtype=np.uint32)
        rc_off[1:] = np.cumsum([len(c) for c in self.read_cigar])
        batch = abi.BatchData(
            read_is_reverse=np.zeros(n, dtype=np.uint8), read_seq_len=seq_len, read_seq_off=seq_off,
            seq=np.concatenate(self.read_seq) if n else np.zeros(0, np.uint8), seq_fmt=abi.SEQ_ASCII,
            seg_read=np.arange(n, dtype=np.uint32), seg_contig=np.arange(n, dtype=np.uint32),
            seg_pos=np.array(self.read_pos, dtype=np.int64), seg_is_fwd_strand=np.ones(n, dtype=np.uint8),
            seg_cigar_off=rc_off, cigar=np.concatenate(self.read_cigar) if n else np.zeros(0, np.uint32),
            item_seg=np.arange(n, dtype=np.uint32), item_cseg=np.zeros(n, dtype=np.uint32))
        return index, batch

    def run(self, stages: int, backend: Optional[Backend] = None):
        """-> list of None | (pos, cigar) per case, in insertion order"""
        backend = backend or hip_backend()
        index, batch = self.build()
        res = backend(index, batch, stages)
        assert res.n_items == len(self.contig_len)
        out = [None] * res.n_items
        for i in range(res.n_items):
            k = int(res.item_seg[i])
            st = int(res.item_status[i])
            if st == abi.ITEM_LIFTED:
                out[k] = (int(res.item_ref_pos[i]), res.item_cigar(i).copy())
            elif st == abi.ITEM_NO_LIFTOVER:
                out[k] = None
            else:
                raise PortelloError(abi.PLO_ERR_INTERNAL, f"case {k}: item status {st} (the reference would panic)")
        return out


def liftover_read_alignment(ref1_to_ref2_map: Tuple[int, Sequence[int]], ref1_cigar_segment_start_pos: int,
                            ref1_cigar: Sequence[int], backend: Optional[Backend] = None):
    """liftover_read_alignment(&ReadToRefTreeMap, i64, &[Cigar]) -> Option<(i64, Vec<Cigar>)>
    (src/liftover_read_alignment.rs:137-141).  The map is given by what it is built from: ``(ref_pos, cigar)`` of
    the contig segment alignment, i.e. the arguments of get_read_segment_to_ref_pos_tree_map."""
    cs = CaseSet()
    cs.add_liftover(ref1_to_ref2_map[0], ref1_to_ref2_map[1], ref1_cigar_segment_start_pos, ref1_cigar)
    return cs.run(abi.STAGE_LIFTOVER, backend)[0]


def simplify_alignment_indels(ref_pos: int, cigar: Sequence[int], ref_seq: bytes, read_seq: bytes,
                              backend: Optional[Backend] = None):
    """simplify_alignment_indels(i64, &[Cigar], &[u8], &[u8]) -> (i64, Vec<Cigar>) (src/simplify_alignment_indels.rs:119-124)"""
    cs = CaseSet()
    cs.add_simplify(ref_pos, cigar, ref_seq, read_seq)
    return cs.run(abi.STAGE_SIMPLIFY, backend)[0]


def left_shift_indels(ref_pos: int, cigar: Sequence[int], ref_seq: bytes, read_seq: bytes, backend: Optional[Backend] = None):
    """left_shift_indels(i64, &[Cigar], &[u8], &[u8]) -> (i64, Vec<Cigar>)
    (lib/rust-vc-utils/src/bam_utils/cigar/shift_indels/left_shift_indels.rs:17-22)"""
    cs = CaseSet()
    cs.add_left_shift(ref_pos, cigar, ref_seq, read_seq)
    return cs.run(abi.STAGE_LSHIFT, backend)[0]


def assemble_sa_values(seg_off: np.ndarray, text: np.ndarray, item_read: np.ndarray) -> list:
    """SA:Z value of every item from the per-item segments (src/read_alignment_scanner.rs:352-364): the segments of the
    read's other items, in item order; None where the record gets no SA tag.  Items of a read are consecutive."""
    n = len(item_read)
    t = text.tobytes()
    segs = [t[int(seg_off[i]):int(seg_off[i + 1])] for i in range(n)]
    out = [None] * n
    i0 = 0
    while i0 < n:
        i1 = i0
        while i1 < n and item_read[i1] == item_read[i0]:
            i1 += 1
        for i in range(i0, i1):
            if segs[i]:
                v = b"".join(segs[j] for j in range(i0, i1) if j != i)
                out[i] = v if v else None
        i0 = i1
    return out


class PinnedArray:
    """numpy view of page-locked host memory from plo_host_alloc (released on close / garbage collection)"""

    def __init__(self, like: np.ndarray):
        like = np.ascontiguousarray(like)
        self.lib = load_library()
        self.ptr = C.c_void_p()
        rc = self.lib.plo_host_alloc(max(16, like.nbytes), C.byref(self.ptr))
        if rc != abi.PLO_OK:
            raise PortelloError(rc, "plo_host_alloc failed")
        buf = (C.c_uint8 * max(16, like.nbytes)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=like.dtype, count=like.size).reshape(like.shape)
        self.array[...] = like

    def close(self):
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            self.array = None
            self.lib.plo_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
