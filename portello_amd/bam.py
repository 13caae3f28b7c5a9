"""ctypes binding of include/portello_bam.h: BAM/BGZF input windows, batches built from records, output record bytes,
BGZF output.  Mirrors the reference's reader / writer use in src/read_alignment_scanner.rs (see the header for the
line-by-line correspondence)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import abi, api

ERR_IO, ERR_DATA = 7, 8


class PloRecordsParams(C.Structure):
    _fields_ = [("index", C.POINTER(abi.PloIndexDesc)), ("contig_names", C.POINTER(C.c_char_p)), ("ref_names", C.POINTER(C.c_char_p)),
                ("is_target_region", C.c_int32), ("n_threads", C.c_int32)]


class PloRecordBuf(C.Structure):
    _fields_ = [("bytes", C.POINTER(C.c_uint8)), ("n_bytes", C.c_uint64), ("n_records", C.c_uint32), ("record_off", C.POINTER(C.c_uint64)),
                ("n_lifted", C.c_uint32), ("n_unmapped_copies", C.c_uint32)]


_bound = False


def lib():
    global _bound
    L = api.load_library()
    if not _bound:
        vp = C.c_void_p
        L.plo_bam_open.restype = C.c_int
        L.plo_bam_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.plo_bam_close.restype = None
        L.plo_bam_close.argtypes = [vp]
        L.plo_bam_header.restype = C.c_int
        L.plo_bam_header.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_char_p)),
                                     C.POINTER(C.POINTER(C.c_uint32))]
        L.plo_bam_read_window.restype = C.c_int
        L.plo_bam_read_window.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
        L.plo_bam_window_free.restype = None
        L.plo_bam_window_free.argtypes = [vp]
        L.plo_bam_window_n_records.restype = C.c_uint32
        L.plo_bam_window_n_records.argtypes = [vp]
        L.plo_bam_window_eof.restype = C.c_int
        L.plo_bam_window_eof.argtypes = [vp]
        L.plo_bam_window_unmapped.restype = None
        L.plo_bam_window_unmapped.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.plo_bam_window_batch.restype = C.c_int
        L.plo_bam_window_batch.argtypes = [vp, C.POINTER(abi.PloBatchIn), C.POINTER(abi.PloFinishIn)]
        L.plo_bam_window_batch_sparse_strand.restype = C.c_int
        L.plo_bam_window_batch_sparse_strand.argtypes = [vp, C.c_uint32, C.POINTER(abi.PloIndexDesc), C.POINTER(abi.PloBatchIn), C.POINTER(abi.PloFinishIn)]
        L.plo_bam_open_device.restype = C.c_int
        L.plo_bam_open_device.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
        L.plo_bam_window_record.restype = C.c_int
        L.plo_bam_window_record.argtypes = [vp, C.c_uint32, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32)]
        L.plo_bam_open_range.restype = C.c_int
        L.plo_bam_open_range.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.plo_bam_set_device_inflate.restype = None
        L.plo_bam_set_device_inflate.argtypes = [vp, C.c_int]
        L.plo_bam_window_batch_sparse.restype = C.c_int
        L.plo_bam_window_batch_sparse.argtypes = [vp, C.c_uint32, C.POINTER(abi.PloBatchIn), C.POINTER(abi.PloFinishIn)]
        L.plo_sparse_seq_bound.restype = C.c_uint64
        L.plo_sparse_seq_bound.argtypes = [C.POINTER(abi.PloBatchIn)]
        L.plo_sparse_seq_pack.restype = C.c_int
        L.plo_sparse_seq_pack.argtypes = [C.POINTER(abi.PloBatchIn), C.c_uint32, C.c_int, vp, C.c_uint64, vp, C.POINTER(abi.PloBatchIn)]
        L.plo_records_build.restype = C.c_int
        L.plo_records_build.argtypes = [vp, C.POINTER(abi.PloBatchOut), C.POINTER(PloRecordsParams), C.POINTER(PloRecordBuf)]
        L.plo_records_build_finished.restype = C.c_int
        L.plo_records_build_finished.argtypes = [vp, C.POINTER(abi.PloBatchOut), C.POINTER(abi.PloFinishOut), C.POINTER(abi.PloSaOut),
                                                 C.POINTER(PloRecordsParams), C.POINTER(PloRecordBuf)]
        L.plo_bam_output_header.restype = vp
        L.plo_bam_output_header.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_char_p, C.c_char_p, C.c_char_p]
        L.plo_bam_free_text.restype = None
        L.plo_bam_free_text.argtypes = [vp]
        L.plo_bam_writer_open.restype = C.c_int
        L.plo_bam_writer_open.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_int, C.c_int,
                                          C.POINTER(vp)]
        L.plo_bam_write.restype = C.c_int
        L.plo_bam_write.argtypes = [vp, C.c_void_p, C.c_uint64]
        L.plo_bam_writer_close.restype = C.c_int
        L.plo_bam_writer_close.argtypes = [vp]
        L.plo_bam_last_error.restype = C.c_char_p
        L.plo_bam_last_error.argtypes = []
        _bound = True
    return L


def _check(st: int, what: str):
    if st != 0:
        raise api.PortelloError(st, f"{what}: {lib().plo_bam_last_error().decode(errors='replace')}")


def _names(names: Sequence[str]):
    arr = (C.c_char_p * max(1, len(names)))(*[n.encode() if isinstance(n, str) else n for n in names])
    return arr


def sparse_pack(batch: abi.BatchData, margin: int = 32, n_threads: int = 4) -> abi.BatchData:
    """plo_sparse_seq_pack: the batch with its read bases as PLO_SEQ_BAM4_SPARSE (granules within `margin` bases of an indel of a
    read->contig CIGAR); seq_full / read_seq_full_off of the result point at the dense bases of `batch`, which must stay alive."""
    import dataclasses

    d = batch.to_desc()
    cap = int(lib().plo_sparse_seq_bound(C.byref(d)))
    out = np.zeros(cap, np.uint8)
    off = np.zeros(max(1, batch.n_reads), np.uint64)
    sp = abi.PloBatchIn()
    _check(lib().plo_sparse_seq_pack(C.byref(d), int(margin), int(n_threads), out.ctypes.data_as(C.c_void_p), cap, off.ctypes.data_as(C.c_void_p),
                                     C.byref(sp)), "plo_sparse_seq_pack")
    return dataclasses.replace(batch, seq=out[:int(sp.seq_bytes)], seq_fmt=abi.SEQ_BAM4_SPARSE, read_seq_off=off[:batch.n_reads],
                               seq_full=batch.seq, read_seq_full_off=batch.read_seq_off)


class Window:
    """a decoded window of primary records (plo_bam_window)"""

    def __init__(self, handle):
        self.handle = handle
        self._batch = None

    @property
    def n_records(self) -> int:
        return int(lib().plo_bam_window_n_records(self.handle))

    @property
    def eof(self) -> bool:
        """the reader reached the end of the file while collecting this window: nothing follows it"""
        return bool(lib().plo_bam_window_eof(self.handle))

    def record_bytes(self, i: int) -> bytes:
        """primary record i as it stands in the BAM stream (block_size word included)"""
        p, n = C.POINTER(C.c_uint8)(), C.c_uint32()
        _check(lib().plo_bam_window_record(self.handle, i, C.byref(p), C.byref(n)), "plo_bam_window_record")
        return C.string_at(p, n.value)

    def unmapped_bytes(self) -> Tuple[bytes, int]:
        p, n, k = C.POINTER(C.c_uint8)(), C.c_uint64(), C.c_uint32()
        lib().plo_bam_window_unmapped(self.handle, C.byref(p), C.byref(n), C.byref(k))
        return (C.string_at(p, n.value) if n.value else b""), int(k.value)

    def batch_desc(self, with_finish: bool = False, sparse_margin: Optional[int] = None, index_desc: Optional[abi.PloIndexDesc] = None):
        """plo_batch_in (host arrays owned by the window) [+ plo_finish_in].  sparse_margin: read bases as PLO_SEQ_BAM4_SPARSE --
        only the granules within that many bases of an indel of a read->contig CIGAR, the complete bases stay in the records
        (plo_bam_window_batch_sparse)"""
        b = abi.PloBatchIn()
        f = abi.PloFinishIn()
        if sparse_margin is not None and index_desc is not None:
            # (read segments that touch no reverse-mapped contig segment send the bases of their insertions only)
            _check(lib().plo_bam_window_batch_sparse_strand(self.handle, int(sparse_margin), C.byref(index_desc), C.byref(b), C.byref(f) if with_finish else None),
                   "plo_bam_window_batch_sparse_strand")
        elif sparse_margin is not None:
            _check(lib().plo_bam_window_batch_sparse(self.handle, int(sparse_margin), C.byref(b), C.byref(f) if with_finish else None),
                   "plo_bam_window_batch_sparse")
        else:
            _check(lib().plo_bam_window_batch(self.handle, C.byref(b), C.byref(f) if with_finish else None), "plo_bam_window_batch")
        self._batch = b
        return (b, f) if with_finish else b

    def batch_data(self) -> abi.BatchData:
        """numpy copy of the batch (for the oracle / the host-buffer entry point)"""
        b = self.batch_desc()
        n, ns = int(b.n_reads), int(b.n_segs)

        def cp(p, dt, cnt):
            return np.ctypeslib.as_array(p, shape=(cnt,)).astype(dt, copy=True) if cnt else np.zeros(0, dt)

        coff = cp(b.seg_cigar_off, np.uint32, ns + 1) if ns else np.zeros(1, np.uint32)
        return abi.BatchData(read_is_reverse=cp(b.read_is_reverse, np.uint8, n), read_seq_len=cp(b.read_seq_len, np.uint32, n),
                             read_seq_off=cp(b.read_seq_off, np.uint64, n), seq=cp(b.seq, np.uint8, int(b.seq_bytes)), seq_fmt=int(b.seq_fmt),
                             seg_read=cp(b.seg_read, np.uint32, ns), seg_contig=cp(b.seg_contig, np.uint32, ns),
                             seg_pos=cp(b.seg_pos, np.int64, ns), seg_is_fwd_strand=cp(b.seg_is_fwd_strand, np.uint8, ns),
                             seg_cigar_off=coff, cigar=cp(b.cigar, np.uint32, int(coff[-1])))

    def build_records(self, lift_out: abi.PloBatchOut, index_desc: abi.PloIndexDesc, contig_names: Sequence[str], ref_names: Sequence[str],
                      is_target_region: bool = False, n_threads: int = 0) -> Tuple[bytes, np.ndarray, int, int]:
        """BAM record bytes of the window's output (lifted records / unmapped copies): (bytes, record offsets, n lifted, n unmapped)"""
        cn, rn = _names(contig_names), _names(ref_names)
        pr = PloRecordsParams(C.pointer(index_desc), cn, rn, 1 if is_target_region else 0, n_threads)
        rb = PloRecordBuf()
        _check(lib().plo_records_build(self.handle, C.byref(lift_out), C.byref(pr), C.byref(rb)), "plo_records_build")
        data = C.string_at(rb.bytes, rb.n_bytes) if rb.n_bytes else b""
        off = np.ctypeslib.as_array(rb.record_off, shape=(int(rb.n_records) + 1,)).copy() if rb.n_records else np.zeros(1, np.uint64)
        return data, off, int(rb.n_lifted), int(rb.n_unmapped_copies)

    def build_records_raw(self, lift_out, index_desc, contig_names, ref_names, is_target_region=False, n_threads=0) -> PloRecordBuf:
        """the same without copying the bytes out (valid until the window's next plo_records_build / free)"""
        cn, rn = _names(contig_names), _names(ref_names)
        pr = PloRecordsParams(C.pointer(index_desc), cn, rn, 1 if is_target_region else 0, n_threads)
        rb = PloRecordBuf()
        _check(lib().plo_records_build(self.handle, C.byref(lift_out), C.byref(pr), C.byref(rb)), "plo_records_build")
        return rb

    def build_records_finished_raw(self, lift_out, fin_out, sa_out, index_desc, contig_names, ref_names, is_target_region=False,
                                   n_threads=0) -> PloRecordBuf:
        """the records from a batch finished on the device (plo_records_build_finished): `lift_out`, `fin_out`, `sa_out` are HOST
        copies of plo_liftover_batch_dev's, plo_finish_batch_dev's and plo_sa_segments_dev's results (devbatch.HostResults)"""
        cn, rn = _names(contig_names), _names(ref_names)
        pr = PloRecordsParams(C.pointer(index_desc), cn, rn, 1 if is_target_region else 0, n_threads)
        rb = PloRecordBuf()
        _check(lib().plo_records_build_finished(self.handle, C.byref(lift_out), C.byref(fin_out), C.byref(sa_out) if sa_out is not None else None,
                                                C.byref(pr), C.byref(rb)), "plo_records_build_finished")
        return rb

    def close(self):
        if self.handle:
            lib().plo_bam_window_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BamReader:
    def __init__(self, path: str, n_threads: int = 4, device_inflate: Optional[int] = None, part: Optional[int] = None, n_parts: int = 1):
        """device_inflate: HIP device index = BGZF blocks are inflated on that GPU (plo_bam_set_device_inflate), -1 / False = on the
        host; None = the environment (PLO_BGZF_DEVICE) decides, host inflate by default.
        part / n_parts: this reader's share of the file (plo_bam_open_range: a split by compressed offset, no index needed) -- one rank's
        or worker's records when several GPUs lift one BAM"""
        h = C.c_void_p()
        dev = -2 if device_inflate is None else (-1 if device_inflate is False else int(device_inflate))
        if part is None:
            _check(lib().plo_bam_open_device(path.encode(), n_threads, dev, C.byref(h)), f"plo_bam_open({path})")
        else:
            _check(lib().plo_bam_open_range(path.encode(), n_threads, dev, int(part), int(n_parts), C.byref(h)), f"plo_bam_open_range({path}, {part}/{n_parts})")
        self.handle = h
        text, lt, n = C.c_char_p(), C.c_uint32(), C.c_uint32()
        names, lens = C.POINTER(C.c_char_p)(), C.POINTER(C.c_uint32)()
        _check(lib().plo_bam_header(h, C.byref(text), C.byref(lt), C.byref(n), C.byref(names), C.byref(lens)), "plo_bam_header")
        self.header_text = C.string_at(text, lt.value).decode(errors="replace") if lt.value else ""
        self.ref_names: List[str] = [names[i].decode() for i in range(n.value)]
        self.ref_lens: List[int] = [int(lens[i]) for i in range(n.value)]

    def read_window(self, max_records: int) -> Optional[Window]:
        """next window of at most max_records primary records; None at the end of the file (a final window may carry only
        unmapped records)"""
        h = C.c_void_p()
        _check(lib().plo_bam_read_window(self.handle, max_records, C.byref(h)), "plo_bam_read_window")
        w = Window(h)
        if w.n_records == 0 and w.unmapped_bytes()[1] == 0:
            w.close()
            return None
        return w

    def close(self):
        if self.handle:
            lib().plo_bam_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def output_header(ref_names: Sequence[str], ref_lens: Sequence[int], program_name="portello", program_version="0.6.1", cmdline="") -> str:
    rn = _names(ref_names)
    rl = (C.c_uint32 * max(1, len(ref_lens)))(*[int(x) for x in ref_lens])
    p = lib().plo_bam_output_header(len(ref_names), rn, rl, program_name.encode(), program_version.encode(), cmdline.encode())
    s = C.string_at(p).decode()
    lib().plo_bam_free_text(p)
    return s


class BamWriter:
    def __init__(self, path: str, header_text: str, ref_names: Sequence[str], ref_lens: Sequence[int], level: int = 0, n_threads: int = 4):
        h = C.c_void_p()
        rn = _names(ref_names)
        rl = (C.c_uint32 * max(1, len(ref_lens)))(*[int(x) for x in ref_lens])
        _check(lib().plo_bam_writer_open(path.encode(), header_text.encode(), len(ref_names), rn, rl, level, n_threads, C.byref(h)),
               f"plo_bam_writer_open({path})")
        self.handle = h

    def write(self, data):
        if isinstance(data, (bytes, bytearray)):
            _check(lib().plo_bam_write(self.handle, data, len(data)), "plo_bam_write")
        elif isinstance(data, np.ndarray):
            a = np.ascontiguousarray(data, dtype=np.uint8)
            _check(lib().plo_bam_write(self.handle, a.ctypes.data_as(C.c_void_p), a.nbytes), "plo_bam_write")
        else:  # (pointer, n_bytes)
            _check(lib().plo_bam_write(self.handle, C.cast(data[0], C.c_void_p), int(data[1])), "plo_bam_write")

    def close(self):
        if self.handle:
            h, self.handle = self.handle, None
            _check(lib().plo_bam_writer_close(h), "plo_bam_writer_close")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- phase 1 (index construction) ----------------------------------------------------------------------------------------------
class PloTargetRegion(C.Structure):
    _fields_ = [("chrom_index", C.c_uint32), ("start", C.c_int64), ("end", C.c_int64)]


class Phase1:
    """plo_phase1_scan: the contig->reference index built from the assembly->reference BAM (scan_contig_bam,
    src/contig_alignment_scanner/mod.rs:290-459)"""

    def __init__(self, asm_to_ref_bam: str, contig_names: Sequence[str], contig_lens: Sequence[int], target_region=None, n_threads: int = 4):
        L = lib()
        L.plo_phase1_scan.restype = C.c_int
        L.plo_phase1_scan.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(PloTargetRegion), C.c_int,
                                      C.POINTER(C.c_void_p)]
        L.plo_phase1_index_desc.restype = C.c_int
        L.plo_phase1_index_desc.argtypes = [C.c_void_p, C.POINTER(abi.PloIndexDesc)]
        L.plo_phase1_info.restype = C.c_int
        L.plo_phase1_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_uint32)),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.plo_phase1_free.restype = None
        L.plo_phase1_free.argtypes = [C.c_void_p]
        cn = _names(contig_names)
        cl = (C.c_int64 * max(1, len(contig_lens)))(*[int(x) for x in contig_lens])
        tr = PloTargetRegion(*target_region) if target_region is not None else None
        h = C.c_void_p()
        _check(L.plo_phase1_scan(asm_to_ref_bam.encode(), len(contig_names), cn, cl, C.byref(tr) if tr is not None else None, n_threads,
                                 C.byref(h)), "plo_phase1_scan")
        self.handle = h
        n, names, lens = C.c_uint32(), C.POINTER(C.c_char_p)(), C.POINTER(C.c_uint32)()
        a, b, r = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(L.plo_phase1_info(h, C.byref(n), C.byref(names), C.byref(lens), C.byref(a), C.byref(b), C.byref(r)), "plo_phase1_info")
        self.ref_names = [names[i].decode() for i in range(n.value)]
        self.ref_lens = [int(lens[i]) for i in range(n.value)]
        self.segments_clipped, self.segments_joined, self.n_records = int(a.value), int(b.value), int(r.value)

    def index_data(self, chrom_seq) -> abi.IndexData:
        """numpy copy of the index description; chrom_seq = the reference sequences (ASCII arrays or device pointers)"""
        d = abi.PloIndexDesc()
        _check(lib().plo_phase1_index_desc(self.handle, C.byref(d)), "plo_phase1_index_desc")
        nc, ns = int(d.n_contigs), int(d.n_segments)

        def cp(p, dt, cnt):
            return np.ctypeslib.as_array(p, shape=(cnt,)).astype(dt, copy=True) if cnt else np.zeros(0, dt)

        coff = cp(d.seg_cigar_off, np.uint32, ns + 1) if ns else np.zeros(1, np.uint32)
        clen = cp(d.contig_len, np.int64, nc)
        revs = []
        for c in range(nc):
            p = d.rev_contig_seq[c]
            revs.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(clen[c]),)).copy() if p else None)
        return abi.IndexData(contig_len=clen, contig_seg_off=cp(d.contig_seg_off, np.uint32, nc + 1), seg_chrom_index=cp(d.seg_chrom_index, np.uint32, ns),
                             seg_pos=cp(d.seg_pos, np.int64, ns), seg_is_fwd_strand=cp(d.seg_is_fwd_strand, np.uint8, ns),
                             seg_mapq=cp(d.seg_mapq, np.uint8, ns), seg_seq_order_start=cp(d.seg_seq_order_start, np.int64, ns),
                             seg_seq_order_end=cp(d.seg_seq_order_end, np.int64, ns), seg_cigar_off=coff,
                             seg_cigar=cp(d.seg_cigar, np.uint32, int(coff[-1])), chrom_seq=list(chrom_seq), rev_contig_seq=revs)

    def close(self):
        if getattr(self, "handle", None):
            lib().plo_phase1_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
