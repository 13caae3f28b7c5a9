"""Device-resident batches: builds a ``plo_batch_in`` whose arrays are torch tensors already in HBM (the form the
bench times: inputs resident on the GPU when the timed region starts), and downloads ``plo_batch_out`` arrays."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import abi


def _p(t: torch.Tensor, ctype):
    return C.cast(C.c_void_p(t.data_ptr()), C.POINTER(ctype))


@dataclass
class DeviceBatch:
    read_is_reverse: torch.Tensor  # uint8
    read_seq_len: torch.Tensor     # int32 (bit pattern of uint32)
    read_seq_off: torch.Tensor     # int64 (bit pattern of uint64)
    seq: torch.Tensor              # uint8
    seq_fmt: int
    seg_read: torch.Tensor         # int32
    seg_contig: torch.Tensor       # int32
    seg_pos: torch.Tensor          # int64
    seg_is_fwd_strand: torch.Tensor  # uint8
    seg_cigar_off: torch.Tensor    # int32 [n_segs+1]
    cigar: torch.Tensor            # int32 (bit pattern of uint32)

    @property
    def n_reads(self) -> int:
        return int(self.read_seq_len.numel())

    @property
    def n_segs(self) -> int:
        return int(self.seg_read.numel())

    @classmethod
    def from_workload(cls, w, read_lo: int = 0, read_hi: Optional[int] = None) -> "DeviceBatch":
        """Slice reads [read_lo, read_hi) of a synth.Workload living on the GPU (offsets re-based)."""
        dev = w.device
        assert dev.type == "cuda"
        hi = w.n_reads if read_hi is None else read_hi
        lo = read_lo
        seg_lo = int(torch.searchsorted(w.seg_read, torch.tensor(lo, device=dev)).item())
        seg_hi = int(torch.searchsorted(w.seg_read, torch.tensor(hi, device=dev)).item())
        co = w.seg_cigar_off_r[seg_lo: seg_hi + 1]
        c0 = int(co[0].item()) if co.numel() else 0
        c1 = int(co[-1].item()) if co.numel() else 0
        so = w.read_seq_off[lo:hi]
        sl = w.read_seq_len[lo:hi]
        if hi > lo:
            b0 = int(so[0].item())
            last = int(sl[-1].item())
            b1 = int(so[-1].item()) + ((last + 1) // 2 if w.cfg.seq_fmt == abi.SEQ_BAM4 else last)
        else:
            b0 = b1 = 0
        return cls(
            read_is_reverse=w.read_is_reverse[lo:hi].contiguous(), read_seq_len=sl.to(torch.int32).contiguous(),
            read_seq_off=(so - b0).to(torch.int64).contiguous(), seq=w.seq[b0:b1].contiguous(), seq_fmt=w.cfg.seq_fmt,
            seg_read=(w.seg_read[seg_lo:seg_hi] - lo).to(torch.int32).contiguous(),
            seg_contig=w.seg_contig[seg_lo:seg_hi].to(torch.int32).contiguous(),
            seg_pos=w.seg_pos_r[seg_lo:seg_hi].to(torch.int64).contiguous(),
            seg_is_fwd_strand=w.seg_is_fwd_r[seg_lo:seg_hi].contiguous(),
            seg_cigar_off=(co - c0).to(torch.int32).contiguous(), cigar=w.cigar[c0:c1].to(torch.int32).contiguous())

    @classmethod
    def from_batch_data(cls, b: abi.BatchData, device="cuda") -> "DeviceBatch":
        """upload of a host-side batch (numpy arrays) as it is"""
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt) if a.dtype.itemsize == np.dtype(dt).itemsize else
                                           np.ascontiguousarray(a).astype(dt)).to(device)
        return cls(read_is_reverse=t(b.read_is_reverse, np.uint8), read_seq_len=t(b.read_seq_len, np.int32), read_seq_off=t(b.read_seq_off, np.int64),
                   seq=t(b.seq if len(b.seq) else np.zeros(1, np.uint8), np.uint8), seq_fmt=b.seq_fmt, seg_read=t(b.seg_read, np.int32),
                   seg_contig=t(b.seg_contig, np.int32), seg_pos=t(b.seg_pos, np.int64), seg_is_fwd_strand=t(b.seg_is_fwd_strand, np.uint8),
                   seg_cigar_off=t(b.seg_cigar_off, np.int32), cigar=t(b.cigar if len(b.cigar) else np.zeros(1, np.uint32), np.int32))

    @classmethod
    def from_read_ranges(cls, w, ranges) -> "DeviceBatch":
        """The reads of several ranges [lo, hi) of a workload as ONE batch (a rank's windows, portello_amd/shard.py):
        the slices are concatenated, offsets re-based.  Read r of range k becomes read (reads of ranges < k) + r - lo."""
        parts = [cls.from_workload(w, lo, hi) for lo, hi in ranges if hi > lo]
        if not parts:
            return cls.from_workload(w, 0, 0)
        if len(parts) == 1:
            return parts[0]
        rb = cb = sb = 0
        seg_read, coff, soff = [], [], []
        for p_ in parts:
            seg_read.append(p_.seg_read + rb)
            coff.append(p_.seg_cigar_off[:-1] + cb)
            soff.append(p_.read_seq_off + sb)
            rb += p_.n_reads
            cb += int(p_.cigar.numel())
            sb += int(p_.seq.numel())
        coff.append(torch.tensor([cb], dtype=torch.int32, device=parts[0].cigar.device))
        cat = lambda name: torch.cat([getattr(p_, name) for p_ in parts]).contiguous()
        return cls(read_is_reverse=cat("read_is_reverse"), read_seq_len=cat("read_seq_len"), read_seq_off=torch.cat(soff).contiguous(),
                   seq=cat("seq"), seq_fmt=parts[0].seq_fmt, seg_read=torch.cat(seg_read).to(torch.int32).contiguous(),
                   seg_contig=cat("seg_contig"), seg_pos=cat("seg_pos"), seg_is_fwd_strand=cat("seg_is_fwd_strand"),
                   seg_cigar_off=torch.cat(coff).to(torch.int32).contiguous(), cigar=cat("cigar"))

    def desc(self) -> abi.PloBatchIn:
        b = abi.PloBatchIn()
        b.n_reads = self.n_reads
        b.read_is_reverse = _p(self.read_is_reverse, C.c_uint8)
        b.read_seq_len = _p(self.read_seq_len, C.c_uint32)
        b.read_seq_off = _p(self.read_seq_off, C.c_uint64)
        b.seq = _p(self.seq, C.c_uint8)
        b.seq_bytes = self.seq.numel()
        b.seq_fmt = self.seq_fmt
        b.n_segs = self.n_segs
        b.seg_read = _p(self.seg_read, C.c_uint32)
        b.seg_contig = _p(self.seg_contig, C.c_uint32)
        b.seg_pos = _p(self.seg_pos, C.c_int64)
        b.seg_is_fwd_strand = _p(self.seg_is_fwd_strand, C.c_uint8)
        b.seg_cigar_off = _p(self.seg_cigar_off, C.c_uint32)
        b.cigar = _p(self.cigar, C.c_uint32)
        b.n_items = 0
        b.item_seg = C.cast(None, C.POINTER(C.c_uint32))
        b.item_cseg = C.cast(None, C.POINTER(C.c_uint32))
        return b


def download(eng, out: abi.PloBatchOut) -> abi.BatchResult:
    n, nc = int(out.n_items), int(out.n_cigar)
    return abi.BatchResult(
        item_seg=eng.download(out.item_seg, np.uint32, n), item_cseg=eng.download(out.item_cseg, np.uint32, n),
        item_status=eng.download(out.item_status, np.uint8, n), item_need_flipped=eng.download(out.item_need_flipped, np.uint8, n),
        item_mapq=eng.download(out.item_mapq, np.uint8, n), item_chrom_index=eng.download(out.item_chrom_index, np.uint32, n),
        item_ref_pos=eng.download(out.item_ref_pos, np.int64, n), item_cigar_off=eng.download(out.item_cigar_off, np.uint64, n),
        item_cigar_len=eng.download(out.item_cigar_len, np.uint32, n), cigar=eng.download(out.cigar, np.uint32, nc))


def run_and_download(eng, db: DeviceBatch, stages: int = abi.STAGES_ALL) -> abi.BatchResult:
    torch.cuda.synchronize()
    out = eng.liftover_batch_dev(db.desc(), stages)
    return download(eng, out)


def finish_inputs(w, db: DeviceBatch, seed: int = 1):
    """Synthetic record flags + base qualities for the record-finishing stage (device tensors + the plo_finish_in)."""
    dev = db.seq.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lens = db.read_seq_len.to(torch.int64)
    qoff = torch.cumsum(lens, 0) - lens
    total = int(lens.sum().item())
    qual = torch.randint(0, 94, (max(1, total),), generator=g, device=dev, dtype=torch.uint8)
    flags = (db.read_is_reverse.to(torch.int16) * 0x10) | (torch.randint(0, 2, (db.n_reads,), generator=g, device=dev).to(torch.int16) * 0x400)
    fin = abi.PloFinishIn(_p(flags, C.c_uint16), _p(qual, C.c_uint8), _p(qoff, C.c_uint64), total)
    return fin, dict(flags=flags, qual=qual, qoff=qoff)


def download_finish(eng, fo: abi.PloFinishOut, n_items: int, n_reads: int) -> dict:
    res = {name: eng.download(getattr(fo, name), dt, n_items) for name, dt in abi.FINISH_ITEM_FIELDS}
    res.update({name: eng.download(getattr(fo, name), dt, n_reads) for name, dt in abi.FINISH_READ_FIELDS})
    res["rev_seq"] = eng.download(fo.rev_seq, np.uint8, int(fo.rev_seq_bytes))
    res["rev_qual"] = eng.download(fo.rev_qual, np.uint8, int(fo.rev_qual_bytes))
    return res


def sa_inputs(names, dev):
    """chromosome labels as device arrays + the plo_sa_in"""
    enc = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
    off = np.zeros(len(enc) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(e) for e in enc])
    blob = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy()
    t_off = torch.from_numpy(off).to(dev)
    t_blob = torch.from_numpy(blob).to(dev)
    return abi.PloSaIn(len(enc), _p(t_off, C.c_uint32), _p(t_blob, C.c_uint8)), dict(off=t_off, blob=t_blob)


def download_sa(eng, so: abi.PloSaOut):
    off = eng.download(so.item_sa_off, np.uint32, int(so.n_items) + 1)
    text = eng.download(so.sa_text, np.uint8, int(so.sa_bytes))
    return off, text


# ---- a BAM window's batch on the device, finished there, results back on the host (pipeline.run_bam_to_bam, device_finish) --------

def _host_view(ptr, dtype, count: int) -> np.ndarray:
    if not count:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dtype).itemsize,)).view(dtype)


@dataclass
class UploadedWindow:
    """device copies of a window's plo_batch_in (dense bases) and plo_finish_in; keeps the tensors alive"""
    batch: DeviceBatch
    flags: torch.Tensor
    qual: torch.Tensor
    qual_off: torch.Tensor
    qual_bytes: int

    def finish_in(self) -> abi.PloFinishIn:
        return abi.PloFinishIn(_p(self.flags, C.c_uint16), _p(self.qual, C.c_uint8), _p(self.qual_off, C.c_uint64), self.qual_bytes)


def upload_window(desc: abi.PloBatchIn, fin: abi.PloFinishIn, dev) -> UploadedWindow:
    """copies the host arrays of a window's batch descriptor (bam.Window.batch_desc(with_finish=True)) to the device, on torch's
    current stream"""
    n, ns = int(desc.n_reads), int(desc.n_segs)

    def up(ptr, dtype, count, as_dtype=None):
        a = _host_view(ptr, dtype, count)
        if as_dtype is not None:
            a = a.view(as_dtype)
        t = torch.from_numpy(a) if count else torch.zeros(0, dtype=torch.from_numpy(np.zeros(1, as_dtype or dtype)).dtype)
        return t.to(dev, non_blocking=True) if count else torch.zeros(1, dtype=t.dtype, device=dev)[:0]

    coff = _host_view(desc.seg_cigar_off, np.uint32, ns + 1)
    n_ops = int(coff[-1]) if ns else 0
    b = DeviceBatch(read_is_reverse=up(desc.read_is_reverse, np.uint8, n), read_seq_len=up(desc.read_seq_len, np.uint32, n, np.int32),
                    read_seq_off=up(desc.read_seq_off, np.uint64, n, np.int64), seq=up(desc.seq, np.uint8, int(desc.seq_bytes)),
                    seq_fmt=int(desc.seq_fmt), seg_read=up(desc.seg_read, np.uint32, ns, np.int32), seg_contig=up(desc.seg_contig, np.uint32, ns, np.int32),
                    seg_pos=up(desc.seg_pos, np.int64, ns), seg_is_fwd_strand=up(desc.seg_is_fwd_strand, np.uint8, ns),
                    seg_cigar_off=up(desc.seg_cigar_off, np.uint32, ns + 1, np.int32), cigar=up(desc.cigar, np.uint32, n_ops, np.int32))
    return UploadedWindow(b, up(fin.read_flags, np.uint16, n, np.int16), up(fin.qual, np.uint8, int(fin.qual_bytes)),
                          up(fin.read_qual_off, np.uint64, n, np.int64), int(fin.qual_bytes))


class PinnedArena:
    """A worker's page-locked landing area for the results of one window at a time (round 5): every array of a HostResults is a slice of ONE
    pinned block that is reused from window to window, filled by asynchronous copies on the worker's stream and waited for once.  (Round 4
    downloaded each of the ~25 arrays into a fresh numpy array with a stream synchronisation of its own: page faults on 90 MB of new
    memory per 7 500-read window and staged copies into pageable memory.  Measured per 60 k reads: lift stage 0.317 -> 0.302 s, record
    assembly -- which reads these arrays -- 0.163 -> 0.102 s; EXPERIMENTS.md 5.15.)"""

    def __init__(self, nbytes: int = 64 << 20):
        self.buf = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        self.used = 0
        self.retired = []  # blocks outgrown while a window's arrays still point into them

    def reset(self):
        self.used = 0
        self.retired = []

    def take(self, nbytes: int) -> torch.Tensor:
        at = (self.used + 63) & ~63
        if at + nbytes > self.buf.numel():
            self.retired.append(self.buf)
            self.buf = torch.empty(max(at + nbytes, self.buf.numel() * 3 // 2), dtype=torch.uint8, pin_memory=True)
            at = 0
        self.used = at + nbytes
        return self.buf[at:at + nbytes]


class HostResults:
    """host copies of a context's device results for one batch -- plo_batch_out (compacted), plo_finish_out, plo_sa_out -- as the
    structs plo_records_build_finished takes (the numpy arrays behind the pointers live as long as this object; with an `arena`: until the
    arena's next reset)"""

    def __init__(self, eng, out: abi.PloBatchOut, fo: abi.PloFinishOut, so: Optional[abi.PloSaOut], n_reads: int, arena: Optional[PinnedArena] = None, dev=None):
        self._keep = []
        pending = []

        def dl(ptr, dtype, count, ctype):
            if arena is None or not count:
                a = eng.download(ptr, dtype, count)
            else:
                from .gather import device_view

                nb = count * np.dtype(dtype).itemsize
                dst = arena.take(nb)
                dst.copy_(device_view(ptr, nb, torch.uint8, dev), non_blocking=True)  # (torch's current stream = the engine's)
                pending.append(dst)
                a = dst.numpy().view(dtype)
            self._keep.append(a)
            return abi._ptr(a, ctype)

        n, nc = int(out.n_items), int(out.n_cigar)
        self.lift = abi.PloBatchOut()
        self.lift.n_items = n
        self.lift.n_cigar = nc
        for name, dt, ct, cnt in (("item_seg", np.uint32, C.c_uint32, n), ("item_cseg", np.uint32, C.c_uint32, n), ("item_status", np.uint8, C.c_uint8, n),
                                  ("item_need_flipped", np.uint8, C.c_uint8, n), ("item_mapq", np.uint8, C.c_uint8, n),
                                  ("item_chrom_index", np.uint32, C.c_uint32, n), ("item_ref_pos", np.int64, C.c_int64, n),
                                  ("item_cigar_off", np.uint64, C.c_uint64, n), ("item_cigar_len", np.uint32, C.c_uint32, n), ("cigar", np.uint32, C.c_uint32, nc)):
            setattr(self.lift, name, dl(getattr(out, name), dt, cnt, ct))
        ctypes_of = {np.uint16: C.c_uint16, np.int64: C.c_int64, np.uint8: C.c_uint8, np.uint64: C.c_uint64, np.uint32: C.c_uint32}
        self.fin = abi.PloFinishOut()
        for name, dt in abi.FINISH_ITEM_FIELDS:
            setattr(self.fin, name, dl(getattr(fo, name), dt, n, ctypes_of[dt]))
        for name, dt in abi.FINISH_READ_FIELDS:
            setattr(self.fin, name, dl(getattr(fo, name), dt, n_reads, ctypes_of[dt]))
        self.fin.rev_seq_bytes, self.fin.rev_qual_bytes = int(fo.rev_seq_bytes), int(fo.rev_qual_bytes)
        self.fin.rev_seq = dl(fo.rev_seq, np.uint8, int(fo.rev_seq_bytes), C.c_uint8)
        self.fin.rev_qual = dl(fo.rev_qual, np.uint8, int(fo.rev_qual_bytes), C.c_uint8)
        self.fin.finish_ms, self.fin.revcomp_ms = fo.finish_ms, fo.revcomp_ms
        self.fin.n_items, self.fin.n_reads = n, n_reads
        self.sa = None
        if so is not None:
            self.sa = abi.PloSaOut()
            self.sa.n_items = int(so.n_items)
            self.sa.item_sa_off = dl(so.item_sa_off, np.uint32, int(so.n_items) + 1, C.c_uint32)
            self.sa.sa_text = dl(so.sa_text, np.uint8, int(so.sa_bytes), C.c_uint8)
            self.sa.sa_bytes = int(so.sa_bytes)
            self.sa.sa_ms = so.sa_ms
        if pending:
            torch.cuda.current_stream().synchronize()  # one wait for all the copies
