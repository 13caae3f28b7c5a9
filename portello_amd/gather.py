"""Final record gather across ranks (the path's only exchange step).

Reads shard across GPUs with no data-path collective; after a batch every rank holds compact result records
(per item: read segment, contig segment, status/flags, chromosome, position, CIGAR).  They are gathered to the writer
rank with direct peer -> root ``isend``/``irecv`` posted as one group (one xGMI link per peer, all links concurrently;
a ring all-gather would be per-link bound for no benefit) after a 16-byte-per-rank size exchange; every field is sent
straight from the engine's output array (no packing copy).  ``torch.distributed`` backend "nccl" is RCCL on ROCm; the
same code runs on CPU tensors over gloo (tests).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import abi

# (field, numpy dtype, torch dtype used as the raw carrier)
ITEM_FIELDS = [
    ("item_seg", np.uint32, torch.int32),
    ("item_cseg", np.uint32, torch.int32),
    ("item_status", np.uint8, torch.uint8),
    ("item_need_flipped", np.uint8, torch.uint8),
    ("item_mapq", np.uint8, torch.uint8),
    ("item_chrom_index", np.uint32, torch.int32),
    ("item_ref_pos", np.int64, torch.int64),
    ("item_cigar_off", np.uint64, torch.int64),
    ("item_cigar_len", np.uint32, torch.int32),
]
_TYPESTR = {torch.int32: "<i4", torch.uint8: "|u1", torch.int64: "<i8"}


class _DevArray:
    """zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)"""

    def __init__(self, ptr: int, n: int, tdtype):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": _TYPESTR[tdtype], "data": (ptr, False), "version": 2}


def device_view(ptr, n: int, tdtype, dev) -> torch.Tensor:
    addr = C.cast(ptr, C.c_void_p).value
    if n == 0 or not addr:
        return torch.zeros(0, dtype=tdtype, device=dev)
    return torch.as_tensor(_DevArray(addr, n, tdtype), device=dev)


def tensors_from_out(out: abi.PloBatchOut, dev) -> Dict[str, torch.Tensor]:
    n, nc = int(out.n_items), int(out.n_cigar)
    d = {name: device_view(getattr(out, name), n, tdt, dev) for name, _, tdt in ITEM_FIELDS}
    d["cigar"] = device_view(out.cigar, nc, torch.int32, dev)
    return d


def tensors_from_result(res: abi.BatchResult) -> Dict[str, torch.Tensor]:
    """CPU tensors from a host BatchResult (gloo tests)"""
    d = {}
    for name, npdt, tdt in ITEM_FIELDS:
        a = np.ascontiguousarray(getattr(res, name), dtype=npdt)
        d[name] = torch.from_numpy(a.view(np.dtype(_TYPESTR[tdt]))).clone()
    d["cigar"] = torch.from_numpy(np.ascontiguousarray(res.cigar, dtype=np.uint32).view(np.int32)).clone()
    return d


def _pad8(t: torch.Tensor) -> torch.Tensor:
    b = t.contiguous().view(torch.uint8)
    r = (-b.numel()) % 8
    if r:
        b = torch.cat([b, torch.zeros(r, dtype=torch.uint8, device=b.device)])
    return b


def pack(t: Dict[str, torch.Tensor]) -> torch.Tensor:
    """one contiguous byte payload: every array padded to 8 bytes, fixed field order"""
    return torch.cat([_pad8(t[name]) for name, _, _ in ITEM_FIELDS] + [_pad8(t["cigar"])])


def unpack(payload: torch.Tensor, n_items: int, n_cigar: int) -> Dict[str, torch.Tensor]:
    out = {}
    off = 0
    for name, _, tdt in ITEM_FIELDS + [("cigar", np.uint32, torch.int32)]:
        n = n_cigar if name == "cigar" else n_items
        nbytes = n * torch.tensor([], dtype=tdt).element_size()
        out[name] = payload[off: off + nbytes].view(tdt)
        off += nbytes + ((-nbytes) % 8)
    return out


def gather_payloads(t: Dict[str, torch.Tensor], dist, rank: int, world: int, root: int = 0) -> Optional[List[Dict[str, torch.Tensor]]]:
    """Gather every rank's records on `root`.  Returns the per-rank dicts on root, None elsewhere.

    Every field travels as its own message straight from the engine's output array (no packing copy); all messages of the
    step are posted as ONE group so that the peers' links run concurrently into the root, and messages between one pair
    of ranks match in posting order."""
    dev = t["cigar"].device
    fields = ITEM_FIELDS + [("cigar", np.uint32, torch.int32)]
    sizes = torch.tensor([t["item_seg"].numel(), t["cigar"].numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    if rank != root:
        ops = [dist.P2POp(dist.isend, t[name].contiguous(), root) for name, _, _ in fields if t[name].numel()]
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        return None
    res: List[Dict[str, torch.Tensor]] = []
    ops = []
    for r in range(world):
        if r == root:
            res.append({name: t[name] for name, _, _ in fields})
            continue
        ni, nc = (int(x) for x in all_sizes[r].tolist())
        d = {}
        for name, _, tdt in fields:
            n = nc if name == "cigar" else ni
            d[name] = torch.empty(n, dtype=tdt, device=dev)
            if n:
                ops.append(dist.P2POp(dist.irecv, d[name], r))
        res.append(d)
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return res


class PendingGather:
    """An exchange posted by gather_payloads_async: the requests, the tensors that must stay alive until they complete, and
    (on the root) the per-rank results."""

    def __init__(self, works, keep, result):
        self.works, self.keep, self.result = works, keep, result

    def wait(self):
        for q in self.works:
            q.wait()
        self.works = []
        return self.result


def gather_payloads_async(t: Dict[str, torch.Tensor], dist, rank: int, world: int, root: int = 0) -> PendingGather:
    """gather_payloads without the final wait: the size exchange is synchronous (16 bytes per rank), the payload messages are
    posted and returned as a PendingGather.  The caller keeps computing the next batch (into OTHER output buffers) and calls
    wait() before it reuses the buffers `t` views -- the exchange of batch i then overlaps the compute of batch i+1.  Every
    rank must post its exchanges in the same order (single thread per rank)."""
    dev = t["cigar"].device
    fields = ITEM_FIELDS + [("cigar", np.uint32, torch.int32)]
    sizes = torch.tensor([t["item_seg"].numel(), t["cigar"].numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    if rank != root:
        keep = [t[name].contiguous() for name, _, _ in fields if t[name].numel()]
        ops = [dist.P2POp(dist.isend, x, root) for x in keep]
        return PendingGather(dist.batch_isend_irecv(ops) if ops else [], keep, None)
    res: List[Dict[str, torch.Tensor]] = []
    ops = []
    for r in range(world):
        if r == root:
            res.append({name: t[name] for name, _, _ in fields})
            continue
        ni, nc = (int(x) for x in all_sizes[r].tolist())
        d = {}
        for name, _, tdt in fields:
            n = nc if name == "cigar" else ni
            d[name] = torch.empty(n, dtype=tdt, device=dev)
            if n:
                ops.append(dist.P2POp(dist.irecv, d[name], r))
        res.append(d)
    return PendingGather(dist.batch_isend_irecv(ops) if ops else [], res, res)


def gather_results(out: abi.PloBatchOut, dev, dist, rank: int, world: int, root: int = 0):
    """GPU form: zero-copy views of the engine's device outputs -> gather on `root`."""
    return gather_payloads(tensors_from_out(out, dev), dist, rank, world, root)


def gather_results_async(out: abi.PloBatchOut, dev, dist, rank: int, world: int, root: int = 0) -> PendingGather:
    """GPU form of gather_payloads_async (zero-copy views of the engine's device outputs)."""
    return gather_payloads_async(tensors_from_out(out, dev), dist, rank, world, root)


def to_result(t: Dict[str, torch.Tensor]) -> abi.BatchResult:
    def np_(name, npdt):
        return t[name].cpu().numpy().view(npdt).copy()

    return abi.BatchResult(**{name: np_(name, npdt) for name, npdt, _ in ITEM_FIELDS}, cigar=np_("cigar", np.uint32))


# ---- order-independent comparison of gathered records with a single-rank result (bench.py --scaling strong, tests) --------

def combine(parts: List[Dict[str, torch.Tensor]], seg_maps: Optional[List[Optional[torch.Tensor]]] = None) -> Dict[str, torch.Tensor]:
    """One record set from the per-rank dicts of gather_payloads: item arrays concatenated, CIGAR offsets re-based onto the
    concatenated CIGAR buffer, and -- with seg_maps[r] = global read-segment index of rank r's local segment -- item_seg mapped
    back to the unsharded numbering."""
    out: Dict[str, List[torch.Tensor]] = {name: [] for name, _, _ in ITEM_FIELDS}
    cig, base = [], 0
    for r, t in enumerate(parts):
        for name, _, _ in ITEM_FIELDS:
            v = t[name]
            if name == "item_cigar_off":
                v = v + base
            elif name == "item_seg" and seg_maps is not None and seg_maps[r] is not None:
                v = seg_maps[r][v.long()].to(torch.int32)
            out[name].append(v)
        cig.append(t["cigar"])
        base += int(t["cigar"].numel())
    res = {name: torch.cat(v) for name, v in out.items()}
    res["cigar"] = torch.cat(cig)
    return res


def canonical_tensors(t: Dict[str, torch.Tensor]):
    """records sorted by (read segment, contig segment): (keys, item fields in key order, CIGAR ops in key order)"""
    key = (t["item_seg"].long() << 20) | t["item_cseg"].long()
    order = torch.argsort(key)
    lens = t["item_cigar_len"].long()[order]
    offs = t["item_cigar_off"].long()[order]
    emit = (t["item_status"][order] == abi.ITEM_LIFTED) | (t["item_status"][order] == abi.ITEM_LEN_MISMATCH)
    lens = torch.where(emit, lens, torch.zeros_like(lens))
    total = int(lens.sum().item())
    starts = torch.cumsum(lens, 0) - lens
    flat = torch.repeat_interleave(offs - starts, lens) + torch.arange(total, device=key.device)
    fields = {name: t[name][order] for name, _, _ in ITEM_FIELDS if name not in ("item_cigar_off",)}
    return key[order], fields, t["cigar"][flat] if total else t["cigar"][:0]


def same_records(a: Dict[str, torch.Tensor], b: Dict[str, torch.Tensor]) -> bool:
    """bit-exact, order-independent equality of two record sets (status, flags, chromosome, position, CIGAR ops of every item)"""
    ka, fa, ca = canonical_tensors(a)
    kb, fb, cb = canonical_tensors(b)
    if ka.numel() != kb.numel() or not torch.equal(ka, kb):
        return False
    lifted = (fa["item_status"] == abi.ITEM_LIFTED) | (fa["item_status"] == abi.ITEM_LEN_MISMATCH)
    for name in fa:
        x, y = fa[name], fb[name]
        if name in ("item_ref_pos", "item_cigar_len"):  # defined for lifted records only
            x, y = x[lifted], y[lifted]
        if not torch.equal(x, y):
            return False
    return ca.numel() == cb.numel() and bool(torch.equal(ca, cb))


def local_to_global_segments(w, read_ranges) -> torch.Tensor:
    """global read-segment index of every local segment of DeviceBatch.from_read_ranges(w, read_ranges)"""
    dev = w.seg_read.device
    parts = []
    for lo, hi in read_ranges:
        if hi <= lo:
            continue
        s0 = int(torch.searchsorted(w.seg_read, torch.tensor(lo, device=dev)).item())
        s1 = int(torch.searchsorted(w.seg_read, torch.tensor(hi, device=dev)).item())
        parts.append(torch.arange(s0, s1, device=dev))
    return torch.cat(parts) if parts else torch.zeros(0, dtype=torch.long, device=dev)


# ---- the same exchange through the library's C ABI (plo_gather_*: what a host that is not Python binds) ----

class AbiGather:
    """plo_gather_create / plo_gather_records / plo_gather_wait.  The communicator is the library's own (ncclCommInitRank on an id that
    rank 0 makes and `dist` -- any initialised torch.distributed group, gloo will do -- carries to the others)."""

    def __init__(self, lib, dist, rank: int, world: int, device: int, root: int = 0):
        import ctypes as C

        self.lib, self.rank, self.world, self.root = lib, rank, world, root
        ident = (C.c_uint8 * 128)()
        if rank == root:
            st = lib.plo_gather_unique_id(ident)
            if st != abi.PLO_OK:
                raise RuntimeError(f"plo_gather_unique_id: {(lib.plo_gather_last_error(None) or b'').decode()}")
        box = [bytes(ident)]
        if world > 1:
            dist.broadcast_object_list(box, src=root)
        ident = (C.c_uint8 * 128)(*box[0])
        h = C.c_void_p()
        st = lib.plo_gather_create(ident, rank, world, device, C.byref(h))
        if st != abi.PLO_OK:
            raise RuntimeError(f"plo_gather_create: {(lib.plo_gather_last_error(None) or b'').decode()}")
        self.handle = h
        self._outs = None

    def gather(self, eng, out: abi.PloBatchOut, dev) -> "AbiPending":
        """posts the exchange of `out` (the engine's last, compacted result) behind the engine's kernels; wait() gives the per-rank tensor
        dicts on the root (zero-copy views of the library's receive buffers, valid until the next gather), None elsewhere"""
        import ctypes as C

        outs = (abi.PloBatchOut * self.world)() if self.rank == self.root else None
        st = self.lib.plo_gather_records(self.handle, eng.handle, C.byref(out), self.root, outs)
        if st != abi.PLO_OK:
            raise RuntimeError(f"plo_gather_records: {(self.lib.plo_gather_last_error(self.handle) or b'').decode()}")
        self._outs = outs
        return AbiPending(self, outs, dev)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.plo_gather_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class AbiPending:
    def __init__(self, g: AbiGather, outs, dev):
        self.g, self.outs, self.dev = g, outs, dev

    def wait(self):
        st = self.g.lib.plo_gather_wait(self.g.handle)
        if st != abi.PLO_OK:
            raise RuntimeError(f"plo_gather_wait: {(self.g.lib.plo_gather_last_error(self.g.handle) or b'').decode()}")
        if self.outs is None:
            return None
        return [tensors_from_out(self.outs[r], self.dev) for r in range(self.g.world)]
