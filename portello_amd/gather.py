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
