"""Sharding of one read set across ranks (BASELINE.json configs[3]).

The reference's unit of parallel work is the *window*: every contig is cut into segments of at most 20 Mb
(``get_region_segments``, lib/rust-vc-utils/src/util.rs:50-67, called at src/read_alignment_scanner.rs:508 with
``segment_size = 20_000_000``, :575) and a primary read belongs to the window its alignment starts in
(:403-406).  Windows share nothing but read-only inputs (:510-534), so they are dealt to the ranks whole, balanced by
the number of input CIGAR ops they carry (sum of n_in): heaviest window first, each to the rank with the least load so
far.  Every rank computes the same deal from the same inputs -- no communication.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

SEGMENT_SIZE = 20_000_000  # src/read_alignment_scanner.rs:575


def region_segments(size: int, segment_size: int = SEGMENT_SIZE) -> List[Tuple[int, int]]:
    """get_region_segments (lib/rust-vc-utils/src/util.rs:50-67): zero-indexed half-open (begin, end) intervals, none larger
    than segment_size, sizes differing by at most one"""
    if size <= 0:
        return []
    count = 1 + (size - 1) // segment_size
    base = size // count
    n_plus_one = size % count
    out = []
    start = 0
    for i in range(count):
        end = min(start + base + (1 if i < n_plus_one else 0), size)
        out.append((start, end))
        start = end
    return out


@dataclass
class Window:
    contig: int
    begin: int
    end: int
    read_lo: int  # reads [read_lo, read_hi) of the (contig, position)-sorted read set start in this window
    read_hi: int
    weight: int   # input CIGAR ops of those reads (all their split segments)


def read_windows(contig_len: Sequence[int], read_contig: np.ndarray, read_start: np.ndarray, read_ops: np.ndarray,
                 segment_size: int = SEGMENT_SIZE) -> List[Window]:
    """Windows of a read set sorted by (contig of the primary alignment, its start).  read_ops[r] = input ops of read r."""
    read_contig = np.asarray(read_contig, dtype=np.int64)
    read_start = np.asarray(read_start, dtype=np.int64)
    n = len(read_contig)
    if n > 1:
        key_ok = (np.diff(read_contig) > 0) | ((np.diff(read_contig) == 0) & (np.diff(read_start) >= 0))
        if not key_ok.all():
            raise ValueError("reads must be sorted by (contig, start) -- a coordinate-sorted read->contig BAM")
    ops_prefix = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.asarray(read_ops, dtype=np.int64), out=ops_prefix[1:])
    c_lo = np.searchsorted(read_contig, np.arange(len(contig_len)), side="left")
    c_hi = np.searchsorted(read_contig, np.arange(len(contig_len)), side="right")
    wins: List[Window] = []
    for c, clen in enumerate(contig_len):
        lo, hi = int(c_lo[c]), int(c_hi[c])
        if hi == lo:
            continue
        starts = read_start[lo:hi]
        for b, e in region_segments(int(clen), segment_size):
            a0 = lo + int(np.searchsorted(starts, b, side="left"))
            a1 = lo + int(np.searchsorted(starts, e, side="left"))
            if a1 > a0:
                wins.append(Window(c, b, e, a0, a1, int(ops_prefix[a1] - ops_prefix[a0])))
        # reads that start at or beyond the contig's end cannot exist in a valid BAM; keep them with the last window
        tail = lo + int(np.searchsorted(starts, int(clen), side="left"))
        if tail < hi:
            wins.append(Window(c, int(clen), int(clen), tail, hi, int(ops_prefix[hi] - ops_prefix[tail])))
    return wins


def deal_windows(windows: Sequence[Window], world: int) -> List[List[int]]:
    """indices of the windows of every rank: heaviest first, each to the rank with the least weight so far (ties: lowest
    rank); the windows of a rank are then put back in input order"""
    order = sorted(range(len(windows)), key=lambda i: (-windows[i].weight, i))
    load = [0] * world
    mine: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        mine[r].append(i)
        load[r] += windows[i].weight
    return [sorted(m) for m in mine]


def rank_read_ranges(windows: Sequence[Window], deal: List[List[int]], rank: int) -> List[Tuple[int, int]]:
    """the rank's read ranges, neighbouring windows merged"""
    out: List[Tuple[int, int]] = []
    for i in deal[rank]:
        w = windows[i]
        if out and out[-1][1] == w.read_lo:
            out[-1] = (out[-1][0], w.read_hi)
        else:
            out.append((w.read_lo, w.read_hi))
    return out


def pipeline_batches(windows: Sequence[Window], deal: List[List[int]], rank: int, k: int) -> List[List[Tuple[int, int]]]:
    """The rank's windows, in input order, cut into `k` consecutive runs of about equal weight -- the batches a rank lifts one after
    the other the way the reference walks its window tasks (src/read_alignment_scanner.rs:508-534), so that the exchange of one batch
    runs under the compute of the next.  Every run as read ranges (neighbouring windows merged); runs may be empty."""
    mine = list(deal[rank])
    total = sum(windows[i].weight for i in mine)
    out: List[List[Tuple[int, int]]] = [[] for _ in range(max(1, k))]
    acc = 0
    for i in mine:
        w = windows[i]
        j = min(len(out) - 1, (acc * len(out)) // total) if total else 0  # the run the window's first op falls into
        if out[j] and out[j][-1][1] == w.read_lo:
            out[j][-1] = (out[j][-1][0], w.read_hi)
        else:
            out[j].append((w.read_lo, w.read_hi))
        acc += w.weight
    return out


def workload_windows(w, segment_size: int = SEGMENT_SIZE) -> List[Window]:
    """windows of a synth.Workload (its first split segment is the primary alignment of a read)"""
    import torch

    seg_read = w.seg_read
    first = torch.ones_like(seg_read, dtype=torch.bool)
    first[1:] = seg_read[1:] != seg_read[:-1]
    idx = torch.nonzero(first).squeeze(1)
    read_contig = w.seg_contig[idx].cpu().numpy()
    read_start = w.seg_pos_r[idx].cpu().numpy()
    nops = (w.seg_cigar_off_r[1:] - w.seg_cigar_off_r[:-1])
    read_ops = torch.zeros(w.n_reads, dtype=torch.long, device=seg_read.device).index_add_(0, seg_read, nops).cpu().numpy()
    return read_windows([int(x) for x in w.contig_len], read_contig, read_start, read_ops, segment_size)
