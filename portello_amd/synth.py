"""Seeded synthetic workloads for the liftover path.

Everything is built from *real edits* so that CIGARs and bases agree: contigs are edited copies of reference
intervals (optionally reverse-complemented and split into several segments), reads are edited copies of contig
intervals.  The code is written with torch tensor ops only, so the same generator runs on the CPU for the parity
tests and on the GPU for the full-size bench workloads (30x-like read sets do not fit a host-side generator).

The generator is bench/test support: it emits exactly the arrays of ``include/portello_liftover.h`` (what the
reference's phase 1 / BAM decode would hand to the hot path) and takes no part in the computation.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import abi

SEED_BASE = 0x504F5254  # "PORT"

_ACGT = torch.tensor([65, 67, 71, 84], dtype=torch.uint8)  # A C G T
# BAM 4-bit codes of "=ACMGRSVTWYHKDBN"
_ASCII_TO_BAM4 = torch.full((256,), 15, dtype=torch.uint8)
for _i, _c in enumerate("=ACMGRSVTWYHKDBN"):
    _ASCII_TO_BAM4[ord(_c)] = _i
_COMP = torch.full((256,), ord("N"), dtype=torch.uint8)
for _a, _b in zip("ATCGNatcgn", "TAGCNtagcn"):
    _COMP[ord(_a)] = ord(_b)

OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)


@dataclass
class EditRates:
    mismatch: float = 3e-4
    ins: float = 4e-4
    dele: float = 3e-4
    hpol_frac: float = 0.7     # fraction of indels snapped to the start of a homopolymer run
    geo_p: float = 0.5         # indel length ~ Geometric(p)
    max_indel: int = 12
    big_indel_prob: float = 0.0  # probability that an indel is a large (50..500 bp) event
    min_gap: int = 2           # minimum number of '=' bases between two edits


@dataclass
class WorkloadConfig:
    name: str = "tiny"
    seed: int = SEED_BASE
    chrom_lens: Tuple[int, ...] = (200_000,)
    n_contigs_per_hap: int = 2
    n_haps: int = 2
    rev_contig_frac: float = 0.5
    max_segments: int = 3
    n_reads: int = 200
    read_len_mean: int = 15_000
    read_len_sd: int = 3_000
    read_len_min: int = 500
    split_read_frac: float = 0.03
    clip_read_frac: float = 0.1
    tract_frac: float = 0.03
    contig_rates: EditRates = field(default_factory=lambda: EditRates(mismatch=1e-3, ins=1e-4, dele=1e-4, hpol_frac=0.3,
                                                                       big_indel_prob=0.02))
    read_rates: EditRates = field(default_factory=EditRates)
    seq_fmt: int = abi.SEQ_BAM4
    sorted_reads: bool = False  # reads of a contig in coordinate order, as a coordinate-sorted read->contig BAM delivers them


def config(name: str, **over) -> WorkloadConfig:
    """Named workloads of BASELINE.json (sizes per SURVEY.md 8(d)); `over` overrides fields."""
    if name == "tiny":
        c = WorkloadConfig(name="tiny")
    elif name == "plumbing":  # configs[0]: 1 k reads x 10 kb, one contig -> chr20 slice
        c = WorkloadConfig(name="plumbing", seed=SEED_BASE + 0, chrom_lens=(2_000_000,), n_contigs_per_hap=1, n_haps=1,
                           rev_contig_frac=0.0, max_segments=1, n_reads=1000, read_len_mean=10_000, read_len_sd=0,
                           split_read_frac=0.0,
                           contig_rates=EditRates(mismatch=1e-3, ins=1e-4, dele=1e-4, hpol_frac=0.3, big_indel_prob=0.01))
    elif name == "chr20":  # configs[1]: 64 Mb slice, 2 haplotypes x ~10 contigs, ~50 k reads x 15 kb
        c = WorkloadConfig(name="chr20", sorted_reads=True, seed=SEED_BASE + 1, chrom_lens=(64_000_000,), n_contigs_per_hap=10, n_haps=2,
                           max_segments=5, n_reads=50_000)
    elif name == "wgs30x":  # configs[2]: 3.1 Gb, 2 x 300 contigs, 2 M reads x 15 kb
        lens = tuple(int(x) for x in np.linspace(248e6, 46e6, 24))
        scale = 3.1e9 / sum(lens)
        lens = tuple(int(x * scale) for x in lens)
        c = WorkloadConfig(name="wgs30x", sorted_reads=True, seed=SEED_BASE + 2, chrom_lens=lens, n_contigs_per_hap=300, n_haps=2,
                           max_segments=5, n_reads=2_000_000)
    elif name == "stress":  # configs[4]: 20 kb reads, 5 % indel-dense CIGARs, same contigs as wgs30x
        # the read count is a parameter (8(d): default 2 M; 200 k keeps the generator's footprint and the test time bounded --
        # a read carries ~2 000 ops, so 200 k reads are 4e8 input ops, six times wgs30x's whole batch)
        base = config("wgs30x")
        c = WorkloadConfig(name="stress", sorted_reads=True, seed=SEED_BASE + 4, chrom_lens=base.chrom_lens, n_contigs_per_hap=300, n_haps=2,
                           max_segments=5, n_reads=200_000, read_len_mean=20_000, read_len_sd=2_000,
                           read_rates=EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1))
    elif name == "stress_small":  # the same read profile on a 64 Mb reference (quick tuning runs)
        c = WorkloadConfig(name="stress_small", sorted_reads=True, seed=SEED_BASE + 4, chrom_lens=(64_000_000,), n_contigs_per_hap=10, n_haps=2,
                           max_segments=5, n_reads=100_000, read_len_mean=20_000, read_len_sd=2_000,
                           read_rates=EditRates(mismatch=5e-3, ins=2.5e-2, dele=2.5e-2, hpol_frac=0.5, min_gap=1))
    else:
        raise ValueError(name)
    for k, v in over.items():
        setattr(c, k, v)
    return c


# ----------------------------------------------------------------------------------------------------------------
# small tensor helpers
# ----------------------------------------------------------------------------------------------------------------

def _lut(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device)


def revcomp(seq: torch.Tensor) -> torch.Tensor:
    return _lut(_COMP, seq.device)[seq.flip(0).long()]


def _rand_bases(n: int, gen: torch.Generator, device) -> torch.Tensor:
    return _lut(_ACGT, device)[torch.randint(0, 4, (n,), generator=gen, device=device)]


def _excl_cumsum(x: torch.Tensor) -> torch.Tensor:
    c = torch.cumsum(x, 0)
    return c - x


def _seg_rank(seg_id: torch.Tensor) -> torch.Tensor:
    """rank of each element inside its (sorted, contiguous) segment"""
    n = seg_id.numel()
    if n == 0:
        return seg_id.clone()
    idx = torch.arange(n, device=seg_id.device)
    is_head = torch.ones(n, dtype=torch.bool, device=seg_id.device)
    is_head[1:] = seg_id[1:] != seg_id[:-1]
    head_idx = torch.where(is_head, idx, torch.zeros_like(idx))
    head_idx = torch.cummax(head_idx, 0).values
    return idx - head_idx


def make_reference(length: int, tract_frac: float, gen: torch.Generator, device) -> torch.Tensor:
    """uniform ACGT with homopolymer (3..20) and 2-6-mer STR tracts covering ~tract_frac of the sequence"""
    seq = _rand_bases(length, gen, device)
    if tract_frac <= 0 or length < 1000:
        return seq
    mean_tract = 14
    n_tr = max(1, int(length * tract_frac / mean_tract))
    # one tract per stride-sized cell (never overlapping: duplicate scatter targets would make the result depend on
    # the thread schedule)
    stride = max(32, length // n_tr)
    n_tr = max(1, (length - 64) // stride)
    start = torch.arange(n_tr, device=device) * stride + torch.randint(0, max(1, stride - 30), (n_tr,), generator=gen, device=device)
    tlen = torch.randint(3, 25, (n_tr,), generator=gen, device=device)
    unit = torch.randint(1, 7, (n_tr,), generator=gen, device=device)
    unit = torch.where(torch.rand(n_tr, generator=gen, device=device) < 0.6, torch.ones_like(unit), unit)
    tr_id = torch.repeat_interleave(torch.arange(n_tr, device=device), tlen)
    off = torch.arange(tr_id.numel(), device=device) - _excl_cumsum(tlen)[tr_id]
    motif = torch.randint(0, 4, (n_tr, 6), generator=gen, device=device)
    b = motif[tr_id, off % unit[tr_id]]
    pos = (start[tr_id] + off).clamp_(max=length - 1)
    seq[pos] = _lut(_ACGT, device)[b]
    return seq


# ----------------------------------------------------------------------------------------------------------------
# the edit engine: derive R sequences from intervals of one source sequence
# ----------------------------------------------------------------------------------------------------------------

@dataclass
class Mutated:
    ops: torch.Tensor        # uint32 BAM-encoded ops (=, X, I, D), flattened over the R derived sequences
    op_off: torch.Tensor     # int64 [R+1]
    seq: torch.Tensor        # uint8 ASCII bases, flattened
    seq_off: torch.Tensor    # int64 [R+1]


def mutate(source: torch.Tensor, starts: torch.Tensor, spans: torch.Tensor, rates: EditRates, gen: torch.Generator) -> Mutated:
    """Derive sequence r from source[starts[r] : starts[r]+spans[r]] by random edits; returns the exact alignment
    (=/X/I/D ops, source as the reference) and the derived bases.  Fully vectorised over r."""
    device = source.device
    R = starts.numel()
    spans = spans.long()
    starts = starts.long()
    span_off = _excl_cumsum(spans)
    T = int(spans.sum().item()) if R else 0
    rate = rates.mismatch + rates.ins + rates.dele
    K = int(torch.poisson(torch.tensor(float(T) * rate), generator=None).item()) if T and rate > 0 else 0
    i64 = dict(dtype=torch.long, device=device)

    if K > 0:
        v = torch.sort(torch.randint(0, T, (K,), generator=gen, device=device)).values
        rid = torch.searchsorted(torch.cumsum(spans, 0), v, right=True)
        off = v - span_off[rid]
        u = torch.rand(K, generator=gen, device=device)
        kind = torch.where(u < rates.mismatch / rate, torch.full((K,), OP_X, **i64),
                           torch.where(u < (rates.mismatch + rates.ins) / rate, torch.full((K,), OP_I, **i64),
                                       torch.full((K,), OP_D, **i64)))
        # geometric length, capped; occasional big events
        g = torch.rand(K, generator=gen, device=device).clamp_(min=1e-12)
        ln = (torch.log(g) / np.log(1.0 - rates.geo_p)).floor().long() + 1
        ln = ln.clamp_(max=rates.max_indel)
        if rates.big_indel_prob > 0:
            big = torch.rand(K, generator=gen, device=device) < rates.big_indel_prob
            ln = torch.where(big, torch.randint(50, 501, (K,), generator=gen, device=device), ln)
        ln = torch.where(kind == OP_X, torch.ones_like(ln), ln)
        # snap indels to the start of the homopolymer run they fall in (left-aligned on the source)
        hp = (torch.rand(K, generator=gen, device=device) < rates.hpol_frac) & (kind != OP_X)
        W = 24
        gpos = starts[rid] + off
        win = gpos[:, None] - torch.arange(0, W, device=device)[None, :]
        ok = win >= starts[rid][:, None]
        wb = source[win.clamp(min=0)]
        same = (wb == wb[:, :1]) & ok
        run_back = torch.cumprod(same.long(), 1).sum(1) - 1  # bases to the left that equal source[gpos]
        off = torch.where(hp, off - run_back, off)
        hp_base = source[gpos]
        # order + spacing filter (vectorised: compare with the previous *candidate*)
        key = rid * (int(spans.max().item()) + 1) + off
        order = torch.argsort(key, stable=True)
        rid, off, kind, ln, hp, hp_base = rid[order], off[order], kind[order], ln[order], hp[order], hp_base[order]
        consumed = torch.where(kind == OP_I, torch.zeros_like(ln), ln)
        prev_end = torch.full((K,), -10**9, **i64)
        same_r = torch.zeros(K, dtype=torch.bool, device=device)
        if K > 1:
            same_r[1:] = rid[1:] == rid[:-1]
            prev_end[1:] = torch.where(same_r[1:], off[:-1] + consumed[:-1], prev_end[1:])
        keep = (off >= rates.min_gap) & (off - prev_end >= rates.min_gap) & (off + consumed + rates.min_gap <= spans[rid])
        # a dropped predecessor could have shielded an overlap: re-check against the previous kept edit
        for _ in range(64):
            kidx = torch.nonzero(keep).squeeze(1)
            if kidx.numel() < 2:
                break
            r2, o2, c2 = rid[kidx], off[kidx], consumed[kidx]
            bad = torch.zeros(kidx.numel(), dtype=torch.bool, device=device)
            bad[1:] = (r2[1:] == r2[:-1]) & (o2[1:] - (o2[:-1] + c2[:-1]) < rates.min_gap)
            if not bool(bad.any()):
                break
            keep[kidx[bad]] = False
        rid, off, kind, ln, hp, hp_base = rid[keep], off[keep], kind[keep], ln[keep], hp[keep], hp_base[keep]
        consumed = torch.where(kind == OP_I, torch.zeros_like(ln), ln)
        K = rid.numel()
    if K == 0:
        rid = torch.zeros(0, **i64)
        off = ln = kind = consumed = rid
        hp = torch.zeros(0, dtype=torch.bool, device=device)
        hp_base = torch.zeros(0, dtype=torch.uint8, device=device)

    n_ed = torch.bincount(rid, minlength=R) if K else torch.zeros(R, **i64)
    ed_off = _excl_cumsum(n_ed)
    j = _seg_rank(rid) if K else rid
    prev_end = torch.zeros(K, **i64)
    if K > 1:
        same_r = rid[1:] == rid[:-1]
        prev_end[1:] = torch.where(same_r, off[:-1] + consumed[:-1], torch.zeros_like(off[1:]))
    eq_len = off - prev_end  # '=' run before each edit (>= min_gap)
    last_end = torch.zeros(R, **i64)
    if K:
        last_idx = ed_off + n_ed - 1
        has = n_ed > 0
        last_end[has] = (off + consumed)[last_idx[has]]
    tail_eq = spans - last_end

    # ---- ops ----
    n_ops = 2 * n_ed + (tail_eq > 0).long()
    op_off = torch.zeros(R + 1, **i64)
    op_off[1:] = torch.cumsum(n_ops, 0)
    ops = torch.zeros(int(op_off[-1].item()), **i64)
    if K:
        slot = op_off[rid] + 2 * j
        ops[slot] = (eq_len << 4) | OP_EQ
        ops[slot + 1] = (ln << 4) | kind
    has_tail = tail_eq > 0
    ops[(op_off[:-1] + 2 * n_ed)[has_tail]] = (tail_eq[has_tail] << 4) | OP_EQ

    # ---- bases: run table (copy runs, edit runs, tail runs) ----
    out_eq = eq_len
    out_ed = torch.where(kind == OP_D, torch.zeros_like(ln), ln) if K else ln
    n_runs = 2 * K + R
    run_len = torch.zeros(n_runs, **i64)
    run_src = torch.zeros(n_runs, **i64)
    run_kind = torch.zeros(n_runs, **i64)  # 0 copy, 1 substitute, 2 fixed base, 3 random
    run_base = torch.zeros(n_runs, dtype=torch.uint8, device=device)
    rslot0 = 2 * ed_off + torch.arange(R, device=device)  # first run slot of each derived sequence
    if K:
        rs = rslot0[rid] + 2 * j
        run_len[rs] = out_eq
        run_src[rs] = starts[rid] + prev_end
        run_len[rs + 1] = out_ed
        run_src[rs + 1] = starts[rid] + off
        run_kind[rs + 1] = torch.where(kind == OP_X, torch.ones_like(kind),
                                       torch.where(hp, torch.full_like(kind, 2), torch.full_like(kind, 3)))
        run_base[rs + 1] = hp_base
    ts = rslot0 + 2 * n_ed
    run_len[ts] = tail_eq
    run_src[ts] = starts + last_end
    run_out = _excl_cumsum(run_len)
    total = int(run_len.sum().item())
    run_id = torch.repeat_interleave(torch.arange(n_runs, device=device), run_len)
    boff = torch.arange(total, device=device) - run_out[run_id]
    src_idx = (run_src[run_id] + boff).clamp_(max=source.numel() - 1)
    base = source[src_idx]
    k = run_kind[run_id]
    if K:
        acgt = _lut(_ACGT, device)
        code = ((base == 67).long() + 2 * (base == 71).long() + 3 * (base == 84).long())
        sub = acgt[(code + torch.randint(1, 4, (total,), generator=gen, device=device)) % 4]
        rnd = acgt[torch.randint(0, 4, (total,), generator=gen, device=device)]
        base = torch.where(k == 1, sub, base)
        base = torch.where(k == 2, run_base[run_id], base)
        base = torch.where(k == 3, rnd, base)
    seq_len = spans + torch.zeros(R, **i64)
    if K:
        delta = torch.where(kind == OP_I, ln, torch.where(kind == OP_D, -ln, torch.zeros_like(ln)))
        seq_len = seq_len + torch.zeros(R, **i64).index_add_(0, rid, delta)
    seq_off = torch.zeros(R + 1, **i64)
    seq_off[1:] = torch.cumsum(seq_len, 0)
    assert int(seq_off[-1].item()) == total
    return Mutated(ops=ops.to(torch.int64), op_off=op_off, seq=base, seq_off=seq_off)


# ----------------------------------------------------------------------------------------------------------------
# workload assembly
# ----------------------------------------------------------------------------------------------------------------

@dataclass
class Workload:
    cfg: WorkloadConfig
    device: torch.device
    # index side (tensors on `device`; small arrays as numpy)
    chrom_seq: List[torch.Tensor]
    contig_len: np.ndarray
    contig_seg_off: np.ndarray
    seg_chrom_index: np.ndarray
    seg_pos: np.ndarray
    seg_is_fwd: np.ndarray
    seg_mapq: np.ndarray
    seg_start: np.ndarray
    seg_end: np.ndarray
    seg_cigar_off: np.ndarray
    seg_cigar: np.ndarray
    rev_contig_seq: List[Optional[torch.Tensor]]
    # batch side (tensors on `device`)
    read_is_reverse: torch.Tensor
    read_seq_len: torch.Tensor
    read_seq_off: torch.Tensor
    seq: torch.Tensor
    seg_read: torch.Tensor
    seg_contig: torch.Tensor
    seg_pos_r: torch.Tensor
    seg_is_fwd_r: torch.Tensor
    seg_cigar_off_r: torch.Tensor
    cigar: torch.Tensor
    contig_fwd: Optional[List[torch.Tensor]] = None  # the contigs' bases (generate(..., keep_contigs=True)): more reads on the same contigs

    @property
    def n_reads(self) -> int:
        return int(self.read_seq_len.numel())

    def index_data(self) -> abi.IndexData:
        """host copy (numpy) of the index side"""
        return abi.IndexData(
            contig_len=self.contig_len, contig_seg_off=self.contig_seg_off, seg_chrom_index=self.seg_chrom_index,
            seg_pos=self.seg_pos, seg_is_fwd_strand=self.seg_is_fwd, seg_mapq=self.seg_mapq,
            seg_seq_order_start=self.seg_start, seg_seq_order_end=self.seg_end, seg_cigar_off=self.seg_cigar_off,
            seg_cigar=self.seg_cigar, chrom_seq=[s.cpu().numpy() for s in self.chrom_seq],
            rev_contig_seq=[None if s is None else s.cpu().numpy() for s in self.rev_contig_seq])

    def index_data_device(self) -> abi.IndexData:
        """index description whose sequences stay on the GPU (borrowed device pointers)"""
        assert self.device.type == "cuda"
        return abi.IndexData(
            contig_len=self.contig_len, contig_seg_off=self.contig_seg_off, seg_chrom_index=self.seg_chrom_index,
            seg_pos=self.seg_pos, seg_is_fwd_strand=self.seg_is_fwd, seg_mapq=self.seg_mapq,
            seg_seq_order_start=self.seg_start, seg_seq_order_end=self.seg_end, seg_cigar_off=self.seg_cigar_off,
            seg_cigar=self.seg_cigar, chrom_seq=[s.data_ptr() for s in self.chrom_seq],
            rev_contig_seq=[None if s is None else s.data_ptr() for s in self.rev_contig_seq],
            chrom_len=np.array([s.numel() for s in self.chrom_seq], dtype=np.int64), seq_mem=abi.MEM_DEVICE)

    def batch_data(self, lo: int = 0, hi: Optional[int] = None) -> abi.BatchData:
        """host copy (numpy) of reads [lo, hi) with re-based offsets"""
        hi = self.n_reads if hi is None else hi
        seg_lo = int(torch.searchsorted(self.seg_read, torch.tensor(lo, device=self.device)).item())
        seg_hi = int(torch.searchsorted(self.seg_read, torch.tensor(hi, device=self.device)).item())
        so = self.read_seq_off[lo:hi].cpu().numpy().astype(np.uint64)
        sl = self.read_seq_len[lo:hi].cpu().numpy()
        if hi > lo:
            b0 = int(so[0])
            nbytes = (int(sl[-1]) + 1) // 2 if self.cfg.seq_fmt == abi.SEQ_BAM4 else int(sl[-1])
            b1 = int(so[-1]) + nbytes
        else:
            b0 = b1 = 0
        co = self.seg_cigar_off_r[seg_lo : seg_hi + 1].cpu().numpy().astype(np.int64)
        c0 = int(co[0]) if len(co) else 0
        c1 = int(co[-1]) if len(co) else 0
        return abi.BatchData(
            read_is_reverse=self.read_is_reverse[lo:hi].cpu().numpy(), read_seq_len=sl, read_seq_off=so - np.uint64(b0),
            seq=self.seq[b0:b1].cpu().numpy(), seq_fmt=self.cfg.seq_fmt,
            seg_read=(self.seg_read[seg_lo:seg_hi] - lo).cpu().numpy(), seg_contig=self.seg_contig[seg_lo:seg_hi].cpu().numpy(),
            seg_pos=self.seg_pos_r[seg_lo:seg_hi].cpu().numpy(), seg_is_fwd_strand=self.seg_is_fwd_r[seg_lo:seg_hi].cpu().numpy(),
            seg_cigar_off=(co - c0).astype(np.uint32), cigar=self.cigar[c0:c1].cpu().numpy().astype(np.uint32))


def _pack_bam4(ascii_seq: torch.Tensor, seq_off: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """ASCII bases + per-read offsets -> BAM 4-bit packed bytes with byte-aligned reads; returns (packed, byte_off)"""
    device = ascii_seq.device
    lens = seq_off[1:] - seq_off[:-1]
    nbytes = (lens + 1) // 2
    byte_off = torch.zeros(lens.numel() + 1, dtype=torch.long, device=device)
    byte_off[1:] = torch.cumsum(nbytes, 0)
    total_bytes = int(byte_off[-1].item())
    codes = _lut(_ASCII_TO_BAM4, device)[ascii_seq.long()]
    n = ascii_seq.numel()
    rid = torch.repeat_interleave(torch.arange(lens.numel(), device=device), lens)
    i = torch.arange(n, device=device) - seq_off[:-1][rid]
    tgt = byte_off[:-1][rid] + (i >> 1)
    packed = torch.zeros(total_bytes, dtype=torch.uint8, device=device)
    hi = (i & 1) == 0
    packed[tgt[hi]] = codes[hi] << 4
    lo = ~hi
    packed[tgt[lo]] |= codes[lo]
    return packed, byte_off


def generate(cfg: WorkloadConfig, device: str | torch.device = "cpu", reuse: Optional["Workload"] = None, keep_contigs: bool = False) -> Workload:
    """`reuse`: a workload generated with keep_contigs=True -- its reference, contigs and contig segments are taken over and only
    the READS are generated (from cfg.seed, cfg.n_reads, the read profile of cfg): the chunks of a read set larger than one
    batch (bench.py --stream, tests) share one index that way."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(cfg.seed)
    rng = np.random.default_rng(cfg.seed)
    torch.manual_seed(cfg.seed)  # torch.poisson draws from the global generator
    if reuse is not None:
        assert reuse.contig_fwd is not None, "generate(..., reuse=w): w must come from generate(..., keep_contigs=True)"
        return _generate_reads(cfg, device, gen, rng, reuse.chrom_seq, reuse.contig_fwd, [int(x) for x in reuse.contig_len], list(reuse.contig_seg_off),
                               list(reuse.seg_chrom_index), list(reuse.seg_pos), list(reuse.seg_is_fwd), list(reuse.seg_mapq), list(reuse.seg_start),
                               list(reuse.seg_end), reuse.seg_cigar_off, reuse.seg_cigar, reuse.rev_contig_seq, keep_contigs)

    chrom_seq = [make_reference(L, cfg.tract_frac, gen, device) for L in cfg.chrom_lens]

    # ---- contigs ----
    contig_fwd: List[torch.Tensor] = []
    contig_len: List[int] = []
    contig_seg_off = [0]
    seg_chrom, seg_pos, seg_fwd, seg_mapq, seg_start, seg_end = [], [], [], [], [], []
    seg_cigs: List[np.ndarray] = []
    rev_contig_seq: List[Optional[torch.Tensor]] = []
    for hap in range(cfg.n_haps):
        # tile the genome with n_contigs_per_hap contigs, lengths lognormal-ish
        w = rng.lognormal(0.0, 0.6, cfg.n_contigs_per_hap)
        chrom_of = rng.choice(len(cfg.chrom_lens), cfg.n_contigs_per_hap, p=np.array(cfg.chrom_lens) / sum(cfg.chrom_lens))
        for ci in range(len(cfg.chrom_lens)):
            idx = np.nonzero(chrom_of == ci)[0]
            if len(idx) == 0:
                continue
            L = cfg.chrom_lens[ci]
            shares = w[idx] / w[idx].sum()
            bounds = np.concatenate([[0], np.cumsum(shares)]) * (L - 200) + 100
            for k, _ in enumerate(idx):
                a, b = int(bounds[k]), int(bounds[k + 1])
                if b - a < 2000:
                    continue
                contig_is_rev = rng.random() < cfg.rev_contig_frac
                n_seg = int(rng.integers(1, cfg.max_segments + 1))
                n_seg = max(1, min(n_seg, (b - a) // 5000))
                cuts = np.sort(rng.choice(np.arange(a + 1000, b - 1000), n_seg - 1, replace=False)) if n_seg > 1 else np.array([], dtype=np.int64)
                edges = np.concatenate([[a], cuts, [b]]).astype(np.int64)
                pieces: List[torch.Tensor] = []
                segs = []
                cpos = 0
                lead = int(rng.integers(0, 200))
                if lead:
                    pieces.append(_rand_bases(lead, gen, device))
                    cpos += lead
                # segment order along the contig follows the contig strand
                order = range(n_seg) if not contig_is_rev else range(n_seg - 1, -1, -1)
                for si in order:
                    ra, rb = int(edges[si]), int(edges[si + 1])
                    # leave a gap in the reference between segments (joiner would have merged <1 kb colinear gaps)
                    if n_seg > 1:
                        ra += int(rng.integers(0, 300))
                    fwd = not contig_is_rev
                    if n_seg > 1 and rng.random() < 0.2:
                        fwd = not fwd  # inversion
                    m = mutate(chrom_seq[ci], torch.tensor([ra], device=device), torch.tensor([rb - ra], device=device),
                               cfg.contig_rates, gen)
                    sseq = m.seq if fwd else revcomp(m.seq)
                    c0 = cpos
                    pieces.append(sseq)
                    cpos += sseq.numel()
                    segs.append(dict(chrom=ci, pos=ra, fwd=fwd, c0=c0, c1=cpos, ops=m.ops.cpu().numpy(),
                                     mapq=int(rng.choice([60, 60, 60, 30, 5]))))
                    gap = int(rng.integers(0, 300)) if n_seg > 1 else 0
                    if gap:
                        pieces.append(_rand_bases(gap, gen, device))
                        cpos += gap
                trail = int(rng.integers(0, 200))
                if trail:
                    pieces.append(_rand_bases(trail, gen, device))
                    cpos += trail
                cseq = torch.cat(pieces)
                Lc = cseq.numel()
                contig_fwd.append(cseq)
                contig_len.append(Lc)
                any_rev = False
                for k2, s in enumerate(segs):
                    clip_op = OP_S if k2 == 0 else OP_H
                    lead_clip, trail_clip = (s["c0"], Lc - s["c1"]) if s["fwd"] else (Lc - s["c1"], s["c0"])
                    ops = s["ops"].astype(np.uint32)
                    full = []
                    if lead_clip:
                        full.append(np.uint32((lead_clip << 4) | clip_op))
                    full.extend(ops.tolist())
                    if trail_clip:
                        full.append(np.uint32((trail_clip << 4) | clip_op))
                    seg_cigs.append(np.array(full, dtype=np.uint32))
                    seg_chrom.append(s["chrom"])
                    seg_pos.append(s["pos"])
                    seg_fwd.append(1 if s["fwd"] else 0)
                    seg_mapq.append(s["mapq"])
                    seg_start.append(s["c0"])
                    seg_end.append(s["c1"])
                    any_rev |= not s["fwd"]
                contig_seg_off.append(contig_seg_off[-1] + len(segs))
                rev_contig_seq.append(revcomp(cseq) if any_rev else None)
    # one contig that never appeared in the asm->ref BAM: empty segment list (contig_alignment_scanner/mod.rs:364-367)
    orphan = _rand_bases(30_000, gen, device)
    contig_fwd.append(orphan)
    contig_len.append(orphan.numel())
    contig_seg_off.append(contig_seg_off[-1])
    rev_contig_seq.append(None)

    seg_cigar_off = np.zeros(len(seg_cigs) + 1, dtype=np.uint32)
    seg_cigar_off[1:] = np.cumsum([len(c) for c in seg_cigs])
    seg_cigar = np.concatenate(seg_cigs) if seg_cigs else np.zeros(0, np.uint32)
    return _generate_reads(cfg, device, gen, rng, chrom_seq, contig_fwd, contig_len, contig_seg_off, seg_chrom, seg_pos, seg_fwd, seg_mapq, seg_start,
                           seg_end, seg_cigar_off, seg_cigar, rev_contig_seq, keep_contigs)


def _generate_reads(cfg, device, gen, rng, chrom_seq, contig_fwd, contig_len, contig_seg_off, seg_chrom, seg_pos, seg_fwd, seg_mapq, seg_start,
                    seg_end, seg_cigar_off, seg_cigar, rev_contig_seq, keep_contigs) -> "Workload":
    # ---- reads ----
    n_contigs = len(contig_len)
    clen = np.array(contig_len, dtype=np.int64)
    weights = clen / clen.sum()
    reads_per_contig = rng.multinomial(cfg.n_reads, weights)
    parts = dict(is_rev=[], seq=[], seq_off=[], seq_len=[], seg_read=[], seg_contig=[], seg_pos=[], seg_fwd=[], seg_nops=[], ops=[])
    read_base = 0
    seq_byte_base = 0
    i64 = dict(dtype=torch.long, device=device)
    for c in range(n_contigs):
        R = int(reads_per_contig[c])
        if R == 0:
            continue
        Lc = int(clen[c])
        rl = torch.randn(R, generator=gen, device=device) * cfg.read_len_sd + cfg.read_len_mean
        rl = rl.long().clamp_(min=cfg.read_len_min, max=max(cfg.read_len_min, Lc - 2))
        rl = rl.clamp_(max=Lc - 2)
        start = (torch.rand(R, generator=gen, device=device) * (Lc - rl).float()).long().clamp_(min=0)
        if cfg.sorted_reads:
            start, order_ = torch.sort(start)
            rl = rl[order_]
        m = mutate(contig_fwd[c], start, rl, cfg.read_rates, gen)
        is_rev = torch.rand(R, generator=gen, device=device) < 0.5
        # soft clips (random bases) on a fraction of reads, supplementary part on a fraction
        u = torch.rand(R, generator=gen, device=device)
        lead = torch.where(u < cfg.clip_read_frac, torch.randint(1, 200, (R,), generator=gen, device=device), torch.zeros(R, **i64))
        u2 = torch.rand(R, generator=gen, device=device)
        trail = torch.where(u2 < cfg.clip_read_frac, torch.randint(1, 200, (R,), generator=gen, device=device), torch.zeros(R, **i64))
        is_split = torch.rand(R, generator=gen, device=device) < cfg.split_read_frac
        if Lc < 8000:
            is_split = torch.zeros_like(is_split)
        n_split = int(is_split.sum().item())
        a_len = m.seq_off[1:] - m.seq_off[:-1]
        # supplementary parts: 1..3 kb from elsewhere on the contig, either strand
        sidx = torch.nonzero(is_split).squeeze(1)
        if n_split:
            bl = torch.randint(1000, 3000, (n_split,), generator=gen, device=device).clamp_(max=Lc - 2)
            bstart = (torch.rand(n_split, generator=gen, device=device) * (Lc - bl).float()).long()
            mb = mutate(contig_fwd[c], bstart, bl, cfg.read_rates, gen)
            b_opp = torch.rand(n_split, generator=gen, device=device) < 0.5
            b_len_s = mb.seq_off[1:] - mb.seq_off[:-1]
            trail[sidx] = 0  # the supplementary part takes the place of the trailing clip
        b_len = torch.zeros(R, **i64)
        if n_split:
            b_len[sidx] = b_len_s
        total_len = lead + a_len + trail + b_len
        # ---- bases: concatenate [lead clip][A][trail clip | B] per read via a run table ----
        nrun = 4
        run_len = torch.stack([lead, a_len, trail, b_len], 1).reshape(-1)
        run_out = _excl_cumsum(run_len)
        tot = int(run_len.sum().item())
        run_id = torch.repeat_interleave(torch.arange(R * nrun, device=device), run_len)
        boff = torch.arange(tot, device=device) - run_out[run_id]
        rr = run_id // nrun
        kind = run_id % nrun
        bases = _rand_bases(tot, gen, device)
        isA = kind == 1
        bases[isA] = m.seq[(m.seq_off[:-1][rr[isA]] + boff[isA])]
        if n_split:
            isB = kind == 3
            sp_rank = torch.zeros(R, **i64)
            sp_rank[sidx] = torch.arange(n_split, device=device)
            q = sp_rank[rr[isB]]
            o = boff[isB]
            opp = b_opp[q]
            src_i = torch.where(opp, mb.seq_off[:-1][q] + (b_len_s[q] - 1 - o), mb.seq_off[:-1][q] + o)
            bb = mb.seq[src_i]
            bb = torch.where(opp, _lut(_COMP, device)[bb.long()], bb)
            bases[isB] = bb
        # ---- segments + ops ----
        # primary: [lead S] opsA [trail+b_len S]
        a_nops = m.op_off[1:] - m.op_off[:-1]
        tclip = trail + b_len
        p_nops = a_nops + (lead > 0).long() + (tclip > 0).long()
        if n_split:
            b_nops = mb.op_off[1:] - mb.op_off[:-1]
            s_nops = b_nops + 1
        seg_per_read = 1 + is_split.long()
        seg_off = _excl_cumsum(seg_per_read)
        n_seg_c = int(seg_per_read.sum().item())
        seg_nops = torch.zeros(n_seg_c, **i64)
        seg_nops[seg_off] = p_nops
        if n_split:
            seg_nops[seg_off[sidx] + 1] = s_nops
        seg_op_off = _excl_cumsum(seg_nops)
        ops = torch.zeros(int(seg_nops.sum().item()), **i64)
        po = seg_op_off[seg_off]
        hasl = lead > 0
        ops[po[hasl]] = (lead[hasl] << 4) | OP_S
        # copy A ops
        aid = torch.repeat_interleave(torch.arange(R, device=device), a_nops)
        ao = torch.arange(int(a_nops.sum().item()), device=device) - m.op_off[:-1][aid]
        ops[po[aid] + hasl.long()[aid] + ao] = m.ops
        hast = tclip > 0
        ops[(po + hasl.long() + a_nops)[hast]] = (tclip[hast] << 4) | OP_S
        seg_read_c = torch.repeat_interleave(torch.arange(R, device=device), seg_per_read) + read_base
        seg_pos_c = torch.zeros(n_seg_c, **i64)
        seg_pos_c[seg_off] = start
        seg_fwd_c = torch.zeros(n_seg_c, dtype=torch.uint8, device=device)
        seg_fwd_c[seg_off] = (~is_rev).to(torch.uint8)
        if n_split:
            so = seg_op_off[seg_off[sidx] + 1]
            pre = (lead + a_len)[sidx]  # bases of the record before B
            # same strand: [pre S] opsB ; opposite strand (segment sees revcomp(record)): opsB [pre S]
            bid = torch.repeat_interleave(torch.arange(n_split, device=device), b_nops)
            bo = torch.arange(int(b_nops.sum().item()), device=device) - mb.op_off[:-1][bid]
            shift = (~b_opp).long()
            ops[so[bid] + shift[bid] + bo] = mb.ops
            clip_slot = torch.where(b_opp, so + b_nops, so)
            ops[clip_slot] = (pre << 4) | OP_S
            seg_pos_c[seg_off[sidx] + 1] = bstart
            # SA entry strand: same as the primary unless the part is on the opposite strand
            prim_fwd = (~is_rev)[sidx]
            seg_fwd_c[seg_off[sidx] + 1] = torch.where(b_opp, ~prim_fwd, prim_fwd).to(torch.uint8)
        parts["is_rev"].append(is_rev.to(torch.uint8))
        # pack per contig so that the full-size workloads never hold all bases as ASCII + int64 indices at once
        so_c = torch.zeros(R + 1, **i64)
        so_c[1:] = torch.cumsum(total_len, 0)
        if cfg.seq_fmt == abi.SEQ_BAM4:
            packed, boff = _pack_bam4(bases, so_c)
        else:
            packed, boff = bases, so_c
        parts["seq"].append(packed)
        parts["seq_off"].append(boff[:-1] + seq_byte_base)
        seq_byte_base += int(packed.numel())
        del bases, run_id, boff, rr, kind
        parts["seq_len"].append(total_len)
        parts["seg_read"].append(seg_read_c)
        parts["seg_contig"].append(torch.full((n_seg_c,), c, **i64))
        parts["seg_pos"].append(seg_pos_c)
        parts["seg_fwd"].append(seg_fwd_c)
        parts["seg_nops"].append(seg_nops)
        parts["ops"].append(ops)
        read_base += R

    def cat(key, dtype):
        if not parts[key]:
            return torch.zeros(0, dtype=dtype, device=device)
        return torch.cat(parts[key]).to(dtype)

    seq_len = cat("seq_len", torch.long)
    seq = cat("seq", torch.uint8)
    read_seq_off = cat("seq_off", torch.long)
    seg_nops = cat("seg_nops", torch.long)
    seg_cigar_off_r = torch.zeros(seg_nops.numel() + 1, **i64)
    seg_cigar_off_r[1:] = torch.cumsum(seg_nops, 0)

    return Workload(
        cfg=cfg, device=device, chrom_seq=chrom_seq, contig_len=clen, contig_seg_off=np.array(contig_seg_off, dtype=np.uint32),
        seg_chrom_index=np.array(seg_chrom, dtype=np.uint32), seg_pos=np.array(seg_pos, dtype=np.int64),
        seg_is_fwd=np.array(seg_fwd, dtype=np.uint8), seg_mapq=np.array(seg_mapq, dtype=np.uint8),
        seg_start=np.array(seg_start, dtype=np.int64), seg_end=np.array(seg_end, dtype=np.int64),
        seg_cigar_off=seg_cigar_off, seg_cigar=seg_cigar, rev_contig_seq=rev_contig_seq,
        read_is_reverse=cat("is_rev", torch.uint8), read_seq_len=seq_len.to(torch.int32), read_seq_off=read_seq_off,
        seq=seq, seg_read=cat("seg_read", torch.long), seg_contig=cat("seg_contig", torch.long),
        seg_pos_r=cat("seg_pos", torch.long), seg_is_fwd_r=cat("seg_fwd", torch.uint8),
        seg_cigar_off_r=seg_cigar_off_r, cigar=cat("ops", torch.long), contig_fwd=contig_fwd if keep_contigs else None)
