"""Shared checks of the full-size GPU tests (BASELINE.json configs 2, 3 and 5): size-independent properties of every
lifted record + oracle parity on a *strided* sample (blocks of consecutive reads spread evenly over the whole read
set, hence over all contigs and both strands -- not the first N reads of a coordinate-sorted set)."""
import numpy as np

from portello_amd import abi

_FIELDS = ("item_seg", "item_cseg", "item_status", "item_need_flipped", "item_mapq", "item_chrom_index", "item_ref_pos",
           "item_cigar_off", "item_cigar_len", "cigar")


def flat_ops(res):
    lens = res.item_cigar_len.astype(np.int64)
    idx = np.repeat(res.item_cigar_off.astype(np.int64), lens) + (np.arange(lens.sum()) - np.repeat(np.cumsum(lens) - lens, lens))
    ops = res.cigar[idx]
    item = np.repeat(np.arange(res.n_items), lens)
    return ops & 15, (ops >> 4).astype(np.int64), item


def check_properties(w, res, min_lifted=0.95):
    """properties 1-4 of tests/test_gpu_parity.py::test_full_size_properties_chr20"""
    assert res.n_items >= w.n_reads * 0.9
    lifted = res.item_status == abi.ITEM_LIFTED
    assert lifted.mean() > min_lifted and (res.item_status <= abi.ITEM_NO_LIFTOVER).all()
    t, L, item = flat_ops(res)
    # (1) read length of the lifted CIGAR == seq_len (the reference's own sanity check, read_alignment_scanner.rs:206-207)
    rl = np.bincount(item, weights=L * np.isin(t, [0, 1, 4, 5, 7, 8]), minlength=res.n_items).astype(np.int64)
    seq_len = w.read_seq_len.cpu().numpy()[w.seg_read.cpu().numpy()[res.item_seg]]
    assert (rl[lifted] == seq_len[lifted]).all()
    # (2) canonical form: no zero-length op, no equal neighbours, only M I D N S H (=/X become M)
    assert (L > 0).all() and np.isin(t, [0, 1, 2, 3, 4, 5]).all()
    same_item = item[1:] == item[:-1]
    assert not (same_item & (t[1:] == t[:-1])).any()
    # (3) no indel at the alignment edges: the first and the last non-clip op of every record is a match
    nonclip = ~np.isin(t, [4, 5])
    first = np.full(res.n_items, -1)
    last = np.full(res.n_items, -1)
    pos = np.nonzero(nonclip)[0]
    first[item[pos][::-1]] = t[pos][::-1]
    last[item[pos]] = t[pos]
    assert (first[lifted] == 0).all() and (last[lifted] == 0).all()
    # (4) the lifted alignment stays inside its chromosome
    ref_span = np.bincount(item, weights=L * np.isin(t, [0, 2, 3]), minlength=res.n_items).astype(np.int64)
    clen = np.array([s.numel() for s in w.chrom_seq])[res.item_chrom_index]
    assert (res.item_ref_pos[lifted] >= 0).all() and ((res.item_ref_pos + ref_span)[lifted] <= clen[lifted]).all()
    return lifted


def sub_result(res, seg_lo, seg_hi):
    """items of read segments [seg_lo, seg_hi), segment indices re-based"""
    keep = (res.item_seg >= seg_lo) & (res.item_seg < seg_hi)
    d = {f: (getattr(res, f)[keep] if f != "cigar" else res.cigar) for f in _FIELDS}
    d["item_seg"] = (d["item_seg"] - seg_lo).astype(np.uint32)
    return abi.BatchResult(**d)


def strided_blocks(n_reads, n_blocks, block):
    if n_reads <= n_blocks * block:
        return [(0, n_reads)]
    stride = n_reads // n_blocks
    return [(i * stride, i * stride + block) for i in range(n_blocks)]


def check_strided_parity(w, res, oracle, n_blocks=40, block=400, threads=8, ix=None, blocks=None):
    """oracle parity of `n_blocks` blocks of `block` consecutive reads spread evenly over the read set; returns the number of
    items compared, how many sat on reverse-mapped contig segments, and the distinct contigs touched"""
    import torch

    ix = w.index_data() if ix is None else ix
    n_cmp = 0
    contigs = set()
    n_flip = 0
    for lo, hi in (blocks if blocks is not None else strided_blocks(w.n_reads, n_blocks, block)):  # (blocks: explicit read ranges, e.g. a tiling of the whole set)
        b = w.batch_data(lo, hi)
        ref = oracle.liftover_batch(ix, b, abi.STAGES_ALL, threads)
        seg_lo = int(torch.searchsorted(w.seg_read, torch.tensor(lo, device=w.device)).item())
        seg_hi = int(torch.searchsorted(w.seg_read, torch.tensor(hi, device=w.device)).item())
        got = sub_result(res, seg_lo, seg_hi)
        a, g = ref.canonical(), got.canonical()
        if a != g:
            bad = [(x[:7], y[:7]) for x, y in zip(a, g) if x != y]
            raise AssertionError(f"reads [{lo},{hi}): {len(bad)} of {len(a)} items differ (ref {len(a)} / got {len(g)}), first: {bad[:1]}")
        n_cmp += len(a)
        n_flip += int(ref.item_need_flipped.sum())
        contigs.update(int(c) for c in b.seg_contig)
    return n_cmp, n_flip, len(contigs)


def check_properties_device(w, out, dev, min_lifted=0.9):
    """properties (1)-(3) of check_properties computed on the GPU over every op of a result that is still in the engine's device
    buffers (a 250 k-read batch of the stress profile has 0.5 G output ops: too many to walk in numpy): read length of every
    lifted CIGAR == seq_len, canonical form, status range.  Returns (#items, #lifted)."""
    import torch

    from portello_amd import gather

    t = gather.tensors_from_out(out, dev)
    n = int(out.n_items)
    status = t["item_status"].long()
    assert int((status > abi.ITEM_NO_LIFTOVER).sum().item()) == 0
    lifted = status == abi.ITEM_LIFTED
    assert float(lifted.float().mean().item()) > min_lifted
    lens = t["item_cigar_len"].long()
    off = t["item_cigar_off"].long()
    total = int(lens.sum().item())
    item = torch.repeat_interleave(torch.arange(n, device=dev), lens)
    start = torch.cumsum(lens, 0) - lens
    idx = off[item] + (torch.arange(total, device=dev) - start[item])
    ops = t["cigar"].long()[idx]
    del idx
    ty, ln = ops & 15, ops >> 4
    del ops
    # (2) canonical form: no zero-length op, only M I D N S H, no equal neighbours inside a CIGAR
    assert int((ln <= 0).sum().item()) == 0 and int((ty > 5).sum().item()) == 0
    assert int(((item[1:] == item[:-1]) & (ty[1:] == ty[:-1])).sum().item()) == 0
    # (1) read bases consumed == seq_len (src/read_alignment_scanner.rs:206-207)
    readc = (ty == 0) | (ty == 1) | (ty == 4) | (ty == 5)
    rl = torch.zeros(n, dtype=torch.long, device=dev).scatter_add_(0, item, ln * readc.long())
    seq_len = w.read_seq_len.long()[w.seg_read.long()[t["item_seg"].long()]]
    assert int(((rl != seq_len) & lifted).sum().item()) == 0
    return n, int(lifted.sum().item())
