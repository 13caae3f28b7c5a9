"""Random adversarial batches for the parity tests: every CIGAR op code (M I D N S H P = X), zero-length ops, edge
indels, clips inside CIGARs, tiny alphabets (long homologies), inconsistent lengths (LEN_MISMATCH), out-of-range lifted
positions (PANIC), empty maps, reads crossing several contig segments, reverse-mapped segments."""
import numpy as np

from portello_amd import abi

OPS_W = np.array([30, 14, 14, 3, 6, 3, 2, 18, 10], dtype=np.float64)  # M I D N S H P = X


def rand_cigar(rng, n_ops, max_len=12, zero_prob=0.08, ops_w=OPS_W):
    if n_ops == 0:
        return np.zeros(0, np.uint32)
    ops = rng.choice(9, size=n_ops, p=ops_w / ops_w.sum())
    lens = rng.integers(1, max_len + 1, size=n_ops)
    lens[rng.random(n_ops) < zero_prob] = 0
    return ((lens.astype(np.uint32) << 4) | ops.astype(np.uint32)).astype(np.uint32)


def ref_len(c):
    t = c & 15
    return int(((c >> 4) * np.isin(t, [0, 2, 3, 7, 8])).sum())


def make(seed: int, n_contigs=4, n_reads=40, alphabet=b"ACGT", explicit=False, seq_fmt=abi.SEQ_ASCII):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    n_chroms = 2
    chrom_seq = [alpha[rng.integers(0, len(alpha), size=int(rng.integers(150, 500)))] for _ in range(n_chroms)]
    contig_len, contig_seg_off, rev = [], [0], []
    seg_chrom, seg_pos, seg_fwd, seg_mapq, seg_s, seg_e, seg_cig = [], [], [], [], [], [], []
    for c in range(n_contigs):
        L = int(rng.integers(120, 400))
        contig_len.append(L)
        nseg = int(rng.integers(0, 4)) if c else 2
        cuts = np.sort(rng.integers(0, L + 1, size=2 * nseg))
        any_rev = False
        for s in range(nseg):
            seg_chrom.append(int(rng.integers(0, n_chroms)))
            seg_pos.append(int(rng.integers(0, 200)))
            fwd = bool(rng.random() < 0.5)
            any_rev |= not fwd
            seg_fwd.append(int(fwd))
            seg_mapq.append(int(rng.integers(0, 61)))
            seg_s.append(int(cuts[2 * s]))
            seg_e.append(int(cuts[2 * s + 1]))
            seg_cig.append(rand_cigar(rng, int(rng.integers(0, 14)), max_len=40, zero_prob=0.03,
                                      ops_w=np.array([40, 8, 8, 2, 6, 4, 1, 20, 10], dtype=np.float64)))
        contig_seg_off.append(len(seg_pos))
        # rev_contig_seq present for contigs with a reverse segment (and sometimes otherwise)
        rev.append(alpha[rng.integers(0, len(alpha), size=L)] if (any_rev or rng.random() < 0.3) else None)
    seg_cigar_off = np.zeros(len(seg_cig) + 1, dtype=np.uint32)
    seg_cigar_off[1:] = np.cumsum([len(c) for c in seg_cig])
    index = abi.IndexData(
        contig_len=contig_len, contig_seg_off=contig_seg_off, seg_chrom_index=seg_chrom, seg_pos=seg_pos, seg_is_fwd_strand=seg_fwd,
        seg_mapq=seg_mapq, seg_seq_order_start=seg_s, seg_seq_order_end=seg_e, seg_cigar_off=seg_cigar_off,
        seg_cigar=np.concatenate(seg_cig) if seg_cig else np.zeros(0, np.uint32), chrom_seq=chrom_seq, rev_contig_seq=rev)

    read_rev, read_len, read_off, seqs = [], [], [], []
    sg_read, sg_contig, sg_pos, sg_fwd, sg_cig = [], [], [], [], []
    off = 0
    for r in range(n_reads):
        nsegs = int(rng.choice([1, 1, 1, 2, 3]))
        cigs = []
        for _ in range(nsegs):
            cg_ = rand_cigar(rng, int(rng.integers(0, 16)))
            cigs.append(cg_)
        # most reads are length-consistent with their first segment's CIGAR, some are not
        t = cigs[0] & 15
        rl = int(((cigs[0] >> 4) * np.isin(t, [0, 1, 4, 5, 7, 8])).sum())
        if rng.random() < 0.15:
            rl = int(rng.integers(0, 80))
        nbytes = (rl + 1) // 2 if seq_fmt == abi.SEQ_BAM4 else rl
        if seq_fmt == abi.SEQ_BAM4:
            seqs.append(rng.integers(0, 256, size=nbytes).astype(np.uint8))
        else:
            seqs.append(alpha[rng.integers(0, len(alpha), size=rl)])
        read_rev.append(int(rng.random() < 0.5))
        read_len.append(rl)
        read_off.append(off)
        off += nbytes
        for cg_ in cigs:
            c = int(rng.integers(0, n_contigs))
            span = ref_len(cg_)
            hi = contig_len[c] - span
            if hi < 0:  # keep the read inside the contig (rev_pos >= 0)
                cg_ = cg_[:0]
                hi = contig_len[c]
            sg_read.append(r)
            sg_contig.append(c)
            sg_pos.append(int(rng.integers(0, hi + 1)))
            sg_fwd.append(int(rng.random() < 0.5))
            sg_cig.append(cg_)
    coff = np.zeros(len(sg_cig) + 1, dtype=np.uint32)
    coff[1:] = np.cumsum([len(c) for c in sg_cig])
    item_seg = item_cseg = None
    if explicit:
        item_seg, item_cseg = [], []
        for s, c in enumerate(sg_contig):
            ns = contig_seg_off[c + 1] - contig_seg_off[c]
            for k in range(ns):
                if rng.random() < 0.7:
                    item_seg.append(s)
                    item_cseg.append(k)
    batch = abi.BatchData(
        read_is_reverse=read_rev, read_seq_len=read_len, read_seq_off=read_off,
        seq=np.concatenate(seqs) if seqs else np.zeros(0, np.uint8), seq_fmt=seq_fmt, seg_read=sg_read, seg_contig=sg_contig,
        seg_pos=sg_pos, seg_is_fwd_strand=sg_fwd, seg_cigar_off=coff,
        cigar=np.concatenate(sg_cig) if sg_cig else np.zeros(0, np.uint32), item_seg=item_seg, item_cseg=item_cseg)
    return index, batch
