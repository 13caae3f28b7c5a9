"""Record finishing: flags / alignment end / bin / primary selection / unmapped copy and
reverse_alignment_seq_and_qual.  CPU: oracle pins.  GPU: HIP vs oracle, byte for byte on the meaningful bytes."""
import numpy as np
import pytest

from portello_amd import abi, synth


def sam_spec_reg2bin(beg, end):
    """SAM specification 5.3, C code of reg2bin (independent pin of the restated bam_reg2bin)"""
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def test_reg2bin_matches_the_sam_specification(oracle):
    rng = np.random.default_rng(5)
    L = oracle.lib()
    for _ in range(3000):
        b = int(rng.integers(0, 2**29 - 2))
        e = b + int(rng.choice([1, 2, 100, 16384, 20000, 10**6, 10**8]))
        e = min(e, 2**29)
        assert L.orc_bam_reg2bin(b, e) == sam_spec_reg2bin(b, e)
    assert L.orc_bam_reg2bin(0, 1) == 4681 and L.orc_bam_reg2bin(16383, 16385) == 585


def _host_case(seed, seq_fmt):
    w = synth.generate(synth.config("tiny", n_reads=120, seed=seed, split_read_frac=0.2, read_len_mean=3000, read_len_sd=900,
                                    seq_fmt=seq_fmt))
    b = w.batch_data()
    rng = np.random.default_rng(seed)
    lens = b.read_seq_len.astype(np.int64)
    qoff = np.cumsum(lens) - lens
    qual = rng.integers(0, 94, size=int(lens.sum()), dtype=np.uint8)
    flags = (b.read_is_reverse.astype(np.uint16) * 0x10) | (rng.integers(0, 2, size=b.n_reads).astype(np.uint16) * 0x400)
    return w, b, flags, qual, qoff


def test_oracle_finish_semantics(oracle):
    w, b, flags, qual, qoff = _host_case(401, abi.SEQ_BAM4)
    lift = oracle.liftover_batch(w.index_data(), b, abi.STAGES_ALL, 1)
    f = oracle.finish_batch(b, flags, qual, qoff, lift)
    reads = b.seg_read[lift.item_seg]
    for r in range(b.n_reads):
        items = np.nonzero((reads == r) & (lift.item_status == 0))[0]
        assert f["read_n_lifted"][r] == len(items)
        if len(items):
            assert f["item_is_primary"][items].sum() == 1
            p = items[f["item_is_primary"][items] == 1][0]
            assert lift.item_mapq[p] == lift.item_mapq[items].max() and p == items[lift.item_mapq[items] == lift.item_mapq[p]][0]
            assert not (f["item_flag"][p] & 0x800) and all(f["item_flag"][i] & 0x800 for i in items if i != p)
        else:
            assert f["read_unmapped_flag"][r] & 0x4 and not f["read_unmapped_flag"][r] & 0x10
    # a flipped record: reverse complement of the decoded read, re-encoded; qualities reversed
    i = int(np.nonzero(f["item_seq_off"] != abi.NO_FLIP)[0][0])
    r = int(reads[i])
    L = int(b.read_seq_len[r])
    src = oracle.decode_bam4(b.seq[int(b.read_seq_off[r]):], L)
    got = oracle.decode_bam4(f["rev_seq"][int(f["item_seq_off"][i]):], L)
    assert got == oracle.rev_comp(src)
    assert (f["rev_qual"][int(f["item_qual_off"][i]): int(f["item_qual_off"][i]) + L] == qual[qoff[r]: qoff[r] + L][::-1]).all()
    assert (f["item_flag"][i] ^ flags[r]) & 0x10


def _compare_finish(got, ref, lift, b, seq_fmt):
    lifted = lift.item_status == 0
    for name, _ in abi.FINISH_ITEM_FIELDS:
        assert (got[name][lifted] == ref[name][lifted]).all(), name
    for name, _ in abi.FINISH_READ_FIELDS:
        sel = slice(None) if name != "read_unmapped_flag" else (ref["read_n_lifted"] == 0)
        assert (got[name][sel] == ref[name][sel]).all(), name
    assert len(got["rev_seq"]) == len(ref["rev_seq"]) and len(got["rev_qual"]) == len(ref["rev_qual"])
    reads = b.seg_read[lift.item_seg]
    n_checked = 0
    for offs_s, offs_q, rd in ((got["item_seq_off"], got["item_qual_off"], reads), (got["read_seq_off"], got["read_qual_off"], np.arange(b.n_reads))):
        for k in np.nonzero(offs_s != abi.NO_FLIP)[0]:
            L = int(b.read_seq_len[rd[k]])
            nb = (L + 1) // 2 if seq_fmt == abi.SEQ_BAM4 else L
            so, qo = int(offs_s[k]), int(offs_q[k])
            assert (got["rev_seq"][so: so + nb] == ref["rev_seq"][so: so + nb]).all()
            assert (got["rev_qual"][qo: qo + L] == ref["rev_qual"][qo: qo + L]).all()
            n_checked += 1
    return n_checked


@pytest.mark.parametrize("seq_fmt", [abi.SEQ_BAM4, abi.SEQ_ASCII])
def test_finish_device_code_on_host_vs_oracle(oracle, seq_fmt):
    """finish_core.hpp (the code the GPU runs) executed with host loops, odd thread counts, every length residue"""
    import emu_lib

    for seed in (411, 412):
        w, b, flags, qual, qoff = _host_case(seed, seq_fmt)
        lift = oracle.liftover_batch(w.index_data(), b, abi.STAGES_ALL, 1)
        ref = oracle.finish_batch(b, flags, qual, qoff, lift)
        got = emu_lib.finish_batch(b, flags, qual, qoff, lift, nthreads=7 if seed == 411 else 64)
        assert _compare_finish(got, ref, lift, b, seq_fmt) > 20


@pytest.mark.gpu
@pytest.mark.parametrize("seq_fmt", [abi.SEQ_BAM4, abi.SEQ_ASCII])
def test_finish_hip_vs_oracle(oracle, seq_fmt):
    import ctypes as C

    import torch

    from portello_amd import api, devbatch

    w = synth.generate(synth.config("tiny", n_reads=400, seed=402 + seq_fmt, split_read_frac=0.2, read_len_mean=3000, read_len_sd=1200,
                                    seq_fmt=seq_fmt), device="cuda")
    index = api.Index(w.index_data_device())
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    db = devbatch.DeviceBatch.from_workload(w)
    desc = db.desc()
    fin, keep = devbatch.finish_inputs(w, db, seed=9)
    torch.cuda.synchronize()
    out = eng.liftover_batch_dev(desc)
    fo = eng.finish_batch_dev(desc, fin)
    lift = devbatch.download(eng, out)
    got = devbatch.download_finish(eng, fo, lift.n_items, db.n_reads)
    b = w.batch_data()
    ref = oracle.finish_batch(b, keep["flags"].cpu().numpy().view(np.uint16), keep["qual"].cpu().numpy(), keep["qoff"].cpu().numpy(),
                              oracle.liftover_batch(w.index_data(), b, abi.STAGES_ALL, 2))
    n_checked = _compare_finish(got, ref, lift, b, seq_fmt)
    assert n_checked > 50
    eng.close()
    index.close()


# ---- SA tag text (get_sa_tag_segment, src/read_alignment_scanner.rs:292-301, :348-364) --------------------------------------

def _chrom_names(n):
    return [f"chr{k + 1}" if k % 3 else f"chrUn_scaffold{k:04d}v1" for k in range(n)]


def test_oracle_sa_format(oracle):
    """hand-checked SA values on a two-record read (1-based position, strand from the record's reverse bit, CIGAR text)"""
    C_ = abi
    batch = C_.BatchData(read_is_reverse=[0], read_seq_len=[10], read_seq_off=[0], seq=np.zeros(5, np.uint8), seq_fmt=C_.SEQ_BAM4,
                         seg_read=[0, 0], seg_contig=[0, 0], seg_pos=[0, 0], seg_is_fwd_strand=[1, 1], seg_cigar_off=[0, 1, 2],
                         cigar=[(10 << 4) | 0, (10 << 4) | 0])
    lift = C_.BatchResult(item_seg=np.array([0, 1], np.uint32), item_cseg=np.array([0, 0], np.uint32), item_status=np.array([0, 0], np.uint8),
                          item_need_flipped=np.array([0, 1], np.uint8), item_mapq=np.array([60, 7], np.uint8),
                          item_chrom_index=np.array([1, 0], np.uint32), item_ref_pos=np.array([99, 12344], np.int64),
                          item_cigar_off=np.array([0, 3], np.uint64), item_cigar_len=np.array([3, 2], np.uint32),
                          cigar=np.array([(4 << 4) | 4, (5 << 4) | 0, (1 << 4) | 2, (7 << 4) | 0, (3 << 4) | 1], np.uint32))
    vals = oracle.sa_values(batch, lift, np.array([0x0, 0x810], np.uint16), ["chrA", "chrB"])
    assert vals == [b"chrA,12345,-,7M3I,7,0;", b"chrB,100,+,4S5M1D,60,0;"]
    # a read with a single lifted record gets no SA tag
    lift.item_status = np.array([0, 1], np.uint8)
    assert oracle.sa_values(batch, lift, np.array([0x0, 0x810], np.uint16), ["chrA", "chrB"]) == [None, None]


def test_sa_device_code_on_host_vs_oracle(oracle):
    import emu_lib
    from portello_amd import api

    n_with = 0
    for seed in (421, 422):
        w, b, flags, qual, qoff = _host_case(seed, abi.SEQ_BAM4)
        ix = w.index_data()
        lift = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 1)
        f = oracle.finish_batch(b, flags, qual, qoff, lift)
        names = _chrom_names(len(ix.chrom_len))
        ref = oracle.sa_values(b, lift, f["item_flag"], names)
        off, text, item_read = emu_lib.sa_segments(b, lift, f["item_flag"], f["read_n_lifted"], names)
        got = api.assemble_sa_values(off, text, item_read)
        assert got == ref
        n_with += sum(v is not None for v in ref)
    assert n_with > 10


@pytest.mark.gpu
def test_sa_hip_vs_oracle(oracle):
    import torch

    from portello_amd import api, devbatch

    w = synth.generate(synth.config("tiny", n_reads=600, seed=431, split_read_frac=0.3, read_len_mean=3000, read_len_sd=1200), device="cuda")
    index = api.Index(w.index_data_device())
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    db = devbatch.DeviceBatch.from_workload(w)
    desc = db.desc()
    fin, keep = devbatch.finish_inputs(w, db, seed=9)
    ix = w.index_data()
    names = _chrom_names(len(ix.chrom_len))
    sa_in, keep_sa = devbatch.sa_inputs(names, db.seq.device)
    torch.cuda.synchronize()
    out = eng.liftover_batch_dev(desc)
    with pytest.raises(api.PortelloError):  # needs the finishing results (record flags, records per read)
        eng.sa_segments_dev(sa_in)
    fo = eng.finish_batch_dev(desc, fin)
    so = eng.sa_segments_dev(sa_in)
    lift = devbatch.download(eng, out)
    off, text = devbatch.download_sa(eng, so)
    b = w.batch_data()
    item_read = np.asarray(b.seg_read, dtype=np.uint32)[lift.item_seg]
    got = api.assemble_sa_values(off, text, item_read)
    olift = oracle.liftover_batch(ix, b, abi.STAGES_ALL, 2)
    of = oracle.finish_batch(b, keep["flags"].cpu().numpy().view(np.uint16), keep["qual"].cpu().numpy(), keep["qoff"].cpu().numpy(), olift)
    ref = oracle.sa_values(b, olift, of["item_flag"], names)
    assert got == ref
    assert sum(v is not None for v in ref) > 20
    eng.close()
    index.close()
