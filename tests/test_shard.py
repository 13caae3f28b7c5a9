"""Window maths and dealing of the strong-scaling mode (portello_amd/shard.py): pinned by the reference's own vectors for
get_region_segments (lib/rust-vc-utils/src/util.rs:187-193)."""
import numpy as np

from portello_amd import shard, synth


def test_region_segments_reference_vectors():
    assert shard.region_segments(100, 200) == [(0, 100)]
    assert shard.region_segments(100, 49) == [(0, 34), (34, 67), (67, 100)]


def test_region_segments_properties():
    rng = np.random.default_rng(5)
    for _ in range(200):
        size = int(rng.integers(1, 10**9))
        seg = int(rng.integers(1, 10**8))
        s = shard.region_segments(size, seg)
        assert s[0][0] == 0 and s[-1][1] == size
        assert all(a[1] == b[0] for a, b in zip(s, s[1:]))
        lens = [e - b for b, e in s]
        assert max(lens) <= seg and max(lens) - min(lens) <= 1


def test_windows_partition_the_reads_and_the_deal_is_balanced():
    w = synth.generate(synth.config("tiny", n_reads=400, seed=77, split_read_frac=0.1, sorted_reads=True))
    wins = shard.workload_windows(w, segment_size=25_000)
    # every read in exactly one window, windows in read order
    assert wins[0].read_lo == 0 and wins[-1].read_hi == w.n_reads
    assert all(a.read_hi == b.read_lo for a, b in zip(wins, wins[1:]))
    total = int(w.cigar.numel())
    assert sum(x.weight for x in wins) == total
    # a read belongs to the window its primary alignment starts in (src/read_alignment_scanner.rs:403-406)
    first = np.ones(w.seg_read.numel(), dtype=bool)
    sr = w.seg_read.numpy()
    first[1:] = sr[1:] != sr[:-1]
    start, contig = w.seg_pos_r.numpy()[first], w.seg_contig.numpy()[first]
    for x in wins:
        assert (contig[x.read_lo:x.read_hi] == x.contig).all()
        assert (start[x.read_lo:x.read_hi] >= x.begin).all() and (start[x.read_lo:x.read_hi] < max(x.end, x.begin + 1)).all()
    for world in (1, 2, 3, 8):
        deal = shard.deal_windows(wins, world)
        assert sorted(i for d in deal for i in d) == list(range(len(wins)))
        loads = [sum(wins[i].weight for i in d) for d in deal]
        # greedy heaviest-first: no rank exceeds the mean by more than the heaviest window
        assert max(loads) <= total / world + max(x.weight for x in wins)
        ranges = [shard.rank_read_ranges(wins, deal, r) for r in range(world)]
        assert sum(hi - lo for rr in ranges for lo, hi in rr) == w.n_reads
