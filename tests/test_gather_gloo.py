"""N > 1 path on CPU: reads shard across ranks with no data-path collective, result records are gathered to rank 0
(gloo, world_size 2).  The oracle stands in for the per-rank engine here (no GPU); what is under test is the
sharding + pack/send/recv/unpack logic that bench.py runs over RCCL."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from portello_amd import abi, gather, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    """strong scaling as bench.py does it: one read set, the reference's windows dealt to the ranks by input ops, every rank
    lifts its windows (several read ranges), records gathered to rank 0 and compared with the unsharded result"""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle
    from portello_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.generate(synth.config("tiny", n_reads=120, seed=301, split_read_frac=0.2, sorted_reads=True))
    ix = w.index_data()
    wins = shard.workload_windows(w, segment_size=30_000)
    deal = shard.deal_windows(wins, world)
    assert len(wins) > 2 * world
    ranges = shard.rank_read_ranges(wins, deal, rank)
    # the rank's batch = its ranges one after the other; the oracle stands in for the engine, one call per range
    parts, seg_base = [], 0
    for lo, hi in ranges:
        b = w.batch_data(lo, hi)
        t = gather.tensors_from_result(pyoracle.liftover_batch(ix, b, abi.STAGES_ALL, 1))
        t["item_seg"] = t["item_seg"] + seg_base
        seg_base += b.n_segs
        parts.append(t)
    mine = gather.combine(parts)
    got = gather.gather_payloads(mine, dist, rank, world)
    if rank == 0:
        whole = gather.tensors_from_result(pyoracle.liftover_batch(ix, w.batch_data(), abi.STAGES_ALL, 1))
        seg_maps = [gather.local_to_global_segments(w, shard.rank_read_ranges(wins, deal, r)) for r in range(world)]
        assert sum(int(m.numel()) for m in seg_maps) == int(w.seg_read.numel())
        allr = gather.combine(got, seg_maps)
        assert gather.same_records(allr, whole)
        allr["item_ref_pos"][3] += 1  # and the comparison does notice a difference
        assert not gather.same_records(allr, whole)
        open(os.path.join(outdir, "ok"), "w").write("ok")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def gather_shard(n, rank, world):
    return (n * rank) // world, (n * (rank + 1)) // world


def test_pack_unpack_roundtrip(oracle):
    w = synth.generate(synth.config("tiny", n_reads=30, seed=302))
    res = oracle.liftover_batch(w.index_data(), w.batch_data(), abi.STAGES_ALL, 1)
    t = gather.tensors_from_result(res)
    back = gather.to_result(gather.unpack(gather.pack(t), res.n_items, len(res.cigar)))
    assert back.canonical() == res.canonical()


def test_shard_and_gather_world_size_2():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        assert os.path.exists(os.path.join(d, "ok"))


def _worker_async(rank, world, port, outdir):
    """two batches in flight: the exchange of batch 0 is still pending when batch 1 is computed and posted"""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pend, wholes = [], []
    for step in range(3):
        w = synth.generate(synth.config("tiny", n_reads=60 + 10 * step, seed=310 + step, split_read_frac=0.2))
        ix = w.index_data()
        lo, hi = gather_shard(w.n_reads, rank, world)
        res = pyoracle.liftover_batch(ix, w.batch_data(lo, hi), abi.STAGES_ALL, 1)
        if len(pend) == 2:  # double buffering: wait for the exchange posted two steps ago before "reusing its buffers"
            first = pend.pop(0)
            got = first[0].wait()
            if rank == 0:
                _check_gathered(got, first[1], first[2], world)
        pend.append((gather.gather_payloads_async(gather.tensors_from_result(res), dist, rank, world), w,
                     pyoracle.liftover_batch(ix, w.batch_data(), abi.STAGES_ALL, 1) if rank == 0 else None))
    for p_, w, whole in pend:
        got = p_.wait()
        if rank == 0:
            _check_gathered(got, w, whole, world)
        else:
            assert got is None
    if rank == 0:
        open(os.path.join(outdir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def _check_gathered(got, w, whole, world):
    rows = []
    for r in range(world):
        rlo, _ = gather_shard(w.n_reads, r, world)
        seg_base = int(torch.searchsorted(w.seg_read, torch.tensor(rlo)).item())
        part = gather.to_result(got[r])
        part.item_seg = part.item_seg + np.uint32(seg_base)
        rows += part.canonical()
    assert rows == whole.canonical()


def test_async_gather_world_size_2():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_async, args=(2, _free_port(), d), nprocs=2, join=True)
        assert os.path.exists(os.path.join(d, "ok"))


def _worker_pipeline8(rank, world, port, outdir):
    """BASELINE configs[3] / [4] name eight ranks: the deal for world size 8, every rank's windows as three consecutive batches
    (shard.pipeline_batches: the reference's window loop, src/read_alignment_scanner.rs:508-534), the exchange of a batch posted while the
    next one is lifted, rank 0 maps every batch of every rank back to the unsharded numbering.  The oracle stands in for the engine."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle
    from portello_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = 3
    w = synth.generate(synth.config("tiny", n_reads=400, seed=303, split_read_frac=0.2, sorted_reads=True))
    ix = w.index_data()
    wins = shard.workload_windows(w, segment_size=6_000)
    deal = shard.deal_windows(wins, world)
    assert len(wins) >= 3 * world and all(len(d) > 0 for d in deal)
    groups = [shard.pipeline_batches(wins, deal, r, k) for r in range(world)]
    # a rank's batches tile its read ranges exactly
    for r in range(world):
        flat = sorted(x for g in groups[r] for x in g)
        merged = []
        for lo, hi in flat:
            if merged and merged[-1][1] == lo:
                merged[-1] = (merged[-1][0], hi)
            else:
                merged.append((lo, hi))
        assert merged == shard.rank_read_ranges(wins, deal, r)
    pend = []
    for j in range(k):
        parts, seg_base = [], 0
        for lo, hi in groups[rank][j]:
            b = w.batch_data(lo, hi)
            t = gather.tensors_from_result(pyoracle.liftover_batch(ix, b, abi.STAGES_ALL, 1))
            t["item_seg"] = t["item_seg"] + seg_base
            seg_base += b.n_segs
            parts.append(t)
        mine = gather.combine(parts) if parts else gather.tensors_from_result(pyoracle.liftover_batch(ix, w.batch_data(0, 0), abi.STAGES_ALL, 1))
        if len(pend) == 2:  # two contexts' worth of buffers: the exchange posted two batches ago must be through
            pend[-2][1].wait()
        pend.append((j, gather.gather_payloads_async(mine, dist, rank, world)))
    got = {j: p_.wait() for j, p_ in pend}
    if rank == 0:
        whole = gather.tensors_from_result(pyoracle.liftover_batch(ix, w.batch_data(), abi.STAGES_ALL, 1))
        per_batch = []
        for j in range(k):
            maps_j = [gather.local_to_global_segments(w, groups[r][j]) for r in range(world)]
            per_batch.append(gather.combine(got[j], maps_j))
        allr = gather.combine(per_batch)
        assert int(allr["item_seg"].numel()) == int(whole["item_seg"].numel())
        assert gather.same_records(allr, whole)
        open(os.path.join(outdir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_window_pipeline_world_size_8():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_pipeline8, args=(8, _free_port(), d), nprocs=8, join=True)
        assert os.path.exists(os.path.join(d, "ok"))
