"""World size 2 with the REAL engine on every rank (missing #2 / next #7): two processes share the box's one GPU for
compute (RCCL refuses two ranks on one device -- profiles/r02_rccl_2proc_1gpu_refused.txt -- compute does not), gloo carries the
record gather.  Each rank: the reference's windows (get_region_segments, lib/rust-vc-utils/src/util.rs:50-67; a read belongs to the
window its primary alignment starts in, src/read_alignment_scanner.rs:403-406) dealt by input ops -> plo_liftover_batch_dev on its
read ranges -> plo_compact_output_dev -> host tensors -> gather.gather_payloads -> rank 0 compares the gathered record set, bit for
bit and order-independently, with its own HIP result of the whole read set AND with the oracle on a sample."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, n_reads, segment_size):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from portello_amd import abi, api, devbatch, gather, shard, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = synth.generate(synth.config("chr20", n_reads=n_reads), device=dev)  # same seed on every rank: the same read set
    index = api.Index(w.index_data_device())
    eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
    wins = shard.workload_windows(w, segment_size=segment_size)
    deal = shard.deal_windows(wins, world)
    assert len(wins) >= 2 * world and all(len(d) > 0 for d in deal)
    ranges = shard.rank_read_ranges(wins, deal, rank)
    db = devbatch.DeviceBatch.from_read_ranges(w, ranges)
    torch.cuda.synchronize()  # (torch's default stream is handle 0 = "create a private stream" for the engine: the inputs must be complete)
    out = eng.liftover_batch_dev(db.desc())
    eng.compact_output_dev(out)
    eng.sync()
    t = eng.timing()
    assert t.n_lane_items > 0  # (the HIP lane kernel ran on this rank)
    mine = {k: v.cpu() for k, v in gather.tensors_from_out(out, dev).items()}  # host tensors: gloo
    got = gather.gather_payloads(mine, dist, rank, world)
    if rank == 0:
        seg_maps = [gather.local_to_global_segments(w, shard.rank_read_ranges(wins, deal, r)).cpu() for r in range(world)]
        assert sum(int(m.numel()) for m in seg_maps) == int(w.seg_read.numel())  # every read segment belongs to exactly one rank
        allr = gather.combine(got, seg_maps)
        whole_db = devbatch.DeviceBatch.from_workload(w)
        torch.cuda.synchronize()
        whole_out = eng.liftover_batch_dev(whole_db.desc())
        eng.compact_output_dev(whole_out)
        eng.sync()
        whole = {k: v.cpu() for k, v in gather.tensors_from_out(whole_out, dev).items()}
        assert int(allr["item_seg"].numel()) == int(whole["item_seg"].numel())
        assert gather.same_records(allr, whole)
        allr["item_ref_pos"][5] += 1  # (the comparison notices a difference)
        assert not gather.same_records(allr, whole)
        # ... and the single-GPU result it is compared with is the oracle's, on blocks of reads spread over the set
        from oracle import pyoracle
        import fullsize

        pyoracle.build()
        res = devbatch.download(eng, whole_out)
        n_cmp, _, _ = fullsize.check_strided_parity(w, res, pyoracle, 10, 200, ix=w.index_data())
        assert n_cmp >= 2000
        open(os.path.join(outdir, "ok"), "w").write(f"{int(whole['item_seg'].numel())} {len(wins)} {n_cmp}")
    else:
        assert got is None
    dist.barrier()
    eng.close()
    index.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_shard_hip_engine_gather_world_size_2():
    n_reads = 30_000
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d, n_reads, 5_000_000), nprocs=2, join=True)
        ok = os.path.join(d, "ok")
        assert os.path.exists(ok)
        items, n_win, n_cmp = (int(x) for x in open(ok).read().split())
        assert items >= n_reads and n_win >= 4
