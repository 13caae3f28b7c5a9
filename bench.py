#!/usr/bin/env python3
"""bench.py -- lifted HiFi reads/sec of the MI355X liftover engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload wgs30x|chr20|stress|plumbing] [--reads R]

One *step* = one pass of the hot path (plo_liftover_batch_dev: item enumeration, strand preparation, left-shift,
liftover, length check, simplify) over one batch of synthetic reads that is already resident in HBM when the timed
region starts.  At N > 1 (launched by torch.distributed.run, one rank per GPU) every rank lifts its own shard of
the same size (weak scaling, reads shard with no data-path collective) and the compact result records are then
gathered to rank 0 over RCCL (peer -> root send/recv, the path's only exchange step).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

# the host driver of this pool only supports dmabuf IPC (RCCL / cross-process tensor sharing fail without it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from portello_amd import abi, api, devbatch, synth  # noqa: E402
from portello_amd import gather as plo_gather  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBS = 6300.0  # measured copy ceiling (same guide)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(w, eng, db, got_out, budget_s: float = 12.0):
    """Times the oracle (CPU restatement of the reference algorithm) on a bounded sample of the same workload and
    checks the GPU result of those reads against it.  Rank 0, N = 1 only."""
    from oracle import pyoracle

    pyoracle.build()
    cores = os.cpu_count() or 1
    ixd = w.index_data()  # host copy of the index
    n_reads = w.n_reads
    # calibrate on a small slice, then size the sample for ~budget_s of wall time on all cores
    probe = min(n_reads, 2000)
    b = w.batch_data(0, probe)
    t0 = time.perf_counter()
    pyoracle.liftover_batch(ixd, b, abi.STAGES_ALL, cores)
    t_probe = max(1e-4, time.perf_counter() - t0)
    rate = probe / t_probe
    sample = int(min(n_reads, 300_000, max(probe, rate * budget_s)))
    b = w.batch_data(0, sample)
    t0 = time.perf_counter()
    ref = pyoracle.liftover_batch(ixd, b, abi.STAGES_ALL, cores)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    pyoracle.liftover_batch(ixd, w.batch_data(0, probe), abi.STAGES_ALL, 1)
    rate1 = probe / max(1e-4, time.perf_counter() - t0)
    # parity of the sampled reads: GPU items of reads [0, sample) vs oracle (item order is (segment, contig segment))
    got = devbatch.download(eng, got_out)
    n_seg_sample = b.n_segs
    keep = got.item_seg < n_seg_sample
    sub = abi.BatchResult(*(getattr(got, f)[keep] if f != "cigar" else got.cigar for f in
                            ("item_seg", "item_cseg", "item_status", "item_need_flipped", "item_mapq", "item_chrom_index",
                             "item_ref_pos", "item_cigar_off", "item_cigar_len", "cigar")))
    ok = sub.canonical() == ref.canonical()
    return {"value": sample / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {sample} reads of the workload, oracle/liboracle.so (C restatement of the reference algorithm, "
                      f"not the reference binary), {cores} threads; 1 thread: {rate1:.0f} reads/s",
            "single_thread_value": rate1, "seconds": dt}, ok, int(keep.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("PLO_BENCH_WORKLOAD", "wgs30x"))
    ap.add_argument("--reads", type=int, default=0, help="override the workload's read count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workers", type=int, default=int(os.environ.get("PLO_BENCH_WORKERS", "1")),
                    help="host worker threads per GPU, each with its own context and HIP stream (INTEGRATION.md: one plo_ctx per "
                         "rayon worker); batches are dealt to them in turn, so one worker's enumerate pass and host syncs overlap the "
                         "other's tile kernel.  Default 1: the HIP-event kernel times of the roofline object then measure execution "
                         "only (with several streams they include the wait behind the other stream's kernel)")
    ap.add_argument("--async-gather", action="store_true", default=bool(os.environ.get("PLO_BENCH_ASYNC_GATHER")),
                    help="N > 1: two contexts alternate and the record gather of batch i stays in flight while batch i+1 is "
                         "computed (single thread per rank, same posting order on all ranks).  Opt-in: covered by the gloo "
                         "tests, not yet exercised over RCCL on several GPUs")
    ap.add_argument("--overlap-workers", type=int, default=0,
                    help="after the timed region, repeat the same K steps with this many workers and report the rate as the "
                         "supplementary object `overlap` (N = 1 only; 0/1 = skip, the default: the extra launches would enter a "
                         "rocprofv3 kernel summary of the same command with their longer, overlapped durations)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("PLO_BENCH_FORCE_DIST"):  # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    over = {"seed": synth.config(args.workload).seed + 1000 * rank}
    if args.reads:
        over["n_reads"] = args.reads
    cfg = synth.config(args.workload, **over)
    t0 = time.perf_counter()
    w = synth.generate(cfg, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    if rank == 0:
        log(f"[bench] workload {cfg.name}: {w.n_reads} reads, {w.seg_read.numel()} read segments, {int(w.cigar.numel())} input ops, "
            f"{len(w.contig_len)} contigs / {len(w.seg_pos)} contig segments, generated on GPU in {t_gen:.1f}s")

    index = api.Index(w.index_data_device(), device=local_rank)
    n_workers = max(1, args.workers)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_workers)]
    engs = [api.Engine(index, stream=s.cuda_stream) for s in streams]
    eng = engs[0]
    db = devbatch.DeviceBatch.from_workload(w)
    desc = db.desc()
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    gather_lock = threading.Lock()  # one exchange at a time per rank: the n-th exchange of every rank pairs up
    last_out = [None] * n_workers

    def step(k):
        out = engs[k].liftover_batch_dev(desc)
        if dist is not None:
            engs[k].compact_output_dev(out)  # no slab gaps over xGMI
            with gather_lock, torch.cuda.stream(streams[k]):
                plo_gather.gather_results(out, dev, dist, rank, world)
        last_out[k] = out
        return out

    lift_ms, enum_ms, big_ms, mid_ms, retry_ms = [], [], [], [], []

    def run_steps(n_steps, record):
        """n_steps batches, dealt to the workers in turn; every worker drives its own context on its own stream"""
        errors = []

        def worker(k):
            try:
                torch.cuda.set_device(local_rank)
                for _ in range(k, n_steps, n_workers):
                    step(k)
                    if record:
                        tm = engs[k].timing()  # HIP events on the worker's stream (waits for the step)
                        lift_ms.append(tm.lift_ms)
                        enum_ms.append(tm.enumerate_ms)
                        big_ms.append(tm.big_ms)
                        mid_ms.append(tm.mid_ms)
                        retry_ms.append(tm.retry_ms)
            except BaseException as e:  # noqa: BLE001 -- re-raised on the main thread
                errors.append(e)

        if n_workers == 1:
            worker(0)
        else:
            th = [threading.Thread(target=worker, args=(k,)) for k in range(n_workers)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errors:
            raise errors[0]

    async_gather = bool(args.async_gather and dist is not None and n_workers == 1)
    if async_gather:
        engs.append(api.Engine(index, stream=torch.cuda.Stream(device=dev).cuda_stream))  # second set of output buffers

        def run_steps(n_steps, record):  # noqa: F811 -- pipelined variant of the loop above
            pending = [None, None]
            for i in range(n_steps):
                k = i & 1
                if pending[k] is not None:  # the exchange that still reads context k's outputs
                    pending[k].wait()
                    pending[k] = None
                out_k = engs[k].liftover_batch_dev(desc)
                engs[k].compact_output_dev(out_k)
                last_out[0] = out_k
                pending[k] = plo_gather.gather_results_async(out_k, dev, dist, rank, world)
                if record:
                    tm = engs[k].timing()
                    lift_ms.append(tm.lift_ms)
                    enum_ms.append(tm.enumerate_ms)
                    big_ms.append(tm.big_ms)
                    mid_ms.append(tm.mid_ms)
                    retry_ms.append(tm.retry_ms)
            for p_ in pending:
                if p_ is not None:
                    p_.wait()

    run_steps(max(args.warmup, n_workers), False)  # every context sizes its buffers outside the timed region
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = last_out[0]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nr = torch.tensor([w.n_reads], dtype=torch.float64, device=dev)
        dist.all_reduce(nr, op=dist.ReduceOp.SUM)
        total_reads = float(nr.item())
    else:
        total_reads = float(w.n_reads)

    # SURVEY.md 8(e) "report both": the same K batches without the final record gather (every rank keeps / writes its own shard)
    no_gather = None
    if dist is not None:
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            engs[0].liftover_batch_dev(desc)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        no_gather = {"value": total_reads * args.steps / float(t.item()), "unit": "reads/s", "ms_per_step": float(t.item()) / args.steps * 1e3,
                     "note": "same K steps without the record gather (each rank keeps its shard); supplementary"}
        out = last_out[0] = engs[0].liftover_batch_dev(desc)

    overlap = None
    if dist is None and n_workers == 1 and args.overlap_workers > 1:
        # supplementary: the same K batches dealt to several workers (not `value`: see --workers)
        ow = args.overlap_workers
        o_streams = [torch.cuda.Stream(device=dev) for _ in range(ow)]
        o_engs = [api.Engine(index, stream=s_.cuda_stream) for s_ in o_streams]

        def o_run(n_steps):
            th = [threading.Thread(target=lambda k=k: [o_engs[k].liftover_batch_dev(desc) for _ in range(k, n_steps, ow)]) for k in range(ow)]
            for t in th:
                t.start()
            for t in th:
                t.join()

        o_run(ow)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        o_run(args.steps)
        torch.cuda.synchronize()
        odt = time.perf_counter() - t1
        overlap = {"host_workers_per_gpu": ow, "value": w.n_reads * args.steps / odt, "unit": "reads/s", "ms_per_step": odt / args.steps * 1e3,
                   "note": "same K batches dealt to several host workers (one context + HIP stream each): enumerate pass and host syncs of one "
                           "worker overlap the other's tile kernel; supplementary, the headline value is the single-worker rate"}
        for e in o_engs:
            e.close()
    tm = eng.timing()
    kms = {"k_lift_mid": float(np.mean(mid_ms)), "k_lift_tiles": float(np.mean(lift_ms)), "k_lift_big": float(np.mean(big_ms)),
           "k_lift_retry": float(np.mean(retry_ms))}
    dominant = max(kms, key=kms.get)
    dom_ms = kms[dominant]
    # algorithmic bytes are counted by the kernels themselves (SURVEY.md 8(d) formula), summed over all lift kernels;
    # attribute them to the dominant kernel in proportion to its share of the lift time
    share = dom_ms / max(1e-9, sum(kms.values()))
    achieved = (tm.algo_bytes * share) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = None
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof):
        try:
            tr = json.load(open(prof))
            traffic = tr.get(cfg.name, {}).get(dominant)
        except Exception:
            traffic = None

    result = {
        "metric": "lifted HiFi reads/sec (whole node)",
        "value": total_reads * args.steps / dt,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": cfg.name, "reads_per_gpu": w.n_reads, "read_len_mean": cfg.read_len_mean,
                   "items_per_gpu": int(tm.n_items), "in_ops_per_gpu": int(tm.n_in_ops), "out_ops_per_gpu": int(tm.n_out_ops),
                   "large_items_per_gpu": int(tm.n_big_items), "mid_items_per_gpu": int(tm.n_mid_items), "retry_items_per_gpu": int(tm.n_retry_items), "seq_fmt": "bam4", "parallelism": f"shard{world}", "host_workers_per_gpu": n_workers,
                   "gather": ("rccl send/recv to rank 0" + (", overlapped with the next batch" if async_gather else "")) if world > 1 else "none"},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "frac_of_copy_ceiling": achieved / HBM_COPY_CEILING_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(tm.algo_bytes * share), "kernel_ms": dom_ms,
                     "enumerate_ms": float(np.mean(enum_ms)), "lift_tiles_ms": float(np.mean(lift_ms)),
                     "lift_big_ms": float(np.mean(big_ms)), "lift_mid_ms": float(np.mean(mid_ms)),
                     "lift_retry_ms": float(np.mean(retry_ms))},
    }
    if overlap is not None:
        result["overlap"] = overlap
    if no_gather is not None:
        result["no_gather"] = no_gather
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cb, ok, n_checked = cpu_baseline(w, eng, db, out)
            result["cpu_baseline"] = cb
            result["parity_sample_items"] = n_checked
            result["parity_sample_ok"] = bool(ok)
            if not ok:
                log("[bench] PARITY FAILURE on the sampled reads")
        except Exception as e:  # the baseline must never hide the measurement
            log(f"[bench] cpu_baseline failed: {e!r}")
            result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result), flush=True)
    for e in engs:
        e.close()
    index.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
