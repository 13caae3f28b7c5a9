"""BAM in -> lifted BAM out: the host-side mirror of the reference's phase 2 driver (scan_and_remap_reads,
src/read_alignment_scanner.rs:566-661) on top of the C ABI.

The reference spawns one rayon task per <= 20 Mb contig window; every task reads its records, lifts them one at a time and
writes through a shared, mutex-protected writer.  Here a *reader* thread decodes windows of primary records and builds
their batches (plo_bam_read_window + plo_bam_window_batch), `n_workers` *lift* threads -- each with its own plo_ctx and HIP
stream, the arrangement of INTEGRATION.md -- run plo_liftover_batch (page-locked H2D, kernels, D2H) and assemble the
output records (plo_records_build), and a *writer* thread emits them (plo_bam_write; BGZF level 0 = the reference's
stdout mode).  The three stages overlap; ctypes releases the GIL inside every native call.  Unmapped input records are
passed through to the "unassembled" writer (scan_unmapped_reads, :537-559).
"""
from __future__ import annotations

import os
import queue
import threading
import time
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from . import abi, api, bam


def effective_cpus() -> int:
    """host cores this process can actually use: the smallest of the CPU count, the affinity mask and the cgroup CPU quota"""
    import os

    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


@dataclass
class PipelineStats:
    reads: int = 0
    windows: int = 0
    records_out: int = 0
    lifted: int = 0
    unmapped_copies: int = 0
    unmapped_passed_through: int = 0
    bytes_out: int = 0
    seconds: float = 0.0
    read_s: float = 0.0      # BGZF inflate + record walk (reader thread busy time)
    batch_s: float = 0.0     # batch construction (batcher thread busy time)
    lift_s: float = 0.0      # plo_liftover_batch, summed over workers
    build_s: float = 0.0     # plo_records_build, summed over workers
    write_s: float = 0.0     # BGZF output (writer thread busy time)
    device_ms: float = 0.0   # HIP-event time of the lift calls
    finish_device_ms: float = 0.0  # device_finish: HIP-event time of the finishing, reverse-complement and SA-text kernels
    stage_done_s: dict = field(default_factory=dict)  # when each stage's thread ended, and the closes behind them (seconds after the start)
    lift_detail_s: dict = field(default_factory=dict)  # device_finish: the lift stage by step (host clock; the steps that wait for the device carry its time)
    out_paths: List[str] = field(default_factory=list)  # the output file, or the shards (out_shards > 1)
    errors: List[str] = field(default_factory=list)


def run_bam_to_bam(in_path: str, out_path: str, index: api.Index, index_data: abi.IndexData, contig_names: Sequence[str],
                   ref_names: Sequence[str], ref_lens: Sequence[int], window_reads: int = 50_000, n_workers: int = 2,
                   io_threads: int = 16, level: int = 0, unassembled_path: Optional[str] = None, is_target_region: bool = False,
                   cmdline: str = "", sparse_margin: Optional[int] = 32, device_inflate: Optional[bool] = True,
                   device_finish: bool = False, read_threads: Optional[int] = None, build_threads: Optional[int] = None,
                   write_threads: Optional[int] = None, ramp: bool = True, part: Optional[int] = None, n_parts: int = 1,
                   out_shards: int = 1, n_readers: int = 1) -> PipelineStats:  # noqa: E501
    """device_finish: the records are finished on the device -- the window's batch goes up with all its bases and qualities
    (sparse_margin is ignored), plo_finish_batch_dev (flags, bin, primary record, reverse_alignment_seq_and_qual) and
    plo_sa_segments_dev (SA text) run behind the lift kernels, their results come back and plo_records_build_finished only copies
    them into place.  Same bytes as the host finishing (tests/test_bam.py).
    sparse_margin: the windows' read bases go to the device as PLO_SEQ_BAM4_SPARSE (granules within that many bases of an indel;
    the complete bases stay in the window's records for the engine's second look); None = dense bases.  device_inflate: the BGZF
    blocks of the input are inflated on the GPU (leaves the host cores to record assembly and output; falls back to the host without
    a device), None = as the environment says.
    part / n_parts: this process's share of the input (plo_bam_open_range: a split by compressed offset) -- with several GPUs every rank
    runs the pipeline over its part and writes its own output shard (the reference's output order is unspecified: the shards'
    concatenation is a valid result)
    n_readers > 1: the input (or this process's part of it) is cut again into that many parts by compressed offset (plo_bam_open_range),
    every part with a reader and a batcher thread of its own feeding the same lift workers: with the output in shards the reader -- one
    chain of refills: stage, inflate on the device, copy back, walk -- is what the run waits for (390 k reads/s alone on the bench sample).
    out_shards > 1: the lifted records go into that many files (`out_path` with .0, .1, ... in front of its extension), one writer thread
    each, a window's records to whichever writer is free -- the reference's output order is unspecified (docs/user_guide.md:227-230), so the
    shards' union is the output (`samtools cat` joins them).  Buffered writes into ONE file are serialised by its inode lock (9.5 GB/s
    from any number of threads on the GPU box, 61-126 GB/s into a file per thread: tools/write_bench.cpp), which is what bounds the
    one-file pipeline at ~320 k reads/s of 15 kb HiFi records."""
    # threads inside the stages (inflate / batch construction, record assembly per worker, BGZF output).  The stages run at the same
    # time: half of io_threads each by default (tools/bench_e2e_threads.py on the 16-core GPU box, best of three runs: 80.6-81.4 k reads/s
    # with 8 / 4-8 / 6-8 threads against 77.1 k with 16 each; the input and output stages are bound by the page cache either way)
    half = max(2, io_threads // 2)
    read_threads = read_threads or half
    build_threads = build_threads or half
    write_threads = write_threads or half
    st = PipelineStats()
    ixd = index_data.to_desc()
    n_readers = max(1, int(n_readers))
    dev_arg = (index.device if device_inflate else (-1 if device_inflate is False else None))
    if n_readers == 1:
        rds = [bam.BamReader(in_path, read_threads, device_inflate=dev_arg, part=part, n_parts=n_parts)]
    else:
        p0, np0 = (part or 0), max(1, n_parts)
        rds = [bam.BamReader(in_path, max(1, read_threads // n_readers), device_inflate=dev_arg, part=p0 * n_readers + i, n_parts=np0 * n_readers) for i in range(n_readers)]
    rd = rds[0]
    if list(rd.ref_names) != list(contig_names):
        raise ValueError("the read->contig BAM's @SQ list differs from the contig names of the index")
    out_shards = max(1, int(out_shards))
    if out_shards == 1:
        out_paths = [out_path]
    else:
        stem, ext = os.path.splitext(out_path)
        out_paths = [f"{stem}.{k}{ext}" for k in range(out_shards)]
    st.out_paths = list(out_paths)
    hdr_out = bam.output_header(ref_names, ref_lens, cmdline=cmdline)
    wrs = [bam.BamWriter(p_, hdr_out, ref_names, ref_lens, level=level, n_threads=max(1, write_threads // out_shards)) for p_ in out_paths]
    un = None
    if unassembled_path:
        un = bam.BamWriter(unassembled_path, bam.output_header(ref_names, ref_lens, cmdline=cmdline), ref_names, ref_lens, level=level,
                           n_threads=max(1, write_threads // 2))
    q_in: "queue.Queue" = queue.Queue(maxsize=2 * n_workers)
    q_out: "queue.Queue" = queue.Queue(maxsize=2 * n_workers)
    lock = threading.Lock()
    abort = threading.Event()  # a stage that fails releases the others instead of leaving them blocked on a full / empty queue
    t0 = time.perf_counter()

    def put(q, item):
        while not abort.is_set():
            try:
                q.put(item, timeout=0.2)
                return
            except queue.Full:
                pass

    def get(q):
        while not abort.is_set():
            try:
                return q.get(timeout=0.2)
            except queue.Empty:
                pass
        return None

    # the input side is two stages: BGZF inflate + record walk (read_window), then the window's batch arrays (batch_desc: CIGARs,
    # bases, qualities gathered into the plo_batch_in layout) -- about half of the reader's time each
    q_wins = [queue.Queue(maxsize=2) for _ in rds]
    chains_left = [len(rds)]  # batcher chains still running: the last one posts the lift workers' sentinels and starts the readers' teardown

    def reader(ci):
        rd, q_win = rds[ci], q_wins[ci]
        try:
            # the first windows are small and double up to window_reads: the stages behind the reader start after milliseconds instead of
            # after a whole window's decode -- a five-stage pipeline over a handful of full windows is mostly ramp otherwise
            n_win = 0
            while True:
                t = time.perf_counter()
                size = window_reads if not ramp else min(window_reads, max(256, window_reads >> max(0, 4 - n_win)))
                n_win += 1
                win = rd.read_window(size)
                if win is None:
                    break
                with lock:
                    st.read_s += time.perf_counter() - t
                put(q_win, win)
        except BaseException as e:  # noqa: BLE001
            st.errors.append(f"reader {ci}: {e!r}")
            abort.set()
        finally:
            st.stage_done_s["reader" if len(rds) == 1 else f"reader {ci}"] = time.perf_counter() - t0
            put(q_win, None)

    def batcher(ci):
        q_win = q_wins[ci]
        try:
            while True:
                win = get(q_win)
                if win is None:
                    break
                t = time.perf_counter()
                if not win.n_records:
                    desc = None
                elif device_finish:
                    desc = win.batch_desc(with_finish=True)  # (plo_batch_in, plo_finish_in), dense bases
                else:
                    desc = win.batch_desc(sparse_margin=sparse_margin, index_desc=ixd if sparse_margin is not None else None)
                with lock:
                    st.batch_s += time.perf_counter() - t
                put(q_in, (win, desc))
        except BaseException as e:  # noqa: BLE001
            st.errors.append(f"batcher {ci}: {e!r}")
            abort.set()
        finally:
            st.stage_done_s["batcher" if len(rds) == 1 else f"batcher {ci}"] = time.perf_counter() - t0
            with lock:
                chains_left[0] -= 1
                last = chains_left[0] == 0
            if last:
                for _ in range(n_workers):
                    put(q_in, None)
                # the readers' teardown (page-locked stream buffers and device buffers: ~0.1 s for a 4 GB input) runs beside the last
                # windows' lifting and writing: no window needs its reader once its batch is built
                closer.start()

    def lifter(k):
        eng = None
        try:
            # the set-up is inside the try: a failure here (no CUDA torch, out of memory, a bad device) must set `abort` and still post
            # the worker's sentinel, or the writer waits for it forever and the run hangs instead of raising
            if device_finish:
                import torch

                from . import devbatch
                dev = torch.device("cuda", index.device)
                tstream = torch.cuda.Stream(device=dev)  # uploads, kernels and downloads of this worker, in order
                eng = api.Engine(index, stream=tstream.cuda_stream)
                sa_in, _sa_keep = devbatch.sa_inputs(ref_names, dev)
                arena = devbatch.PinnedArena() if not os.environ.get("PLO_PIPELINE_PAGEABLE_RESULTS") else None
            else:
                eng = api.Engine(index)
            while True:
                item = get(q_in)
                if item is None:
                    break
                win, desc = item
                rb = None
                if desc is not None:
                    t = time.perf_counter()
                    if device_finish:
                        marks = [("start", t)]
                        with torch.cuda.stream(tstream):
                            up = devbatch.upload_window(desc[0], desc[1], dev)
                            marks.append(("upload (issue)", time.perf_counter()))
                            ddesc = up.batch.desc()
                            out = eng.liftover_batch_dev(ddesc)
                            marks.append(("liftover", time.perf_counter()))
                            eng.compact_output_dev(out)
                            marks.append(("compact", time.perf_counter()))
                            fo = eng.finish_batch_dev(ddesc, up.finish_in())
                            marks.append(("finish", time.perf_counter()))
                            so = eng.sa_segments_dev(sa_in)
                            marks.append(("sa text", time.perf_counter()))
                            if arena is not None:
                                arena.reset()
                            host = devbatch.HostResults(eng, out, fo, so, win.n_records, arena=arena, dev=dev)
                            marks.append(("download", time.perf_counter()))
                        t1 = time.perf_counter()
                        with lock:
                            for (_, a), (name, b) in zip(marks, marks[1:]):
                                st.lift_detail_s[name] = st.lift_detail_s.get(name, 0.0) + (b - a)
                        rb = win.build_records_finished_raw(host.lift, host.fin, host.sa, ixd, contig_names, ref_names, is_target_region, build_threads)
                        with lock:
                            st.finish_device_ms += float(fo.finish_ms) + float(fo.revcomp_ms) + float(so.sa_ms)
                        del up
                    else:
                        lift = eng.liftover_batch_host(desc)
                        t1 = time.perf_counter()
                        rb = win.build_records_raw(lift, ixd, contig_names, ref_names, is_target_region, build_threads)
                    t2 = time.perf_counter()
                    tm = eng.timing()
                    with lock:
                        st.lift_s += t1 - t
                        st.build_s += t2 - t1
                        st.device_ms += tm.total_ms
                        st.reads += win.n_records
                        st.windows += 1
                        st.records_out += int(rb.n_records)
                        st.lifted += int(rb.n_lifted)
                        st.unmapped_copies += int(rb.n_unmapped_copies)
                        st.bytes_out += int(rb.n_bytes)
                put(q_out, (win, rb))
        except BaseException as e:  # noqa: BLE001
            st.errors.append(f"lift worker {k}: {e!r}")
            abort.set()
        finally:
            st.stage_done_s[f"lift worker {k}"] = time.perf_counter() - t0
            if eng is not None:
                eng.close()

    un_lock = threading.Lock()

    def close_reader():
        t = time.perf_counter()
        for r_ in rds:
            r_.close()
        st.stage_done_s["reader closed"] = time.perf_counter() - t0
        st.stage_done_s["reader close took"] = time.perf_counter() - t

    closer = threading.Thread(target=close_reader)

    def writer(k):
        # (one sentinel per writer, posted by the main thread when every lift worker has ended)
        try:
            while not abort.is_set():
                item = get(q_out)
                if item is None:
                    break
                win, rb = item
                t = time.perf_counter()
                if rb is not None and rb.n_bytes:
                    wrs[k].write((rb.bytes, rb.n_bytes))
                ub, nu = win.unmapped_bytes()
                if nu:
                    with un_lock:
                        st.unmapped_passed_through += nu
                        if un is not None:
                            un.write(ub)
                win.close()
                with lock:
                    st.write_s += time.perf_counter() - t
        except BaseException as e:  # noqa: BLE001
            st.errors.append(f"writer {k}: {e!r}")
            abort.set()
        st.stage_done_s["writer" if out_shards == 1 else f"writer {k}"] = time.perf_counter() - t0

    st.stage_done_s["set up"] = time.perf_counter() - t0
    front = ([threading.Thread(target=reader, args=(ci,)) for ci in range(len(rds))] + [threading.Thread(target=batcher, args=(ci,)) for ci in range(len(rds))] +
             [threading.Thread(target=lifter, args=(k,)) for k in range(n_workers)])
    writers = [threading.Thread(target=writer, args=(k,)) for k in range(out_shards)]
    for t in front + writers:
        t.start()
    for t in front:
        t.join()
    for _ in writers:
        put(q_out, None)
    for t in writers:
        t.join()
    for wr in wrs:
        wr.close()
    st.stage_done_s["output closed"] = time.perf_counter() - t0
    if un is not None:
        un.close()
    if closer.ident is not None:  # (started by the last batcher)
        closer.join()
    else:
        for r_ in rds:
            r_.close()
    st.seconds = time.perf_counter() - t0
    st.stage_done_s["all closed"] = st.seconds
    if st.errors:
        raise RuntimeError("; ".join(st.errors))
    return st
