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
    read_s: float = 0.0      # decode + batch construction (reader thread busy time)
    lift_s: float = 0.0      # plo_liftover_batch, summed over workers
    build_s: float = 0.0     # plo_records_build, summed over workers
    write_s: float = 0.0     # BGZF output (writer thread busy time)
    device_ms: float = 0.0   # HIP-event time of the lift calls
    errors: List[str] = field(default_factory=list)


def run_bam_to_bam(in_path: str, out_path: str, index: api.Index, index_data: abi.IndexData, contig_names: Sequence[str],
                   ref_names: Sequence[str], ref_lens: Sequence[int], window_reads: int = 50_000, n_workers: int = 2,
                   io_threads: int = 16, level: int = 0, unassembled_path: Optional[str] = None, is_target_region: bool = False,
                   cmdline: str = "", sparse_margin: Optional[int] = 32, device_inflate: Optional[bool] = True) -> PipelineStats:  # noqa: E501
    """sparse_margin: the windows' read bases go to the device as PLO_SEQ_BAM4_SPARSE (granules within that many bases of an indel;
    the complete bases stay in the window's records for the engine's second look); None = dense bases.  device_inflate: the BGZF
    blocks of the input are inflated on the GPU (leaves the host cores to record assembly and output; falls back to the host without
    a device), None = as the environment says"""
    st = PipelineStats()
    ixd = index_data.to_desc()
    rd = bam.BamReader(in_path, io_threads, device_inflate=(index.device if device_inflate else (-1 if device_inflate is False else None)))
    if list(rd.ref_names) != list(contig_names):
        raise ValueError("the read->contig BAM's @SQ list differs from the contig names of the index")
    wr = bam.BamWriter(out_path, bam.output_header(ref_names, ref_lens, cmdline=cmdline), ref_names, ref_lens, level=level, n_threads=io_threads)
    un = None
    if unassembled_path:
        un = bam.BamWriter(unassembled_path, bam.output_header(ref_names, ref_lens, cmdline=cmdline), ref_names, ref_lens, level=level,
                           n_threads=max(1, io_threads // 2))
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

    def reader():
        try:
            while True:
                t = time.perf_counter()
                win = rd.read_window(window_reads)
                if win is None:
                    break
                desc = win.batch_desc(sparse_margin=sparse_margin) if win.n_records else None
                st.read_s += time.perf_counter() - t
                put(q_in, (win, desc))
        except BaseException as e:  # noqa: BLE001
            st.errors.append(f"reader: {e!r}")
            abort.set()
        finally:
            for _ in range(n_workers):
                put(q_in, None)

    def lifter(k):
        eng = api.Engine(index)
        try:
            while True:
                item = get(q_in)
                if item is None:
                    break
                win, desc = item
                rb = None
                if desc is not None:
                    t = time.perf_counter()
                    lift = eng.liftover_batch_host(desc)
                    t1 = time.perf_counter()
                    rb = win.build_records_raw(lift, ixd, contig_names, ref_names, is_target_region, io_threads)
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
            eng.close()
            put(q_out, None)

    def writer():
        done = 0
        try:
            while done < n_workers and not abort.is_set():
                item = get(q_out)
                if item is None:
                    done += 1
                    continue
                win, rb = item
                t = time.perf_counter()
                if rb is not None and rb.n_bytes:
                    wr.write((rb.bytes, rb.n_bytes))
                ub, nu = win.unmapped_bytes()
                if nu:
                    st.unmapped_passed_through += nu
                    if un is not None:
                        un.write(ub)
                win.close()
                st.write_s += time.perf_counter() - t
        except BaseException as e:  # noqa: BLE001
            st.errors.append(f"writer: {e!r}")
            abort.set()

    threads = [threading.Thread(target=reader), threading.Thread(target=writer)] + [threading.Thread(target=lifter, args=(k,)) for k in range(n_workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wr.close()
    if un is not None:
        un.close()
    rd.close()
    st.seconds = time.perf_counter() - t0
    if st.errors:
        raise RuntimeError("; ".join(st.errors))
    return st
