"""BAM in -> lifted BAM out with the ORACLE lifting on the host cores: the CPU counterpart of portello_amd/pipeline.py's
run_bam_to_bam, for bench.py's `cpu_baseline.end_to_end` figure (missing #3 / next #6a).

TEST / BENCH INFRASTRUCTURE ONLY (never imported by portello_amd/).  The reference's phase-2 loop is read -> lift -> clone -> write
under a mutex (src/read_alignment_scanner.rs:393-488, 566-661) on N rayon workers; here the SAME reader, batch builder, record
builder and BGZF writer stages as the GPU pipeline (the host C++ behind include/portello_bam.h), with orc_liftover_batch
(oracle/portello_oracle.c, `lift_threads` threads) in place of the engine.  Stages overlap as in the GPU pipeline.
"""
from __future__ import annotations

import ctypes as C
import queue
import threading
import time
from typing import Optional, Sequence

from portello_amd import abi, bam

from . import pyoracle


def run_bam_to_bam_cpu(in_path: str, out_path: str, index_data: abi.IndexData, contig_names: Sequence[str], ref_names: Sequence[str],
                       ref_lens: Sequence[int], window_reads: int = 7500, io_threads: int = 16, lift_threads: int = 16, level: int = 0,
                       unassembled_path: Optional[str] = None, max_reads: Optional[int] = None) -> dict:
    L = pyoracle.lib()
    ixd = index_data.to_desc()
    half = max(2, io_threads // 2)
    rd = bam.BamReader(in_path, half, device_inflate=-1)  # host inflate
    wr = bam.BamWriter(out_path, bam.output_header(ref_names, ref_lens), ref_names, ref_lens, level=level, n_threads=half)
    un = bam.BamWriter(unassembled_path, bam.output_header(ref_names, ref_lens), ref_names, ref_lens, level=level, n_threads=2) if unassembled_path else None
    st = {"reads": 0, "records_out": 0, "read_s": 0.0, "batch_s": 0.0, "lift_s": 0.0, "build_s": 0.0, "write_s": 0.0, "bytes_out": 0, "windows": 0}
    errors = []
    abort = threading.Event()
    q_win: "queue.Queue" = queue.Queue(maxsize=2)
    q_in: "queue.Queue" = queue.Queue(maxsize=2)
    q_out: "queue.Queue" = queue.Queue(maxsize=2)

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
            seen, n_win = 0, 0
            while max_reads is None or seen < max_reads:
                t = time.perf_counter()
                size = min(window_reads, max(256, window_reads >> max(0, 4 - n_win)))  # the GPU pipeline's ramp of window sizes
                n_win += 1
                win = rd.read_window(size)
                if win is None:
                    break
                seen += win.n_records
                st["read_s"] += time.perf_counter() - t
                put(q_win, win)
        except BaseException as e:  # noqa: BLE001
            errors.append(f"reader: {e!r}")
            abort.set()
        finally:
            put(q_win, None)

    def batcher():
        try:
            while True:
                win = get(q_win)
                if win is None:
                    break
                t = time.perf_counter()
                desc = win.batch_desc() if win.n_records else None  # dense bases: the oracle compares whole reads
                st["batch_s"] += time.perf_counter() - t
                put(q_in, (win, desc))
        except BaseException as e:  # noqa: BLE001
            errors.append(f"batcher: {e!r}")
            abort.set()
        finally:
            put(q_in, None)

    def lifter():
        try:
            while True:
                item = get(q_in)
                if item is None:
                    break
                win, desc = item
                rb = None
                if desc is not None:
                    t = time.perf_counter()
                    out = abi.PloBatchOut()
                    if L.orc_liftover_batch(C.byref(ixd), C.byref(desc), abi.STAGES_ALL, lift_threads, C.byref(out)) != 0:
                        raise RuntimeError("orc_liftover_batch failed")
                    t1 = time.perf_counter()
                    rb = win.build_records_raw(out, ixd, contig_names, ref_names, False, half)
                    L.orc_batch_free(C.byref(out))
                    t2 = time.perf_counter()
                    st["lift_s"] += t1 - t
                    st["build_s"] += t2 - t1
                    st["reads"] += win.n_records
                    st["windows"] += 1
                    st["records_out"] += int(rb.n_records)
                    st["bytes_out"] += int(rb.n_bytes)
                put(q_out, (win, rb))
        except BaseException as e:  # noqa: BLE001
            errors.append(f"lifter: {e!r}")
            abort.set()
        finally:
            put(q_out, None)

    def writer():
        try:
            while not abort.is_set():
                item = get(q_out)
                if item is None:
                    break
                win, rb = item
                t = time.perf_counter()
                if rb is not None and rb.n_bytes:
                    wr.write((rb.bytes, rb.n_bytes))
                ub, nu = win.unmapped_bytes()
                if nu and un is not None:
                    un.write(ub)
                win.close()
                st["write_s"] += time.perf_counter() - t
        except BaseException as e:  # noqa: BLE001
            errors.append(f"writer: {e!r}")
            abort.set()

    t0 = time.perf_counter()
    ths = [threading.Thread(target=f) for f in (reader, batcher, lifter, writer)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wr.close()
    if un is not None:
        un.close()
    rd.close()
    st["seconds"] = time.perf_counter() - t0
    if errors:
        raise RuntimeError("; ".join(errors))
    return st
