#!/usr/bin/env python3
"""bench.py -- lifted HiFi reads/sec of the MI355X liftover engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload wgs30x|chr20|stress|plumbing] [--reads R] [--scaling strong|weak]

One *step* = one pass of the hot path (plo_liftover_batch_dev: item enumeration, strand preparation, left-shift,
liftover, length check, simplify) over one batch of synthetic reads that is already resident in HBM when the timed
region starts.

N > 1: one rank per GPU.  Under a launcher (torch.distributed.run sets WORLD_SIZE / RANK / LOCAL_RANK) bench.py is one rank of it;
WITHOUT one, `python bench.py --gpus N` starts the N ranks itself as a child `python -m torch.distributed.run` (launch_plan) and
relays rank 0's line and the exit code; a WORLD_SIZE that disagrees with --gpus is an error.  `--dist-backend gloo` runs the same
shard -> lift -> gather -> verify with host-tensor payloads (one-GPU boxes, where RCCL refuses two ranks on a device).
Default `--scaling strong` = BASELINE.json configs[3]: ONE
read set (same seed on every rank) is cut into the reference's own windows (<= 20 Mb of a contig,
src/read_alignment_scanner.rs:508), the windows are dealt to the ranks balanced by their input CIGAR ops
(portello_amd/shard.py), every rank lifts its windows with no data-path collective, and the compact result records are
gathered to rank 0 over RCCL (peer -> root send/recv, the path's only exchange step).  After the timed region rank 0 checks
that the gathered records equal its own single-GPU result of the whole read set, bit for bit.  `--scaling weak` gives every
rank its own read set of the full size instead.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
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

from portello_amd import abi, api, devbatch, shard, synth  # noqa: E402
from portello_amd import build as plo_build  # noqa: E402
from portello_amd import gather as plo_gather  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBS = 6300.0  # measured copy ceiling (same guide)
_FIELDS = ("item_seg", "item_cseg", "item_status", "item_need_flipped", "item_mapq", "item_chrom_index", "item_ref_pos",
           "item_cigar_off", "item_cigar_len", "cigar")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(w, got, budget_s: float = 12.0, ixd=None):
    """Times the oracle (CPU restatement of the reference algorithm: sorted-array block maps + bisection instead of the
    reference's BTreeMap, read segments split evenly over pthreads instead of rayon tasks per 20 Mb window) on a bounded sample
    of the same workload -- blocks of consecutive reads spread evenly over the whole coordinate-sorted read set -- and checks the
    GPU result of those reads against it.  Rank 0, N = 1 only."""
    from oracle import pyoracle

    pyoracle.build()
    # threads = the cores this process may actually use (the GPU boxes report 256 logical CPUs under a cgroup quota of 16)
    cores = pipeline_cpus()
    ixd = w.index_data() if ixd is None else ixd  # host copy of the index
    n_reads = w.n_reads
    # calibrate on a small slice, then size the sample for ~budget_s of wall time on all cores
    probe = min(n_reads, 2000)
    b = w.batch_data(0, probe)
    t0 = time.perf_counter()
    pyoracle.liftover_batch(ixd, b, abi.STAGES_ALL, cores)
    rate = probe / max(1e-4, time.perf_counter() - t0)
    sample = int(min(n_reads, 300_000, max(probe, rate * budget_s)))
    n_blocks = 24 if n_reads >= 24 * 64 else 1
    block = max(1, sample // n_blocks)
    stride = n_reads // n_blocks
    dt = 0.0
    n_done = n_items = 0
    ok = True
    for k in range(n_blocks):
        lo = k * stride
        hi = min(n_reads, lo + block)
        b = w.batch_data(lo, hi)
        t0 = time.perf_counter()
        ref = pyoracle.liftover_batch(ixd, b, abi.STAGES_ALL, cores)
        dt += time.perf_counter() - t0
        n_done += hi - lo
        # parity of the block: GPU items of its read segments vs the oracle (item order is (segment, contig segment))
        s0 = int(torch.searchsorted(w.seg_read, torch.tensor(lo, device=w.device)).item())
        s1 = int(torch.searchsorted(w.seg_read, torch.tensor(hi, device=w.device)).item())
        keep = (got.item_seg >= s0) & (got.item_seg < s1)
        d = {f: (getattr(got, f)[keep] if f != "cigar" else got.cigar) for f in _FIELDS}
        d["item_seg"] = (d["item_seg"] - s0).astype(np.uint32)
        ok = ok and abi.BatchResult(**d).canonical() == ref.canonical()
        n_items += int(keep.sum())
    t0 = time.perf_counter()
    pyoracle.liftover_batch(ixd, w.batch_data(0, probe), abi.STAGES_ALL, 1)
    rate1 = probe / max(1e-4, time.perf_counter() - t0)
    return {"value": n_done / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "block_map": "array-based (sorted key / value arrays + bisection): no slower than the reference's BTreeMap -- the baseline errs on the conservative side; "
                         "the look-ups are a few per cent of its time either way (three bisections per alignment-match op against ~60 us per read; nine look-ups "
                         "instead of one did not move the single-thread time beyond run-to-run noise, DESIGN section 6)",
            "sample": f"{n_blocks} blocks of {block} consecutive reads spread evenly over the read set ({n_done} reads), "
                      f"oracle/liboracle.so = C restatement of the reference algorithm, not the reference binary (sorted-array block "
                      f"maps with bisection where the reference has a BTreeMap; segments split evenly over {cores} pthreads where the "
                      f"reference spawns rayon tasks per 20 Mb window); 1 thread: {rate1:.0f} reads/s",
            "single_thread_value": rate1, "seconds": dt, "host_cpus_reported": os.cpu_count()}, ok, n_items


def pipeline_cpus() -> int:
    from portello_amd import pipeline

    return pipeline.effective_cpus()



def host_rates(threads: int, tmpdir: str = None) -> dict:
    """what the box's host cores move per second, beside the pipeline's per-stage rates: memcpy and CRC-32 (zlib's, and libdeflate's
    carry-less-multiply one the BGZF reader / writer use when the runtime is there), one thread and `threads` threads"""
    import ctypes
    import threading
    import zlib

    import numpy as np

    n = 256 << 20
    src = np.full(n, 7, dtype=np.uint8)
    dst = np.empty_like(src)
    libc = ctypes.CDLL(None)
    libc.memcpy.restype = ctypes.c_void_p
    libc.memcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    ld = None
    for name in ("libdeflate.so.0", "libdeflate.so", "libdeflate.so.1"):
        try:
            ld = ctypes.CDLL(name)
            ld.libdeflate_crc32.restype = ctypes.c_uint32
            ld.libdeflate_crc32.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
            break
        except OSError:
            ld = None

    def run(fn, nt):
        part = n // nt
        ths = [threading.Thread(target=fn, args=(k * part, part)) for k in range(nt)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return n / (time.perf_counter() - t0) / 1e9

    cp = lambda o, m: libc.memcpy(dst.ctypes.data + o, src.ctypes.data + o, m)
    crc_z = lambda o, m: zlib.crc32(memoryview(src)[o:o + m])
    out = {"threads": threads, "buffer_MB": n >> 20}
    cp(0, n)  # touch the pages
    out["memcpy_GBps_1_thread"] = max(run(cp, 1) for _ in range(2))
    out["memcpy_GBps_all_threads"] = max(run(cp, threads) for _ in range(2))
    out["crc32_zlib_GBps_1_thread"] = run(crc_z, 1)
    if ld is not None:
        crc_l = lambda o, m: ld.libdeflate_crc32(0, src.ctypes.data + o, m)
        out["crc32_libdeflate_GBps_1_thread"] = run(crc_l, 1)
        out["crc32_libdeflate_GBps_all_threads"] = run(crc_l, threads)
    if tmpdir:  # what the output file's page cache takes: positional writes of 8 MB pieces from `threads` threads (the BGZF writer's pattern)
        path = os.path.join(tmpdir, "rate_probe.bin")
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            mv = memoryview(src)

            def wr(o, m):
                p = o
                while p < o + m:
                    k = min(8 << 20, o + m - p)
                    os.pwrite(fd, mv[p:p + k], p)
                    p += k

            out["page_cache_pwrite_GBps_all_threads"] = max(run(wr, threads) for _ in range(2))
        finally:
            os.close(fd)
            os.unlink(path)
    return out


def end_to_end(w, index, ixd, sample_reads: int, window_reads: int, n_workers: int, io_threads: int, pcie_reads: int = 60000, verify: bool = True):
    """BAM file in -> lifted BAM file out on a bounded sample of the workload (a block of consecutive reads from the middle of
    the coordinate-sorted read set, written as a real BGZF-compressed read->contig BAM first): BGZF inflate + record parsing +
    batch construction -> page-locked H2D -> lift kernels -> D2H -> BAM record assembly (tags, flags, reversed seq/qual) ->
    BGZF output (level 0, what the reference uses for stdout).  Also times the host-buffer entry point alone on the same
    windows (`pcie_inclusive`: H2D + kernels + D2H, no BAM work)."""
    import shutil
    import tempfile

    from portello_amd import bam, bamsynth, pipeline

    n = min(sample_reads, w.n_reads)
    lo = max(0, (w.n_reads - n) // 2)
    d = tempfile.mkdtemp(prefix="plo_e2e_")
    try:
        inp, outp = os.path.join(d, "reads.bam"), os.path.join(d, "lifted.bam")
        t0 = time.perf_counter()
        meta = bamsynth.write_read_bam(w, inp, lo, lo + n, level=1, n_threads=io_threads)
        t_write = time.perf_counter() - t0
        cn, rn = meta["contig_names"], bamsynth.ref_names(w)
        rl = [int(s.numel()) for s in w.chrom_seq]
        in_bytes = os.path.getsize(inp)
        # warm-up into a file of its own, removed before the timed runs: every timed run creates its output.  (Rounds 3-5 rewrote the warm-up's
        # path: O_TRUNC on a file with gigabytes of dirty pages, then a forced flush of the rewritten file inside close() -- 0.55 s of a 1.5 s
        # run on the GPU box's overlay / ext4 root, tools/e2e_stages.py -- which a user writing a new file never pays.)
        warm = os.path.join(d, "warm.bam")
        pipeline.run_bam_to_bam(inp, warm, index, ixd, cn, rn, rl, window_reads=min(window_reads, 2000), n_workers=1, io_threads=io_threads)
        os.unlink(warm)
        st = pipeline.run_bam_to_bam(inp, outp, index, ixd, cn, rn, rl, window_reads=window_reads, n_workers=n_workers, io_threads=io_threads,
                                     unassembled_path=os.path.join(d, "unassembled.bam"))
        e2e = {"value": st.reads / st.seconds, "unit": "reads/s", "reads": st.reads, "seconds": st.seconds, "windows": st.windows,
               "window_reads": window_reads, "lift_workers": n_workers, "io_threads": io_threads, "host_cpus_reported": os.cpu_count(),
               "host_cpus_effective": pipeline_cpus(), "records_out": st.records_out,
               "lifted_records": st.lifted, "unmapped_copies": st.unmapped_copies, "unmapped_passed_through": st.unmapped_passed_through,
               "input_bam_MB": in_bytes / 1e6, "output_MB": st.bytes_out / 1e6,
               "stage_busy_s": {"decode": st.read_s, "batch construction": st.batch_s, "lift (H2D+kernels+D2H), summed over workers": st.lift_s,
                                "record assembly, summed over workers": st.build_s, "bgzf write": st.write_s, "device (HIP events)": st.device_ms / 1e3},
               "sample": f"reads [{lo}, {lo + n}) of the workload as a BGZF level-1 read->contig BAM (synthetic qualities / aux tags, "
                         f"written in {t_write:.1f} s outside the timed run); output BGZF level 0",
               "note": "supplementary: one GPU, one node's host cores; `value` of the bench line stays the HBM-resident kernel rate"}
        e2e["stage_GBps"] = {"decode: compressed BAM in": in_bytes / 1e9 / max(st.read_s, 1e-9), "record assembly: record bytes out": st.bytes_out / 1e9 / max(st.build_s, 1e-9),
                             "bgzf write: record bytes out": st.bytes_out / 1e9 / max(st.write_s, 1e-9), "whole run: bytes in + out": (in_bytes + st.bytes_out) / 1e9 / st.seconds}
        try:
            e2e["host_rates"] = host_rates(pipeline_cpus(), d)
        except Exception as e:  # noqa: BLE001
            e2e["host_rates"] = {"error": repr(e)}
        # the same run with the records finished on the device (flags, bin, primary, reversed bases / qualities: plo_finish_batch_dev;
        # SA text: plo_sa_segments_dev; the host only copies them into place, plo_records_build_finished): dense bases go up, the
        # reversed ones come back
        outp_dev = os.path.join(d, "lifted_device_finished.bam")
        try:
            sd = pipeline.run_bam_to_bam(inp, outp_dev, index, ixd, cn, rn, rl, window_reads=window_reads, n_workers=n_workers, io_threads=io_threads,
                                         unassembled_path=os.path.join(d, "unassembled_dev.bam"), device_finish=True)
            e2e["device_finished"] = {"value": sd.reads / sd.seconds, "unit": "reads/s", "seconds": sd.seconds, "records_out": sd.records_out,
                                      "output_MB": sd.bytes_out / 1e6,
                                      "stage_busy_s": {"decode": sd.read_s, "batch construction": sd.batch_s, "lift + finish (H2D, kernels, D2H), summed over workers": sd.lift_s,
                                                       "record assembly, summed over workers": sd.build_s, "bgzf write": sd.write_s,
                                                       "device (HIP events): lift": sd.device_ms / 1e3, "device (HIP events): finish + revcomp + SA": sd.finish_device_ms / 1e3},
                                      "lift_stage_steps_s": dict(sd.lift_detail_s),
                                      "same_records_out_and_bytes": bool(sd.records_out == st.records_out and sd.bytes_out == st.bytes_out)}
        except Exception as e:  # noqa: BLE001
            log(f"[bench] end_to_end with device finishing failed: {e!r}")
            e2e["device_finished"] = {"value": None, "error": repr(e)}
        # ... and with one output shard per writer thread ("each rank writes its own shard"; the reference's output order is
        # unspecified): what the ONE-file run is bound by is the file's inode lock, not the device or the record assembly
        shards_paths = None
        try:
            n_sh = max(2, min(8, io_threads // 4))
            ss = pipeline.run_bam_to_bam(inp, os.path.join(d, "sharded.bam"), index, ixd, cn, rn, rl, window_reads=window_reads, n_workers=max(n_workers, 3), io_threads=io_threads,
                                         unassembled_path=os.path.join(d, "unassembled_sh.bam"), device_finish=True, out_shards=n_sh)
            shards_paths = list(ss.out_paths)
            e2e["output_shards"] = {"value": ss.reads / ss.seconds, "unit": "reads/s", "seconds": ss.seconds, "shards": n_sh, "lift_workers": max(n_workers, 3),
                                    "records_out": ss.records_out, "output_MB": ss.bytes_out / 1e6,
                                    "stage_busy_s": {"decode": ss.read_s, "batch construction": ss.batch_s, "lift + finish, summed over workers": ss.lift_s,
                                                     "record assembly, summed over workers": ss.build_s, "bgzf write, summed over writers": ss.write_s},
                                    "stage_done_s": {k_: round(v_, 3) for k_, v_ in ss.stage_done_s.items()},
                                    "same_records_out_and_bytes": bool(ss.records_out == st.records_out and ss.bytes_out == st.bytes_out),
                                    "note": "device-finished records into one BGZF file per writer thread; the shards' union is the output (samtools cat joins them).  (Two reader chains -- pipeline n_readers=2 -- do not help on 16 host cores: 279 k against 301 k reads/s, the lift workers' record assembly becomes the tail)"}
        except Exception as e:  # noqa: BLE001
            log(f"[bench] end_to_end with output shards failed: {e!r}")
            e2e["output_shards"] = {"value": None, "error": repr(e)}
        if verify:
            # after the timed run: the written BAM re-read with the independent reader and compared, record for record, with the
            # expectation (oracle alignments + the Python restatement of the record logic) for a strided sample of >= 5 000 reads
            try:
                from oracle import expect

                every = max(1, (n // 500) // 12)
                t0 = time.perf_counter()
                v = expect.verify_lifted_bam(inp, outp, ixd, cn, rn, window=500, every=every, threads=min(16, io_threads),
                                             unassembled_bam=os.path.join(d, "unassembled.bam"))
                v["seconds"] = time.perf_counter() - t0
                e2e["records_verified"] = v["records_verified"] if v["ok"] else 0
                e2e["verification"] = v
                if e2e.get("device_finished", {}).get("value"):
                    vd = expect.verify_lifted_bam(inp, outp_dev, ixd, cn, rn, window=500, every=every, threads=min(16, io_threads),
                                                  unassembled_bam=os.path.join(d, "unassembled_dev.bam"))
                    e2e["device_finished"]["records_verified"] = vd["records_verified"] if vd["ok"] else 0
                    if not vd["ok"]:
                        log("[bench] END-TO-END VERIFICATION FAILURE (device-finished records)")
                if shards_paths and (e2e.get("output_shards") or {}).get("value"):
                    vs = expect.verify_lifted_bam(inp, shards_paths, ixd, cn, rn, window=500, every=every, threads=min(16, io_threads),
                                                  unassembled_bam=os.path.join(d, "unassembled_sh.bam"))
                    e2e["output_shards"]["records_verified"] = vs["records_verified"] if vs["ok"] else 0
                    if not vs["ok"]:
                        log("[bench] END-TO-END VERIFICATION FAILURE (output shards)")
                if not v["ok"]:
                    log("[bench] END-TO-END VERIFICATION FAILURE: the written BAM differs from the expected records")
            except Exception as e:  # noqa: BLE001
                log(f"[bench] end_to_end verification could not run: {e!r}")
                e2e["records_verified"] = None
            # the CPU pipeline beside it (cpu_baseline leg only): the same reader / batch / record / writer stages with the oracle lifting on
            # the host cores, same input file, same window size, all of the box's cores for the lift (oracle/cpu_pipeline.py)
            try:
                from oracle import cpu_pipeline, expect

                outp_cpu = os.path.join(d, "lifted_cpu.bam")
                cores = pipeline_cpus()
                cs = cpu_pipeline.run_bam_to_bam_cpu(inp, outp_cpu, ixd, cn, rn, rl, window_reads=window_reads, io_threads=io_threads, lift_threads=cores,
                                                     unassembled_path=os.path.join(d, "unassembled_cpu.bam"))
                vc = expect.verify_lifted_bam(inp, outp_cpu, ixd, cn, rn, window=500, every=max(1, (n // 500) // 12), threads=min(16, io_threads),
                                              unassembled_bam=os.path.join(d, "unassembled_cpu.bam"))
                e2e["cpu_pipeline"] = {"value": cs["reads"] / cs["seconds"], "unit": "reads/s", "cores": cores, "kind": "port", "seconds": cs["seconds"],
                                       "reads": cs["reads"], "records_out": cs["records_out"], "records_verified": vc["records_verified"] if vc["ok"] else 0,
                                       "stage_busy_s": {"decode": cs["read_s"], "batch construction": cs["batch_s"], f"lift (oracle, {cores} threads)": cs["lift_s"],
                                                        "record assembly": cs["build_s"], "bgzf write": cs["write_s"]},
                                       "note": "BAM -> BAM with the CPU restatement of the reference algorithm lifting on the host cores (not the reference "
                                               "binary): same input file, windows, reader, record builder and BGZF writer as the GPU pipeline"}
            except Exception as e:  # noqa: BLE001
                log(f"[bench] cpu pipeline baseline failed: {e!r}")
                e2e["cpu_pipeline"] = None
        # the host-buffer entry point alone: whole sample in one window, dense bases and sparse bases (margin 32)
        rd = bam.BamReader(inp, io_threads)
        eng = api.Engine(index)
        wins = []
        win = rd.read_window(pcie_reads)
        pcie = None
        ixd_c = ixd.to_desc()  # (read segments that touch no reverse-mapped contig segment send their insertions' bases only)
        if win is not None and win.n_records:
            wins.append((win, None))

            def h2d_bytes(b0):
                return (int(b0.seq_bytes) + 4 * int(np.ctypeslib.as_array(b0.seg_cigar_off, shape=(int(b0.n_segs) + 1,))[-1]) + 25 * int(b0.n_segs)
                        + 13 * int(b0.n_reads))

            def run(sparse_margin):
                tb = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    desc = win.batch_desc(sparse_margin=sparse_margin, index_desc=(ixd_c if sparse_margin is not None else None))
                    tb.append(time.perf_counter() - t0)
                eng.liftover_batch_host(desc)
                reps = 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    eng.liftover_batch_host(desc)
                dt = (time.perf_counter() - t0) / reps
                t = eng.timing()
                return {"value": win.n_records / dt, "unit": "reads/s", "ms_per_call": dt * 1e3, "h2d_MB_per_call": h2d_bytes(desc) / 1e6,
                        "device_ms_per_call": t.total_ms, "second_look_items_per_call": int(t.n_miss_items), "second_look_ms": t.miss_ms,
                        "batch_build_ms": min(tb) * 1e3}

            dense = run(None)
            sparse = run(32)
            # two callers, one context each (the arrangement of INTEGRATION.md): one's transfers run under the other's kernels
            two = None
            try:
                import threading as _th
                desc2 = win.batch_desc(sparse_margin=32, index_desc=ixd_c)  # (the window's arrays as the last run(32) left them)
                eng2 = api.Engine(index)
                eng2.liftover_batch_host(desc2)
                reps2 = 6
                def _loop(e_):
                    for _ in range(reps2):
                        e_.liftover_batch_host(desc2)
                ths = [_th.Thread(target=_loop, args=(e_,)) for e_ in (eng, eng2)]
                t0 = time.perf_counter()
                for t_ in ths:
                    t_.start()
                for t_ in ths:
                    t_.join()
                dt2 = time.perf_counter() - t0
                two = {"value": 2 * reps2 * win.n_records / dt2, "unit": "reads/s", "ms_per_call_per_worker": dt2 / reps2 * 1e3,
                       "note": "two callers share the one PCIe link and the runtime's copy queue: depending on how their transfers "
                               "interleave a second context changes the rate by -25 % to +45 % from run to run (29-53 M reads/s measured); "
                               "the steady gain of several contexts is with device-resident batches (`overlap`)"}
                eng2.close()
            except Exception as e:  # noqa: BLE001
                log(f"[bench] two-worker host-buffer measurement failed: {e!r}")
            pcie = dict(sparse)
            pcie.update({"two_workers": two, "reads_per_call": win.n_records, "seq_fmt": "bam4_sparse (granules of 32 bases within 32 bases of a read->contig indel; read segments that touch no reverse-mapped contig segment: "
                                                                                                  "the granules of their insertions only, plo_bam_window_batch_sparse_strand)",
                         "dense_bases": dense,
                         "note": "plo_liftover_batch on page-locked host arrays, one context, synchronous: H2D of read bases + CIGARs, kernels, "
                                 "second look at items whose comparisons left the granules sent, D2H of the dense result.  batch_build_ms = "
                                 "plo_bam_window_batch(_sparse) on the decoded window (not in `value`; part of the decode stage of end_to_end)"})
        for win, _ in wins:
            win.close()
        eng.close()
        rd.close()
        # the faster of the two verified runs is the one reported on top (normally the device-finished one); the other stays beside it
        dfin = e2e.get("device_finished") or {}
        e2e["finishing"] = "host (plo_records_build)"
        if dfin.get("value") and dfin["value"] > e2e["value"] and (not verify or dfin.get("records_verified")):
            host = {k: e2e[k] for k in ("value", "seconds", "stage_busy_s", "records_verified") if k in e2e}
            for k in ("value", "seconds", "stage_busy_s", "records_verified"):
                if k in dfin:
                    e2e[k] = dfin[k]
            e2e["finishing"] = "device (plo_finish_batch_dev + plo_sa_segments_dev, copied into place by plo_records_build_finished)"
            e2e["host_finished"] = host
            del e2e["device_finished"]
        return e2e, pcie
    finally:
        shutil.rmtree(d, ignore_errors=True)


def end_to_end_sharded(w, index, sample_reads: int, window_reads: int, n_workers: int, dist, rank: int, world: int, verify: bool = True):
    """N > 1: ONE read->contig BAM lifted by all ranks -- BASELINE configs[3] as a BAM run.  Rank 0 writes the sample as a BGZF level-1 BAM
    on the node's file system; every rank opens ITS PART of it (plo_bam_open_range: a split by compressed offset, first block and first
    record found without an index -- the reference gives every worker an IndexedReader, src/worker_thread_data.rs:21-30), runs the whole
    pipeline over it (inflate -> batches -> lift + finish on its GPU -> record bytes -> BGZF level 0) and writes its own output shard; no
    collective on the data path.  Timed between two barriers, max over ranks.  Rank 0 then checks the shards' union against the expectation.
    Every collective below is reached by every rank whatever happens to its own work (a rank that fails says so in the next reduction)."""
    import shutil
    import tempfile

    from portello_amd import bamsynth, pipeline

    on_gpu = dist.get_backend() == "nccl"

    def reduce(vals, op):
        t = torch.tensor(vals, dtype=torch.float64)
        if on_gpu:
            t = t.cuda()
        dist.all_reduce(t, op=op)
        return [float(x) for x in t.cpu()]

    n = min(sample_reads, w.n_reads)
    lo = max(0, (w.n_reads - n) // 2)
    box = [None]
    if rank == 0:
        try:
            box[0] = tempfile.mkdtemp(prefix="plo_e2e_sharded_")
        except Exception as e:  # noqa: BLE001
            log(f"[bench] end_to_end_sharded: no temporary directory: {e!r}")
    dist.broadcast_object_list(box, src=0)
    d = box[0]
    io_threads = max(2, min(64, pipeline_cpus()) // world)
    ok, ixd, cn, rn, rl, inp = d is not None, None, None, None, None, None
    try:
        if ok:
            inp = os.path.join(d, "reads.bam")
            ixd = w.index_data()
            cn, rn = bamsynth.contig_names(w), bamsynth.ref_names(w)
            rl = [int(s_.numel()) for s_ in w.chrom_seq]
            if rank == 0:
                bamsynth.write_read_bam(w, inp, lo, lo + n, level=1, n_threads=max(2, min(64, pipeline_cpus())))
    except Exception as e:  # noqa: BLE001
        log(f"[bench] end_to_end_sharded set-up failed on rank {rank}: {e!r}")
        ok = False
    ok = reduce([1.0 if ok else 0.0], dist.ReduceOp.MIN)[0] > 0.5  # (also the barrier behind rank 0's file)
    res = None
    st = None
    if ok:
        outp, unp = os.path.join(d, f"lifted.{rank}.bam"), os.path.join(d, f"unassembled.{rank}.bam")
        kw = dict(window_reads=window_reads, n_workers=n_workers, io_threads=io_threads, part=rank, n_parts=world)
        try:
            warm = os.path.join(d, f"warm.{rank}.bam")  # (a file of its own, removed: the timed run creates its output -- see end_to_end)
            pipeline.run_bam_to_bam(inp, warm, index, ixd, cn, rn, rl, **dict(kw, window_reads=min(window_reads, 2000), n_workers=1))
            os.unlink(warm)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] end_to_end_sharded warm-up failed on rank {rank}: {e!r}")
            ok = False
        dist.barrier()
        t0 = time.perf_counter()
        try:
            if ok:
                st = pipeline.run_bam_to_bam(inp, outp, index, ixd, cn, rn, rl, unassembled_path=unp, **kw)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] end_to_end_sharded run failed on rank {rank}: {e!r}")
            ok = False
        dist.barrier()
        dt = time.perf_counter() - t0
        ok = reduce([1.0 if ok else 0.0], dist.ReduceOp.MIN)[0] > 0.5
        mine = [dt, float(st.reads) if st else 0.0, float(st.records_out) if st else 0.0, float(st.bytes_out) if st else 0.0]
        tmax = reduce(mine, dist.ReduceOp.MAX)
        tsum = reduce(mine, dist.ReduceOp.SUM)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, int(st.reads) if st else 0)
        if rank == 0 and ok:
            res = {"value": tsum[1] / tmax[0], "unit": "reads/s", "reads": int(tsum[1]), "seconds": tmax[0], "ranks": world,
                   "reads_per_rank": per_rank, "records_out": int(tsum[2]), "output_MB": tsum[3] / 1e6, "io_threads_per_rank": io_threads,
                   "input_bam_MB": os.path.getsize(inp) / 1e6,
                   "note": "one input BAM, every rank lifts its part (plo_bam_open_range) and writes its own output shard; no data-path collective"}
            if verify:
                try:
                    from oracle import expect

                    every = max(1, (n // 500) // 12)
                    v = expect.verify_lifted_bam(inp, [os.path.join(d, f"lifted.{r}.bam") for r in range(world)], ixd, cn, rn, window=500, every=every,
                                                 threads=min(16, max(2, pipeline_cpus())), unassembled_bam=[os.path.join(d, f"unassembled.{r}.bam") for r in range(world)])
                    res["records_verified"] = v["records_verified"] if v["ok"] and int(tsum[1]) == n else 0
                    res["verification"] = v
                except Exception as e:  # noqa: BLE001
                    log(f"[bench] end_to_end_sharded verification could not run: {e!r}")
                    res["records_verified"] = None
        elif rank == 0:
            res = {"value": None, "note": "a rank's pipeline failed (see stderr)"}
    dist.barrier()
    if rank == 0 and d is not None:
        shutil.rmtree(d, ignore_errors=True)
    return res


def stream_main(args, cfg, dev, dev_index, chunk_reads):
    """A read set larger than one batch (stress at 2 M reads, wgs30x at 6.2 M): chunks generated on one set of
    contigs, all resident in HBM, lifted batch after batch.  One step = one pass over all batches.  The timed passes use one host
    worker (clean HIP-event kernel times for the roofline object); `overlap` repeats them with two."""
    from portello_amd import stream

    chunks = stream.generate_chunks(synth, cfg.name, cfg.n_reads, chunk_reads, dev, log)
    torch.cuda.synchronize()
    index = api.Index(chunks[0].index_data_device(), device=dev_index)
    dbs = [devbatch.DeviceBatch.from_workload(w) for w in chunks]
    descs = [db.desc() for db in dbs]
    total_reads = sum(w.n_reads for w in chunks)
    times = {k_: [] for k_ in ("lift", "lanes", "enum", "big", "mid", "retry", "heavy")}
    agg = {"items": 0, "in_ops": 0, "out_ops": 0, "algo": 0, "mid_items": 0, "big_items": 0, "lane_items": 0, "retry_items": 0}

    def record(tm):
        times["lift"].append(tm.lift_ms)
        times["lanes"].append(tm.lanes_ms)
        times["enum"].append(tm.enumerate_ms)
        times["big"].append(tm.big_ms)
        times["mid"].append(tm.mid_ms)
        times["retry"].append(tm.retry_ms)
        times["heavy"].append(tm.heavy_lanes_ms)
        agg["items"] += int(tm.n_items)
        agg["in_ops"] += int(tm.n_in_ops)
        agg["out_ops"] += int(tm.n_out_ops)
        agg["mid_items"] += int(tm.n_mid_items)
        agg["big_items"] += int(tm.n_big_items)
        agg["lane_items"] += int(tm.n_lane_items)
        agg["retry_items"] += int(tm.n_retry_items)
        if int(tm.heavy_kernel):
            agg["heavy_kernel"] = int(tm.heavy_kernel)

    # algorithmic bytes: one pass of the counting kernels outside the timed region (the production light-item kernel carries no counters)
    cnt_run = stream.StreamRunner(index, dev, 1)
    for e_ in cnt_run.engines:
        e_.set_stats()

    def record_algo(tm):
        agg["algo"] += int(tm.algo_bytes)

    cnt_run.run(descs, record=record_algo)
    cnt_run.sync()
    for e_ in cnt_run.engines:
        e_.close()
    one = stream.StreamRunner(index, dev, 1)
    for _ in range(max(1, args.warmup)):
        one.run(descs)
    one.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one.run(descs, record=record)
    one.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_calls = args.steps * len(descs)
    kms = {"k_lift_lanes": float(np.sum(times["lanes"])) / n_calls, "k_lift_mid": float(np.sum(times["mid"])) / n_calls,
           "k_lift_tiles": float(np.sum(times["lift"])) / n_calls, "k_lift_big": float(np.sum(times["big"])) / n_calls,
           "k_lift_retry": float(np.sum(times["retry"])) / n_calls, "k_lift_lanes_g": float(np.sum(times["heavy"])) / n_calls}
    dominant = max(kms, key=kms.get)
    dom_ms = kms[dominant]
    share = dom_ms / max(1e-9, sum(kms.values()))
    algo_per_call = agg["algo"] / len(descs)
    achieved = (algo_per_call * share) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    heavy_name = {2: "k_lift_lanes_g_w3", 3: "k_lift_stream"}.get(agg.get("heavy_kernel", 0), "k_lift_lanes_g")  # (the batches' last heavy-item kernel)
    dom_name = {"k_lift_mid": "k_lift_mid<16>", "k_lift_lanes_g": heavy_name}.get(dominant, dominant)
    result = {
        "metric": "lifted HiFi reads/sec (whole node)", "value": total_reads * args.steps / dt, "unit": "reads/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": cfg.name, "reads_total": int(total_reads), "reads_this_rank": int(total_reads), "read_len_mean": cfg.read_len_mean,
                   "streamed": True, "batches": len(descs), "reads_per_batch": [w.n_reads for w in chunks],
                   "items_per_gpu": agg["items"] // args.steps, "in_ops_per_gpu": agg["in_ops"] // args.steps, "out_ops_per_gpu": agg["out_ops"] // args.steps,
                   "large_items_per_gpu": agg["big_items"] // args.steps, "mid_items_per_gpu": agg["mid_items"] // args.steps,
                   "lane_items_per_gpu": agg["lane_items"] // args.steps, "retry_items_per_gpu": agg["retry_items"] // args.steps, "seq_fmt": "bam4",
                   "parallelism": "one read set on one set of contigs, lifted as consecutive batches through one context (one step = one pass over all batches)",
                   "host_workers_per_gpu": 1, "kernel_source_hash": plo_build.source_hash(), "gather": "none"},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "frac_of_copy_ceiling": achieved / HBM_COPY_CEILING_GBS, "traffic": None, "algorithmic_bytes_per_launch": int(algo_per_call * share),
                     "kernel_ms": dom_ms, "launches_per_step": len(descs), "enumerate_ms": float(np.sum(times["enum"])) / n_calls,
                     "lift_lanes_ms": kms["k_lift_lanes"], "lift_tiles_ms": kms["k_lift_tiles"], "lift_big_ms": kms["k_lift_big"],
                     "lift_mid_ms": kms["k_lift_mid"], "lift_retry_ms": kms["k_lift_retry"], "lift_heavy_ms": kms["k_lift_lanes_g"], "heavy_kernel": heavy_name if agg.get("heavy_kernel", 0) else None},
    }
    one.close()
    if args.overlap_workers > 1:
        two = stream.StreamRunner(index, dev, args.overlap_workers)
        two.run(descs)
        two.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            two.run(descs)
        two.sync()
        torch.cuda.synchronize()
        odt = time.perf_counter() - t1
        result["overlap"] = {"host_workers_per_gpu": args.overlap_workers, "value": total_reads * args.steps / odt, "unit": "reads/s",
                             "ms_per_step": odt / args.steps * 1e3,
                             "note": "the same passes with the batches dealt to several host workers (one context + HIP stream each)"}
        two.close()
    if not args.no_cpu_baseline:
        # parity on every batch's strided sample + the CPU baseline on the first batch
        try:
            chk = api.Engine(index)
            ixd_host = chunks[0].index_data()
            ok_all, n_checked, cb = True, 0, None
            for k_, (w_, db_) in enumerate(zip(chunks, dbs)):
                got = devbatch.run_and_download(chk, db_)
                b_, ok, n_items = cpu_baseline(w_, got, budget_s=(15.0 if k_ == 0 else 1.0), ixd=ixd_host)
                cb = cb or b_
                ok_all = ok_all and ok
                n_checked += n_items
            chk.close()
            result["cpu_baseline"] = cb
            result["parity_sample_items"] = n_checked
            result["parity_sample_ok"] = bool(ok_all)
            if not ok_all:
                log("[bench] PARITY FAILURE on the sampled reads")
        except Exception as e:  # noqa: BLE001 -- the baseline must never hide the measurement
            log(f"[bench] cpu_baseline failed: {e!r}")
            result["cpu_baseline"] = None
    print(json.dumps(result), flush=True)
    index.close()



def launch_plan(n_gpus: int, argv, port: int = 0, base_env=None):
    """The N-rank launch of `python bench.py --gpus N ...` when no launcher started it (WORLD_SIZE unset): the command line and the
    environment of ONE child process -- `python -m torch.distributed.run`, one rank per GPU, rendezvous on 127.0.0.1 -- exactly the
    command the driver uses for N > 1.  The reference starts its own workers the same way (a rayon pool of --threads workers inside
    scan_and_remap_reads, src/read_alignment_scanner.rs:606-660, one reader per worker, src/worker_thread_data.rs:21-30).  A child
    process, never os.exec*: this process may already have loaded the HIP runtime."""
    import socket

    if port <= 0:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ if base_env is None else base_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)  # (torch.distributed.run sets them per rank)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["PLO_BENCH_LAUNCHED"] = "1"
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, int(n_gpus)))))
    return cmd, env


def launch_ranks(n_gpus: int, argv) -> int:
    """runs the plan as a child process; rank 0's JSON line is the child's stdout, relayed as it is; returns the child's exit code"""
    import subprocess

    cmd, env = launch_plan(n_gpus, argv)
    log(f"[bench] --gpus {n_gpus} without a launcher: starting {n_gpus} ranks: {' '.join(cmd)}")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in js:
            log(ln)
    if js:
        print(js[-1], flush=True)  # ONE JSON line: rank 0's
    elif p.returncode == 0:
        log("[bench] the ranks printed no result line")
        return 5
    return p.returncode


def finalize(result) -> int:
    """A result whose records differ from the checker's is not a measurement: `parity_sample_ok` false, a failed `end_to_end`
    verification or a gathered record set that differs from the single-GPU one => "parity_failed": true, "value": null (the number
    stays in `value_unverified`) and a non-zero exit code."""
    result.setdefault("value_kind", "hbm_resident_kernel_rate")  # `value` is NOT an end-to-end BAM->BAM rate (that is `end_to_end`)
    why = []
    if result.get("parity_sample_ok") is False:
        why.append("sampled items differ from the oracle")
    e2e = result.get("end_to_end") or {}
    if e2e.get("records_verified") == 0 and "verification" in e2e:
        why.append("end_to_end: the written BAM differs from the expected records")
    if (e2e.get("device_finished") or {}).get("records_verified") == 0:
        why.append("end_to_end (device-finished records): the written BAM differs from the expected records")
    if (e2e.get("output_shards") or {}).get("records_verified") == 0:
        why.append("end_to_end (output shards): the written shards differ from the expected records")
    if (result.get("verify") or {}).get("gathered_equals_single_gpu_result") is False:
        why.append("gathered records differ from the single-GPU result")
    if (result.get("end_to_end_sharded") or {}).get("records_verified") == 0 and "verification" in (result.get("end_to_end_sharded") or {}):
        why.append("end_to_end_sharded: the ranks' output shards differ from the expected records")
    if why:
        result["parity_failed"] = True
        result["parity_failure"] = why
        result["value_unverified"] = result.get("value")
        result["value"] = None
        return 3
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("PLO_BENCH_WORKLOAD", "wgs30x"))
    ap.add_argument("--reads", type=int, default=0, help="override the workload's read count")
    ap.add_argument("--scaling", choices=("strong", "weak"), default=os.environ.get("PLO_BENCH_SCALING", "strong"),
                    help="N > 1: strong = one read set sharded by windows over the ranks (BASELINE configs[3]); weak = every rank "
                         "its own read set of the full size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window-calls", type=int, default=200,
                    help="N = 1: calls of the supplementary `window_50k` measurement (0 = skip: traced / counter runs skip it, its launches "
                         "would enter the per-kernel averages)")
    ap.add_argument("--chunk-reads", type=int, default=0,
                    help="N = 1: read sets of more than this many reads are lifted as a stream of batches (portello_amd/stream.py); default "
                         "250 k for the stress profile (0.5 G input ops per batch; a batch is bounded by 31-bit op indices), 8 M otherwise")
    ap.add_argument("--e2e-reads", type=int, default=int(os.environ.get("PLO_BENCH_E2E_READS", "180000")),
                    help="N = 1: size of the BAM-to-BAM end-to-end sample (0 = skip the end_to_end / pcie_inclusive objects)")
    ap.add_argument("--e2e-window", type=int, default=7500,
                    help="primary records per window of the end-to-end run (a 60 k-read sample is 8 windows: with fewer, larger ones the "
                         "pipeline spends most of the run filling and draining; the default sample of 180 k reads is 24)")
    ap.add_argument("--e2e-workers", type=int, default=2, help="lift worker threads (contexts) of the end-to-end run")
    ap.add_argument("--pipeline-batches", type=int, default=int(os.environ.get("PLO_BENCH_PIPELINE_BATCHES", "4")),
                    help="N > 1, strong scaling: every rank also lifts its windows as this many batches on two contexts, the record gather of "
                         "batch i under the compute of batch i + 1 (`gather_modes.window_pipeline`; 1: off)")
    ap.add_argument("--no-verify", action="store_true", help="strong scaling: skip rank 0's comparison of the gathered records "
                                                             "with its own single-GPU result (after the timed region)")
    ap.add_argument("--workers", type=int, default=int(os.environ.get("PLO_BENCH_WORKERS", "1")),
                    help="host worker threads per GPU, each with its own context and HIP stream (INTEGRATION.md: one plo_ctx per "
                         "rayon worker); batches are dealt to them in turn, so one worker's enumerate pass and host syncs overlap the "
                         "other's tile kernel.  Default 1: the HIP-event kernel times of the roofline object then measure execution "
                         "only (with several streams they include the wait behind the other stream's kernel)")
    ap.add_argument("--overlap-workers", type=int, default=int(os.environ.get("PLO_BENCH_OVERLAP_WORKERS", "2")),
                    help="after the timed region, repeat the same K steps with this many host workers (one context + HIP stream each, the "
                         "arrangement INTEGRATION.md recommends) and report the rate as the supplementary object `overlap` (N = 1 only; "
                         "0/1 = skip: traced runs -- tools/profile_round.sh -- skip it, its launches would enter a rocprofv3 kernel summary "
                         "with their longer, overlapped durations)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default=os.environ.get("PLO_BENCH_DIST_BACKEND", "nccl"),
                    help="N > 1: nccl = RCCL over xGMI (device tensors, the production path); gloo = the same shard -> lift -> gather -> "
                         "verify run with the rank payloads passing through host tensors -- for boxes with ONE GPU, where RCCL refuses two "
                         "ranks on a device (PLO_BENCH_SHARE_GPU=1 lets the ranks share it)")
    args = ap.parse_args()

    # N > 1 without a launcher: start the N ranks ourselves (before anything touches the GPU), relay rank 0's line and the exit code
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"error: --gpus {args.gpus} but WORLD_SIZE {world}: the launcher's world size and --gpus must agree "
            f"(unset WORLD_SIZE to let bench.py start the ranks itself)")
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    n_dev = torch.cuda.device_count()
    # (several ranks on one GPU only happen in the single-GPU exercise of the distributed path, PLO_BENCH_SHARE_GPU=1)
    share_gpu = bool(os.environ.get("PLO_BENCH_SHARE_GPU")) or (args.dist_backend == "gloo" and world > n_dev)
    dev_index = local_rank if not share_gpu else local_rank % max(1, n_dev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    force_dist = bool(os.environ.get("PLO_BENCH_FORCE_DIST"))  # exercises the RCCL path with one rank on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")
    # where the collectives' tensors live: the GPU for RCCL; host memory for gloo (payloads are copied down on the engine's stream)
    host_comm = dist is not None and args.dist_backend == "gloo"
    comm_dev = torch.device("cpu") if host_comm else dev
    strong = args.scaling == "strong"

    over = {"seed": synth.config(args.workload).seed + (0 if strong else 1000 * rank)}
    if args.reads:
        over["n_reads"] = args.reads
    cfg = synth.config(args.workload, **over)
    chunk_reads = args.chunk_reads or (500_000 if cfg.name.startswith("stress") else 8_000_000)  # (stress: 16.0 M reads/s at 500 k per batch, 14.7 M at 250 k)
    if world == 1 and dist is None and cfg.n_reads > chunk_reads:
        rc = stream_main(args, cfg, dev, dev_index, chunk_reads)
        if rc:
            sys.exit(rc)
        return
    t0 = time.perf_counter()
    w = synth.generate(cfg, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    if rank == 0:
        log(f"[bench] workload {cfg.name}: {w.n_reads} reads, {w.seg_read.numel()} read segments, {int(w.cigar.numel())} input ops, "
            f"{len(w.contig_len)} contigs / {len(w.seg_pos)} contig segments, generated on GPU in {t_gen:.1f}s")

    index = api.Index(w.index_data_device(), device=dev_index)
    n_workers = max(1, args.workers)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_workers)]
    engs = [api.Engine(index, stream=s.cuda_stream) for s in streams]
    eng = engs[0]

    # ---- this rank's batch ---------------------------------------------------------------------------------------------
    shard_info = None
    my_ranges = [(0, w.n_reads)]
    deal = wins = None
    if strong and dist is not None:
        wins = shard.workload_windows(w)
        deal = shard.deal_windows(wins, world)
        my_ranges = shard.rank_read_ranges(wins, deal, rank)
        loads = [sum(wins[i].weight for i in d) for d in deal]
        shard_info = {"windows": len(wins), "segment_size": shard.SEGMENT_SIZE, "windows_per_rank": [len(d) for d in deal],
                      "in_ops_per_rank": loads, "imbalance": (max(loads) / (sum(loads) / world)) if sum(loads) else 1.0}
        db = devbatch.DeviceBatch.from_read_ranges(w, my_ranges)
    else:
        db = devbatch.DeviceBatch.from_workload(w)
    desc = db.desc()
    my_reads = db.n_reads
    torch.cuda.synchronize()

    # The statistics of the batch -- algorithmic bytes (SURVEY.md 8(d)'s B_item, counted per item on the device) and lane utilisation -- come
    # from ONE call on a context of its own that asks for the counting instantiation of the light-item kernel (plo_ctx_set_stats); the
    # timed steps below run the production kernel, which is compiled without the counters (VERDICT r5, next #1a).
    eng_stats = api.Engine(index, stream=streams[0].cuda_stream).set_stats()
    eng_stats.liftover_batch_dev(desc)
    stats_tm = eng_stats.timing()
    stats = {"algo_bytes": int(stats_tm.algo_bytes), "lane_utilisation": float(stats_tm.lane_utilisation), "lanes_ms_counting_kernel": float(stats_tm.lanes_ms)}
    eng_stats.close()
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    gather_lock = threading.Lock()  # one exchange at a time per rank: the n-th exchange of every rank pairs up
    last_out = [None] * n_workers
    last_gather = [None]

    def step(k, gather=True):
        out = engs[k].liftover_batch_dev(desc)
        if dist is not None and gather:
            engs[k].compact_output_dev(out)  # no slab gaps over xGMI; completes on the engine's stream, not at return
            with gather_lock, torch.cuda.stream(streams[k]):  # the exchange is ordered behind the engine's kernels
                if host_comm:
                    mine = {k_: v.cpu() for k_, v in plo_gather.tensors_from_out(out, dev).items()}
                    last_gather[0] = plo_gather.gather_payloads(mine, dist, rank, world)
                else:
                    last_gather[0] = plo_gather.gather_results(out, dev, dist, rank, world)
        last_out[k] = out
        return out

    times = {k_: [] for k_ in ("lift", "lanes", "enum", "big", "mid", "retry", "heavy")}

    def record(tm):
        times["lift"].append(tm.lift_ms)
        times["lanes"].append(tm.lanes_ms)
        times["enum"].append(tm.enumerate_ms)
        times["big"].append(tm.big_ms)
        times["mid"].append(tm.mid_ms)
        times["retry"].append(tm.retry_ms)
        times["heavy"].append(tm.heavy_lanes_ms)

    def run_steps(n_steps, rec, gather=True):
        """n_steps batches, dealt to the workers in turn; every worker drives its own context on its own stream"""
        errors = []

        def worker(k):
            try:
                torch.cuda.set_device(dev_index)
                for _ in range(k, n_steps, n_workers):
                    step(k, gather)
                    if rec:
                        record(engs[k].timing())  # HIP events on the worker's stream (waits for the step)
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

    def timed(fn, n_steps):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n_steps)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def make_result(dt_, gather_desc):
        tm = eng.timing()
        # the heavy items' kernel under its own name (plo_timing::heavy_kernel says which one `heavy_lanes_ms` is the time of)
        heavy_name = {2: "k_lift_lanes_g_w3", 3: "k_lift_stream"}.get(int(tm.heavy_kernel), "k_lift_lanes_g")
        kms = {"k_lift_lanes": float(np.mean(times["lanes"])), "k_lift_mid": float(np.mean(times["mid"])), "k_lift_tiles": float(np.mean(times["lift"])),
               "k_lift_big": float(np.mean(times["big"])), "k_lift_retry": float(np.mean(times["retry"])), heavy_name: float(np.mean(times["heavy"]))}
        dominant = max(kms, key=kms.get)
        dom_ms = kms[dominant]
        # the name rocprofv3 lists it under: the tile kernel of batches without heavy items has its slice capacity compiled in
        dom_name = {"k_lift_tiles": "k_lift_tiles_c256" if int(tm.tile_cap) == 256 else "k_lift_tiles", "k_lift_mid": "k_lift_mid<16>"}.get(dominant, dominant)
        # algorithmic bytes are counted by the kernels themselves (SURVEY.md 8(d) formula), summed over all lift kernels;
        # attribute them to the dominant kernel in proportion to its share of the lift time
        share = dom_ms / max(1e-9, sum(kms.values()))
        algo_bytes = stats["algo_bytes"]  # (the statistics call above: same batch, same routing)
        achieved = (algo_bytes * share) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM traffic from the PMC counters (profiles/hbm_traffic.json, refreshed by tools/save_profiles.py): reported only when the
        # entry was collected on this workload, this read count and these very kernel sources -- else null
        traffic = None
        issue = None  # wave-instructions per launch of the dominant kernel (same keyed profile): what the SIMDs must issue at the least
        src_hash = plo_build.source_hash()
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            try:
                ent = json.load(open(prof)).get(cfg.name, {})
                if ent.get("_reads") == int(my_reads) and ent.get("_source_hash") == src_hash and ent.get("_n_gpus", 1) == world:
                    traffic = ent.get(dom_name, ent.get(dominant))
                    issue = ent.get("_insts")
            except Exception:
                traffic = None

        # the enumerate pass beside it (k_seg_count, scans, k_item_emit, class order): every input op read once (4 B), the batch's
        # per-segment (25 B) and per-read (13 B) arrays, 100 B of descriptors + result header fields written per item
        enum_ms = float(np.mean(times["enum"]))
        enum_bytes = 4 * int(tm.n_in_ops) + 25 * int(db.n_segs) + 13 * int(db.n_reads) + 100 * int(tm.n_items)
        enum_obj = {"ms": enum_ms, "algorithmic_bytes": enum_bytes, "achieved": enum_bytes / (enum_ms * 1e-3) / 1e9 if enum_ms > 0 else 0.0,
                    "unit": "GB/s", "frac": (enum_bytes / (enum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if enum_ms > 0 else 0.0}
        # The issue floor beside the HBM fraction, so that the line itself says which resource binds: a wave-instruction occupies its SIMD's
        # issue slot for four cycles (wave64 on 16 lanes); `issue_floor_ms` counts the vector instructions only (what VERDICT r4 asked for),
        # `issue_floor_all_ms` every instruction class -- measured on this engine's kernels, scalar instructions do not hide behind the vector
        # ones of the few waves a SIMD holds: all three lane kernels run at 65-92 % of THAT floor
        n_simds = 4 * int(torch.cuda.get_device_properties(dev).multi_processor_count)
        clock_hz = 2.4e9  # MI355X engine clock (MI355X_MICROARCH.md)
        issue_obj = {"issue_floor_ms": None, "frac_of_issue_floor": None}
        if issue and issue.get("valu") and dom_ms > 0:
            fl = issue["valu"] * 4.0 / (n_simds * clock_hz) * 1e3
            allc = sum(issue.get(k_, 0.0) for k_ in ("valu", "salu", "lds", "vmem_rd", "vmem_wr", "branch"))
            issue_obj = {"issue_floor_ms": fl, "frac_of_issue_floor": fl / dom_ms, "issue_floor_all_ms": allc * 4.0 / (n_simds * clock_hz) * 1e3,
                         "frac_of_issue_floor_all": allc * 4.0 / (n_simds * clock_hz) * 1e3 / dom_ms,
                         "wave_instructions_per_launch": {k_: issue[k_] for k_ in ("valu", "salu", "lds", "vmem_rd", "vmem_wr", "branch") if k_ in issue},
                         "wave_instructions_per_item": allc / max(1, int(tm.n_items)), "simds": n_simds, "clock_ghz": clock_hz / 1e9,
                         # (round 6, tools/issue_model.hip: neither the vector unit -- this fraction -- nor the CU's scalar unit is saturated; a wave's
                         # group is a dependent chain and three waves share a SIMD: DESIGN.md section 6)
                         "binding_resource": "dependent instruction chains of the 3 waves per SIMD (vector issue floor and memory-side traffic beside it)"}
        # what the memory side moves while the kernel runs (the counters' traffic, 128-byte lines for 20-byte homology probes included)
        traffic_obj = {}
        if traffic and dom_ms > 0:
            traffic_obj = {"traffic_GBps": traffic / (dom_ms * 1e-3) / 1e9, "traffic_frac_of_peak": traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "traffic_over_algorithmic": traffic / max(1.0, algo_bytes * share)}
        result = {
            "metric": "lifted HiFi reads/sec (whole node)",
            "value": total_reads * args.steps / dt_,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_ / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "dtype_note": "the reference computes in i64 / usize; every batch is range-checked on the device (PLO_ERR_RANGE beyond the 31-bit BAM domain), inside the domain 32-bit results are bit-identical",
            "data": "synthetic",
            "config": {"workload": cfg.name, "reads_total": int(total_reads), "reads_this_rank": my_reads, "read_len_mean": cfg.read_len_mean,
                       "items_per_gpu": int(tm.n_items), "in_ops_per_gpu": int(tm.n_in_ops), "out_ops_per_gpu": int(tm.n_out_ops),
                       "large_items_per_gpu": int(tm.n_big_items), "mid_items_per_gpu": int(tm.n_mid_items),
                       "retry_items_per_gpu": int(tm.n_retry_items), "lane_items_per_gpu": int(tm.n_lane_items), "heavy_lane_items_per_gpu": int(tm.n_heavy_lane_items), "seq_fmt": "bam4",
                       "tile_geometry": {"slice_elements": int(tm.tile_cap), "window": int(tm.tile_window)},
                       "parallelism": (f"one read set, 20 Mb windows dealt to {world} ranks by input ops" if strong else f"{world} independent read sets"),
                       "host_workers_per_gpu": n_workers, "kernel_source_hash": src_hash,
                       "gather": gather_desc, "dist_backend": (args.dist_backend if dist is not None else None),
                       "launched_by": ("bench.py (child torch.distributed.run)" if os.environ.get("PLO_BENCH_LAUNCHED") else
                                       ("external launcher" if world > 1 else "single process"))},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_copy_ceiling": achieved / HBM_COPY_CEILING_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(algo_bytes * share), "kernel_ms": dom_ms,
                         "enumerate_ms": float(np.mean(times["enum"])), "enumerate_pass": enum_obj, "lift_lanes_ms": kms["k_lift_lanes"], "lift_tiles_ms": kms["k_lift_tiles"],
                         "lift_big_ms": kms["k_lift_big"], "lift_mid_ms": kms["k_lift_mid"], "lift_retry_ms": kms["k_lift_retry"], "lift_heavy_ms": kms[heavy_name], "heavy_kernel": heavy_name if int(tm.heavy_kernel) else None,
                         "lane_utilisation": stats["lane_utilisation"],  # lanes at work / (64 x trips) of the lane kernels' liftover loop and shift rounds
                         "statistics_from": "one call of the counting kernel outside the timed region (plo_ctx_set_stats); the timed kernel carries no counters",
                         "lanes_ms_counting_kernel": stats["lanes_ms_counting_kernel"],
                         "host_syncs_per_call": int(tm.host_syncs), **issue_obj, **traffic_obj},
        }
        return result

    dt_ng = None
    gather_modes = None
    if dist is None:
        run_steps(max(args.warmup, n_workers), False)  # every context sizes its buffers outside the timed region
        dt = timed(lambda n: run_steps(n, True), args.steps)
        total_reads = float(my_reads)
        result = make_result(dt, "none")
    else:
        nr = torch.tensor([my_reads], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(nr, op=dist.ReduceOp.SUM)
        total_reads = float(nr.item())
        # First the same K steps without the record gather (no collective on the data path): a complete measurement that rank 0 can
        # print whatever happens to the exchange afterwards.  Then the headline: every step's records gathered on rank 0.
        run_steps(max(args.warmup, n_workers), False, gather=False)
        dt_ng = timed(lambda n: run_steps(n, True, gather=False), args.steps)
        fallback = make_result(dt_ng, "none: the record gather to rank 0 did not complete in this run (see stderr); every rank keeps its shard")

        pending_print = [fallback]  # what rank 0 prints if the next RCCL section does not come back

        # A collective that hangs or fails must not look like a healthy N-GPU run: the line rank 0 prints then has
        # "gather_failed": true at top level and "value": null (the complete no-gather measurement stays in `no_gather`), and
        # every rank exits non-zero.
        def failed_line(reason):
            r = dict(pending_print[0])
            r["no_gather"] = {"value": r["value"], "unit": "reads/s", "ms_per_step": r["ms_per_step"],
                              "note": "same K steps without the record gather (each rank keeps / writes its own shard)"}
            r["value"] = None
            r["gather_failed"] = True
            r["gather_failure"] = reason
            return r

        def gather_bail():
            log("[bench] an RCCL record gather did not finish in time")
            if rank == 0:
                print(json.dumps(failed_line("timeout: a record gather did not come back within PLO_BENCH_GATHER_TIMEOUT")), flush=True)
            os._exit(3)

        gather_error = [None]

        def supp_bail():
            # a SUPPLEMENTARY gather mode hung: the complete measurement made so far goes out as it is (marked), the process cannot go on
            log("[bench] a supplementary gather mode did not finish in time")
            if rank == 0:
                r_ = dict(pending_print[0])
                r_["supplementary_timed_out"] = True
                r_["degraded"] = True
                print(json.dumps(r_), flush=True)
            os._exit(4)

        def guarded(fn, fatal=True):
            if gather_error[0] is not None:  # (after an RCCL error the next collective would only hang until the watchdog fires)
                return None
            wd = threading.Timer(float(os.environ.get("PLO_BENCH_GATHER_TIMEOUT", "150")), gather_bail if fatal else supp_bail)
            wd.daemon = True
            wd.start()
            try:
                return fn()
            except Exception as e:  # noqa: BLE001
                log(f"[bench] RCCL record gather failed: {e!r}")
                gather_error[0] = repr(e)
                return None
            finally:
                wd.cancel()

        # (a) every step's records gathered on rank 0 before the next step starts
        def sync_run():
            run_steps(max(args.warmup, n_workers), False)
            return timed(lambda n: run_steps(n, False), args.steps)

        dt_sync = guarded(sync_run)
        result = fallback
        if dt_sync is None:
            if rank == 0:
                print(json.dumps(failed_line(gather_error[0] or "the synchronous record gather failed")), flush=True)
            os._exit(3)
        if dt_sync is not None:
            result = make_result(dt_sync, ("gloo send/recv of host copies" if host_comm else "rccl send/recv") + " to rank 0 after every step")
            pending_print[0] = result
        # (b) the gather of batch i overlapped with the compute of batch i+1: two contexts alternate, so that the exchange reads one
        # context's buffers while the other computes.  Both the posting and the wait happen under the owning engine's stream: RCCL
        # starts the sends behind the compaction kernel, and the engine's next kernels start behind the sends that read its buffers.
        dt_async = None
        if n_workers == 1 and dt_sync is not None and not host_comm:
            def async_run():
                s2 = torch.cuda.Stream(device=dev)
                a_streams = [streams[0], s2]
                a_engs = [engs[0], api.Engine(index, stream=s2.cuda_stream)]
                a_engs[1].liftover_batch_dev(desc)  # sizes its buffers
                a_engs[1].sync()

                def run_async(n_steps):
                    pending = [None, None]
                    for i in range(n_steps):
                        k = i & 1
                        if pending[k] is not None:
                            with torch.cuda.stream(a_streams[k]):
                                pending[k].wait()
                            pending[k] = None
                        out_k = a_engs[k].liftover_batch_dev(desc)
                        a_engs[k].compact_output_dev(out_k)
                        with torch.cuda.stream(a_streams[k]):
                            pending[k] = plo_gather.gather_results_async(out_k, dev, dist, rank, world)
                    for k, p_ in enumerate(pending):
                        if p_ is not None:
                            with torch.cuda.stream(a_streams[k]):
                                p_.wait()

                run_async(2)
                d_ = timed(run_async, args.steps)
                a_engs[1].close()
                engs[0].liftover_batch_dev(desc)  # (this rank's own result back in the first context)
                engs[0].sync()
                return d_

            dt_async = guarded(async_run)
        # (a') the same synchronous gather through the library's own C ABI (plo_gather_*: ncclAllGather of the sizes + grouped ncclSend / ncclRecv
        # on a communicator the library creates -- what a Rust host binds, INTEGRATION.md section 6 route (b)); torch.distributed only
        # carries the 128-byte id
        dt_abi = None
        abi_parts = None
        # (on by default with one rank -- what the one-GPU boxes of this pool can run, RCCL refuses two ranks on one device -- and with
        # PLO_BENCH_ABI_GATHER=1: a multi-rank exchange through it has not run on hardware yet, and a collective that hangs would take the
        # complete measurements above with it)
        abi_on = os.environ.get("PLO_BENCH_ABI_GATHER", "1" if world == 1 else "0") == "1"
        if dt_sync is not None and not host_comm and n_workers == 1 and abi_on:
            def abi_run():
                ag = plo_gather.AbiGather(index.lib, dist, rank, world, dev_index)
                last = [None]

                def run(n_steps):
                    for _ in range(n_steps):
                        out_ = eng.liftover_batch_dev(desc)
                        eng.compact_output_dev(out_)
                        last[0] = ag.gather(eng, out_, dev).wait()

                run(1)
                d_ = timed(run, args.steps)
                parts_ = [{k_: v.clone() for k_, v in t_.items()} for t_ in last[0]] if rank == 0 else None
                ag.close()
                return d_, parts_

            got_abi = guarded(abi_run, fatal=False)
            if got_abi is not None:
                dt_abi, abi_parts = got_abi
            else:
                gather_error[0] = None  # (supplementary: the modes below still run)
        # (c) the reference's window loop on every rank (src/read_alignment_scanner.rs:508-534: window tasks one after the other, one locked
        # writer behind them): the rank's windows as k batches on two contexts, the exchange of batch i under the compute of batch i + 1 --
        # across the steps' boundaries too; only the last batch's exchange of the last step is waited for with nothing to hide it under.
        # (gloo with host copies has no asynchronous form here: the batches and their exchanges then simply follow each other.)
        dt_pipe = None
        pipe_parts = None  # rank 0: the gathered parts of one pass, batch by batch (verification)
        k_pipe = max(1, args.pipeline_batches)
        if strong and k_pipe > 1 and n_workers == 1 and dt_sync is not None:
            pipe_groups = [shard.pipeline_batches(wins, deal, r, k_pipe) for r in range(world)]
            sub_dbs = [devbatch.DeviceBatch.from_read_ranges(w, g) for g in pipe_groups[rank]]
            sub_descs = [d_.desc() for d_ in sub_dbs]

            def pipe_run_outer():
                s2 = torch.cuda.Stream(device=dev)
                p_streams = [streams[0], s2]
                p_engs = [engs[0], api.Engine(index, stream=s2.cuda_stream)]

                def snap(got_):  # (the root's own part is a view of a context's buffers, which the context's next batch overwrites)
                    return None if got_ is None else [{k_: v.clone() for k_, v in t_.items()} for t_ in got_]

                def pipe_run(n_steps, keep=None):
                    pending = [None, None]
                    for i in range(n_steps * k_pipe):
                        j, c_ = i % k_pipe, i & 1
                        if pending[c_] is not None:
                            with torch.cuda.stream(p_streams[c_]):
                                got_ = pending[c_][1].wait()
                            if keep is not None:
                                keep[pending[c_][0]] = snap(got_)
                            pending[c_] = None
                        out_j = p_engs[c_].liftover_batch_dev(sub_descs[j])
                        p_engs[c_].compact_output_dev(out_j)
                        with torch.cuda.stream(p_streams[c_]):
                            if host_comm:
                                mine = {k_: v.cpu() for k_, v in plo_gather.tensors_from_out(out_j, dev).items()}
                                got_ = plo_gather.gather_payloads(mine, dist, rank, world)
                                if keep is not None:
                                    keep[j] = snap(got_)
                            else:
                                pending[c_] = (j, plo_gather.gather_results_async(out_j, dev, dist, rank, world))
                    for c_ in (0, 1):
                        if pending[c_] is not None:
                            with torch.cuda.stream(p_streams[c_]):
                                got_ = pending[c_][1].wait()
                            if keep is not None:
                                keep[pending[c_][0]] = snap(got_)

                pipe_run(1)  # (sizes the second context's buffers)
                d_ = timed(pipe_run, args.steps)
                kept = {}
                pipe_run(1, kept)  # one more pass whose gathered parts rank 0 keeps (cloned: the contexts' buffers are reused)
                torch.cuda.synchronize()
                parts_ = None
                if rank == 0:
                    parts_ = [kept[j] for j in range(k_pipe)]
                p_engs[1].close()
                engs[0].liftover_batch_dev(desc)  # (this rank's own whole-share result back in the first context)
                engs[0].sync()
                return d_, parts_

            got_pipe = guarded(pipe_run_outer, fatal=False)
            if got_pipe is not None:
                dt_pipe, pipe_parts = got_pipe
            else:
                gather_error[0] = None  # (supplementary: what follows still runs)
        gather_modes = {}  # reads/s of the whole job with the records of every step on rank 0 when the clock stops
        if dt_sync is not None:
            gather_modes["after_every_step"] = {"value": total_reads * args.steps / dt_sync, "unit": "reads/s", "ms_per_step": dt_sync / args.steps * 1e3}
        if dt_async is not None:
            gather_modes["overlapped_with_next_step"] = {"value": total_reads * args.steps / dt_async, "unit": "reads/s",
                                                         "ms_per_step": dt_async / args.steps * 1e3}
            if dt_async < dt_sync:  # the headline is the faster complete pipeline: K steps, all K record sets on rank 0 when the clock stops
                result = make_result(dt_async, "rccl send/recv to rank 0, the exchange of batch i overlapped with the compute of batch i+1 "
                                               "(two contexts alternate)")
        if dt_abi is not None:
            gather_modes["c_abi"] = {"value": total_reads * args.steps / dt_abi, "unit": "reads/s", "ms_per_step": dt_abi / args.steps * 1e3,
                                     "note": "plo_gather_records / plo_gather_wait after every step (the library's own RCCL communicator)"}
        if dt_pipe is not None:
            gather_modes["window_pipeline"] = {"value": total_reads * args.steps / dt_pipe, "unit": "reads/s", "ms_per_step": dt_pipe / args.steps * 1e3,
                                               "batches_per_rank_and_step": k_pipe, "reads_per_batch_this_rank": [int(d_.n_reads) for d_ in sub_dbs],
                                               "note": "every rank lifts its windows as consecutive batches on two contexts, the gather of batch i under the compute "
                                                       "of batch i+1 (the reference's window loop, read_alignment_scanner.rs:508-534)"}
            if dt_pipe < min(x for x in (dt_sync, dt_async) if x is not None):
                result = make_result(dt_pipe, ("gloo send/recv of host copies" if host_comm else "rccl send/recv") + f" to rank 0, every rank's windows as {k_pipe} "
                                     "batches per step on two contexts, the exchange of batch i under the compute of batch i+1")

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

    # a reference-sized window (the reference lifts ~50 k reads per 20 Mb window task, src/read_alignment_scanner.rs:508-534,575): the
    # per-call fixed cost -- launches, host round trips -- against a batch forty times smaller than the headline's
    window_50k = None
    if dist is None and n_workers == 1 and w.n_reads >= 100_000 and args.window_calls > 0:
        try:
            lo = w.n_reads // 2
            wdb = devbatch.DeviceBatch.from_workload(w, lo, lo + 50_000)
            wdesc = wdb.desc()
            torch.cuda.synchronize()
            for _ in range(5):
                eng.liftover_batch_dev(wdesc)
            eng.sync()
            n_calls = args.window_calls
            t1 = time.perf_counter()
            for _ in range(n_calls):
                eng.liftover_batch_dev(wdesc)
            eng.sync()
            wdt = time.perf_counter() - t1
            tmw = eng.timing()
            # the same calls as a production caller that never asks for the phase times makes them (plo_ctx_set_phase_events(ctx, 0): no event
            # records between the phases, two ~6 us bubbles fewer per call on the stream)
            eng.set_phase_events(False)
            for _ in range(5):
                eng.liftover_batch_dev(wdesc)
            eng.sync()
            t1 = time.perf_counter()
            for _ in range(n_calls):
                eng.liftover_batch_dev(wdesc)
            eng.sync()
            wdt_ne = time.perf_counter() - t1
            eng.set_phase_events(True)
            window_50k = {"value": 50_000 * n_calls / wdt, "unit": "reads/s", "reads_per_call": 50_000, "calls": n_calls, "ms_per_call": wdt / n_calls * 1e3,
                          "device_ms_per_call": float(tmw.total_ms), "host_workers_per_gpu": 1,
                          "without_phase_events": {"value": 50_000 * n_calls / wdt_ne, "ms_per_call": wdt_ne / n_calls * 1e3},
                          "note": "plo_liftover_batch_dev on a device-resident 50 k-read window, one context, back to back; `without_phase_events`: the "
                                  "same on a context with plo_ctx_set_phase_events(ctx, 0)"}
            last_out[0] = eng.liftover_batch_dev(desc)  # (the headline batch's result back in the context: the parity sample below reads it)
            eng.sync()
        except Exception as e:  # noqa: BLE001 -- supplementary
            log(f"[bench] window_50k failed: {e!r}")

    # Everything from here on is supplementary.  At N > 1 it runs more collectives (gather variants, verification): if any of that
    # does not finish, every rank gives up after a while and rank 0 still prints the headline measurement made above.
    watchdog = None
    if dist is not None:
        def bail():
            log("[bench] the supplementary distributed measurements did not finish in time: printing the headline result without them")
            result["supplementary_timed_out"] = True
            result["degraded"] = True
            if rank == 0:
                print(json.dumps(result), flush=True)
            os._exit(4)  # the headline (gathered) measurement is complete and printed; a collective of the supplementary part hung

        watchdog = threading.Timer(float(os.environ.get("PLO_BENCH_SUPP_TIMEOUT", "900")), bail)
        watchdog.daemon = True
        watchdog.start()
    e2e_sh = None
    try:
        no_gather = verify = None
        # ---- supplementary distributed numbers (SURVEY.md 8(e) "report both") ---------------------------------------------------
        if dist is not None:
            no_gather = {"value": total_reads * args.steps / dt_ng, "unit": "reads/s", "ms_per_step": dt_ng / args.steps * 1e3,
                         "note": "same K steps without the record gather (each rank keeps / writes its own shard)"}
            # one more synchronous step whose gathered records rank 0 compares with its own result of the WHOLE read set
            step(0)
            torch.cuda.synchronize()
            if strong and not args.no_verify:
                if rank == 0:
                    seg_maps = [plo_gather.local_to_global_segments(w, shard.rank_read_ranges(wins, deal, r)).to(comm_dev) for r in range(world)]
                    got_all = plo_gather.combine(last_gather[0], seg_maps)
                    got_all = {k_: v.clone() for k_, v in got_all.items()}
                    whole_db = devbatch.DeviceBatch.from_workload(w)
                    eng.liftover_batch_dev(whole_db.desc())  # (sizes the context's buffers for the whole set: its events would span the allocations)
                    whole_out = eng.liftover_batch_dev(whole_db.desc())
                    whole_ms = float(eng.timing().total_ms)  # device time of the WHOLE read set on this one GPU
                    eng.compact_output_dev(whole_out)
                    eng.sync()
                    whole = {k_: v.to(comm_dev) for k_, v in plo_gather.tensors_from_out(whole_out, dev).items()}
                    same = plo_gather.same_records(got_all, whole)
                    verify = {"gathered_equals_single_gpu_result": bool(same), "items": int(whole["item_seg"].numel()),
                              "reads": int(w.n_reads), "single_gpu_device_ms": whole_ms,
                              "this_rank": {"reads": int(my_reads), "ms_per_step_no_gather": dt_ng / args.steps * 1e3,
                                            "single_gpu_pro_rata_ms": whole_ms * my_reads / max(1.0, total_reads)}}
                    if abi_parts is not None:  # the C ABI's gather
                        same_a = plo_gather.same_records(plo_gather.combine([{k_: v.to(comm_dev) for k_, v in t_.items()} for t_ in abi_parts], seg_maps), whole)
                        verify["c_abi_gather_equals_single_gpu_result"] = bool(same_a)
                        same = same and same_a
                        verify["gathered_equals_single_gpu_result"] = bool(same)
                    if pipe_parts is not None:  # the window pipeline's batches, every one mapped back to the unsharded numbering
                        per_batch = []
                        for j in range(k_pipe):
                            maps_j = [plo_gather.local_to_global_segments(w, pipe_groups[r][j]).to(comm_dev) for r in range(world)]
                            per_batch.append(plo_gather.combine([{k_: v.to(comm_dev) for k_, v in t_.items()} for t_ in pipe_parts[j]], maps_j))
                        same_p = plo_gather.same_records(plo_gather.combine(per_batch), whole)
                        verify["window_pipeline_equals_single_gpu_result"] = bool(same_p)
                        same = same and same_p
                        verify["gathered_equals_single_gpu_result"] = bool(same)
                    if not same:
                        log("[bench] VERIFY FAILURE: gathered records differ from the single-GPU result")
                    step(0, gather=False)  # restore this rank's own result in the context (read by the roofline object below)
                barrier()

        # BASELINE configs[3] as a BAM run: one input BAM, every rank its part of it (plo_bam_open_range), its own output shard
        if dist is not None and strong and args.e2e_reads > 0:
            e2e_sh = end_to_end_sharded(w, index, args.e2e_reads, args.e2e_window, args.e2e_workers, dist, rank, world, verify=not args.no_cpu_baseline)
    except Exception as e:  # noqa: BLE001 -- supplementary objects must never hide the measurement
        log(f"[bench] supplementary distributed measurements failed: {e!r}")
    if watchdog is not None:
        watchdog.cancel()
    for name, obj in (("shard", shard_info), ("overlap", overlap), ("window_50k", window_50k), ("no_gather", no_gather), ("gather_modes", gather_modes),
                      ("verify", verify), ("end_to_end_sharded", e2e_sh)):
        if obj is not None:
            result[name] = obj
    ixd_host = None
    if rank == 0 and world == 1 and dist is None and args.e2e_reads > 0:
        try:
            ixd_host = w.index_data()
            e2e, pcie = end_to_end(w, index, ixd_host, args.e2e_reads, args.e2e_window, args.e2e_workers, int(os.environ.get("PLO_BENCH_IO_THREADS", "0")) or max(2, min(64, pipeline_cpus())),
                                   verify=not args.no_cpu_baseline)
            result["end_to_end"] = e2e
            if pcie is not None:
                result["pcie_inclusive"] = pcie
        except Exception as e:  # supplementary objects must never hide the measurement
            log(f"[bench] end_to_end failed: {e!r}")
            result["end_to_end"] = None
    if rank == 0 and world == 1 and dist is None and not args.no_cpu_baseline:
        try:
            got = devbatch.download(eng, last_out[0])
            cb, ok, n_checked = cpu_baseline(w, got, ixd=ixd_host)
            if cb is not None and (result.get("end_to_end") or {}).get("cpu_pipeline"):
                cb["end_to_end"] = result["end_to_end"]["cpu_pipeline"]  # the CPU figure beside the GPU `end_to_end`
            result["cpu_baseline"] = cb
            result["parity_sample_items"] = n_checked
            result["parity_sample_ok"] = bool(ok)
            if not ok:
                log("[bench] PARITY FAILURE on the sampled reads")
        except Exception as e:  # the baseline must never hide the measurement
            log(f"[bench] cpu_baseline failed: {e!r}")
            result["cpu_baseline"] = None
    rc = 0
    if rank == 0:
        rc = finalize(result)
        print(json.dumps(result), flush=True)
    for e in engs:
        e.close()
    index.close()
    if dist is not None:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
