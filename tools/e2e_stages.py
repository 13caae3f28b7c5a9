#!/usr/bin/env python3
"""The BAM -> BAM pipeline's stages alone and together on one sample: reader alone, reader + batch construction, then the whole pipeline
(device finish) for several worker / thread shares.  Tells a stage that is slow by itself from stages that slow each other down.  GPU only.
usage: tools/e2e_stages.py [reads]"""
import os
import shutil
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, bam, bamsynth, pipeline, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 240000
dev = torch.device("cuda", 0)
w = synth.generate(synth.config("wgs30x", n_reads=400000), device=dev)
index = api.Index(w.index_data_device(), 0)
ixd = w.index_data()
d = tempfile.mkdtemp(prefix="plo_e2e_")
try:
    inp = os.path.join(d, "reads.bam")
    lo = (w.n_reads - n) // 2
    meta = bamsynth.write_read_bam(w, inp, lo, lo + n, level=1, n_threads=16)
    cn, rn = meta["contig_names"], bamsynth.ref_names(w)
    rl = [int(s.numel()) for s in w.chrom_seq]
    pipeline.run_bam_to_bam(inp, os.path.join(d, "o.bam"), index, ixd, cn, rn, rl, window_reads=2000, n_workers=1, device_finish=True)
    for threads in (() if os.environ.get("PLO_E2E_PIPELINE_ONLY") else (8, 16)):
        for with_batch in (False, True):
            best = None
            for _ in range(3):
                rd = bam.BamReader(inp, threads, device_inflate=0)
                t0 = time.perf_counter()
                tb = 0.0
                k = 0
                while True:
                    win = rd.read_window(7500)
                    if win is None:
                        break
                    k += win.n_records
                    if with_batch and win.n_records:
                        t1 = time.perf_counter()
                        win.batch_desc(with_finish=True)
                        tb += time.perf_counter() - t1
                    win.close()
                t = time.perf_counter() - t0
                rd.close()
                if best is None or t < best[0]:
                    best = (t, tb)
            print(f"reader alone ({threads} threads){' + batch construction in the same thread' if with_batch else ''}: {k} reads in {best[0]:.3f} s = "
                  f"{k / best[0] / 1e3:.0f} k reads/s" + (f" (batch construction {best[1]:.3f} s)" if with_batch else ""), flush=True)
    if os.environ.get("PLO_E2E_WRITER_AB"):
        # the writer's variants, same pipeline otherwise (two lift workers, a new output file per run)
        for env in ({}, {"PLO_BGZF_CHUNK_MB": "1024"}, {"PLO_BGZF_CHUNK_MB": "1024", "PLO_BGZF_FALLOCATE": "1"}, {"PLO_BGZF_CHUNK_MB": "2048", "PLO_BGZF_FALLOCATE": "1"}, {}):
            for k in ("PLO_BGZF_FALLOCATE", "PLO_BGZF_COPY_BLOCKS", "PLO_BGZF_CHUNK_MB"):
                os.environ.pop(k, None)
            os.environ.update(env)
            best = None
            for r in range(4):
                outp = os.path.join(d, f"w_{r}.bam")
                st = pipeline.run_bam_to_bam(inp, outp, index, ixd, cn, rn, rl, device_finish=True, window_reads=7500, io_threads=16, n_workers=2)
                os.unlink(outp)
                if best is None or st.seconds < best.seconds:
                    best = st
            print(f"writer {env or 'default'}: {best.reads / best.seconds / 1e3:.1f} k reads/s ({best.seconds:.3f} s; busy: read {best.read_s:.2f} batch {best.batch_s:.2f} "
                  f"lift {best.lift_s:.2f} build {best.build_s:.2f} write {best.write_s:.2f}); done at: " + ", ".join(f"{k} {v:.3f}" for k, v in best.stage_done_s.items()), flush=True)
        sys.exit(0)
    for kw in (dict(n_workers=2), dict(n_workers=4), dict(n_workers=3, read_threads=16, build_threads=4, write_threads=16)):
        a = dict(window_reads=7500, io_threads=16)
        a.update(kw)
        best = None
        for fresh in ((False, True) if kw == dict(n_workers=2) else (True,)):
            best = None
            for r in range(3):
                # fresh: a new output file per run -- rewriting one path means O_TRUNC on a file with gigabytes of dirty pages and (ext4,
                # auto_da_alloc) a forced flush when the rewritten file is closed
                outp = os.path.join(d, f"o_{len(a)}_{r}.bam" if fresh else "o.bam")
                if fresh and os.path.exists(outp):
                    os.unlink(outp)
                st = pipeline.run_bam_to_bam(inp, outp, index, ixd, cn, rn, rl, device_finish=True, **a)
                if fresh:
                    os.unlink(outp)
                if best is None or st.seconds < best.seconds:
                    best = st
            if not fresh:
                print("(same output path rewritten)", flush=True)
                det = ", ".join(f"{k} {v:.3f}" for k, v in best.lift_detail_s.items())
                print(f"pipeline {kw}: {best.reads / best.seconds / 1e3:.1f} k reads/s ({best.seconds:.3f} s)\n    done at: " + ", ".join(f"{k} {v:.3f}" for k, v in best.stage_done_s.items()), flush=True)
        det = ", ".join(f"{k} {v:.3f}" for k, v in best.lift_detail_s.items())
        print(f"pipeline {kw}: {best.reads / best.seconds / 1e3:.1f} k reads/s ({best.seconds:.3f} s; busy: read {best.read_s:.2f} batch {best.batch_s:.2f} "
              f"lift {best.lift_s:.2f} build {best.build_s:.2f} write {best.write_s:.2f})\n    lift stage: {det}\n    done at: "
              + ", ".join(f"{k} {v:.3f}" for k, v in best.stage_done_s.items()), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
