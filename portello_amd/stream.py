"""Read sets larger than one batch: consecutive plo_liftover_batch_dev calls over a list of batches, dealt to a few host workers
with one context (plo_ctx) and one HIP stream each -- the arrangement INTEGRATION.md gives the reference's rayon workers -- so
that one batch's enumerate pass and host round trips run under another's lift kernels.  The reference walks its read set the
same way, window by window (src/read_alignment_scanner.rs:495-535); a batch here is bounded by the engine's 31-bit op indices
(about 10^9 input ops: 500 k indel-dense 20 kb reads, or tens of millions of HiFi reads)."""
import threading
from typing import Callable, List, Optional

import torch

from portello_amd import api


class StreamRunner:
    def __init__(self, index: api.Index, device: torch.device, n_workers: int = 2):
        self.device = device
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(1, n_workers))]
        self.engines = [api.Engine(index, stream=s.cuda_stream) for s in self.streams]

    def run(self, descs: List, consume: Optional[Callable] = None, record: Optional[Callable] = None):
        """lifts every batch (abi.PloBatchIn descriptors of device-resident batches) once; `consume(k, engine, out)` is called on
        the worker's thread while batch k's result is still in the engine's buffers (download, checks, finishing);
        `record(engine.timing())` after every call"""
        n_w = len(self.engines)
        errors = []

        def worker(wi):
            try:
                torch.cuda.set_device(self.device)
                eng = self.engines[wi]
                for k in range(wi, len(descs), n_w):
                    out = eng.liftover_batch_dev(descs[k])
                    if record is not None:
                        record(eng.timing())
                    if consume is not None:
                        consume(k, eng, out)
            except BaseException as e:  # noqa: BLE001 -- re-raised on the caller's thread
                errors.append(e)

        if n_w == 1:
            worker(0)
        else:
            th = [threading.Thread(target=worker, args=(i,)) for i in range(n_w)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errors:
            raise errors[0]

    def sync(self):
        for e in self.engines:
            e.sync()

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []


def chunk_reads_default(workload_name: str) -> int:
    """reads per batch of a streamed run: about 0.5 G input ops for the indel-dense profile, 2 M reads (the bench batch) otherwise"""
    return 250_000 if workload_name.startswith("stress") else 2_000_000


def generate_chunks(synth, cfg_name: str, total_reads: int, chunk_reads: int, device, log=None):
    """the read set in chunks on ONE set of contigs (synth.generate(..., reuse=)): [(workload chunk)], first one carries the index"""
    import time

    sizes = [chunk_reads] * (total_reads // chunk_reads) + ([total_reads % chunk_reads] if total_reads % chunk_reads else [])
    base_seed = synth.config(cfg_name).seed
    chunks = []
    for i, n in enumerate(sizes):
        t0 = time.perf_counter()
        cfg = synth.config(cfg_name, n_reads=n, seed=base_seed + 7919 * i)
        w = synth.generate(cfg, device=device, keep_contigs=True) if i == 0 else synth.generate(cfg, device=device, reuse=chunks[0])
        chunks.append(w)
        if log:
            log(f"[stream] chunk {i + 1}/{len(sizes)}: {w.n_reads} reads, {int(w.cigar.numel())} input ops, generated in {time.perf_counter() - t0:.1f} s")
    return chunks
