#!/usr/bin/env python3
"""Which items does the lane kernel hand to the retry list?  (GPU)  Prints their CIGARs and saves the batch of their reads for the emulator."""
import ctypes as C, os, sys, pickle
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from portello_amd import api, devbatch, synth, abi, cigar as cg
L = api.load_library()
L.plo_ctx_debug_retry_list.restype = C.c_uint
L.plo_ctx_debug_retry_list.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_uint]
w = synth.generate(synth.config("chr20", n_reads=30000), device="cuda")
index = api.Index(w.index_data_device())
eng = api.Engine(index, stream=torch.cuda.current_stream().cuda_stream)
lo, hi = 0, 8000
db = devbatch.DeviceBatch.from_workload(w, lo, hi)
torch.cuda.synchronize()
got = devbatch.run_and_download(eng, db)
t = eng.timing()
print("items", t.n_items, "retried", t.n_retry_items, "big", t.n_big_items, "mid", t.n_mid_items, "syncs", t.host_syncs)
buf = (C.c_uint * 4096)()
n = L.plo_ctx_debug_retry_list(eng.handle, buf, 4096)
b = w.batch_data(lo, hi)
for i in list(buf[:n])[:10]:
    seg = int(got.item_seg[i]); cs = int(got.item_cseg[i])
    c = b.cigar[b.seg_cigar_off[seg]:b.seg_cigar_off[seg + 1]]
    print("item", i, "seg", seg, "cseg", cs, "read", int(b.seg_read[seg]), "n_in", len(c), "cigar", cg.to_string(c) if hasattr(cg, "to_string") else c[:40], "status", int(got.item_status[i]))
