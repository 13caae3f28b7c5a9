"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/portello_liftover.h declares, and
fails loudly (no CPU fallback) when no HIP device is usable.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from portello_amd import abi, api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    return build.build()


def declared_symbols():
    out = set()
    for h in ("portello_liftover.h", "portello_bam.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        out |= set(re.findall(r"^(?:plo_status|void|int|uint32_t|uint64_t|char \*|const char \*)\s*\*?(plo_[a-z0-9_]+)\(", text, flags=re.M))
    return sorted(out)


def test_library_exports_every_declared_symbol(lib_path):
    L = C.CDLL(lib_path)
    syms = declared_symbols()
    assert {"plo_index_create", "plo_ctx_create", "plo_liftover_batch", "plo_liftover_batch_dev", "plo_selftest", "plo_bam_open",
            "plo_bam_read_window", "plo_bam_window_batch", "plo_records_build", "plo_records_build_finished", "plo_bam_writer_open", "plo_bam_output_header",
            "plo_bam_window_n_records", "plo_bam_window_batch_sparse", "plo_sparse_seq_pack", "plo_sparse_seq_bound"} <= set(syms)
    for s in syms:
        assert hasattr(L, s), f"{s} declared in the header but not exported"


def test_version_string(lib_path):
    L = api.load_library(lib_path)
    assert b"gfx950" in L.plo_version()


def test_api_version_matches_header(lib_path):
    text = open(os.path.join(ROOT, "include", "portello_liftover.h")).read()
    v = int(re.search(r"#define PLO_API_VERSION (\d+)", text).group(1))
    L = api.load_library(lib_path)
    assert L.plo_api_version() == v == abi.PLO_API_VERSION


def test_timing_getter_respects_struct_size(lib_path):
    # API version 4: the callee writes no more than the caller's struct holds; without a size it refuses.  (No device needed:
    # a NULL context is rejected first, so only the argument contract is visible from here.)
    L = api.load_library(lib_path)
    t = abi.PloTiming()
    assert L.plo_ctx_timing(None, C.byref(t)) == abi.PLO_ERR_INVALID_ARG


@pytest.mark.gpu
def test_timing_struct_size_contract_on_a_real_context(lib_path):
    """API version 4 on a live context (the NULL-context test above never reaches the size check): struct_size 0 is refused; a caller
    with a SMALLER (older) struct gets a truncated copy -- not one byte beyond it is written -- and the size that was filled in"""
    from portello_amd import synth

    w = synth.generate(synth.config("tiny", n_reads=50, seed=3))
    index = api.Index(w.index_data(), 0)
    eng = api.Engine(index)
    eng.liftover_batch(w.batch_data())
    L = eng.lib

    class Guarded(C.Structure):  # a plo_timing followed by a canary
        _fields_ = [("t", abi.PloTiming), ("canary", C.c_uint8 * 64)]

    g = Guarded()
    C.memset(C.byref(g), 0xA5, C.sizeof(g))
    g.t.struct_size = 0
    assert L.plo_ctx_timing(eng.handle, C.byref(g.t)) == abi.PLO_ERR_INVALID_ARG
    assert bytes(g.canary) == b"\xa5" * 64
    # an older caller: the struct ends after n_in_ops (40 bytes)
    small = abi.PloTiming.n_in_ops.offset + 8
    C.memset(C.byref(g), 0xA5, C.sizeof(g))
    g.t.struct_size = small
    assert L.plo_ctx_timing(eng.handle, C.byref(g.t)) == abi.PLO_OK
    raw = bytes((C.c_uint8 * C.sizeof(g)).from_buffer(g))
    assert g.t.struct_size == small and g.t.n_in_ops > 0
    assert raw[small:] == b"\xa5" * (C.sizeof(g) - small)  # nothing past the caller's struct was touched
    # the full struct
    g.t.struct_size = C.sizeof(abi.PloTiming)
    assert L.plo_ctx_timing(eng.handle, C.byref(g.t)) == abi.PLO_OK
    assert g.t.struct_size == C.sizeof(abi.PloTiming) and g.t.n_items > 0 and bytes(g.canary) == b"\xa5" * 64
    eng.close()
    index.close()


def test_ctypes_struct_layout_matches_header():
    # sizes implied by the header on LP64
    assert C.sizeof(abi.PloBatchOut) == 8 + 10 * 8 + 8
    # struct_size + 4 floats + 2 u32 | 3 u64 | 4 + 2 + 2 + 2 + 2 + 1 + 2 four-byte fields (+ tail padding to 8)
    assert C.sizeof(abi.PloTiming) == 8 * 4 + 3 * 8 + 16 * 4
    assert abi.PloTiming.struct_size.offset == 0 and abi.PloTiming.n_in_ops.offset == 32
    assert C.sizeof(abi.PloBatchIn) == 8 + 4 * 8 + 8 + 8 + 6 * 8 + 8 + 2 * 8 + 2 * 8
    assert C.sizeof(abi.PloIndexDesc) == 8 + 2 * 8 + 8 + 8 * 8 + 8 + 3 * 8 + 8


def test_fails_loudly_without_gpu(lib_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cs = api.CaseSet()
    cs.add_liftover(1000, np.array([(100 << 4) | 0], dtype=np.uint32), 10, np.array([(10 << 4) | 0], dtype=np.uint32))
    with pytest.raises(api.PortelloError) as ei:
        cs.run(abi.STAGE_LIFTOVER)
    assert ei.value.status == abi.PLO_ERR_NO_DEVICE


def test_integration_doc_binds_every_declared_symbol():
    """INTEGRATION.md shows the reference-side binding a maintainer would add: every entry point the headers declare appears in it"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in declared_symbols() if not re.search(r"\b" + s + r"\b", doc)]
    assert not missing, missing
