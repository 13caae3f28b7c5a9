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


def test_ctypes_struct_layout_matches_header():
    # sizes implied by the header on LP64
    assert C.sizeof(abi.PloBatchOut) == 8 + 10 * 8 + 8
    # struct_size + 4 floats + 2 u32 | 3 u64 | 4 + 2 + 2 + 2 + 2 + 1 four-byte fields (+ tail padding to 8)
    assert C.sizeof(abi.PloTiming) == 8 * 4 + 3 * 8 + 14 * 4
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
