"""ctypes binding of tests/emu/libplo_emu.so: the device algorithm executed under the CPU wave64 emulator.
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from portello_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
_LIB = os.path.join(_HERE, "emu", "libplo_emu.so")
_lib = None


def build(force=False, sanitize=False):
    srcs = [os.path.join(_HERE, "emu", "emu_harness.cpp"), os.path.join(_HERE, "emu", "plo_wave.hpp")] + [
        os.path.join(ROOT, "portello_amd", "csrc", f) for f in ("lift_core.hpp", "lane_core.hpp", "lift_types.hpp", "index_pack.hpp", "enumerate.hpp")]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-I" + os.path.join(_HERE, "emu"),
               "-o", _LIB, srcs[0]]
        if sanitize:
            cmd[1:1] = ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"]
        subprocess.check_call(cmd)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.emu_liftover_batch.restype = C.c_int
        L.emu_liftover_batch.argtypes = [C.POINTER(abi.PloIndexDesc), C.POINTER(abi.PloBatchIn), C.c_uint32, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_uint, C.c_int, C.POINTER(abi.PloBatchOut), C.POINTER(C.c_ulonglong)]
        L.emu_free_last.restype = None
        _lib = L
    return _lib


def liftover_batch(index: abi.IndexData, batch: abi.BatchData, stages=abi.STAGES_ALL, cap=768, window=256, big_thresh=256,
                   big_cap=1 << 16, order_seed=0, lane_max_in=40):
    d = index.to_desc()
    b = batch.to_desc()
    out = abi.PloBatchOut()
    counters = (C.c_ulonglong * 24)()
    rc = lib().emu_liftover_batch(C.byref(d), C.byref(b), stages, cap, window, big_thresh, big_cap, order_seed, lane_max_in, C.byref(out), counters)
    res = abi.result_from_out(out)
    lib().emu_free_last()
    return rc, res, list(counters)
