"""Builds libportello_liftover.so (HIP, gfx950) in-tree.  `python -m portello_amd.build`"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libportello_liftover.so")
SOURCES = ["engine.hip", "bam_host.cpp", "phase1.cpp", "gather_rccl.cpp"]
LIBS = ["-lz", "-ldl"]
HEADERS = ["plo_wave.hpp", "lift_core.hpp", "lane_core.hpp", "lane_stream.hpp", "inflate.hpp", "finish_core.hpp", "lift_types.hpp", "index_pack.hpp", "enumerate.hpp", "bam_internal.hpp"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def source_hash() -> str:
    """identifies the kernel sources a measurement was made with (profiles/hbm_traffic.json entries carry it: bench.py reports a
    counter-derived traffic figure only for the very code, workload and read count it was collected on)"""
    import hashlib

    h = hashlib.sha256()
    # the device code: engine.hip and what it includes (the host-side BAM / phase-1 sources do not enter a kernel's traffic)
    for f in sorted(["engine.hip"] + [x for x in HEADERS if x != "bam_internal.hpp"]):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def build_timing() -> str:
    """instrumented variant (per-phase s_memtime accumulation) used by tools/tune.py only"""
    out = os.path.join(HERE, "libportello_liftover_timing.so")
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DPLO_PHASE_TIMING", "-I" + CSRC, "-o", out,
           os.path.join(CSRC, "engine.hip"), os.path.join(CSRC, "bam_host.cpp"), os.path.join(CSRC, "phase1.cpp"), os.path.join(CSRC, "gather_rccl.cpp")] + LIBS
    subprocess.check_call(cmd)
    return out


def build(force: bool = False, verbose: bool = False, extra: list[str] | None = None) -> str:
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "portello_liftover.h"),
                                                                os.path.join(HERE, "..", "include", "portello_bam.h")]
    stale = (not os.path.exists(LIB)) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)
    if not (force or stale):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical", "-Wno-unused-const-variable",
           "-I" + CSRC, "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES] + LIBS + (extra or [])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, extra=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force"]))
