"""CIGAR helpers on the BAM encoding (``len << 4 | op``) shared by the host API, the tests and the generator.

Op codes follow rust-htslib's ``Cigar`` enum / the BAM specification: M0 I1 D2 N3 S4 H5 P6 =7 X8.
"""
from __future__ import annotations

import re
from typing import Iterable, List, Sequence, Tuple

import numpy as np

OPS = "MIDNSHP=X"
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)
_OP_CODE = {c: i for i, c in enumerate(OPS)}
_CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=X])")

# which ops advance the reference / the read (lib/rust-vc-utils/src/bam_utils/cigar/mod.rs:26-47,
# ignore_hard_clip = false as on every hot-path call site)
REF_CONSUMING = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1], dtype=np.uint32)
READ_CONSUMING = np.array([1, 1, 0, 0, 1, 1, 0, 1, 1], dtype=np.uint32)


def encode(ops: Iterable[Tuple[str, int]] | str) -> np.ndarray:
    """``"10M2D"`` or ``[("M", 10), ("D", 2)]`` -> uint32 array."""
    if isinstance(ops, str):
        pairs = [(m.group(2), int(m.group(1))) for m in _CIGAR_RE.finditer(ops)]
        if "".join(f"{n}{c}" for c, n in pairs) != ops:
            raise ValueError(f"malformed CIGAR string: {ops!r}")
    else:
        pairs = list(ops)
    return np.array([(n << 4) | _OP_CODE[c] for c, n in pairs], dtype=np.uint32)


def decode(cigar: Sequence[int]) -> str:
    return "".join(f"{int(c) >> 4}{OPS[int(c) & 0xF]}" for c in cigar)


def to_pairs(cigar: Sequence[int]) -> List[Tuple[str, int]]:
    return [(OPS[int(c) & 0xF], int(c) >> 4) for c in cigar]


def ref_len(cigar: np.ndarray) -> int:
    """get_cigar_ref_offset (cigar/mod.rs:174-180)."""
    cigar = np.asarray(cigar, dtype=np.uint32)
    return int(((cigar >> 4) * REF_CONSUMING[cigar & 0xF]).sum())


def read_len(cigar: np.ndarray) -> int:
    """get_cigar_read_offset(cigar, ignore_hard_clip=false) (cigar/mod.rs:164-170)."""
    cigar = np.asarray(cigar, dtype=np.uint32)
    return int(((cigar >> 4) * READ_CONSUMING[cigar & 0xF]).sum())
