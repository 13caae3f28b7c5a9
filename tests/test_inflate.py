"""The device's DEFLATE decoder (portello_amd/csrc/inflate.hpp: one GPU thread per BGZF block) executed on the host against
zlib: stored / fixed / dynamic blocks, all levels, multi-block streams, and corrupted input (an error code, never a crash or a
write outside the output)."""
import ctypes as C
import zlib

import numpy as np
import pytest

import emu_lib


def _inflate(comp: bytes, out_len: int, slack: int = 0, wave: bool = False):
    L = emu_lib.lib()
    L.emu_inflate.restype = C.c_int
    L.emu_inflate.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint32)]
    L.emu_inflate_wave.restype = C.c_int
    L.emu_inflate_wave.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint]
    out = (C.c_uint8 * (out_len + slack + 16))()
    for k in range(16):
        out[out_len + slack + k] = 0xA5  # guard
    w = C.c_uint32(0)
    if wave:  # the 64 lanes of an emulated wave decode together (lane-strided table fills and match copies), lanes shuffled
        rc = L.emu_inflate_wave(comp, len(comp), out, out_len + slack, C.byref(w), 4711)
    else:
        rc = L.emu_inflate(comp, len(comp), out, out_len + slack, C.byref(w))
    assert bytes(out[out_len + slack:out_len + slack + 16]) == b"\xa5" * 16, "write beyond the output buffer"
    return rc, bytes(out[:w.value]) if rc == 0 else b""


def _payloads():
    rng = np.random.default_rng(3)
    yield b""
    yield b"a"
    yield b"ACGT" * 5000
    yield rng.integers(0, 256, 40000, dtype=np.uint8).tobytes()  # incompressible
    yield bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), 65280))
    yield (b"the quick brown fox " * 700)[:13001]
    q = rng.integers(0, 94, 30000, dtype=np.uint8).tobytes()
    yield q + q[::-1] + b"\0" * 3000 + q[:999]


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                            (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_against_zlib(level, strategy):
    for data in _payloads():
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        comp = c.compress(data) + c.flush()
        rc, out = _inflate(comp, len(data))
        assert rc == 0 and out == data, (level, strategy, len(data), rc)
        # several deflate blocks in one stream (sync flushes insert empty stored blocks)
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        third = max(1, len(data) // 3)
        comp = c.compress(data[:third]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(data[third:]) + c.flush()
        rc, out = _inflate(comp, len(data), slack=7)
        assert rc == 0 and out == data


def test_corrupt_input_is_an_error_not_a_crash():
    rng = np.random.default_rng(8)
    data = (b"GATTACA" * 3000) + rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    comp = zlib.compressobj(6, zlib.DEFLATED, -15).compress(data) + zlib.compressobj(6, zlib.DEFLATED, -15).flush()
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    assert _inflate(comp, len(data))[0] == 0
    assert _inflate(comp[: len(comp) // 2], len(data))[0] < 0           # truncated input
    assert _inflate(comp, len(data) - 100)[0] < 0                        # output too small
    n_err = 0
    for k in range(300):  # random bit flips: mostly errors, sometimes a different valid stream -- never a crash / overrun
        b = bytearray(comp)
        i = int(rng.integers(0, len(b)))
        b[i] ^= 1 << int(rng.integers(0, 8))
        rc, out = _inflate(bytes(b), len(data))
        n_err += rc < 0 or out != data
    assert n_err > 250
    for k in range(100):  # random garbage
        g = rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8).tobytes()
        _inflate(g, 1000)


@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)])
def test_wave_cooperative_decode_against_zlib(level, strategy):
    """the same decoder run by 64 emulated lanes at once, as the GPU kernel runs it"""
    rng = np.random.default_rng(5)
    q = rng.integers(0, 94, 3000, dtype=np.uint8).tobytes()
    for data in (b"", b"ACGT" * 700 + b"N" * 300, q + q[::-1] + q[:500] + bytes(200), rng.integers(0, 256, 2500, dtype=np.uint8).tobytes()):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        comp = c.compress(data) + c.flush()
        rc, out = _inflate(comp, len(data), wave=True)
        assert rc == 0 and out == data, (level, strategy, len(data), rc)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(q * 3) + c.flush()
    assert _inflate(comp[: len(comp) // 2], len(q) * 3, wave=True)[0] < 0


def _big_payloads():
    """block-sized inputs that make the wave's LDS windows wrap: the 8 KiB output ring (write-back in 4 KiB halves), matches
    nearer and farther than the ring holds, stored runs that bypass the ring and are matched into afterwards"""
    rng = np.random.default_rng(17)
    text = (b"@read/%d/ccs\tACGTTGCA\tRG:Z:x\tnp:i:12\n" * 40)
    rnd = rng.integers(0, 256, 21000, dtype=np.uint8).tobytes()          # incompressible: zlib emits stored blocks for it
    qual = bytes(rng.choice(np.arange(33, 74, dtype=np.uint8), 15000))
    # far repeats: 9 000, 20 000 and 31 000 bytes back; near ones; an overlapping run
    a = text + qual + rnd[:9000] + text + b"A" * 700 + qual[:4000] + rnd[:3000] + qual[5000:9000] + text
    yield a[:65280]
    yield rnd + rnd[100:8000] + b"xyz" * 50 + rnd[20000:] + rnd[:500]   # matches into (and across the end of) stored runs
    yield (bytes(rng.integers(0, 4, 65280, dtype=np.uint8) + 65))        # 2 bits of entropy per byte: long Huffman-only stretches
    yield rng.integers(0, 256, 65280, dtype=np.uint8).tobytes()


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_wave_decode_of_full_blocks(level):
    for data in _big_payloads():
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        rc, out = _inflate(comp, len(data), wave=True)
        assert rc == 0 and out == data, (level, len(data), rc)
        rc, out = _inflate(comp, len(data))
        assert rc == 0 and out == data
        # a sync flush in the middle: an empty stored block between two dynamic ones
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(data[:30000]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(data[30000:]) + c.flush()
        rc, out = _inflate(comp, len(data), wave=True)
        assert rc == 0 and out == data


def test_wave_decode_rejects_truncated_and_damaged_blocks():
    rng = np.random.default_rng(23)
    data = next(_big_payloads())
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    assert _inflate(comp[:len(comp) * 2 // 3], len(data), wave=True)[0] < 0
    assert _inflate(comp, len(data) - 5000, wave=True)[0] < 0
    bad = 0
    for k in range(12):
        b = bytearray(comp)
        i = int(rng.integers(0, len(b)))
        b[i] ^= 1 << int(rng.integers(0, 8))
        rc, out = _inflate(bytes(b), len(data), wave=True)
        bad += rc < 0 or out != data
    assert bad >= 10


def test_crc32_by_a_wave_equals_zlib():
    """k_bgzf_crc's crc32_wave (inflate.hpp): 64 chunk CRCs combined with x^(8 n) products over GF(2) -- against zlib.crc32 for every
    length around the chunking's edges (0, 1, 63, 64, 65, ... a full 64 KiB block) and odd alignments"""
    import ctypes as C
    import zlib

    L = emu_lib.lib()
    L.emu_crc32_wave.restype = C.c_uint32
    L.emu_crc32_wave.argtypes = [C.c_void_p, C.c_uint32, C.c_uint]
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, size=65536 + 64, dtype=np.uint8)
    for n in (0, 1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 1000, 4095, 4096, 4097, 40000, 65535, 65536):
        for off in (0, 1, 3):
            buf = np.ascontiguousarray(data[off:off + n])
            got = L.emu_crc32_wave(buf.ctypes.data, n, 0 if n % 2 else 11)
            assert got == (zlib.crc32(buf.tobytes()) & 0xffffffff), (n, off)
