"""Independent reader of BAM files for the tests: Python's gzip (BGZF = concatenated gzip members) + struct.
TEST INFRASTRUCTURE ONLY -- shares no code with portello_amd/csrc/bam_host.cpp."""
import gzip
import struct


def bgzf_blocks(path):
    """(block sizes, uncompressed sizes) of every BGZF member, checking the BC extra field"""
    data = open(path, "rb").read()
    out, i = [], 0
    while i < len(data):
        assert data[i:i + 4] == b"\x1f\x8b\x08\x04", "not a BGZF member"
        xlen = struct.unpack_from("<H", data, i + 10)[0]
        extra = data[i + 12:i + 12 + xlen]
        bsize, j = None, 0
        while j < len(extra):
            si1, si2, slen = extra[j], extra[j + 1], struct.unpack_from("<H", extra, j + 2)[0]
            if (si1, si2) == (66, 67):
                bsize = struct.unpack_from("<H", extra, j + 4)[0] + 1
            j += 4 + slen
        assert bsize is not None
        isize = struct.unpack_from("<I", data, i + bsize - 4)[0]
        out.append((bsize, isize))
        i += bsize
    assert i == len(data)
    return out


def read_bam(path):
    """(header text, [(name, length)], [record bytes with block_size prefix])"""
    with gzip.open(path, "rb") as fh:
        d = fh.read()
    assert d[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<I", d, 4)[0]
    text = d[8:8 + l_text].decode()
    o = 8 + l_text
    n_ref = struct.unpack_from("<I", d, o)[0]
    o += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<I", d, o)[0]
        name = d[o + 4:o + 4 + ln - 1].decode()
        o += 4 + ln
        refs.append((name, struct.unpack_from("<I", d, o)[0]))
        o += 4
    recs = []
    while o < len(d):
        bs = struct.unpack_from("<I", d, o)[0]
        recs.append(d[o:o + 4 + bs])
        o += 4 + bs
    assert o == len(d)
    return text, refs, recs
