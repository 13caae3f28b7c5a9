"""Pins the CPU restatement (oracle/) against every known-answer vector the reference's unit tests hold for the
hot path (tests/golden/reference_vectors.json).  CPU only."""
import numpy as np
import pytest

from portello_amd import cigar as cg


def _map(oracle, m, ihc=False):
    if m is None:
        return np.zeros(0, np.uint64), np.zeros(0, np.int64)
    return oracle.map_build(m["pos"], cg.encode(m["cigar"]), ihc)


def test_liftover_vectors(golden, oracle):
    for v in golden["liftover"]:
        keys, vals = _map(oracle, v["map"])
        r = oracle.liftover_read_alignment(keys, vals, v["start"], cg.encode(v["cigar"]))
        if v["expect"] is None:
            assert r is None, v["id"]
        else:
            assert r is not None, v["id"]
            assert r[0] == v["expect"]["pos"], v["id"]
            assert cg.decode(r[1]) == v["expect"]["cigar"], v["id"]


def test_simplify_vectors(golden, oracle):
    for v in golden["simplify"]:
        r = oracle.simplify_alignment_indels(v["pos"], cg.encode(v["cigar"]), v["ref"].encode(), v["read"].encode())
        assert r is not None
        assert r[0] == v["expect"]["pos"], v["id"]
        assert cg.decode(r[1]) == v["expect"]["cigar"], v["id"]


def test_shift_vectors(golden, oracle):
    for v in golden["shift"]:
        r = oracle.shift_indels(v["dir"], v["pos"], cg.encode(v["cigar"]), v["ref"].encode(), v["read"].encode())
        assert r is not None
        assert r[0] == v["expect"]["pos"], v["id"]
        if v["expect"]["cigar"] is not None:
            assert cg.decode(r[1]) == v["expect"]["cigar"], v["id"]


def test_homology_vectors(golden, oracle):
    for v in golden["homology"]:
        r = oracle.indel_breakend_homology(v["ref"].encode(), v["ref_range"], v["read"].encode(), v["read_range"])
        assert list(r) == v["expect"], v["id"]


def test_cigar_helper_vectors(golden, oracle):
    for v in golden["compress"]:
        assert cg.decode(oracle.compress_cigar(cg.encode(v["cigar"]))) == v["expect"], v["id"]
    for v in golden["edge_cleanup"]:
        shift, c = oracle.clean_up_cigar_edge_indels(cg.encode(v["cigar"]))
        assert shift == v["expect_shift"]
        assert cg.decode(c) == v["expect"]
    L = oracle.lib()
    for v in golden["position_walk"]:
        ref_pos, read_pos = v["ref_start"], v["read_start"]
        for i, c in enumerate(cg.encode(v["cigar"])):
            read_pos += L.orc_cigarseg_read_offset(int(c), int(v["ignore_hard_clip"]))
            ref_pos += L.orc_cigarseg_ref_offset(int(c))
            if v["expect_ref"] is not None:
                assert ref_pos == v["expect_ref"][i], v["id"]
            assert read_pos == v["expect_read"][i], v["id"]
    for v in golden["clip_positions"]:
        assert list(oracle.read_clip_positions(cg.encode(v["cigar"]), v["ignore_hard_clip"])) == v["expect"], v["id"]
    for v in golden["alignment_end"]:
        assert v["pos"] + cg.ref_len(cg.encode(v["cigar"])) == v["expect"]


def test_tree_map_vectors(golden, oracle):
    for v in golden["tree_map"]:
        keys, vals = oracle.map_build(v["pos"], cg.encode(v["cigar"]), v["ignore_hard_clip"])
        for read_pos, expect in v["lookups"]:
            assert oracle.map_get_ref_pos(keys, vals, read_pos) == expect, v["id"]
        got = oracle.map_get_ref_range(keys, vals, v["range"]["a"], v["range"]["b"])
        assert [list(x) for x in got] == v["range"]["expect"], v["id"]


def test_rev_comp_vectors(golden, oracle):
    for v in golden["rev_comp"]:
        assert oracle.rev_comp(v["seq"].encode()) == v["expect"].encode()


def test_decode_bam4(oracle):
    packed = bytes([0x12, 0x48, 0xF0])
    assert oracle.decode_bam4(packed, 5) == b"ACGTN"


def test_map_builder_shapes(oracle):
    """Consequences of the builder spelled out in SURVEY.md A.5: leading clips leave no entry, a contig->ref
    deletion yields two adjacent Some blocks, the last entry is always None."""
    keys, vals = oracle.map_build(1000, cg.encode("30S70M"), False)
    assert list(keys) == [30, 100] and list(vals) == [1000, oracle.NONE]
    keys, vals = oracle.map_build(1000, cg.encode("10M1D5M"), False)
    assert list(keys) == [0, 10, 15] and list(vals) == [1000, 1011, oracle.NONE]
    keys, vals = oracle.map_build(1000, cg.encode("5M10I10M"), False)
    assert list(keys) == [0, 5, 15, 25] and list(vals) == [1000, oracle.NONE, 1005, oracle.NONE]
    keys, vals = oracle.map_build(0, cg.encode(""), False)
    assert len(keys) == 0
