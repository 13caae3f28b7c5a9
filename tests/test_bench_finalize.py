"""bench.py's rule that a result whose records differ from the checker's is not a measurement."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("plo_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_parity_failure_voids_the_value():
    b = _bench()
    r = {"value": 1.0, "parity_sample_ok": False}
    assert b.finalize(r) != 0
    assert r["value"] is None and r["parity_failed"] is True and r["value_unverified"] == 1.0
    r = {"value": 2.0, "parity_sample_ok": True, "end_to_end": {"records_verified": 0, "verification": {"ok": False}}}
    assert b.finalize(r) != 0 and r["value"] is None
    r = {"value": 2.0, "end_to_end": {"records_verified": 10, "verification": {}, "device_finished": {"value": 1.0, "records_verified": 0}}}
    assert b.finalize(r) != 0 and r["value"] is None
    r = {"value": 2.0, "verify": {"gathered_equals_single_gpu_result": False}}
    assert b.finalize(r) != 0 and r["value"] is None


def test_clean_result_keeps_its_value_and_says_what_it_is():
    b = _bench()
    r = {"value": 3.0, "parity_sample_ok": True, "end_to_end": {"records_verified": 6000, "verification": {"ok": True}},
         "verify": {"gathered_equals_single_gpu_result": True}}
    assert b.finalize(r) == 0
    assert r["value"] == 3.0 and "parity_failed" not in r
    assert r["value_kind"] == "hbm_resident_kernel_rate"
    # a verification that could not run (records_verified None) is not a parity failure
    r = {"value": 3.0, "end_to_end": {"records_verified": None}}
    assert b.finalize(r) == 0
