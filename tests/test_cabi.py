"""CPU: the C-ABI library loads, exports every symbol include/cofhe_hip.h declares, and its
host-only format conversion matches the reference's binary tensor format.  No kernel runs."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, load_json

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as P  # noqa: E402


def hx(s):
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from cofhe_amd import load_library
    return load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "cofhe_hip.h")).read()
    names = set(re.findall(r"\b(cofhe_hip_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib, n), n
    assert lib.cofhe_hip_record_words() == 168


def test_format_roundtrip_matches_reference_format(lib, golden):
    prm, vec = golden
    from cofhe_amd.engine import Engine
    eng = Engine.__new__(Engine)          # conversion helpers need no GPU context
    eng.L = lib
    for key in ("add_valid", "add_edge", "scal_1d"):
        data = bytes.fromhex(vec[key]["out"])
        shape, recs = eng.bytes_to_records(data)
        sh2, cts = P.deserialize_ciphertext_tensor(data)
        assert shape == sh2
        import simlib as S
        want = np.concatenate([S.form_record(f.a, f.b, f.c) for ct in cts for f in ct])
        assert np.array_equal(recs, want)
        assert eng.records_to_bytes(recs, shape) == data
    s = bytes.fromhex(vec["scal_1d"]["s"])
    shape, ex = eng.bytes_to_exponents(s)
    vals = [hx(e) for e in vec["scal_1d"]["s_list"]]
    ex = ex.reshape(-1, 32)
    for row, v in zip(ex, vals):
        assert int.from_bytes(row[:31].tobytes(), "little") == abs(v)
        assert int(row[31]) == (1 if v < 0 else 0)


def test_part_decryption_tensor_format(lib):
    """serialize_part_decryption_result_tensor layout (3 integers per element) round-trips"""
    th = load_json("threshold_s128_k128.json")
    from cofhe_amd.engine import Engine
    import simlib as S
    eng = Engine.__new__(Engine)
    eng.L = lib
    data = bytes.fromhex(th["cases"][0]["parts"][0])
    shape, recs = eng.pdr_bytes_to_records(data)
    assert shape == [2, 2] and recs.size == 4 * 168
    _, cts = P.deserialize_ciphertext_tensor(P.serialize_ciphertext_tensor([2], [(S_form(recs, 0), S_form(recs, 1)),
                                                                                 (S_form(recs, 2), S_form(recs, 3))]))
    assert P.serialize_form_tensor([2, 2], [f for ct in cts for f in ct]) == data
    assert eng.pdr_records_to_bytes(recs, shape) == data
    with pytest.raises(Exception):
        eng.pdr_records_to_bytes(recs, [3])            # shape / record count mismatch


def S_form(recs, i):
    import simlib as S
    a, b, c = S.record_form(recs[i * 168:(i + 1) * 168])
    return P.Form(a, b, c)


def test_malformed_buffers_are_rejected(lib):
    from cofhe_amd.engine import Engine, CofheHipError
    eng = Engine.__new__(Engine)
    eng.L = lib
    with pytest.raises(CofheHipError):
        eng.bytes_to_records(b"\x01\x00")
    big = P.Form(1 << 1290, 1, 1 << 100)     # a beyond the 1280-bit plane
    data = P.serialize_ciphertext_tensor([1], [(big, big)])
    with pytest.raises(CofheHipError, match="outside the supported range"):
        eng.bytes_to_records(data)


def test_no_gpu_means_loud_failure(lib):
    """on a machine without a GPU the context constructor must fail, never fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cofhe_amd import Engine, CofheHipError
    prm = load_json("params_tiny_k8.json")
    with pytest.raises(CofheHipError):
        Engine(hx(prm["delta"]))


def test_shard_rows_matches_python_partition():
    """cofhe_hip_shard_rows (the C ABI's row partition) == cofhe_amd.shard.row_partition, ragged cases included;
    pure host arithmetic, no GPU"""
    import ctypes as C
    from cofhe_amd import load_library, shard
    L = load_library()
    L.cofhe_hip_shard_rows.restype = None
    for n_rows in (0, 1, 7, 128, 129, 1000, 1 << 33):
        for world in (1, 2, 3, 8):
            want = shard.row_partition(n_rows, world)
            for rank in range(world):
                r0, nl = C.c_uint64(), C.c_uint64()
                L.cofhe_hip_shard_rows(C.c_uint64(n_rows), C.c_uint32(world), C.c_uint32(rank), C.byref(r0), C.byref(nl))
                assert (r0.value, r0.value + nl.value) == want[rank], (n_rows, world, rank)
