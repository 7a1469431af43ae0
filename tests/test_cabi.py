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


REC_BYTES = 168 * 4
WG_GROUPS = 32
WNAF_POSITIONS = 31 * 32 + 2


def _check_layout(regs, total):
    """regions in workspace order, 256-byte aligned, disjoint, inside the total"""
    end = 0
    for name, off, nbytes in regs:
        assert off % 256 == 0 and off >= end, (name, off, end)
        end = off + nbytes
    assert end == total
    return {name: (off, nbytes) for name, off, nbytes in regs}


@pytest.mark.parametrize("n_ct", [1, 63, 64, 65, 16384])
@pytest.mark.parametrize("shared", [0, 1])
def test_workspace_carving_of_decrypt_and_part_decrypt(n_ct, shared):
    """cofhe_hip_decrypt_records / cofhe_hip_part_decrypt_records: sizes and offsets of every workspace region, for operand
    counts either side of the 64-ciphertext threshold of the shared-c1 test and of a workgroup boundary.  What each kernel
    indexes (cofhe_hip.hip): k_pow_shared writes slot g0 * (tw + 2) + s, s <= tw + 1, for EVERY group g0 of its grid
    (idle groups own slots too) and its result to front record g < n_ladders; k_spread_records fills front records
    1 .. n_ct - 1 when one ladder stands for all; k_decrypt reads front record g < n_ct as its one partial decryption;
    k_wnaf_digits writes at most WNAF_POSITIONS digit bytes and the length word behind them."""
    from cofhe_amd import engine
    tw = 16
    ladders = 1 if (shared and n_ct >= 64) else n_ct
    grid_groups = (ladders + WG_GROUPS - 1) // WG_GROUPS * WG_GROUPS
    for op in ("decrypt", "part_decrypt"):
        regs, total = engine.workspace_plan(op, n_ct, shared)
        r = _check_layout(regs, total)
        assert [name for name, _, _ in regs] == ["front", "table", "digits", "maxlen", "pairctl"]
        assert r["front"][0] == 0
        assert r["front"][1] == (n_ct * REC_BYTES if op == "decrypt" else 0)
        assert r["table"][1] >= grid_groups * (tw + 2) * REC_BYTES                 # every group of the grid, not only the alive ones
        assert r["digits"][1] >= WNAF_POSITIONS and r["maxlen"][1] >= 4
        # the memset of the digits runs up to the end of the pair ladder's counts (256 ladders x 16 bytes) and must stay inside
        # the three regions, which follow each other
        assert r["maxlen"][0] >= r["digits"][0] + r["digits"][1] and r["pairctl"][0] >= r["maxlen"][0] + r["maxlen"][1]
        assert r["pairctl"][1] >= 256 * 16 and r["pairctl"][0] + 256 * 16 <= total
        # the rings of the pair ladder (4 records per ladder, at most 256 ladders) live in the table region
        assert r["table"][1] >= min(ladders, 256) * 4 * REC_BYTES
    # the generic form: a front of the caller's choosing is left alone by everything else
    regs, total = engine.workspace_plan("pow_shared", ladders, 2 * n_ct * REC_BYTES + 100)
    r = _check_layout(regs, total)
    assert r["front"] == (0, 2 * n_ct * REC_BYTES + 100) and r["table"][0] >= 2 * n_ct * REC_BYTES + 100


def test_workspace_carving_of_the_other_entry_points():
    """matrix product (tables, digit matrix, per-column op lists and counts, partial products and their tree), accumulation
    tree, one encryption chunk, fixed-base powers: disjoint regions, each at least what its kernels index"""
    from cofhe_amd import engine
    for (n, m, p, bits, w, segs) in [(16, 16, 16, 9, 5, 1), (8, 64, 64, 13, 7, 16), (256, 256, 256, 17, 8, 1), (2, 33, 2, 128, 4, 2), (1, 7, 1, 992, 2, 1)]:
        regs, total = engine.workspace_plan("scal_matmul", n, m, p, bits, w, segs)
        r = _check_layout(regs, total)
        tw = 1 << (w - 2)
        seglen = (m + segs - 1) // segs
        rcap = (bits + 2) * (seglen + 1) + 2
        assert r["table"][1] == (n * m * 2 * tw * REC_BYTES if tw > 1 else 0)          # k_pow_table: tw slots per base
        assert r["digits"][1] >= WNAF_POSITIONS * m * p                               # k_wnaf_digits: [position][j p + k]
        assert r["ops"][1] >= segs * p * rcap * 4 and r["counts"][1] >= segs * p * 4     # k_matmul_schedule: rcap words per column
        # every position holds at most one squaring and seglen products, plus the closing entry: the cap is never short
        assert rcap >= (bits + 1) * (seglen + 1) + 1
        if segs > 1:
            assert r["partial"][1] == n * segs * p * 2 * REC_BYTES                    # k_scal_matmul_wnaf: out[i][seg][k][h]
            assert r["tree"][1] == 2 * n * ((segs + 1) // 2) * 2 * p * REC_BYTES       # two levels of k_compose_pairs
        else:
            assert r["partial"][1] == 0 and r["tree"][1] == 0
    for (n, m, p) in [(16, 64, 16), (1, 4, 1), (3, 5, 4)]:
        regs, total = engine.workspace_plan("accumulate_tree", n, m, p)
        r = _check_layout(regs, total)
        half = n * ((m + 1) // 2) * 2 * p * REC_BYTES                                  # the first level is the largest
        assert r["level_a"][1] == half and r["level_b"][1] == half
    for ne, k in [(1, 128), (65536, 128), (16384, 256), (7, 8)]:
        regs, total = engine.workspace_plan("encrypt_chunk", ne, k)
        r = _check_layout(regs, total)
        cap = k // 2 + 3
        assert r["header"][1] >= 24 and r["idx"][1] >= cap * ne * 4
        assert r["level_a"][1] >= cap * ne * REC_BYTES and r["level_b"][1] >= (cap + 1) // 2 * ne * REC_BYTES
    for n, mmax in [(1, 1), (2, 330), (4, 500)]:
        regs, total = engine.workspace_plan("fixed_base", n, mmax)
        r = _check_layout(regs, total)
        assert r["level_a"][1] == n * mmax * REC_BYTES and r["level_b"][1] >= n * ((mmax + 1) // 2) * REC_BYTES
        assert r["gather"][1] >= n * 8 + n * mmax * 4
    with pytest.raises(Exception):
        engine.workspace_plan("no_such_op", 1)
