"""GPU: HIP kernels (through the C ABI) against the C++/GMP oracle and the golden vectors."""
import json
import os
import sys

import pytest

import oracle_lib as O
from conftest import ROOT, load_json

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as P  # noqa: E402

pytestmark = pytest.mark.gpu


def hx(s):
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


_engines = {}


def engine(delta):
    from cofhe_amd import Engine
    if delta not in _engines:
        _engines[delta] = Engine(delta)
    return _engines[delta]


def test_golden_add(golden):
    prm, vec = golden
    d = hx(prm["delta"])
    E = engine(d)
    for key in ("add_valid", "add_edge"):
        v = vec[key]
        out = E.add_ciphertext_tensors(bytes.fromhex(v["ct1"]), bytes.fromhex(v["ct2"]))
        assert out == bytes.fromhex(v["out"]), key


def test_golden_scal_1d(golden):
    prm, vec = golden
    d = hx(prm["delta"])
    v = vec["scal_1d"]
    out = engine(d).scal_ciphertext_tensors(bytes.fromhex(v["s"]), bytes.fromhex(v["cts"]))
    assert out == bytes.fromhex(v["out"])


def test_golden_scal_2d(golden):
    prm, vec = golden
    d = hx(prm["delta"])
    v = vec["scal_2d"]
    out = engine(d).scal_ciphertext_tensors(bytes.fromhex(v["s"]), bytes.fromhex(v["cts"]), bytes.fromhex(v["zero"]))
    assert out == bytes.fromhex(v["out"])


def _random_tensor(d, n, seed, nbase=24):
    rng = P.SplitMix64(seed)
    base = [P.random_form(d, rng) for _ in range(nbase)]
    cts = []
    for i in range(n):
        a = base[rng.below(nbase)]
        b = base[rng.below(nbase)]
        cts.append((a, b))
    return cts


def test_add_16x16_vs_oracle(params128):
    d = hx(params128["delta"])
    E = engine(d)
    x = P.serialize_ciphertext_tensor([16, 16], _random_tensor(d, 256, 1))
    y = P.serialize_ciphertext_tensor([16, 16], _random_tensor(d, 256, 2))
    got = E.add_ciphertext_tensors(x, y)
    assert got == O.add(d, x, y)
    # chain of 5 like the harness (benchmarks/local.cpp:99-117)
    for _ in range(4):
        got = E.add_ciphertext_tensors(got, y)
    sec, want = O.time_matadd_chain(d, x, y, 5, want_out=True)
    assert got == want


def test_errors(params128):
    from cofhe_amd import CofheHipError
    d = hx(params128["delta"])
    E = engine(d)
    cts = _random_tensor(d, 4, 3, nbase=4)
    a = P.serialize_ciphertext_tensor([2, 2], cts)
    b = P.serialize_ciphertext_tensor([4], cts)
    with pytest.raises(CofheHipError, match="Tensor shapes must be equal"):
        E.add_ciphertext_tensors(a, b)
