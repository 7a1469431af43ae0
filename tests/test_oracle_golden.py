"""CPU: the C++/GMP oracle against the committed pure-Python golden vectors, plus the
algebraic identities that hold in any class group."""
import json
import os
import struct
import sys

import pytest

import oracle_lib as O
from conftest import ROOT, load_json

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as P  # noqa: E402


def hx(s):
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


def test_add_valid(golden):
    prm, vec = golden
    d = hx(prm["delta"])
    v = vec["add_valid"]
    for mode in (0, 1):
        out = O.add(d, bytes.fromhex(v["ct1"]), bytes.fromhex(v["ct2"]), mode)
        assert out == bytes.fromhex(v["out"])


def test_add_edge(golden):
    prm, vec = golden
    d = hx(prm["delta"])
    v = vec["add_edge"]
    for mode in (0, 1):
        out = O.add(d, bytes.fromhex(v["ct1"]), bytes.fromhex(v["ct2"]), mode)
        assert out == bytes.fromhex(v["out"])
    assert O.check_tensor(d, bytes.fromhex(v["out"])) == 1


def test_scal_1d(golden):
    prm, vec = golden
    d = hx(prm["delta"])
    v = vec["scal_1d"]
    out = O.scal_1d(d, bytes.fromhex(v["s"]), bytes.fromhex(v["cts"]))
    assert out == bytes.fromhex(v["out"])


def test_scal_2d(golden):
    prm, vec = golden
    d = hx(prm["delta"])
    v = vec["scal_2d"]
    out = O.scal_2d(d, bytes.fromhex(v["s"]), bytes.fromhex(v["cts"]), bytes.fromhex(v["zero"]))
    assert out == bytes.fromhex(v["out"])


def test_qfi_nupow_matches_pow(golden):
    """shared-table wNAF (qfi.inl:1-135) == plain binary powering, incl. negative / zero / wide exponents"""
    prm, vec = golden
    d = hx(prm["delta"])
    v = vec["scal_1d"]
    shape, cts = P.deserialize_ciphertext_tensor(bytes.fromhex(v["cts"]))
    exps = [hx(e) for e in v["s_list"]] + [127, 128, 129, 8191, 12345678901234567890, -97]
    base = cts[0]
    n = len(exps)
    from golden.make_golden import pt_bytes
    s = pt_bytes([n], exps)
    got = O.qfi_nupow(d, s, P.serialize_ciphertext_tensor([1], [base]))
    want = O.scal_1d(d, s, P.serialize_ciphertext_tensor([n], [base] * n))
    assert got == want


def test_errors(golden):
    prm, vec = golden
    d = hx(prm["delta"])
    v = vec["add_valid"]
    sh, cts = P.deserialize_ciphertext_tensor(bytes.fromhex(v["ct1"]))
    with pytest.raises(ValueError, match="Tensor shapes must be equal"):
        O.add(d, bytes.fromhex(v["ct1"]), P.serialize_ciphertext_tensor([4], cts))


def test_group_laws_tiny():
    prm = load_json("params_tiny_k8.json")
    d = hx(prm["delta"])
    rng = P.SplitMix64(99)
    xs = [(P.random_form(d, rng, 20, 16), P.random_form(d, rng, 20, 16)) for _ in range(24)]
    ys = xs[5:] + xs[:5]
    zs = xs[11:] + xs[:11]
    S = lambda v: P.serialize_ciphertext_tensor([len(v)], v)
    xy = O.add(d, S(xs), S(ys))
    yx = O.add(d, S(ys), S(xs))
    assert xy == yx
    assert O.add(d, xy, S(zs)) == O.add(d, S(xs), O.add(d, S(ys), S(zs)))
    # python model agrees element-wise
    assert xy == S(P.add_tensor(xs, ys))
    # x * x^-1 = identity
    inv = [(P.inverse(a), P.inverse(b)) for a, b in xs]
    ident = P.identity(d)
    assert O.add(d, S(xs), S(inv)) == S([(ident, ident)] * len(xs))


def test_serialization_roundtrip_and_quirks():
    # zero gets the sign flag and a 1-byte slot; a value of exactly 8 bits gets 2 bytes
    f0 = P.Form(1, 0, 255)
    f1 = P.Form(256, -1, 65535)
    data = P.serialize_ciphertext_tensor([1], [(f0, f1)])
    (ndim,) = struct.unpack_from("<I", data, 0)
    assert ndim == 1
    offs = struct.unpack_from("<6Q", data, 8)
    m = (1 << 63) - 1
    assert [o & m for o in offs] == [0, 1, 2, 4, 6, 7]
    assert [o >> 63 for o in offs] == [0, 1, 0, 0, 1, 0]
    assert len(data) == 8 + 48 + 10
    assert P.deserialize_ciphertext_tensor(data) == ([1], [(f0, f1)])


def test_plaintext_encoding():
    g = load_json("plaintext_k128.json")
    for c in g["cases"]:
        assert O.make_plaintext(c["x"], 128) == hx(c["pt"]), c
        assert O.get_float(hx(c["pt"]), 128) == pytest.approx(c["back"], rel=1e-6)


def test_dlog_peeling_matches_generic():
    """the bit-peeling discrete log in <f> that the GPU decrypt kernel uses == Pohlig-Hellman"""
    cl = P.CLHSM2k(128, 10, seed=5, disc_bits=60)
    for m in list(range(0, 40)) + [2 ** 10 - 1, 2 ** 9, 2 ** 9 + 2, 768, 1000]:
        g = P.power(cl.f, m, cl.delta)
        assert cl.dlog_in_F(g) == m % cl.M
        assert cl.dlog_in_F_peel(g) == m % cl.M
    prm = load_json("params_s128_k128.json")
    cl2 = P.CLHSM2k.__new__(P.CLHSM2k)
    cl2.k, cl2.M, cl2.delta = 128, 1 << 128, hx(prm["delta"])
    cl2.f = P.Form(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    cl2.id = P.identity(cl2.delta)
    rng = P.SplitMix64(3)
    for m in [0, 1, 2, 6, rng.bits(128), 1 << 127]:
        assert cl2.dlog_in_F_peel(P.power(cl2.f, m, cl2.delta)) == m


@pytest.mark.parametrize("name", ["s128_k128", "tiny_k8"])
def test_threshold_fixtures(name):
    """secret-key shares reconstruct sk with lambda = (1, -1, ..., -1) for every threshold set; the
    C++ oracle's nupow reproduces the partial decryptions c1^share of the fixtures"""
    from itertools import combinations
    prm, th = load_json("params_%s.json" % name), load_json("threshold_%s.json" % name)
    d, sk = hx(prm["delta"]), hx(prm["sk"])
    cts = bytes.fromhex(th["cts"])
    for case in th["cases"]:
        t, n = case["t"], case["n"]
        shares = P.share_secret_key(sk, t, n, [hx(r) for r in case["rho_tail"]])
        assert [[hx(x) for x in sp] for sp in case["shares"]] == shares
        lam = P.combine_lambda(t)
        sets = list(combinations(range(n), t))
        for comb in sets:
            mine = [shares[p][[c for c in sets if p in c].index(comb)] for p in comb]
            assert sum(l * s for l, s in zip(lam, mine)) == sk
        for sh, blob in zip(case["used_shares"], case["parts"]):
            share = hx(sh)
            # 1-D scal with the share as every exponent: component c1 of the result = c1^share
            from golden.make_golden import pt_bytes
            got = O.scal_1d(d, pt_bytes([4], [share] * 4), _reshape(cts, [4]))
            _, got_cts = P.deserialize_ciphertext_tensor(got)
            assert P.serialize_form_tensor([2, 2], [c[0] for c in got_cts]) == bytes.fromhex(blob)


def _reshape(data, shape):
    (ndim,) = struct.unpack_from("<I", data, 0)
    return struct.pack("<I", len(shape)) + b"".join(struct.pack("<I", x) for x in shape) + data[4 + 4 * ndim:]
