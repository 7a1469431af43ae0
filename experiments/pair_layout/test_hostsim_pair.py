"""CPU: the lane-PAIR device arithmetic (cofhe_amd/csrc/pair.hpp) on the host pair simulator against Python integers."""
import ctypes as C
import random

import numpy as np
import pytest

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import sim2lib as S


def rnd(rng, bits):
    return rng.getrandbits(bits) if bits else 0


@pytest.fixture(params=[17, 19])
def LN(request):
    L = S.lib(request.param)
    return L, 2 * request.param


def _vals(rng, W, n):
    full = 32 * W
    sizes = [1, 31, 32, 33, 64, 32 * (W // 2) - 1, 32 * (W // 2), 32 * (W // 2) + 1, full - 40, full - 1, full]
    v = [rnd(rng, rng.choice(sizes)) for _ in range(n)]
    return v + [0, 1, (1 << full) - 1, (1 << (32 * (W // 2))) - 1, 1 << (32 * (W // 2)), (1 << full) - (1 << 32)]


def test_add_sub_cmp_bitlen(LN):
    L, W = LN
    rng = random.Random(1)
    M = 1 << (32 * W)
    xs = _vals(rng, W, 30)
    ys = list(reversed(_vals(rng, W, 30)))
    ys[0] = xs[0]
    n = len(xs)
    out = np.zeros(2 * W * n, dtype=np.uint32)
    misc = np.zeros(5 * n, dtype=np.int32)
    L.sim2_addsub(S.P(S.pack(xs, W)), S.P(S.pack(ys, W)), S.P(out), misc.ctypes.data_as(C.POINTER(C.c_int)), n)
    got = S.unpack(out, W)
    for i, (a, b) in enumerate(zip(xs, ys)):
        assert got[2 * i] == (a + b) % M and got[2 * i + 1] == (a - b) % M
        assert misc[5 * i] == (a + b) // M and misc[5 * i + 1] == (1 if a < b else 0)
        assert misc[5 * i + 2] - 1 == (a > b) - (a < b) and misc[5 * i + 3] == a.bit_length() and misc[5 * i + 4] == (1 if b == 0 else 0)


def test_shifts(LN):
    L, W = LN
    rng = random.Random(2)
    M = 1 << (32 * W)
    xs = _vals(rng, W, 10)
    n = len(xs)
    for sh in [0, 1, 7, 31]:
        out = np.zeros(4 * W * n, dtype=np.uint32)
        L.sim2_shift(S.P(S.pack(xs, W)), sh, S.P(out), n)
        got = S.unpack(out, W)
        for i, a in enumerate(xs):
            assert got[4 * i] == a >> sh and got[4 * i + 1] == (a << sh) % M
            assert got[4 * i + 2] == (a << 32) % M and got[4 * i + 3] == a >> 32


def test_lincomb(LN):
    L, W = LN
    rng = random.Random(3)
    M = 1 << (32 * W)
    xs = _vals(rng, W, 20)
    ys = list(reversed(_vals(rng, W, 20)))
    n = len(xs)
    for A, B in [(0x7FFFFFFF, 12345), (1, 0x7FFFFFFF), (65535, 1), (1, 1), (0x80000000, 0x80000000), (1, 0)]:
        r = np.zeros(W * n, dtype=np.uint32)
        s = np.zeros(W * n, dtype=np.uint32)
        tops = np.zeros(2 * n, dtype=np.uint32)
        L.sim2_lincomb(S.P(S.pack(xs, W)), S.P(S.pack(ys, W)), C.c_uint32(A), C.c_uint32(B), S.P(r), S.P(s), S.P(tops), n)
        rr, ss = S.unpack(r, W), S.unpack(s, W)
        for i, (a, b) in enumerate(zip(xs, ys)):
            assert rr[i] == (A * a - B * b) % M and ss[i] == (A * a + B * b) % M
            assert int(tops[2 * i + 1]) == (A * a + B * b) // M
            assert (A * a - B * b) == rr[i] + (int(tops[2 * i]) - B) * M


def test_mul(LN):
    L, W = LN
    rng = random.Random(4)
    xs = _vals(rng, W, 24)
    ys = list(reversed(_vals(rng, W, 24)))
    n = len(xs)
    out = np.zeros(2 * W * n, dtype=np.uint32)
    L.sim2_mul(S.P(S.pack(xs, W)), S.P(S.pack(ys, W)), S.P(out), n)
    assert S.unpack(out, 2 * W) == [a * b for a, b in zip(xs, ys)]


def test_rem_and_divexact(LN):
    L, W = LN
    rng = random.Random(5)
    full = 32 * W
    # remainder: divisors with their top bit in the same limb (the fast-path precondition), H < D
    dens, los, his, want = [], [], [], []
    for _ in range(24):
        db = rng.choice([full - 44, full - 50, full - 60, full - 33])
        d = rnd(rng, db) | (1 << (db - 1))
        h = rnd(rng, db - rng.choice([1, 5, 40]))
        lo = rnd(rng, full)
        dens.append(d); his.append(h); los.append(lo); want.append((lo + (h << full)) % d)
    n = len(dens)
    rem = np.zeros(W * n, dtype=np.uint32)
    ok = np.zeros(n, dtype=np.int32)
    L.sim2_rem(S.P(S.pack(los, W)), S.P(S.pack(his, W)), S.P(S.pack(dens, W)), S.P(rem), ok.ctypes.data_as(C.POINTER(C.c_int)), n)
    assert list(ok) == [1] * n
    assert S.unpack(rem, W) == want
    # exact division, signed through two's complement: Q = W D^-1 mod 2^(32 nq)
    ws, ds, nqs, wantq = [], [], [], []
    for i in range(40):
        db = rng.choice([full - 44, 522, 33, 64, full - 100])
        qb = rng.choice([1, 31, 32, 33, 522, 545, full // 2, full - 70])
        d = (rnd(rng, db) | (1 << (db - 1)) | 1) << rng.choice([0, 0, 1, 3, 17])
        q = rnd(rng, qb)
        nq = (qb + 1 + 31) // 32 + (i % 2)
        if nq > W or (d * q).bit_length() > 2 * full:
            continue
        neg = i % 3 == 0
        num = -d * q if neg else d * q
        ws.append(num % (1 << full)); ds.append(d); nqs.append(nq)
        wantq.append((-q if neg else q) % (1 << (32 * nq)))
    n = len(ws)
    quot = np.zeros(W * n, dtype=np.uint32)
    ok = np.zeros(n, dtype=np.int32)
    L.sim2_divexact(S.P(S.pack(ws, W)), S.P(S.pack(ds, W)), S.P(quot), np.array(nqs, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int)),
                    ok.ctypes.data_as(C.POINTER(C.c_int)), n)
    assert list(ok) == [1] * n
    assert S.unpack(quot, W) == wantq


def test_word_helpers(LN):
    L, W = LN
    rng = random.Random(6)
    xs = _vals(rng, W, 20)
    wsv = [rng.choice([3, 29, 65537, 0xFFFFFFFF, 223092870, 1 << 31, 1]) for _ in xs]
    n = len(xs)
    small, modw, rem = (np.zeros(n, dtype=np.uint32) for _ in range(3))
    quot = np.zeros(W * n, dtype=np.uint32)
    L.sim2_words(S.P(S.pack(xs, W)), S.P(np.array(wsv, dtype=np.uint32)), S.P(small), S.P(modw), S.P(quot), S.P(rem), n)
    assert [int(v) for v in small] == [x % 223092870 for x in xs]
    assert [int(v) for v in modw] == [x % w for x, w in zip(xs, wsv)]
    assert [int(v) for v in rem] == [x % w for x, w in zip(xs, wsv)]
    assert S.unpack(quot, W) == [x // w for x, w in zip(xs, wsv)]


def test_euclid(LN):
    import math
    L, W = LN
    rng = random.Random(7)
    full = 32 * W
    top = full - 44
    xs = [rnd(rng, top) | (1 << (top - 1)) for _ in range(10)]
    ys = [rnd(rng, top - rng.choice([0, 1, 4, 12])) | 1 for _ in range(10)]
    # shared factors (word-sized gcd) and a few exact-tail shapes
    xs += [(rnd(rng, top - 6) | (1 << (top - 7))) * 29, rnd(rng, 63) | (1 << 62), 97 * 3, 5, rnd(rng, 100) | (1 << 99), (2 ** 20 + 1) * 7]
    ys += [(rnd(rng, top - 8) | 1) * 29, rnd(rng, 61) | 1, 97 * 2, 5, rnd(rng, 97) | 1, 7 * 3]
    n = len(xs)
    out = np.zeros(4 * W * n, dtype=np.uint32)
    signs = np.zeros(2 * n, dtype=np.int32)
    ok = np.zeros(n, dtype=np.int32)
    L.sim2_euclid(S.P(S.pack(xs, W)), S.P(S.pack(ys, W)), -1, S.P(out), signs.ctypes.data_as(C.POINTER(C.c_int)), ok.ctypes.data_as(C.POINTER(C.c_int)), n)
    got = S.unpack(out, W)
    for i, (x0, y0) in enumerate(zip(xs, ys)):
        assert ok[i] == 1, i
        g, z, u, v = got[4 * i: 4 * i + 4]
        assert z == 0 and g == math.gcd(x0, y0)
        assert (int(signs[2 * i]) * u * y0 - g) % x0 == 0
    # partial sequences
    stop = top // 2
    L.sim2_euclid(S.P(S.pack(xs[:10], W)), S.P(S.pack(ys[:10], W)), stop, S.P(out), signs.ctypes.data_as(C.POINTER(C.c_int)),
                  ok.ctypes.data_as(C.POINTER(C.c_int)), 10)
    got = S.unpack(out, W)
    for i in range(10):
        assert ok[i] == 1
        R0, R1, C0, C1 = got[4 * i: 4 * i + 4]
        assert R1.bit_length() <= stop < R0.bit_length()
        assert (int(signs[2 * i]) * C0 * ys[i] - R0) % xs[i] == 0 and (int(signs[2 * i + 1]) * C1 * ys[i] - R1) % xs[i] == 0
    # far apart: leaves the fast path instead of looping
    L.sim2_euclid(S.P(S.pack([xs[0]], W)), S.P(S.pack([12345], W)), -1, S.P(out), signs.ctypes.data_as(C.POINTER(C.c_int)),
                  ok.ctypes.data_as(C.POINTER(C.c_int)), 1)
    assert ok[0] == 0


# ---------------------------------------------------------------------------------------- composition
import json
import os
import sys

from conftest import ROOT, load_json

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as P  # noqa: E402
import simlib as S8  # noqa: E402  (record packing helpers of the 8-lane simulator)


def hx(s):
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


def compose2(n_limbs, forms1, forms2, delta, screen=1):
    L = S.lib(n_limbs)
    n = len(forms1)
    half = ((-delta).bit_length() + 1) // 2
    ad = S8.to_limbs(-delta, 80)
    f1 = np.concatenate([S8.form_record(*f) for f in forms1])
    f2 = np.concatenate([S8.form_record(*f) for f in forms2])
    out = np.zeros(n * S8.REC_WORDS, dtype=np.uint32)
    ok = np.zeros(n, dtype=np.int32)
    L.sim2_compose(S.P(f1), S.P(f2), S.P(out), ok.ctypes.data_as(C.POINTER(C.c_int)), n, half, S.P(ad), screen)
    return [S8.record_form(out[i * S8.REC_WORDS:(i + 1) * S8.REC_WORDS]) for i in range(n)], [int(v) for v in ok]


@pytest.mark.parametrize("name,n_limbs", [("s128_k128", 17), ("s128_k256", 19)])
def test_compose_generic_forms(name, n_limbs):
    """independent random group elements: the pair fast path must handle every one of them (no fallback) and
    agree with the definition (Cohen 5.4.7 + reduction)"""
    prm = load_json("params_%s.json" % name)
    d = hx(prm["delta"])
    rng = P.SplitMix64(21)
    xs = [P.random_form(d, rng, 24, 20) for _ in range(20)]
    ys = [P.random_form(d, rng, 24, 20) for _ in range(20)]
    # squarings and inverse pairs are NOT generic (gcd beyond a word): they must be flagged, never wrong
    xs += [xs[0], xs[1]]
    ys += [xs[0], P.inverse(xs[1])]
    got, ok = compose2(n_limbs, [(f.a, f.b, f.c) for f in xs], [(f.a, f.b, f.c) for f in ys], d)
    nfast = 0
    for g, k, x, y in zip(got, ok, xs, ys):
        if k:
            w = P.compose(x, y)
            assert g == (w.a, w.b, w.c)
            nfast += 1
    assert nfast >= 20 and ok[:20] == [1] * 20


@pytest.mark.parametrize("name,n_limbs", [("s128_k128", 17), ("s128_k256", 19)])
def test_compose_golden_add_pair(name, n_limbs):
    prm = load_json("params_%s.json" % name)
    vec = load_json("vectors_%s.json" % name)
    d = hx(prm["delta"])
    for key in ("add_valid", "add_edge"):
        v = vec[key]
        _, c1 = P.deserialize_ciphertext_tensor(bytes.fromhex(v["ct1"]))
        _, c2 = P.deserialize_ciphertext_tensor(bytes.fromhex(v["ct2"]))
        _, want = P.deserialize_ciphertext_tensor(bytes.fromhex(v["out"]))
        xs = [(f.a, f.b, f.c) for ct in c1 for f in ct]
        ys = [(f.a, f.b, f.c) for ct in c2 for f in ct]
        got, ok = compose2(n_limbs, xs, ys, d)
        wantf = [(f.a, f.b, f.c) for ct in want for f in ct]
        for g, k, w in zip(got, ok, wantf):
            assert (not k) or g == w
        if key == "add_valid":
            assert sum(ok) >= len(ok) - 2


@pytest.mark.parametrize("name,n_limbs", [("s128_k128", 17), ("s128_k256", 19)])
def test_compose_general_gcd_structure(name, n_limbs):
    """with the coprime-representative screen switched off ~40 % of random pairs have gcd(a1, a2) != 1 (small primes):
    the word-sized general structure (d, d1 = gcd(s, d), x2, y2, v1, v2, c2 d1) must agree with the definition"""
    import math
    prm = load_json("params_%s.json" % name)
    d = hx(prm["delta"])
    rng = P.SplitMix64(99)
    xs = [P.random_form(d, rng, 24, 20) for _ in range(40)]
    ys = [P.random_form(d, rng, 24, 20) for _ in range(40)]
    got, ok = compose2(n_limbs, [(f.a, f.b, f.c) for f in xs], [(f.a, f.b, f.c) for f in ys], d, screen=0)
    n_general = n_d1 = 0
    for g, k, x, y in zip(got, ok, xs, ys):
        dd = math.gcd(x.a, y.a)
        if k:
            w = P.compose(x, y)
            assert g == (w.a, w.b, w.c), dd
            if dd != 1:
                n_general += 1
                if math.gcd(dd, (x.b + y.b) // 2) != 1:
                    n_d1 += 1
        else:
            assert dd != 1              # only non-trivial structures may leave the fast path here
    assert n_general >= 8
