"""CPU: the DEVICE arithmetic (cofhe_amd/csrc/{lane,mp,qf}.hpp) compiled for the host lane
simulator (tests/hostsim) against Python big integers, the pure-Python model and the golden
vectors.  Same source the HIP kernels are built from; only the 8-lane primitives differ."""
import ctypes as C
import math
import os
import random
import sys

import numpy as np
import pytest

import simlib as S
from conftest import ROOT, load_json

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as P  # noqa: E402

M2 = 1 << 2560


def hx(s):
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


def rnd(rng, bits):
    return rng.getrandbits(bits) if bits else 0


def test_mul():
    rng = random.Random(1)
    L = S.lib()
    sizes = [1, 31, 32, 33, 160, 161, 640, 1044, 1279, 1280]
    xs = [rnd(rng, rng.choice(sizes)) for _ in range(24)] + [0, 1, (1 << 1280) - 1]
    ys = [rnd(rng, rng.choice(sizes)) for _ in range(24)] + [(1 << 1280) - 1, 0, (1 << 1280) - 1]
    out = np.zeros(80 * len(xs), dtype=np.uint32)
    L.sim_mul11(S.P(S.pack(xs, 40)), S.P(S.pack(ys, 40)), S.P(out), len(xs))
    assert S.unpack(out, 80) == [a * b for a, b in zip(xs, ys)]
    xs2 = [rnd(rng, rng.choice([2086, 2560, 1300, 5])) for _ in range(10)] + [(1 << 2560) - 1]
    ys2 = [rnd(rng, rng.choice([1044, 1280, 17, 522])) for _ in range(10)] + [(1 << 1280) - 1]
    out = np.zeros(120 * len(xs2), dtype=np.uint32)
    L.sim_mul21(S.P(S.pack(xs2, 80)), S.P(S.pack(ys2, 40)), S.P(out), len(xs2))
    assert S.unpack(out, 120) == [a * b for a, b in zip(xs2, ys2)]


def test_lincomb_shift_bitlen():
    rng = random.Random(2)
    L = S.lib()
    n = 12
    xs = [rnd(rng, 2500) for _ in range(n)]
    ys = [rnd(rng, 2400) for _ in range(n)]
    for A, B in [(0x7FFFFFFF, 12345), (1, 0x7FFFFFFF), (65535, 1)]:
        r = np.zeros(80 * n, dtype=np.uint32)
        s = np.zeros(80 * n, dtype=np.uint32)
        L.sim_lincomb(S.P(S.pack(xs, 80)), S.P(S.pack(ys, 80)), C.c_uint32(A), C.c_uint32(B), S.P(r), S.P(s), n)
        assert S.unpack(r, 80) == [(A * a - B * b) % M2 for a, b in zip(xs, ys)]
        assert S.unpack(s, 80) == [(A * a + B * b) % M2 for a, b in zip(xs, ys)]
    for sh in [0, 1, 31, 32, 33, 160, 161, 1279, 1280, 1281, 2000]:
        vals = [rnd(rng, b) for b in (2560, 100, 1280)] + [0, 1, M2 - 1]
        m = len(vals)
        l, r, h = (np.zeros(80 * m, dtype=np.uint32) for _ in range(3))
        bits = np.zeros(m, dtype=np.int32)
        L.sim_shift(S.P(S.pack(vals, 80)), sh, S.P(l), S.P(r), S.P(h), bits.ctypes.data_as(C.POINTER(C.c_int)), m)
        assert S.unpack(l, 80) == [(a << sh) % M2 for a in vals]
        assert S.unpack(r, 80) == [a >> sh for a in vals]
        assert S.unpack(h, 80) == [a >> 1 for a in vals]
        assert [int(b) for b in bits] == [a.bit_length() for a in vals]


def test_divrem_and_xgcd():
    rng = random.Random(3)
    L = S.lib()
    nums = [rnd(rng, rng.choice([2088, 2560, 1566, 1044, 64, 40, 2000])) for _ in range(24)] + [0, 5, M2 - 1, M2 - 1, 1 << 2559]
    dens = [max(1, rnd(rng, rng.choice([1044, 1280, 522, 33, 32, 31, 1, 64, 700]))) for _ in range(24)] + [7, 7, 1, (1 << 1280) - 1, 3]
    n = len(nums)
    q = np.zeros(80 * n, dtype=np.uint32)
    r = np.zeros(80 * n, dtype=np.uint32)
    L.sim_divrem21(S.P(S.pack(nums, 80)), S.P(S.pack(dens, 40)), S.P(q), S.P(r), n)
    assert S.unpack(q, 80) == [a // b for a, b in zip(nums, dens)]
    assert S.unpack(r, 80) == [a % b for a, b in zip(nums, dens)]
    xa = [rnd(rng, 1044) | 1 for _ in range(10)] + [rnd(rng, 1280) for _ in range(3)] + [12, 1 << 1000, 5, 1, 6 << 700]
    ya = [rnd(rng, 1040) for _ in range(10)] + [rnd(rng, 600) for _ in range(3)] + [18, 3, 5, 1, 9 << 650]
    xa, ya = [max(a, b) for a, b in zip(xa, ya)], [min(a, b) for a, b in zip(xa, ya)]
    n = len(xa)
    d = np.zeros(40 * n, dtype=np.uint32)
    u = np.zeros(40 * n, dtype=np.uint32)
    sg = np.zeros(n, dtype=np.int32)
    L.sim_xgcd(S.P(S.pack(xa, 40)), S.P(S.pack(ya, 40)), S.P(d), S.P(u), sg.ctypes.data_as(C.POINTER(C.c_int)), n)
    for dd, uu, ss, a, b in zip(S.unpack(d, 40), S.unpack(u, 40), sg, xa, ya):
        assert dd == math.gcd(a, b)
        assert (int(ss) * uu * b - dd) % a == 0


def _pool(name):
    prm = load_json("params_%s.json" % name)
    d = hx(prm["delta"])
    rng = P.SplitMix64(123)
    f = P.Form(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    base = [P.random_form(d, rng, 12, 10) for _ in range(10)] if name == "tiny_k8" else [P.random_form(d, rng) for _ in range(3)]
    small, ell = [], 3
    while len(small) < 2:
        if P.is_probable_prime(ell) and P.jacobi(d % ell, ell) == 1:
            small.append(P.prime_form(d, ell))
        ell += 2
    amb = P.reduce_form(4, 4, 1 - d // 16)
    pool = base + small + [P.identity(d), f, amb, P.compose(f, f), P.compose(small[0], small[0]),
                           P.compose(base[0], small[1]), P.inverse(base[1])]
    return d, pool


@pytest.mark.parametrize("name", ["tiny_k8", "s128_k128", "s128_k256"])
def test_compose_all_pairs(name):
    """every ordered pair of a pool that contains the identity, f, an ambiguous form, squares,
    inverse pairs and forms sharing factors: all gcd structures of Cohen 5.4.7"""
    d, pool = _pool(name)
    half = ((-d).bit_length() + 1) // 2
    xs = [(x.a, x.b, x.c) for x in pool for _ in pool]
    ys = [(y.a, y.b, y.c) for _ in pool for y in pool]
    got = S.compose(xs, ys, half)
    for g, x, y in zip(got, xs, ys):
        w = P.compose(P.Form(*x), P.Form(*y))
        assert g == (w.a, w.b, w.c)


def test_compose_golden_add(golden):
    prm, vec = golden
    d = hx(prm["delta"])
    half = ((-d).bit_length() + 1) // 2
    for key in ("add_valid", "add_edge"):
        v = vec[key]
        _, c1 = P.deserialize_ciphertext_tensor(bytes.fromhex(v["ct1"]))
        _, c2 = P.deserialize_ciphertext_tensor(bytes.fromhex(v["ct2"]))
        _, want = P.deserialize_ciphertext_tensor(bytes.fromhex(v["out"]))
        xs = [(f.a, f.b, f.c) for ct in c1 for f in ct]
        ys = [(f.a, f.b, f.c) for ct in c2 for f in ct]
        got = S.compose(xs, ys, half)
        assert got == [(f.a, f.b, f.c) for ct in want for f in ct]


def test_pow_ladder_signs_and_zero():
    """qf_pow: zero exponent -> principal form, negative -> inverse, ambiguous / a == c forms"""
    prm = load_json("params_tiny_k8.json")
    d = hx(prm["delta"])
    rng = P.SplitMix64(5)
    g = P.random_form(d, rng, 16, 12)
    amb = P.reduce_form(4, 4, 1 - d // 16)
    forms = [g, g, g, g, g, amb, amb, P.identity(d), g]
    exps = [0, 1, -1, 2, -37, 3, -3, -5, 12345]
    # signed-digit recoding: runs of ones, word boundaries, the widest magnitude a record holds
    more = [(1 << 8) - 1, (1 << 32) - 1, (1 << 32) + 1, (1 << 33) - 1, 0x55555555, 0xAAAAAAAA, 0xFFFFFFFF00000001,
            (1 << 64) - 1, 3 << 31, (1 << 992) - 1, -((1 << 991) + 12345), 0xDB6DB6DB6DB6DB6D, rng.bits(200)]
    forms += [g] * len(more)
    exps += more
    got = S.power([(f.a, f.b, f.c) for f in forms], exps, d)
    for (a, b, c), f, e in zip(got, forms, exps):
        w = P.power(f, e, d)
        assert (a, b, c) == (w.a, w.b, w.c), e


def test_fundamental_odd_discriminant():
    """Delta = -q, q = 3 (mod 4) prime: odd b everywhere, principal form (1, 1, (1+q)/4);
    a small and a 2088-bit discriminant, all pairs of a pool incl. identity / inverses / squares"""
    rng = P.SplitMix64(77)
    for bits in (70, 2088):
        q = P.random_prime(bits, rng, 3 if bits == 70 else 7)
        d = -q
        assert d % 4 == 1
        half = ((-d).bit_length() + 1) // 2
        g, h = P.random_form(d, rng, 24, 16), P.random_form(d, rng, 24, 16)
        pool = [P.identity(d), g, P.inverse(g), P.compose(g, g), h, P.compose(g, h)]
        xs = [(x.a, x.b, x.c) for x in pool for _ in pool]
        ys = [(y.a, y.b, y.c) for _ in pool for y in pool]
        got = S.compose(xs, ys, half, d)
        for gg, x, y in zip(got, xs, ys):
            w = P.compose(P.Form(*x), P.Form(*y))
            assert gg == (w.a, w.b, w.c)
        pw = S.power([(g.a, g.b, g.c)] * 3, [0, -5, (1 << 40) - 3], d)
        for (a, b, c), e in zip(pw, [0, -5, (1 << 40) - 3]):
            w = P.power(g, e, d)
            assert (a, b, c) == (w.a, w.b, w.c)
