"""CPU: the wavefront-wide layout of the latency kernels (cofhe_amd/csrc/wide.hpp: two limbs per lane over 64 lanes, carries
from two 64-bit ballots; qfw.hpp: reduction and the common route of the composition) on the host emulation, against Python
integers and the independent model oracle/pyref.py.  The GPU tier runs the same code through cofhe_hip_compose_wide_records
and the decryption ladder."""
import ctypes as C
import os
import random
import sys

import numpy as np
import pytest

from conftest import ROOT, load_json

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as P  # noqa: E402
import simlib as S8  # noqa: E402  (record packing helpers)
import simwlib as W  # noqa: E402


def hx(s):
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


def rb(rng, b):
    return rng.getrandbits(b) if b else 0


def test_wide_mul_lincomb_shift():
    L = W.lib()
    rng = random.Random(5)
    xs = [rb(rng, rng.choice([1, 31, 32, 33, 64, 65, 522, 1044, 1056, 2088, 2112])) for _ in range(60)] + [0, 1, (1 << 2112) - 1, (1 << 1056) - 1]
    ys = [rb(rng, rng.choice([1, 31, 32, 33, 64, 65, 522, 1044, 1056, 1984])) for _ in range(60)] + [5, 0, (1 << 1984) - 1, (1 << 1056) - 1]
    out = np.zeros(128 * len(xs), dtype=np.uint32)
    L.simw_mul(W.P(W.pack(xs)), W.P(W.pack(ys)), W.P(out), len(xs))
    assert W.unpack(out) == [(a * b) % W.M for a, b in zip(xs, ys)]
    # linear combinations: the two's complement over the whole capacity (long runs of all-ones limbs: every carry ripple the
    # generate / propagate ballots have to carry across lanes), all-ones operands, the widest multipliers
    xs = [rb(rng, rng.choice([4000, 2000, 1044, 64, 63])) for _ in range(40)] + [W.M - 1, (1 << 1600) - 1, 1 << 3000]
    ys = [rb(rng, rng.choice([3990, 1990, 1040, 60, 5])) for _ in range(40)] + [1, 1, 1]
    for A, B in [(1, 1), (0x3FFFFFF, 0x3FFFFFF), (1, 0xFFFFFFFF), (65535, 3)]:
        r = np.zeros(128 * len(xs), dtype=np.uint32)
        s = np.zeros(128 * len(xs), dtype=np.uint32)
        L.simw_lincomb(W.P(W.pack(xs)), W.P(W.pack(ys)), C.c_uint32(A), C.c_uint32(B), W.P(r), W.P(s), len(xs))
        assert W.unpack(r) == [(A * a - B * b) % W.M for a, b in zip(xs, ys)], (A, B)
        assert W.unpack(s) == [(A * a + B * b) % W.M for a, b in zip(xs, ys)], (A, B)
    vals = [rb(rng, 4096), rb(rng, 100), rb(rng, 1280), 0, 1, W.M - 1, rb(rng, 2100)]
    for sh in [0, 1, 31, 32, 33, 63, 64, 65, 95, 96, 97, 160, 1279, 2000, 4095]:
        l = np.zeros(128 * len(vals), dtype=np.uint32)
        r = np.zeros(128 * len(vals), dtype=np.uint32)
        bits = np.zeros(len(vals), dtype=np.int32)
        L.simw_shift(W.P(W.pack(vals)), sh, W.P(l), W.P(r), bits.ctypes.data_as(C.POINTER(C.c_int)), len(vals))
        assert W.unpack(l) == [(a << sh) % W.M for a in vals], sh
        assert W.unpack(r) == [a >> sh for a in vals], sh
        assert [int(b) for b in bits] == [a.bit_length() for a in vals]
    for a, b in [(5, 7), (7, 5), (1 << 4000, 1 << 4000), ((1 << 2000) + 1, 1 << 2000), (0, 0), (0, 1)]:
        assert L.simw_cmp(W.P(W.pack([a])), W.P(W.pack([b]))) == (a > b) - (a < b)


def test_wide_remainder_sequence():
    """qfw.hpp: w_euclid -- both lengths and both windows of a round come from ONE view of the pair's leading lanes
    (wide.hpp: w_top_pair) and the batch is the single-chain form: pairs of equal length, pairs far apart (the shorter
    number ends below the view), equal numbers, powers of two, a zero, single words, full and partial sequences.
    Invariants: x == sx ux y0, y == sy uy y0 (mod x0); the gcd is kept; x >= y; a full sequence ends at y == 0, a partial
    one with the first remainder at or below its stop"""
    from math import gcd
    rng = random.Random(31)
    L = W.lib()
    cases = []
    for bits in (1200, 1279, 640, 130, 129, 128, 127, 65, 64, 63, 33, 20):
        for _ in range(6):
            x = rb(rng, bits) | (1 << (bits - 1))
            cases.append((x, rb(rng, bits), -1))
            cases.append((x, rb(rng, max(1, bits - rng.randrange(1, 40))), -1))
            cases.append((x, rb(rng, bits), bits // 2))
            cases.append((x, rb(rng, bits) | 1, bits // 2 + rng.randrange(-8, 8)))
    for gap in (27, 53, 64, 65, 127, 128, 129, 200, 640, 1100):          # far apart: long-division steps, empty views
        x = rb(rng, 1200) | (1 << 1199)
        cases.append((x, rb(rng, 1200 - gap) | 1, -1))
        cases.append((rb(rng, 1200 - gap) | 1, x, -1))
        cases.append((x, rb(rng, 1200 - gap) | 1, 600))
    x = rb(rng, 900) | 1
    cases += [(x, x, -1), (x, 0, -1), (0, x, -1), (1 << 1000, 1 << 500, -1), ((1 << 1000) - 1, (1 << 64) - 1, -1), (x, 1, -1), (1, 1, -1),
              (x * 7, x * 3, -1), ((1 << 64), (1 << 64) - 1, -1), ((1 << 128) + 1, (1 << 64) + 1, 40), (x << 130, x << 129, -1)]
    out, sg = np.zeros(512, dtype=np.uint32), np.zeros(2, dtype=np.int32)
    for x0, y0, stop in cases:
        ok = L.simw_euclid(W.P(W.pack([x0])), W.P(W.pack([y0])), stop, W.P(out), sg.ctypes.data_as(C.POINTER(C.c_int)))
        assert ok == 1, (x0, y0, stop)
        x, y, ux, uy = (W.unpack(out[128 * k:128 * (k + 1)])[0] for k in range(4))
        sx, sy = int(sg[0]), int(sg[1])
        assert x >= y and gcd(x, y) == gcd(x0, y0), (x0, y0, stop)
        if x0:
            assert (x - sx * ux * y0) % x0 == 0 and (y - sy * uy * y0) % x0 == 0, (x0, y0, stop)
        if stop < 0:
            assert y == 0
        else:
            assert y.bit_length() <= stop or y == 0
            assert x.bit_length() > stop or max(x0, y0).bit_length() <= stop or min(x0, y0).bit_length() <= stop, (x0, y0, stop)
    assert W.lib().simw_status() == 0


def test_wide_divisions():
    L = W.lib()
    rng = random.Random(6)
    cases = []
    for _ in range(200):
        db = rng.choice([1044, 1043, 1056, 1024, 1025, 65, 64, 33, 32, 31, 1, 700])
        nb = rng.choice([2088, 2080, 1044, 1500, db, db + 1, db + 31, db + 32, db + 33, 10, 0])
        d = rb(rng, db) | (1 << (db - 1))
        n = rb(rng, nb)
        k = rng.randrange(5)
        if k == 0:
            d = (1 << db) - 1
        if k == 1:
            n = d * rb(rng, max(nb - db, 1)) + (d - 1)           # remainders at the top of their range
        if k == 2:
            n = d * rb(rng, max(nb - db, 1))                     # and zero
        if n.bit_length() <= 2200:
            cases.append((n, d))
    cases += [(5, 7), ((1 << 2088) - 1, (1 << 1044) - 1), (1 << 1044, 1 << 1043), (1 << 2000, (1 << 1000) + 1)]
    rem = np.zeros(128 * len(cases), dtype=np.uint32)
    assert L.simw_mod(W.P(W.pack([c[0] for c in cases])), W.P(W.pack([c[1] for c in cases])), W.P(rem), len(cases)) == 1
    assert W.unpack(rem) == [n % d for n, d in cases]
    # exact division, 2-adic with 64-bit digits
    cases = []
    for _ in range(250):
        db = rng.choice([1044, 1043, 1280, 65, 64, 63, 97, 160, 33, 32, 1])
        qb = rng.choice([1, 32, 33, 63, 64, 65, 522, 544, 576, 1044, 1056, 1279])
        d = rb(rng, db) | (1 << (db - 1))
        k = rng.randrange(6)
        if k == 0:
            d = (1 << db) - 1
        if k == 1:
            d = (d >> rng.choice([1, 5, 31, 40])) << rng.choice([1, 5, 31, 40]) or 2          # even divisors (< 64 trailing zeros)
        q = rb(rng, qb) | (1 << (qb - 1))
        if rng.random() < 0.3:
            q = (1 << qb) - 1
        if (d * q).bit_length() <= 3900 and (d & ((1 << 64) - 1)):
            cases.append((d * q, d, q))
    nq = np.array([(c[2].bit_length() + 63) // 64 + (i % 3) for i, c in enumerate(cases)], dtype=np.int32)
    quo = np.zeros(128 * len(cases), dtype=np.uint32)
    assert L.simw_divexact(W.P(W.pack([c[0] for c in cases])), W.P(W.pack([c[1] for c in cases])), nq.ctypes.data_as(C.POINTER(C.c_int)),
                           W.P(quo), len(cases)) == 1
    assert W.unpack(quo) == [c[2] for c in cases]
    # a divisor whose low 64 bits are zero is declined (the composition then takes the 8-lane route)
    assert L.simw_divexact(W.P(W.pack([3 << 64])), W.P(W.pack([1 << 64])), np.array([1], dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int)),
                           W.P(quo), 1) == 0


def _compose_w(pairs, d):
    L = W.lib()
    half = ((-d).bit_length() + 1) // 2
    n = len(pairs)
    f1 = np.concatenate([S8.form_record(a.a, a.b, a.c) for a, _ in pairs])
    f2 = np.concatenate([S8.form_record(b.a, b.b, b.c) for _, b in pairs])
    out = np.zeros(n * S8.REC_WORDS, dtype=np.uint32)
    fl = np.zeros(n, dtype=np.int32)
    L.simw_compose(S8.P(f1), S8.P(f2), S8.P(out), fl.ctypes.data_as(C.POINTER(C.c_int)), n, half, S8.P(S8.to_limbs(-d, 80)))
    return [S8.record_form(out[i * S8.REC_WORDS:(i + 1) * S8.REC_WORDS]) for i in range(n)], fl


@pytest.mark.parametrize("name", ["s128_k128", "s128_k256", "tiny_k8"])
def test_wide_composition_against_the_model(name):
    """wf_compose on random pairs, squarings, a ladder-like chain (the common route: taken for all but a few per cent --
    pairs that keep a common factor after the coprime-representative step) and on the lopsided pool (short first
    coefficients, inverse pairs, the identity: mostly declined).  Whatever it accepts must equal the independent model;
    what it declines is left untouched for the 8-lane route, which the GPU tier exercises."""
    from lopsided import lopsided_pool
    prm = load_json("params_%s.json" % name)
    d, k = hx(prm["delta"]), prm["k"]
    rng = P.SplitMix64(199)
    pool = [P.random_form(d, rng) for _ in range(40)]
    pairs = [(pool[rng.below(40)], pool[rng.below(40)]) for _ in range(300)] + [(x, x) for x in pool]
    f = P.Form(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    lop = lopsided_pool(d, k, f)
    pairs += [(lop[rng.below(len(lop))], lop[rng.below(len(lop))]) for _ in range(100)]
    x, chain = pool[0], []
    for i in range(60):
        y = x if i % 3 else pool[1 + i % 7]
        chain.append((x, y))
        x = P.compose(x, y)
    pairs += chain + [(P.identity(d), pool[0]), (pool[1], P.identity(d)), (pool[2], P.inverse(pool[2]))]
    got, fl = _compose_w(pairs, d)
    for g, (a, b), flag in zip(got, pairs, fl):
        if not flag:
            w = P.compose(a, b)
            assert tuple(g) == (w.a, w.b, w.c)
    assert fl[:300].sum() <= 30 and fl[300:340].sum() == 0 and fl[440:500].sum() <= 6        # random pairs, squarings, the chain
    assert W.lib().simw_status() == 0
