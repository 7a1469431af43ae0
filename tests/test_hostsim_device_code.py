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


def test_carry_ripples_across_lanes():
    """long carry / borrow runs: the resolve takes its rare full-chain path (limb 0 of a chunk == 2^32 - 1)"""
    L = S.lib()
    xs = [(1 << 1600) - 1, (1 << 2560) - 1, (1 << 160) - 1, (1 << 1600), (1 << 2400), ((1 << 800) - 1) << 160, (1 << 2559) + (1 << 32) - 1,
          (1 << 1280) - 1, 1 << 1280]
    ys = [1, 1, (1 << 32) - 1, 1, (1 << 161) + 1, 1 << 160, 1, 1, 1]
    # the sparse resolve of sums (mp_resolve_sparse): a word handed over into an all-ones limb 0 next to a limb 1 that is
    # NOT all ones (fast path: the carry stops in limb 1) although limbs 2-4 are; then the same with limb 1 all ones in one
    # lane only (fallback to the general resolve for the whole wavefront)
    pat = lambda l1: sum(((0xFFFFFFFF if i % 5 != 1 else l1(i)) << (32 * i)) for i in range(80))
    top = sum(1 << (32 * i) for i in range(80) if i % 5 == 4)
    xs += [pat(lambda i: 5), pat(lambda i: 0xFFFFFFFF if i == 16 else 7), pat(lambda i: 0xFFFFFFFE)]
    ys += [top, top, top | 1]
    n = len(xs)
    r = np.zeros(80 * n, dtype=np.uint32)
    s = np.zeros(80 * n, dtype=np.uint32)
    for A, B in [(1, 1), (3, 1), (1, 0x7FFFFFFF), (0x3FFFFFF, 0x3FFFFFF)]:
        L.sim_lincomb(S.P(S.pack(xs, 80)), S.P(S.pack(ys, 80)), C.c_uint32(A), C.c_uint32(B), S.P(r), S.P(s), n)
        assert S.unpack(r, 80) == [(A * a - B * b) % M2 for a, b in zip(xs, ys)]
        assert S.unpack(s, 80) == [(A * a + B * b) % M2 for a, b in zip(xs, ys)]


def test_divexact():
    """2-adic exact division against Python: odd / even divisors, word-sized divisors, quotients of every length,
    32 or more trailing zero bits in the divisor (long-division route), zero numerator"""
    rng = random.Random(33)
    L = S.lib()
    cases = []
    for _ in range(40):
        db = rng.choice([1044, 1280, 522, 33, 32, 31, 1, 64, 700, 1043])
        qb = rng.choice([0, 1, 31, 32, 33, 522, 544, 545, 1044, 1279])
        d = max(1, rnd(rng, db)) | (1 << (db - 1))
        if rng.random() < 0.5:
            d = (d >> rng.choice([1, 2, 5, 31])) << rng.choice([1, 2, 5, 31])      # even divisors
            d = max(d, 2)
        q = rnd(rng, qb)
        if (d * q).bit_length() > 2560:
            continue
        cases.append((d * q, d, q))
    cases += [(0, 12345, 0), (7 << 40, 7 << 35, 32), ((1 << 1279) * 3, 3, 1 << 1279), ((1 << 64) * 5, 1 << 64, 5), (1 << 2000, 1 << 1000, 1 << 1000)]
    # the two-digits-per-pass loop (one carry resolve per pair, pending words fed into the second chain): quotients and
    # divisors of all-ones / sparse limbs (every hand-over word and ripple at its largest), divisors whose second limb is 0 or
    # all ones (the 64-bit inverse), 31 trailing zero bits, odd and even digit counts, numerators that fill both planes
    for _ in range(300):
        db = rng.choice([1044, 1043, 1280, 1100, 65, 64, 63, 97, 160, 161, 320])
        d = rnd(rng, db) | (1 << (db - 1)) | 1
        kind = rng.randrange(6)
        if kind == 0:
            d = (1 << db) - 1
        elif kind == 1:
            d = (d >> 64 << 64) | (0xFFFFFFFF << 32) | (d & 0xFFFFFFFF) | 1
        elif kind == 2:
            d = (d >> 64 << 64) | (d & 0xFFFFFFFF) | 1                          # second limb zero
        elif kind == 3:
            d = ((d >> 31) << 31) | (1 << 31) if db > 40 else d                 # 31 trailing zero bits
        qb = rng.choice([1, 32, 33, 63, 64, 65, 95, 96, 97, 522, 544, 545, 576, 1044, 1056, 1216, 1279])
        q = rnd(rng, qb) | (1 << (qb - 1))
        if rng.random() < 0.3:
            q = (1 << qb) - 1
        elif rng.random() < 0.2:
            q = sum(1 << (32 * t_) for t_ in range(0, (qb + 31) // 32, 2)) % (1 << qb) or 1
        if (d * q).bit_length() > 2560 - 2:
            continue
        cases.append((d * q, d, q))
    n = len(cases)
    nq = np.array([(q.bit_length() + 31) // 32 + (i % 3) for i, (_, _, q) in enumerate(cases)], dtype=np.int32)
    out = np.zeros(80 * n, dtype=np.uint32)
    L.sim_divexact21(S.P(S.pack([c[0] for c in cases], 80)), S.P(S.pack([c[1] for c in cases], 40)), S.P(out),
                     nq.ctypes.data_as(C.POINTER(C.c_int)), n)
    assert S.unpack(out, 80) == [c[2] for c in cases]


def _batch_cases(rng, n):
    for i in range(n):
        kind = i % 8
        if kind == 0:       # exact mode, small operands
            x, y, ex, thr = rnd(rng, rng.choice([64, 40, 33, 31, 8, 1])), rnd(rng, rng.choice([63, 33, 30, 20, 3, 0])), 1, 0
        elif kind == 1:     # partial sequence: threshold inside the window
            x, y, ex = rnd(rng, 64) | (1 << 63), rnd(rng, 62), 0
            thr = 1 << rng.randrange(1, 63)
        elif kind == 2:     # equal / adjacent windows
            x = rnd(rng, 64) | (1 << 63)
            y, ex, thr = x - rng.choice([0, 1, 2]), 0, 0
        elif kind == 3:     # lopsided
            x, y, ex, thr = rnd(rng, 64) | (1 << 63), rnd(rng, rng.choice([34, 40, 50])), 0, 0
        elif kind == 4:     # quotient runs of ones and large single quotients
            y = rnd(rng, 40) | (1 << 39)
            x, ex, thr = min(y * rng.choice([1, 2, 3, 1000, 65535, 1 << 20]) + rnd(rng, 30), (1 << 64) - 1), 0, 0
        else:
            x, y, ex, thr = rnd(rng, 64) | (1 << 63), rnd(rng, 64) | (1 << rng.choice([63, 62, 60, 55])), rng.choice([0, 0, 0, 1]), 0
        if x < y:
            x, y = y, x
        yield kind, x, y, ex, thr


def _check_batch_matrix(x, y, ex, thr, M, ok, bound=31):
    A, B, Cc, D = M
    assert max(A, B, Cc, D) < (1 << bound)
    assert A * D - B * Cc == 1, (x, y, ex, thr, M)
    assert ok == (1 if (B | Cc) else 0)
    # corners of the intervals [x, x + 1) x [y, y + 1) with 64 more bits below (exact: the numbers themselves)
    corners = [(x << 64, y << 64)] if ex else [((x << 64) + dx, (y << 64) + dy) for dx in (0, (1 << 64) - 1) for dy in (0, (1 << 64) - 1)]
    for X, Y in corners:
        assert A * X - B * Y >= 0 and D * Y - Cc * X >= 0, (x, y, ex, thr, M)


def test_lehmer_batch_f64_properties():
    """the serving lane's batch (mp.hpp: lehmer_batch -- double precision, 53-bit windows, run-on lanes): every matrix is
    unimodular with cofactors below 2^26 and keeps BOTH remainders non-negative for every pair of numbers the windows can
    stand for; a partial sequence never steps past its threshold by more than one step; the batch makes progress
    (>= 21 cofactor bits on average on full windows at the cap of 8 double-steps, against the 26 the window allows)"""
    rng = random.Random(12)
    L = S.lib()
    L.sim_lehmer_f64.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_uint32)]
    out = np.zeros(4, dtype=np.uint32)
    n_ok, bits, n_full = 0, 0, 0
    for kind, x, y, ex, thr in _batch_cases(rng, 12000):
        x, y, thr = x >> 11, y >> 11, thr >> 11                     # the same families at the product's window width
        if x < y:
            x, y = y, x
        ok = L.sim_lehmer_f64(x, y, ex, thr, S.P(out))
        M = tuple(int(v) for v in out)
        _check_batch_matrix(x, y, ex, thr, M, ok, bound=26)
        n_ok += ok
        if ok and thr == 0 and not ex and kind >= 5:
            bits += max(M).bit_length()
            n_full += 1
    assert n_ok > 7000
    assert bits / n_full >= 21.0, bits / n_full          # 8 double-steps per batch (COFHE_LEHMER_CAP): ~23 of the 26 bits the window allows
    print("cofactor bits per batch:", bits / n_full)


def test_lehmer_batch_uniform_is_the_same_batch():
    """the single-chain form (mp.hpp: lehmer_batch_uniform -- a failing step ends the batch, no snapshots): at the serving
    lane's cap the same matrix as lehmer_batch on every case; at the wide layout's cap of 12 the matrix properties, and
    more cofactor bits per batch"""
    rng = random.Random(13)
    L = S.lib()
    L.sim_lehmer_f64.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_uint32)]
    L.sim_lehmer_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]
    ref, out = np.zeros(4, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
    bits8, bits12, n_full = 0, 0, 0
    for kind, x, y, ex, thr in _batch_cases(rng, 12000):
        x, y, thr = x >> 11, y >> 11, thr >> 11
        if x < y:
            x, y = y, x
        ok0 = L.sim_lehmer_f64(x, y, ex, thr, S.P(ref))
        ok1 = L.sim_lehmer_uniform(x, y, ex, thr, 8, S.P(out))
        assert ok0 == ok1 and (not ok0 or tuple(ref) == tuple(out)), (x, y, ex, thr, tuple(ref), tuple(out))
        ok2 = L.sim_lehmer_uniform(x, y, ex, thr, 12, S.P(out))
        M = tuple(int(v) for v in out)
        _check_batch_matrix(x, y, ex, thr, M, ok2, bound=26)
        if ok0 and ok2 and thr == 0 and not ex and kind >= 5:
            bits8 += max(int(v) for v in ref).bit_length()
            bits12 += max(M).bit_length()
            n_full += 1
    assert bits12 >= bits8 and bits12 / n_full >= 23.0, (bits8 / n_full, bits12 / n_full)
    print("cofactor bits per batch: cap 8 %.2f, cap 12 %.2f" % (bits8 / n_full, bits12 / n_full))


def _serve_sequence(x, y, stop_bits):
    """drives euclid_serve like the workgroup protocol does (mp.hpp: euclid_run_wg) with Python integers standing
    for the client side; returns the final pair, the cofactor column and the number of rounds"""
    L = S.lib()
    ux, uy, sx, sy = 0, 1, -1, 1
    tx, ty, sd = C.c_int(39), C.c_int(39), C.c_int(0)
    w = np.zeros(8, dtype=np.uint32)
    rounds = 0
    while True:
        xy = np.concatenate([S.to_limbs(x, 40), S.to_limbs(y, 40)])
        L.sim_euclid_serve(S.P(xy), stop_bits, C.byref(tx), C.byref(ty), C.byref(sd), S.P(w))
        A, ok = int(w[0]) & 0x7FFFFFFF, int(w[0]) >> 31
        B, dn = int(w[1]) & 0x7FFFFFFF, int(w[1]) >> 31
        Cc, D = int(w[2]), int(w[3])
        if dn:
            assert sd.value == 1
            break
        rounds += 1
        assert rounds < 400
        if ok:
            nx, ny = A * x - B * y, D * y - Cc * x
            assert nx >= 0 and ny >= 0 and (B | Cc) != 0
            ux, uy = A * ux + B * uy, D * uy + Cc * ux
            x, y = nx, ny
        else:           # long-division step as the client does it: order the pair (RENAMES x and y), then one 32-bit
            if x < y:   # digit of the quotient (mp_quot_digit), so a long quotient takes several rounds
                x, y, ux, uy, sx, sy = y, x, uy, ux, sy, sx
            q = x // y
            cut = max(0, q.bit_length() - 32)
            q = (q >> cut) << cut
            x, ux = x - q * y, ux + q * uy
    if x < y:
        x, y, ux, uy, sx, sy = y, x, uy, ux, sy, sx
    return x, y, ux, uy, sx, sy, rounds




def test_euclid_serve_protocol():
    """the serving lane's scalar code (windows, bit lengths, done / long-step decisions, conservative batch) run
    round by round against Python integers: full sequences end at the gcd with a valid cofactor, partial ones stop
    at the bound, lopsided and tiny operands take the long-step route"""
    rng = random.Random(7)
    for bits_x, bits_y in [(1044, 1040), (1044, 1044), (1280, 1270), (700, 690), (64, 60), (33, 2), (1044, 3), (1, 1), (1200, 600), (96, 95),
                           (3, 1044), (600, 1200), (252, 1041), (40, 1280), (1000, 1100), (31, 96)]:      # second operand longer: the client swaps
        for _ in range(3):
            x = rnd(rng, bits_x) | (1 << (bits_x - 1))
            y = (rnd(rng, bits_y) | (1 << (bits_y - 1))) if bits_y else 0
            if bits_x == 252:
                x = 1 << 252                   # a power of two: the first coefficient of f^(2^j)
            g, gy, _gu, _su, _gv, _sv, _r = _serve_sequence(x, y, -1)[:7]
            assert gy == 0 and g == math.gcd(x, y)
    # cofactor: sx * ux * y0 == g (mod x0) with x0 the first operand (ux starts at 0, uy at 1)
    for _ in range(6):
        x0 = rnd(rng, 1044) | (1 << 1043)
        y0 = rnd(rng, 1040) | 1
        g, _z, ux, _uy, sx, _sy, rounds = _serve_sequence(x0, y0, -1)
        assert (sx * ux * y0 - g) % x0 == 0
        assert 15 <= rounds <= 60          # ~38 single-batch rounds, ~21 with two batches per round
    # partial sequences: R1 <= bound < R0, R_i == +-C_i * r (mod v1)
    for _ in range(6):
        v1 = rnd(rng, 1044) | (1 << 1043)
        r = rnd(rng, 1043)
        stop = 522
        R0, R1, C0, C1, s0, s1, rounds = _serve_sequence(v1, r, stop)
        assert R1.bit_length() <= stop < R0.bit_length() or R1 == 0
        assert (s0 * C0 * r - R0) % v1 == 0 and (s1 * C1 * r - R1) % v1 == 0
        assert abs(R0 * C1 + R1 * C0) == v1 or math.gcd(v1, r) != 1


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


def test_malformed_forms_end_with_a_status_not_a_hang():
    """records that are not forms of the discriminant (what a hostile peer could send past the range checks of the
    wire format): every loop of the arithmetic is capped, a zero divisor is flagged -- the call returns (meaningless
    result), nothing divides by zero on the host, nothing spins.  The validation kernel rejects such records before
    they reach the arithmetic (GPU test); this covers the arithmetic itself."""
    prm = load_json("params_s128_k128.json")
    d = hx(prm["delta"])
    half = ((-d).bit_length() + 1) // 2
    L = S.lib()
    L.sim_status.restype = C.c_uint
    L.sim_status()
    bad = [((1 << 900) + 12345, 7, 5), (1, 7, 1), (12, 0, 1 << 2000), ((1 << 1040) + 1, (1 << 1040) - 3, 3), (5, 5, 5)]
    good = P.random_form(d, P.SplitMix64(3))
    xs = [bad[0], bad[2], bad[3], (good.a, good.b, good.c), bad[4]]
    ys = [bad[1], bad[2], (good.a, good.b, good.c), bad[3], bad[4]]
    got = S.compose(xs, ys, half, delta=d)
    assert len(got) == len(xs)            # returned
    st = L.sim_status()
    assert st & 0x7, "at least one safety cap must have been reported"
    # the simulator is still healthy: a valid composition afterwards is exact
    w = P.compose(good, good)
    assert S.compose([(good.a, good.b, good.c)], [(good.a, good.b, good.c)], half, delta=d) == [(w.a, w.b, w.c)]
    assert L.sim_status() == 0


def _fixed_base_chain_pairs(start, f, m, k, d):
    """the compositions acc o f^(+-2^j) of the fixed-base product start o f^m, one per non-zero signed digit of m, with the
    exact partial products as left operands.  For m = 0xe354...a49c and start = pk^r its 8th step (a 1042-bit first
    coefficient against 2^218) is the composition on which the serving lane's per-name hints went stale"""
    pairs, acc, x3 = [], start, 3 * m
    for j in range(k):
        dg = ((x3 >> (j + 1)) & 1) - ((m >> (j + 1)) & 1)
        if dg == 0:
            continue
        y = P.power(f, 1 << j, d)
        if dg < 0:
            y = P.inverse(y)
        pairs.append((acc, y))
        acc = P.compose(acc, y)
    return pairs


def test_compose_through_the_workgroup_protocol():
    """qf_compose<true> in a simulated workgroup (8 groups = one wavefront of host threads, tests/hostsim: run_workgroup):
    the remainder sequences go through euclid_run_wg -- stash, barrier, the serving lanes' euclid_serve, mailbox, barrier,
    client apply or long-division step -- i.e. the code every kernel runs, not the in-group euclid_run.  Random forms,
    squarings, inverse pairs, the principal form, and products with f^(+-2^j) whose short first coefficient makes the
    sequence start lopsided (long-division steps that swap the pair), and the 45-step fixed-base chain whose 8th step once
    broke the serving lane's hints on the GPU (this test fails on that step without the fix in euclid_serve)"""
    prm = load_json("params_s128_k128.json")
    d, k = hx(prm["delta"]), prm["k"]
    half = ((-d).bit_length() + 1) // 2
    F = lambda o: P.Form(hx(o["a"]), hx(o["b"]), hx(o["c"]))
    f, pk, h = F(prm["f"]), F(prm["pk"]), F(prm["h"])
    rng = P.SplitMix64(404)
    rnd_forms = [P.random_form(d, rng, 96, 64) for _ in range(10)]
    r = int("c82e101ee83d683efd4905a925cbbc24f11c50088370731d23689cedb7caca5532b1e56a5bd176f91893d737e90a739d12de7f4321468c73ea174c3"
            "76cae7cb7d7ea367748b2e6efe01e16b79d801488717264fc5d68823f9416023e7bab39b017539f8bea12672556de3b214a20b71dfc4cbbd4e9d2d02e", 16)
    pkr = P.power(pk, r, d)
    one = P.identity(d)
    pairs = _fixed_base_chain_pairs(pkr, f, 0xe35425b964bdb6d05a03893b5c79a49c, k, d)     # the chain tools/bench_ops.py caught
    fj = f
    for j in range(k):
        if j % 9 == 0 or j >= k - 3:
            pairs += [(rnd_forms[j % 10], fj), (P.inverse(fj), pkr)]
        fj = P.compose(fj, fj)
    pairs += [(rnd_forms[0], rnd_forms[0]), (rnd_forms[1], P.inverse(rnd_forms[1])), (one, rnd_forms[2]), (rnd_forms[3], one), (one, one), (f, f)]
    pairs += list(zip(rnd_forms[:5], rnd_forms[5:]))
    n = S.lib().sim_wg_groups()
    want_all = [P.compose(a, b) for a, b in pairs]
    for i0 in range(0, len(pairs), n):
        chunk = pairs[i0:i0 + n]
        got = S.compose_wg([(a.a, a.b, a.c) for a, _ in chunk], [(b.a, b.b, b.c) for _, b in chunk], half, d)
        assert [tuple(g) for g in got] == [(w.a, w.b, w.c) for w in want_all[i0:i0 + n]], i0
    assert S.lib().sim_status() == 0


@pytest.mark.parametrize("name", ["s128_k128", "s128_k256"])
def test_lopsided_pairs_fuzz_on_the_simulator(name):
    """the device composition on operand pairs whose first coefficients differ in length by anything from 0 to ~1040
    bits, both orders (tests/lopsided.py: powers of small prime forms, f^(+-2^j), the identity, full-size elements and
    their inverses): in-group remainder sequence (euclid_run) on 600 pairs and the workgroup-served one (euclid_run_wg,
    the code path of round 2's stale-hint bug) on 160 -- the GPU tier runs 20 000 of the same family"""
    from lopsided import lopsided_pool
    prm = load_json("params_%s.json" % name)
    d, k = hx(prm["delta"]), prm["k"]
    half = ((-d).bit_length() + 1) // 2
    f = P.Form(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    pool = lopsided_pool(d, k, f)
    allf = pool + [P.inverse(x) for x in pool]
    rng = P.SplitMix64(717)
    pairs = [(rng.below(len(allf)), rng.below(len(allf))) for _ in range(300)]
    pairs += [(j, i) for i, j in pairs]
    t3 = lambda x: (x.a, x.b, x.c)
    got = S.compose([t3(allf[i]) for i, _ in pairs], [t3(allf[j]) for _, j in pairs], half, d)
    for g, (i, j) in zip(got, pairs):
        assert tuple(g) == t3(P.compose(allf[i], allf[j])), (i, j)
    n = S.lib().sim_wg_groups()
    wg_pairs = pairs[:80] + pairs[300:380]
    for i0 in range(0, len(wg_pairs), n):
        chunk = wg_pairs[i0:i0 + n]
        got = S.compose_wg([t3(allf[i]) for i, _ in chunk], [t3(allf[j]) for _, j in chunk], half, d)
        assert [tuple(g) for g in got] == [t3(P.compose(allf[i], allf[j])) for i, j in chunk], i0
    assert S.lib().sim_status() == 0


def test_euclid_wg_cofactors_and_stops():
    """the workgroup-served remainder sequence (euclid_run_wg) by itself in a simulated workgroup: gcd and both cofactor
    congruences for operands from 33 to 1200 bits, partial sequences that stop at a bound, lopsided pairs (long-division
    steps), equal operands, y = 0"""
    entry = "sim_euclid_wg"
    import ctypes as C
    L = S.lib()
    rng = random.Random(31)

    def run(pairs, stops):
        n = len(pairs)
        x = np.concatenate([S.to_limbs(a, 40) for a, _ in pairs])
        y = np.concatenate([S.to_limbs(b, 40) for _, b in pairs])
        out = np.zeros(160 * n, dtype=np.uint32)
        sg = np.zeros(2 * n, dtype=np.int32)
        st = np.array(stops, dtype=np.int32)
        getattr(L, entry)(S.P(x), S.P(y), n, st.ctypes.data_as(C.c_void_p), S.P(out), sg.ctypes.data_as(C.c_void_p))
        res = []
        for i in range(n):
            o = out[160 * i:160 * i + 160]
            res.append((S.from_limbs(o[:40]), S.from_limbs(o[40:80]), int(sg[2 * i]) * S.from_limbs(o[80:120]),
                        int(sg[2 * i + 1]) * S.from_limbs(o[120:160])))
        return res

    n = L.sim_wg_groups()
    full, part = [], []
    for bits in (33, 64, 65, 96, 200, 500, 1043, 1044, 1171, 1200):
        for _ in range(2):
            a = rnd(rng, bits) | (1 << (bits - 1))
            full.append((a, rnd(rng, bits - 1) | 1))
    a0 = rnd(rng, 1043) | (1 << 1042)
    full += [(a0, 3), (a0, 5), (a0, 1 << 200), (a0, (1 << 252)), (a0, rnd(rng, 700) | 1), (a0, a0), (a0, 0), (a0, 1), (a0, a0 - 1),
             (6 * (rnd(rng, 500) | 1), 10 * (rnd(rng, 480) | 1))]
    for i0 in range(0, len(full), n):
        chunk = full[i0:i0 + n]
        for (a, b), (x, y, cx, cy) in zip(chunk, run(chunk, [-1] * len(chunk))):
            assert x == math.gcd(a, b) and y == 0, (a.bit_length(), b.bit_length())
            assert (cx * b - x) % a == 0 and (cy * b) % a == 0
    for _ in range(2 * n):
        a = rnd(rng, 1043) | (1 << 1042)
        b = rnd(rng, 1041)
        part.append((a, b, rng.choice([530, 521, 700, 64, 1000, 33])))
    for i0 in range(0, len(part), n):
        chunk = part[i0:i0 + n]
        res = run([(a, b) for a, b, _ in chunk], [s for _, _, s in chunk])
        for (a, b, stop), (x, y, cx, cy) in zip(chunk, res):
            assert x >= y and y.bit_length() <= stop, (stop, x.bit_length(), y.bit_length())
            assert (cx * b - x) % a == 0 and (cy * b - y) % a == 0
            assert x * abs(cy) + y * abs(cx) == a                     # consecutive remainders of one sequence
    assert L.sim_status() == 0


@pytest.mark.parametrize("name", ["s128_k128", "s128_k256"])
def test_compose_with_a_common_word_sized_factor(name):
    """pairs whose first coefficients share a prime >= 29 (what the representative step leaves: 0.8 % of random pairs, one
    in every fifth workgroup): the word-arithmetic route of qf_compose -- r from the coprime case's residue y1 m mod a1
    plus word residues -- and its fall-backs (the factor also divides s, or its square divides a1), against the
    independent model; both through the in-group and the workgroup-served sequences"""
    import math
    prm = load_json("params_%s.json" % name)
    d = hx(prm["delta"])
    half = ((-d).bit_length() + 1) // 2
    rng = P.SplitMix64(929)
    primes = [p for p in (29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 65537, 65539, 1000003, 4294967291)
              if P.jacobi(d % p, p) == 1]
    assert len(primes) >= 8
    pairs, word, plain = [], 0, 0
    for it in range(60):
        p = primes[it % len(primes)]
        g = P.prime_form(d, p)
        gg = P.compose(g, g) if it % 7 == 3 else g                        # p^2 | a1 now and then: the fall-back
        u = P.compose(P.random_form(d, rng), gg)
        v = P.compose(P.random_form(d, rng), g if it % 2 else P.inverse(g))
        k = math.gcd(u.a, v.a)
        if k == 1:
            continue
        pairs += [(u, v), (v, u)]
        word += 2 * (k < (1 << 32))
        s_ = (u.b + v.b) // 2
        plain += 2 * (k < (1 << 32) and math.gcd(s_, k) == 1)
    assert word >= 16 and plain >= 10, (word, plain, len(pairs))
    # common factors beyond a word, and composite ones: forms built directly on a first coefficient P q (P the common
    # factor, q a fresh 700-bit prime), b from the square roots of Delta modulo each prime -- the multi-limb branch of the
    # general formula, and (for P = p1 p2 with the roots combined differently) a factor only part of which divides s
    def crt_form(ps, signs, q):
        mods = list(ps) + [q]
        a = 1
        for p_ in mods:
            a *= p_
        b = 0
        for p_, sg in zip(mods, list(signs) + [1]):
            r = P.sqrt_mod_prime(d % p_, p_) * sg % p_
            n_ = a // p_
            b = (b + r * n_ * pow(n_, -1, p_)) % a
        if (b - d) % 2:
            b += a                      # b == Delta (mod 2); now b in [0, 2a)
        if b > a:
            b -= 2 * a
        assert (b * b - d) % (4 * a) == 0
        return P.reduce_form(a, b, (b * b - d) // (4 * a))
    big = [p for p in (65537, 65539, 65543, 65551, 65557, 65563, 1000003, 1000033, 1000037, 4294967291) if P.jacobi(d % p, p) == 1]
    assert len(big) >= 2
    wide = 0
    for it in range(12):
        ps = (big[it % len(big)], big[(it + 1) % len(big)]) if it % 2 == 0 else (primes[it % 8], primes[(it + 3) % 8])
        if ps[0] == ps[1]:
            continue
        qs = []
        while len(qs) < 2:
            q = P.random_prime(700, rng, 1 + 2 * rng.below(4))
            if P.jacobi(d % q, q) == 1:
                qs.append(q)
        u = crt_form(ps, (1, 1), qs[0])
        v = crt_form(ps, (1, 1) if it % 4 < 2 else (1, -1), qs[1])
        k = math.gcd(u.a, v.a)
        if k == 1:
            continue
        pairs += [(u, v), (v, u)]
        wide += 2 * (k >= (1 << 32))
    assert wide >= 4, wide
    t3 = lambda x: (x.a, x.b, x.c)
    want = [t3(P.compose(a, b)) for a, b in pairs]
    try:
        for word_route in (1, 0):             # the one-composition kernels' route, then the sequence kernels' general formula
            S.lib().sim_set_word_route(word_route)
            got = S.compose([t3(a) for a, _ in pairs], [t3(b) for _, b in pairs], half, d)
            assert [tuple(g_) for g_ in got] == want, word_route
            n = S.lib().sim_wg_groups()
            for i0 in range(0, min(len(pairs), 4 * n), n):
                chunk = pairs[i0:i0 + n]
                got = S.compose_wg([t3(a) for a, _ in chunk], [t3(b) for _, b in chunk], half, d)
                assert [tuple(g_) for g_ in got] == want[i0:i0 + n], (word_route, i0)
    finally:
        S.lib().sim_set_word_route(1)
    assert S.lib().sim_status() == 0


def test_compose_with_the_fifth_and_sixth_representative():
    """pairs for which the first four representatives of the second operand (a, c, a +- b + c) all share a prime <= 23 with
    a1 (0.37 % of random pairs): qf_compose then takes a +- 2 b + 4 c -- (x, y) = (1, +-2) -- and, when even those fail,
    goes on with the common factor; every such pair of a random pool against the independent model, in-group and through
    the workgroup-served sequences"""
    prm = load_json("params_s128_k128.json")
    d = hx(prm["delta"])
    half = ((-d).bit_length() + 1) // 2
    rng = P.SplitMix64(515)
    pool = [P.random_form(d, rng) for _ in range(260)]
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23)
    f = lambda v, x, y: v.a * x * x + v.b * x * y + v.c * y * y
    cands = ((1, 0), (0, 1), (1, 1), (1, -1), (1, 2), (1, -2))
    picks = {4: [], 5: [], None: []}
    for u in pool:
        for v in pool:
            if u is v:
                continue
            ok = [all(not (f(v, x, y) % p == 0 and u.a % p == 0) for p in small) for x, y in cands]
            k = ok.index(True) if True in ok else None
            if k in picks and len(picks[k]) < 24:
                picks[k].append((u, v))
    assert len(picks[4]) >= 12 and len(picks[5]) >= 4, {k: len(v) for k, v in picks.items()}
    pairs = picks[4] + picks[5] + picks[None]
    t3 = lambda x: (x.a, x.b, x.c)
    want = [t3(P.compose(a, b)) for a, b in pairs]
    got = S.compose([t3(a) for a, _ in pairs], [t3(b) for _, b in pairs], half, d)
    assert [tuple(g_) for g_ in got] == want
    n = S.lib().sim_wg_groups()
    for i0 in range(0, len(pairs), n):
        chunk = pairs[i0:i0 + n]
        got = S.compose_wg([t3(a) for a, _ in chunk], [t3(b) for _, b in chunk], half, d)
        assert [tuple(g_) for g_ in got] == want[i0:i0 + n], i0
    assert S.lib().sim_status() == 0


def test_word_route_primitives():
    """mp_mod_word_fast (residues of one- and two-plane numbers modulo a word, through 16-bit half-limbs and tabulated
    powers of 2^16) and word_xgcd16 (float-reciprocal quotients) against Python integers: moduli d^2 for small and large
    16-bit d, arbitrary 32-bit moduli, numbers with all-ones and sparse limbs"""
    import ctypes as C
    import math
    import numpy as np
    rng = P.SplitMix64(616)
    ds = [2, 3, 29, 31, 255, 256, 257, 4099, 32749, 65521, 65535] + [2 + rng.below(65534) for _ in range(40)]
    Ws = [d_ * d_ for d_ in ds] + [1, 2, 3, 0xFFFFFFFF, 0xFFFFFFFB, 0x80000000, 0x10001] + [1 + rng.below(0xFFFFFFFF) for _ in range(40)]
    Ws = [w for w in Ws if 0 < w < (1 << 32)]
    xs = []
    for i, w in enumerate(Ws):
        kind = i % 4
        if kind == 0:
            x = rng.bits(2560)
        elif kind == 1:
            x = (1 << 2560) - 1 - rng.bits(40)
        elif kind == 2:
            x = sum(1 << (32 * rng.below(80)) for _ in range(3)) * (1 + rng.below(1 << 16))
            x %= 1 << 2560
        else:
            x = rng.bits(1044)
        xs.append(x)
    xa = S.pack(xs, 80)
    wa = np.array(Ws, dtype=np.uint32)
    out = np.zeros(2 * len(Ws), dtype=np.uint32)
    S.lib().sim_mod_word_fast(S.P(xa), S.P(wa), S.P(out), len(Ws))
    for i, (x, w) in enumerate(zip(xs, Ws)):
        assert int(out[2 * i]) == (x % (1 << 1280)) % w, (i, w)
        assert int(out[2 * i + 1]) == x % w, (i, w)
    # mp_mod_primorial: the constant modulus 2*3*...*23 of the coprime-representative test (tabulated limb weights)
    M = 223092870
    ps = [0, 1, M - 1, M, M + 1, (1 << 1280) - 1, (1 << 1279), (1 << 1043) - 1] + [rng.bits(1280) for _ in range(40)] + \
         [rng.bits(1044) for _ in range(40)] + [M * rng.bits(1200) for _ in range(8)] + [sum(0xFFFFFFFF << (32 * i) for i in range(0, 40, 3))]
    pa = S.pack(ps, 40)
    pout = np.zeros(len(ps), dtype=np.uint32)
    S.lib().sim_mod_primorial(S.P(pa), S.P(pout), len(ps))
    assert [int(v) for v in pout] == [x % M for x in ps]
    ms, as_ = [], []
    for m_ in [2, 3, 4, 29, 30, 841, 65521, 65535, 46368, 28657] + [2 + rng.below(65534) for _ in range(300)]:
        for a_ in {1, m_ - 1, max(1, m_ // 2), 1 + rng.below(m_ - 1), 1 + rng.below(m_ - 1)}:
            if 0 < a_ < m_:
                ms.append(m_)
                as_.append(a_)
    ms += [46368, 65535, 65534]            # consecutive Fibonacci numbers: the longest remainder sequences of 16-bit operands
    as_ += [28657, 65534, 65533]
    ma, aa = np.array(ms, dtype=np.uint32), np.array(as_, dtype=np.uint32)
    g, inv = np.zeros(len(ms), dtype=np.uint32), np.zeros(len(ms), dtype=np.uint32)
    S.lib().sim_word_xgcd16(S.P(ma), S.P(aa), S.P(g), S.P(inv), len(ms))
    for m_, a_, g_, i_ in zip(ms, as_, g, inv):
        assert int(g_) == math.gcd(m_, a_), (m_, a_)
        assert 0 <= int(i_) < m_ and (int(i_) * a_ - int(g_)) % m_ == 0, (m_, a_, int(i_))


def test_compose_with_four_wavefronts():
    """the kernels' workgroup geometry on the simulator (32 groups = 256 host threads = four wavefronts, one of them serving
    the other three across the workgroup barrier; the default simulated workgroup is a single wavefront): random pairs,
    lopsided pairs (long-division steps), squarings, a ragged last workgroup, against the independent model"""
    from lopsided import lopsided_pool
    prm = load_json("params_s128_k128.json")
    d = hx(prm["delta"])
    half = ((-d).bit_length() + 1) // 2
    rng = P.SplitMix64(3232)
    pool = [P.random_form(d, rng) for _ in range(24)]
    pairs = [(pool[rng.below(24)], pool[rng.below(24)]) for _ in range(40)]
    pairs += [(f_, f_) for f_ in pool[:6]]                                   # squarings
    f = P.Form(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    lop = lopsided_pool(d, prm["k"], f)
    pairs += [(lop[rng.below(len(lop))], lop[rng.below(len(lop))]) for _ in range(14)]
    t3 = lambda x: (x.a, x.b, x.c)
    n = S.lib_wg32().sim_wg_groups()
    assert n == 32
    for i0 in range(0, len(pairs), n):
        chunk = pairs[i0:i0 + n]
        got, status = S.compose_wg32([t3(a) for a, _ in chunk], [t3(b) for _, b in chunk], half, d)
        assert [tuple(g_) for g_ in got] == [t3(P.compose(a, b)) for a, b in chunk], i0
        assert status == 0

