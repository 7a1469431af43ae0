"""Operand pool of the lopsided-pair fuzz tests (CPU tier on the host simulator, GPU tier on the kernels)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyref as P  # noqa: E402


def lopsided_pool(d, k, f):
    """reduced forms whose first coefficient has every bit length from 2 to the full ~bits(Delta)/2: powers of small
    prime forms (a = p^e as long as p^e < sqrt|Delta| / 2), products of two of them, f^(+-2^j) (a = 2^(2(k-j))), and
    full-size random elements"""
    import math
    lim = math.isqrt(-d) // 2
    primes = [p for p in (3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71) if P.jacobi(d % p, p) == 1][:5]
    assert len(primes) >= 3
    pool = []
    for p in primes:
        g = P.prime_form(d, p)
        top = int(math.log(lim, p))
        for e in sorted({1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, top // 2, (3 * top) // 4, top - 7, top - 1, top}):
            if 1 <= e <= top:
                x = P.power(g, e, d)
                assert x.a == p ** e
                pool.append(x)
    mixed = [P.compose(pool[i], pool[-1 - i]) for i in range(0, 24, 3)]
    fj = f
    twos = []
    for j in range(k):
        if j % 9 == 0 or j >= k - 3:
            twos += [fj, P.inverse(fj)]
        fj = P.compose(fj, fj)
    rng = P.SplitMix64(515)
    full = [P.random_form(d, rng) for _ in range(12)]
    return [P.identity(d)] + pool + mixed + twos + full
