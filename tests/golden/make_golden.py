#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/ from oracle/pyref.py (pure-Python
big integers; no third-party arithmetic).  Run from the repo root:

    python tests/golden/make_golden.py

The reference holds no golden vectors of its own ("parity unpinned"), so these
vectors pin the C++/GMP oracle and the HIP kernels to an independent implementation of the
published algorithms.  Every input is determined by the seeds written into the fixtures.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as P  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def hx(x: int) -> str:
    return ("-" if x < 0 else "") + format(abs(x), "x")


def form_d(f: P.Form) -> dict:
    return {"a": hx(f.a), "b": hx(f.b), "c": hx(f.c)}


def pt_bytes(shape, vals) -> bytes:
    """plaintext-tensor format (cpu_cryptosystem.inl:229-267)"""
    import struct
    offs, blobs, last = [], [], 0
    for v in vals:
        offs.append(last | ((1 << 63) if v <= 0 else 0))
        w = max(abs(v).bit_length(), 1) // 8 + 1
        blobs.append(abs(v).to_bytes(w, "little"))
        last += w
    out = struct.pack("<I", len(shape)) + b"".join(struct.pack("<I", d) for d in shape)
    out += b"".join(struct.pack("<Q", o) for o in offs) + b"".join(blobs)
    return out


def make_params(name, sec, k, seed, disc_bits=None):
    cl = P.CLHSM2k(sec, k, seed, disc_bits)
    rng = P.SplitMix64(seed ^ 0xC0FE)
    sk, pk = cl.keygen(rng)
    d = {
        "name": name, "security_level": sec, "k": k, "seed": seed,
        "N": hx(cl.N), "delta": hx(cl.delta), "f": form_d(cl.f), "h": form_d(cl.h),
        "exponent_bound": hx(cl.exponent_bound), "sk": hx(sk), "pk": form_d(pk),
    }
    with open(os.path.join(OUT, "params_%s.json" % name), "w") as fh:
        json.dump(d, fh, indent=1)
    return cl, sk, pk


def edge_forms(cl, rng):
    """forms that force the rare branches: identity, f, small prime forms, inverse pairs,
    equal operands (squaring), a | a' cases, ambiguous forms."""
    d = cl.delta
    ident = cl.id
    g = P.random_form(d, rng)
    ell = 3
    small = []
    while len(small) < 3:
        if P.is_probable_prime(ell) and P.jacobi(d % ell, ell) == 1:
            small.append(P.prime_form(d, ell))
        ell += 2
    amb = P.reduce_form(4, 4, 1 - d // 16)       # ambiguous (order 2): b == a
    assert amb.disc() == d
    pairs = [
        (ident, ident), (ident, g), (g, ident), (g, g), (g, P.inverse(g)),
        (cl.f, cl.f), (cl.f, g), (small[0], small[0]), (small[0], small[1]), (small[1], P.inverse(small[1])),
        (small[2], g), (amb, amb), (amb, g), (P.compose(g, small[0]), small[0]),
        (P.power(cl.f, 2, d), P.power(cl.f, 6, d)), (P.power(g, 3, d), P.power(g, 5, d)),
    ]
    return pairs


def make_vectors(name, cl, sk, pk, seed):
    rng = P.SplitMix64(seed)
    d = cl.delta
    vec = {"params": name, "seed": seed}

    # (1) valid ciphertexts, harness construction (benchmarks/local.cpp:83-98): ramp i+1
    n_el = 4
    r1, r2 = rng.below(cl.exponent_bound), rng.below(cl.exponent_bound)
    ct1 = cl.encrypt_tensor(pk, [i + 1 for i in range(n_el)], r1)
    ct2 = cl.encrypt_tensor(pk, [i + 1 for i in range(n_el)], r2)
    res = P.add_tensor(ct1, ct2)
    vec["add_valid"] = {
        "shape": [2, 2],
        "ct1": P.serialize_ciphertext_tensor([2, 2], ct1).hex(),
        "ct2": P.serialize_ciphertext_tensor([2, 2], ct2).hex(),
        "out": P.serialize_ciphertext_tensor([2, 2], res).hex(),
        "plain_sum": [(2 * (i + 1)) % cl.M for i in range(n_el)],
    }
    # decrypt check of the sum (keeps the CL restatement honest)
    for i in range(n_el):
        assert cl.decrypt(sk, res[i]) == (2 * (i + 1)) % cl.M

    # (2) generic group elements + edge cases, as 1-D tensors of (c1, c2) = pairs of forms
    pairs = edge_forms(cl, rng)
    gen = [(P.random_form(d, rng), P.random_form(d, rng)) for _ in range(8)]
    xs = [(p[0], q[0]) for p, q in zip(pairs[0::2], pairs[1::2])] + [(a, b) for a, b in gen[:4]]
    ys = [(p[1], q[1]) for p, q in zip(pairs[0::2], pairs[1::2])] + [(a, b) for a, b in gen[4:]]
    res = P.add_tensor(xs, ys)
    vec["add_edge"] = {
        "shape": [len(xs)],
        "ct1": P.serialize_ciphertext_tensor([len(xs)], xs).hex(),
        "ct2": P.serialize_ciphertext_tensor([len(xs)], ys).hex(),
        "out": P.serialize_ciphertext_tensor([len(xs)], res).hex(),
    }

    # (3) 1-D scal: exponent edge cases (0, 1, 2, -1, small, k-bit, > k bits, negative big)
    exps = [0, 1, 2, -1, 3, 255, (1 << cl.k) - 1, rng.bits(cl.k), -rng.bits(cl.k // 2), (1 << (cl.k + 3)) + 5, 65536, 7]
    cts = [(P.random_form(d, rng), P.random_form(d, rng)) for _ in range(len(exps) - 2)] + [ct1[0], (cl.id, cl.f)]
    res = P.scal_tensor_1d(exps, cts, d)
    vec["scal_1d"] = {
        "shape": [len(exps)],
        "s": pt_bytes([len(exps)], exps).hex(),
        "s_list": [hx(e) for e in exps],
        "cts": P.serialize_ciphertext_tensor([len(exps)], cts).hex(),
        "out": P.serialize_ciphertext_tensor([len(exps)], res).hex(),
    }

    # (4) 2-D scal (matmul): n=2, m=3, p=2 with harness-style ramp exponents and a given zero
    n, m, p = 2, 3, 2
    cts = cl.encrypt_tensor(pk, [i + 1 for i in range(n * m)], rng.below(cl.exponent_bound))
    s = [j * p + kk + 1 for j in range(m) for kk in range(p)]
    zero = cl.encrypt(pk, 0, rng.below(cl.exponent_bound))
    res = P.scal_tensor_2d(s, cts, zero, n, m, p, d)
    for i in range(n):
        for kk in range(p):
            want = sum((i * m + j + 1) * s[j * p + kk] for j in range(m)) % cl.M
            assert cl.decrypt(sk, res[i * p + kk]) == want
    vec["scal_2d"] = {
        "n": n, "m": m, "p": p,
        "s": pt_bytes([m, p], s).hex(),
        "cts": P.serialize_ciphertext_tensor([n, m], cts).hex(),
        "zero": P.serialize_ciphertext_tensor([1], [zero]).hex(),
        "out": P.serialize_ciphertext_tensor([n, p], res).hex(),
    }
    with open(os.path.join(OUT, "vectors_%s.json" % name), "w") as fh:
        json.dump(vec, fh, indent=1)


def make_threshold_vectors(name, cl, sk, pk, seed):
    """threshold decryption (cpu_cryptosystem_distributed.inl:231-309): shares of sk for
    (t, n) = (2, 3) and (3, 3), the partial decryptions c1^share of a small ciphertext tensor for
    one threshold set each, and the plaintexts their combination must give"""
    from itertools import combinations
    rng = P.SplitMix64(seed)
    ms = [5, cl.M - 1, 0, rng.bits(cl.k)]
    cts = cl.encrypt_tensor(pk, ms, rng.below(cl.exponent_bound))
    out = {"params": name, "seed": seed, "shape": [2, 2], "plain": [hx(m) for m in ms],
           "cts": P.serialize_ciphertext_tensor([2, 2], cts).hex(), "cases": []}
    for t, n, chosen in ((2, 3, (0, 2)), (3, 3, (0, 1, 2))):
        cols = len(P.distribution_matrix(n, t)[0])
        rho = [rng.below(cl.exponent_bound) for _ in range(cols - 1)]
        shares = P.share_secret_key(sk, t, n, rho)
        sets = list(combinations(range(n), t))
        used = []
        for party in chosen:
            idx = [c for c in sets if party in c].index(chosen)
            used.append(shares[party][idx])
        parts = [[P.part_decrypt(cl.delta, sh, ct) for ct in cts] for sh in used]
        for i, ct in enumerate(cts):
            assert P.final_decrypt(cl, ct, [pp[i] for pp in parts]) == ms[i]
        out["cases"].append({
            "t": t, "n": n, "parties": list(chosen), "rho_tail": [hx(r) for r in rho],
            "shares": [[hx(x) for x in sp] for sp in shares],
            "used_shares": [hx(x) for x in used],
            "lambda": P.combine_lambda(t)[:t],
            "parts": [P.serialize_form_tensor([2, 2], pp).hex() for pp in parts],
        })
    with open(os.path.join(OUT, "threshold_%s.json" % name), "w") as fh:
        json.dump(out, fh, indent=1)


def make_plaintext_vectors():
    # negative non-integers: the reference adds 2^k in mpf arithmetic at GMP's default 64-bit precision
    # (cpu_cryptosystem.inl:57-60); pyref.make_plaintext models the limb truncation of that sum
    xs = [0.0, 1.0, 2.5, -1.0, -3.0, 16384.0, -16384.0, 3.9999, 0.5, 1e9, -1e9, 123456.789]
    # round 3: fractional negatives (k = 128: |x| is truncated toward zero before 2^k is added), and magnitudes around
    # 2^100 and up to the float range (truncation toward zero of a 24-bit mantissa is exact)
    xs += [-2.5, -0.75, -123456.789, -3.9999, 1.5 * 2.0 ** 100, -1.5 * 2.0 ** 100, 2.0 ** 100 + 2.0 ** 77, -(2.0 ** 100 + 2.0 ** 77),
           1e20, -1e20, 3e38, -3e38, 2.0 ** 127, -(2.0 ** 127), 2.0 ** -20, -(2.0 ** -20)]
    out = {"k": 128, "cases": [{"x": x, "pt": hx(P.make_plaintext(x, 128)),
                                 "back": P.get_float_from_plaintext(P.make_plaintext(x, 128), 128)} for x in xs]}
    with open(os.path.join(OUT, "plaintext_k128.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    # the reference's local-benchmark parameters (benchmarks/local.cpp:9-12): sec 128, k 128
    cl, sk, pk = make_params("s128_k128", 128, 128, seed=1)
    make_vectors("s128_k128", cl, sk, pk, seed=11)
    make_threshold_vectors("s128_k128", cl, sk, pk, seed=21)
    # examples/node.cpp:33-34 parameters: sec 128, k 256
    cl, sk, pk = make_params("s128_k256", 128, 256, seed=2)
    make_vectors("s128_k256", cl, sk, pk, seed=12)
    # a deliberately tiny group (|DeltaK| = 61 bits, k = 8): collisions, gcd != 1 and
    # short operands are common there
    cl, sk, pk = make_params("tiny_k8", 128, 8, seed=3, disc_bits=58)
    make_vectors("tiny_k8", cl, sk, pk, seed=13)
    make_threshold_vectors("tiny_k8", cl, sk, pk, seed=23)
    make_plaintext_vectors()
    print("fixtures written to", OUT)
