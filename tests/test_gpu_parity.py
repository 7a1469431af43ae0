"""GPU: HIP kernels (through the C ABI) against the C++/GMP oracle and the golden vectors."""
import json
import os
import sys

import pytest

import oracle_lib as O
from conftest import ROOT, load_json

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as P  # noqa: E402
from lopsided import lopsided_pool as _lopsided_pool  # noqa: E402

pytestmark = pytest.mark.gpu


def hx(s):
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


_engines = {}


def _pt_bytes(shape, vals):
    import struct
    offs, blobs, last = [], [], 0
    for v in vals:
        offs.append(last | ((1 << 63) if v <= 0 else 0))
        w = max(abs(v).bit_length(), 1) // 8 + 1
        blobs.append(abs(v).to_bytes(w, "little"))
        last += w
    out = struct.pack("<I", len(shape)) + b"".join(struct.pack("<I", d) for d in shape)
    return out + b"".join(struct.pack("<Q", o) for o in offs) + b"".join(blobs)


@pytest.fixture(autouse=True)
def _device_status_stays_clear():
    """after EVERY GPU test: no kernel of any context the test used hit a safety cap (lane.hpp: CF_ST_*).  The status
    word is what caught round 2's wrong-discriminant bug; a parity test that passes with a cap bit set is not a pass."""
    yield
    for delta, E in list(_engines.items()):
        assert E.device_status(clear=True) == 0, "device status word set on the context of |Delta| = %d bits" % (-delta).bit_length()


def engine(delta):
    # the PyTorch wheel bundles its own HIP runtime: when torch shares the process (the resident
    # tensor tests below) it has to initialise the GPU before libcofhe_hip.so does
    import torch
    torch.cuda.init()
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
    # reference messages of scal_ciphertext_tensors (tensor_ops.inl:273, :284)
    s3 = _pt_bytes([1, 2, 2], [1, 2, 3, 4])
    c3 = P.serialize_ciphertext_tensor([1, 2, 2], cts)
    with pytest.raises(CofheHipError, match="Tensors must be 0D, 1D or 2D for now"):
        E.scal_ciphertext_tensors(s3, c3)
    with pytest.raises(CofheHipError, match="Vector sizes must be equal"):
        E.scal_ciphertext_tensors(_pt_bytes([3], [1, 2, 3]), b)
    with pytest.raises(CofheHipError, match="exponent wider"):
        E.scal_ciphertext_tensors(_pt_bytes([4], [1, 2, 3, 1 << 1000]), b)
    # empty tensors pass through
    e0 = P.serialize_ciphertext_tensor([0], [])
    assert E.add_ciphertext_tensors(e0, e0) == e0


def test_scal_matmul_16x16_config_c1(params128):
    """BASELINE config C1: cts 16x16, s 16x16 with s[j,k] = j*16+k+1 (benchmarks/local.cpp:171-174)"""
    d = hx(params128["delta"])
    E = engine(d)
    n = m = p = 16
    cts = _random_tensor(d, n * m, 5)
    zero = _random_tensor(d, 1, 6, nbase=2)
    s = _pt_bytes([m, p], [j * p + k + 1 for j in range(m) for k in range(p)])
    ct = P.serialize_ciphertext_tensor([n, m], cts)
    z = P.serialize_ciphertext_tensor([1], zero)
    got = E.scal_ciphertext_tensors(s, ct, z)
    assert got == O.scal_2d(d, s, ct, z)


@pytest.mark.parametrize("p,n,m", [(70, 2, 5), (20, 2, 4), (5, 3, 3), (2, 2, 7)])
def test_scal_matmul_signed_window_recoding(params128, p, n, m):
    """2-D scal with every window width the launcher picks (p >= 64: 8, >= 16: 6, >= 4: 4, else 2) and
    exponents that stress the width-w non-adjacent recoding: negative weights as make_plaintext gives
    them (2^k - x), plain negatives, zero, all-ones, word boundaries, the widest record (992 bits)"""
    d, k = hx(params128["delta"]), params128["k"]
    E = engine(d)
    rng = P.SplitMix64(1000 + p)
    M = 1 << k
    special = [0, 1, -1, M - 1, M - 3, M - 100, (1 << 32) - 1, 1 << 32, (1 << 64) + 1, -(M - 7), (1 << 992) - 1,
               -((1 << 991) + 5), 0xFF, 0x80, 0x81, 0x7F, 255 << 24, rng.bits(128), -rng.bits(128), rng.bits(300)]
    exps = [special[(j * p + kk) % len(special)] if (j + kk) % 3 else rng.bits(16) for j in range(m) for kk in range(p)]
    cts = _random_tensor(d, n * m, 60 + p)
    zero = _random_tensor(d, 1, 61, nbase=2)
    s = _pt_bytes([m, p], exps)
    ct = P.serialize_ciphertext_tensor([n, m], cts)
    z = P.serialize_ciphertext_tensor([1], zero)
    assert E.scal_ciphertext_tensors(s, ct, z) == O.scal_2d(d, s, ct, z)


@pytest.mark.parametrize("w", [2, 3, 4, 5, 6, 7, 8])
def test_scal_matmul_every_window_width(params128, w):
    """the launcher picks the window width from the exponent length; the context option "wnaf_width" pins each width
    in turn on a product with short, negative, all-ones and 300-bit exponents"""
    d, k = hx(params128["delta"]), params128["k"]
    E = engine(d)
    E.set_option("wnaf_width", w)
    rng = P.SplitMix64(500 + w)
    n, m, p = 2, 4, 3
    M = 1 << k
    exps = [0, 1, -1, M - 1, M - 3, (1 << 64) + 1, rng.bits(300), -rng.bits(128), 0xFF, 0x81, 255 << 24, rng.bits(16)]
    cts = _random_tensor(d, n * m, 80 + w)
    zero = _random_tensor(d, 1, 81, nbase=2)
    s = _pt_bytes([m, p], exps)
    ct = P.serialize_ciphertext_tensor([n, m], cts)
    z = P.serialize_ciphertext_tensor([1], zero)
    try:
        assert E.scal_ciphertext_tensors(s, ct, z) == O.scal_2d(d, s, ct, z)
    finally:
        E.set_option("wnaf_width", 0)
    from cofhe_amd import CofheHipError
    with pytest.raises(CofheHipError):
        E.set_option("wnaf_width", 9)
    with pytest.raises(CofheHipError):
        E.set_option("no_such_option", 1)


@pytest.mark.parametrize("n,m,p", [(2, 20, 3), (1, 64, 2), (3, 17, 1), (2, 33, 2)])
def test_scal_matmul_split_inner_dimension(params128, n, m, p):
    """few outputs and a long inner dimension: the launcher cuts j into segments (one squaring chain each) and
    folds the partial products with the accumulation tree; ragged last segment, columns of zero exponents
    (empty partial products) and negative exponents included"""
    d, k = hx(params128["delta"]), params128["k"]
    E = engine(d)
    rng = P.SplitMix64(900 + m)
    exps = []
    for j in range(m):
        for kk in range(p):
            r = rng.below(10)
            exps.append(0 if (r < 3 or (kk == 0 and j >= m - 5)) else (-(j * p + kk + 1) if r == 3 else rng.bits(14) + 1))
    cts = _random_tensor(d, n * m, 90 + m)
    zero = _random_tensor(d, 1, 91, nbase=2)
    s = _pt_bytes([m, p], exps)
    ct = P.serialize_ciphertext_tensor([n, m], cts)
    z = P.serialize_ciphertext_tensor([1], zero)
    assert E.scal_ciphertext_tensors(s, ct, z) == O.scal_2d(d, s, ct, z)


@pytest.mark.parametrize("full", [0, 1, 2])
def test_lehmer_batch_on_the_gpu(full):
    """the serving lane's batch where it executes: lehmer_batch (cofhe_amd/csrc/mp.hpp; full = 1: no thresholds; full = 2: the
    single-chain form of the latency layout, lehmer_batch_uniform<12>) on 2^20 window
    pairs, one per lane (v_rcp_f32, v_cvt_u32_f32 and the f32 image of a 53-bit window are the GPU's, not the host
    simulator's 1.0f / x), every matrix checked on the host in exact integers: unimodular, cofactors below 2^26, the ok flag,
    and BOTH remainders non-negative at all four corners of the window intervals (tests/gpu_kernels/lehmer_gpu.hip)"""
    import ctypes as C
    so = os.path.join(ROOT, "tests", "gpu_kernels", "liblehmer_gpu.so")
    assert os.path.exists(so), "tests/gpu_kernels/liblehmer_gpu.so is built by __graft_entry__.build()"
    import torch  # noqa: F401  (PyTorch's HIP runtime first)
    torch.cuda.init()
    L = C.CDLL(so)
    L.lehmer_gpu_check.restype = C.c_long
    L.lehmer_gpu_check.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    stats = (C.c_uint64 * 4)()
    n = 1 << 20
    bad = L.lehmer_gpu_check(n, 20261005 + full, full, stats)
    assert bad == 0, "violations: %d, first at pair %d" % (bad, stats[3])
    assert stats[0] > n // 2
    assert stats[1] / stats[2] >= 21.0            # progress: ~23 of the 26 cofactor bits the window allows


@pytest.mark.parametrize("tree", [0, 1])
@pytest.mark.parametrize("n,m,p,kind", [(2, 20, 3, "mixed"), (1, 64, 2, "mixed"), (3, 17, 1, "mixed"), (2, 33, 2, "mixed"), (16, 16, 16, "ramp"),
                                        (2, 4, 3, "mixed"), (3, 9, 5, "zeros"), (2, 8, 2, "single"), (17, 40, 4, "wide"), (1, 1, 1, "mixed")])
def test_scal_matmul_tree_and_chains(params128, tree, n, m, p, kind):
    """both forms of the matrix product -- the product tree (per-position pairwise trees over all rows at once, then a
    Horner chain over the tree's top level; the default for an inner dimension >= 8) and the lockstep chains (option
    "matmul_tree" = 0) -- on the same inputs against the oracle: zero and negative exponents, columns that are all zero
    (empty trees: the result is Enc(0)), positions with a single entry (copied up every level), one row (a workgroup
    holding several tree elements), 17 rows (a ragged last workgroup), 992-bit exponents (hundreds of positions), and
    inner dimensions below the tree's threshold pinned onto it"""
    d, k = hx(params128["delta"]), params128["k"]
    E = engine(d)
    rng = P.SplitMix64(4000 + 17 * m + p)
    exps = []
    for j in range(m):
        for kk in range(p):
            r = rng.below(10)
            if kind == "ramp":
                e = j * p + kk + 1
            elif kind == "zeros":
                e = 0 if (kk != 1 or j % 4) else rng.bits(9) + 1           # four columns of zeros, one sparse column
            elif kind == "single":
                e = (1 << (3 * j)) if kk == 0 else -(1 << j)               # one non-zero digit per position and base
            elif kind == "wide":
                e = rng.bits(992) if (j + kk) % 7 == 0 else (-rng.bits(128) if r < 2 else rng.bits(20))
            else:
                e = 0 if (r < 3 or (kk == 0 and j >= m - 5)) else (-(j * p + kk + 1) if r == 3 else rng.bits(14) + 1)
            exps.append(e)
    cts = _random_tensor(d, n * m, 300 + m)
    zero = _random_tensor(d, 1, 301, nbase=2)
    s = _pt_bytes([m, p], exps)
    ct = P.serialize_ciphertext_tensor([n, m], cts)
    z = P.serialize_ciphertext_tensor([1], zero)
    E.set_option("matmul_tree", tree)
    try:
        assert E.scal_ciphertext_tensors(s, ct, z) == O.scal_2d(d, s, ct, z)
    finally:
        E.set_option("matmul_tree", -1)


def test_scal_1d_random_128bit_exponents(params128):
    d = hx(params128["delta"])
    E = engine(d)
    rng = P.SplitMix64(8)
    n = 24
    cts = _random_tensor(d, n, 9)
    exps = [rng.bits(128) * (-1 if i % 5 == 0 else 1) for i in range(n)]
    s = _pt_bytes([n], exps)
    ct = P.serialize_ciphertext_tensor([n], cts)
    assert E.scal_ciphertext_tensors(s, ct) == O.scal_1d(d, s, ct)


def test_add_128x128_properties(params128):
    """full C2 size through the resident-record API: commutativity, (x+y)+z == x+(y+z), and a
    sampled byte comparison with the oracle"""
    import numpy as np
    import torch
    d = hx(params128["delta"])
    E = engine(d)
    n = 128 * 128
    base = _random_tensor(d, 64, 10, nbase=16)
    rng = P.SplitMix64(11)
    idx = [[rng.below(64) for _ in range(n)] for _ in range(3)]
    _, recs = E.bytes_to_records(P.serialize_ciphertext_tensor([64], base))
    recs = torch.from_numpy(recs.view(np.int32).reshape(64, 336)).cuda()
    x, y, z = (recs[torch.tensor(ix, device="cuda")].reshape(-1).contiguous() for ix in idx)
    def add(a, b):
        o = torch.empty_like(a)
        E.compose_records(a.data_ptr(), b.data_ptr(), o.data_ptr(), 2 * n)
        torch.cuda.synchronize()
        return o
    xy, yx = add(x, y), add(y, x)
    assert torch.equal(xy, yx)
    assert torch.equal(add(xy, z), add(x, add(y, z)))
    # sample 512 elements against the oracle
    samp = xy[: 512 * 336].cpu().numpy().view(np.uint32)
    got = E.records_to_bytes(samp, [512])
    a = E.records_to_bytes(x[: 512 * 336].cpu().numpy().view(np.uint32), [512])
    b = E.records_to_bytes(y[: 512 * 336].cpu().numpy().view(np.uint32), [512])
    assert got == O.add(d, a, b)


def test_decrypt_golden_ciphertexts(golden):
    """ciphertexts encrypted by the pure-Python CL_HSM2k restatement (fixtures) decrypt on the GPU
    (kernel k_decrypt: c1^sk ladder, division, bit-peeling discrete log in <f>)"""
    import numpy as np
    import torch
    prm, vec = golden
    d, k = hx(prm["delta"]), prm["k"]
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record
    frec = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    ow = (k + 31) // 32 + 1
    for key in ("add_valid", "scal_2d"):
        v = vec[key]
        shape, recs = E.bytes_to_records(bytes.fromhex(v["out"]))
        n = recs.size // 336
        dc = torch.from_numpy(recs.view(np.int32)).cuda()
        dsk = torch.from_numpy(exp_records([hx(prm["sk"])]).view(np.int32)).cuda()
        out = torch.zeros(n * ow, dtype=torch.int32, device="cuda")
        E.decrypt_records(dc.data_ptr(), dsk.data_ptr(), frec, out.data_ptr(), n, k)
        torch.cuda.synchronize()
        o = out.cpu().numpy().view(np.uint32).reshape(n, ow)
        assert not o[:, -1].any()
        got = [int.from_bytes(r[:-1].tobytes(), "little") for r in o]
        if key == "add_valid":
            assert got == v["plain_sum"]
        else:
            nn, m, p = v["n"], v["m"], v["p"]
            s = [j * p + kk + 1 for j in range(m) for kk in range(p)]
            assert got == [sum((i * m + j + 1) * s[j * p + kk] for j in range(m)) % (1 << k) for i in range(nn) for kk in range(p)]


@pytest.mark.parametrize("name", ["s128_k128", "tiny_k8"])
def test_threshold_decrypt_golden(name):
    """part_decrypt_tensor / combine_part_decryption_results_tensor on the GPU against the fixtures
    (pure-Python restatement, cross-checked by the C++ oracle in test_oracle_golden): the partial
    decryptions c1^share byte for byte, then the combined plaintexts"""
    import numpy as np
    import torch
    from conftest import load_json
    prm, th = load_json("params_%s.json" % name), load_json("threshold_%s.json" % name)
    d, k = hx(prm["delta"]), prm["k"]
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record
    frec = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    shape, recs = E.bytes_to_records(bytes.fromhex(th["cts"]))
    n = recs.size // 336
    dc = torch.from_numpy(recs.view(np.int32)).cuda()
    ow = (k + 31) // 32 + 1
    for case in th["cases"]:
        t = case["t"]
        parts = torch.zeros(t * n * 168, dtype=torch.int32, device="cuda")
        for i, sh in enumerate(case["used_shares"]):
            dsh = torch.from_numpy(exp_records([hx(sh)]).view(np.int32)).cuda()
            E.part_decrypt_records(dc.data_ptr(), dsh.data_ptr(), parts.data_ptr() + i * n * 168 * 4, n)
        torch.cuda.synchronize()
        ph = parts.cpu().numpy().view(np.uint32).reshape(t, n * 168)
        for i in range(t):
            assert E.pdr_records_to_bytes(ph[i], shape) == bytes.fromhex(case["parts"][i])
        out = torch.zeros(n * ow, dtype=torch.int32, device="cuda")
        E.combine_part_decryptions_records(dc.data_ptr(), parts.data_ptr(), case["lambda"], frec, out.data_ptr(), n, k)
        torch.cuda.synchronize()
        o = out.cpu().numpy().view(np.uint32).reshape(n, ow)
        assert not o[:, -1].any()
        assert [int.from_bytes(r[:-1].tobytes(), "little") for r in o] == [hx(m) for m in th["plain"]]
    # a wrong share does not land in <f>: status word set, no exception
    bad = torch.from_numpy(exp_records([hx(th["cases"][0]["used_shares"][0]) + 1]).view(np.int32)).cuda()
    parts = torch.zeros(2 * n * 168, dtype=torch.int32, device="cuda")
    E.part_decrypt_records(dc.data_ptr(), bad.data_ptr(), parts.data_ptr(), n)
    good = torch.from_numpy(exp_records([hx(th["cases"][0]["used_shares"][1])]).view(np.int32)).cuda()
    E.part_decrypt_records(dc.data_ptr(), good.data_ptr(), parts.data_ptr() + n * 168 * 4, n)
    out = torch.zeros(n * ow, dtype=torch.int32, device="cuda")
    E.combine_part_decryptions_records(dc.data_ptr(), parts.data_ptr(), [1, -1], frec, out.data_ptr(), n, k)
    torch.cuda.synchronize()
    assert out.cpu().numpy().view(np.uint32).reshape(n, ow)[:, -1].all()
    with pytest.raises(Exception):
        E.combine_part_decryptions_records(dc.data_ptr(), parts.data_ptr(), [1, 2], frec, out.data_ptr(), n, k)


def test_shared_exponent_ladder_edge_exponents(params128):
    """k_pow_shared (part_decrypt_records): zero, unit, negative, short and wide shared exponents against
    the oracle's nupow on the c1 components"""
    import numpy as np
    import torch
    d = hx(params128["delta"])
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records
    n = 5
    cts = _random_tensor(d, n, 71)
    data = P.serialize_ciphertext_tensor([n], cts)
    _, recs = E.bytes_to_records(data)
    dc = torch.from_numpy(recs.view(np.int32)).cuda()
    for e in (0, 1, -1, 2, 5, -37, (1 << 200) + 1, (1 << 64) - 1, -((1 << 991) + 3)):
        de = torch.from_numpy(exp_records([e]).view(np.int32)).cuda()
        out = torch.zeros(n * 168, dtype=torch.int32, device="cuda")
        E.part_decrypt_records(dc.data_ptr(), de.data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        got = E.pdr_records_to_bytes(out.cpu().numpy().view(np.uint32), [n])
        want = O.scal_1d(d, _pt_bytes([n], [e] * n), data)
        _, wcts = P.deserialize_ciphertext_tensor(want)
        assert got == P.serialize_form_tensor([n], [c[0] for c in wcts]), e


@pytest.mark.parametrize("n,m,p", [(3, 5, 4), (2, 3, 2), (2, 8, 3), (1, 7, 1), (2, 1, 2)])
def test_accumulate_vs_oracle(params128, n, m, p):
    """out[i,k] = zero o prod_j x[i,j,k] (accumulation of the ciphertext x ciphertext matrix product):
    the oracle folds the m slices x[:,j,:] in with its element-wise add.  m >= 4 with few outputs takes the
    pairwise-tree path (odd m: unpaired slices), m < 4 the chain kernel"""
    import numpy as np
    import torch
    d = hx(params128["delta"])
    E = engine(d)
    x = _random_tensor(d, n * m * p, 31)
    zero = _random_tensor(d, 1, 32)[0]
    _, xr = E.bytes_to_records(P.serialize_ciphertext_tensor([n * m * p], x))
    _, zr = E.bytes_to_records(P.serialize_ciphertext_tensor([1], [zero]))
    dx = torch.from_numpy(xr.view(np.int32)).cuda()
    dz = torch.from_numpy(zr.view(np.int32)).cuda()
    out = torch.zeros(n * p * 336, dtype=torch.int32, device="cuda")
    E.accumulate_records(dx.data_ptr(), dz.data_ptr(), out.data_ptr(), n, m, p)
    torch.cuda.synchronize()
    got = E.records_to_bytes(out.cpu().numpy().view(np.uint32), [n, p])
    want = P.serialize_ciphertext_tensor([n, p], [zero] * (n * p))
    for j in range(m):
        sl = [x[(i * m + j) * p + k] for i in range(n) for k in range(p)]
        want = O.add(d, want, P.serialize_ciphertext_tensor([n, p], sl))
    assert got == want


def test_negation_power_is_two_to_k_minus_one(params128):
    """negate_ciphertext_tensor raises to make_plaintext(-1) = 2^k - 1 (tensor_ops.inl:137), not to the
    group inverse; the signed-digit ladder must give the oracle's bytes for it and for other runs of ones"""
    d, k = hx(params128["delta"]), params128["k"]
    E = engine(d)
    cts = _random_tensor(d, 6, 41)
    exps = [(1 << k) - 1, (1 << k) - 1, (1 << 64) - 1, 0xFFFF0000FFFF, (1 << 200) - (1 << 100) + 1, -((1 << k) - 1)]
    s = _pt_bytes([6], exps)
    c = P.serialize_ciphertext_tensor([6], cts)
    assert E.scal_ciphertext_tensors(s, c) == O.scal_1d(d, s, c)


def test_wire_format_on_device(golden):
    """serialised tensors unpacked / packed by the GPU kernels (wire.hip) = the host converters =
    the reference's byte layout: ciphertext, partial-decryption and plaintext tensors, incl. the edge
    fixtures (zero b, b = a, a = c, slot widths that are multiples of 8 bits)"""
    import numpy as np
    import torch
    prm, vec = golden
    E = engine(hx(prm["delta"]))

    def roundtrip(data, kind, host_unpack, words):
        d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        shape, want = host_unpack(data)
        n = want.size // words
        rec = torch.full((max(n, 1) * words,), -1, dtype=torch.int32, device="cuda")
        sh2, n2 = E.unpack_tensor_device(d.data_ptr(), len(data), kind, rec.data_ptr(), n)
        assert (sh2, n2) == (shape, n)
        assert np.array_equal(rec.cpu().numpy().view(np.uint32)[: n * words], want)
        cap = E.packed_size_bound(n, kind, len(shape))
        out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        ln = E.pack_tensor_device(rec.data_ptr(), n, kind, shape, out.data_ptr(), cap)
        assert out.cpu().numpy()[:ln].tobytes() == data

    for key in ("add_valid", "add_edge", "scal_1d", "scal_2d"):
        for field in ("ct1", "cts", "out"):
            if field in vec[key]:
                roundtrip(bytes.fromhex(vec[key][field]), 2, E.bytes_to_records, 168)
    roundtrip(bytes.fromhex(vec["scal_1d"]["s"]), 0, E.bytes_to_exponents, 32)
    roundtrip(bytes.fromhex(vec["scal_2d"]["s"]), 0, E.bytes_to_exponents, 32)
    if prm["name"] != "s128_k256":
        from conftest import load_json
        th = load_json("threshold_%s.json" % prm["name"])
        roundtrip(bytes.fromhex(th["cases"][0]["parts"][0]), 1, E.pdr_bytes_to_records, 168)


def test_wire_format_on_device_rejects_malformed(params128):
    import numpy as np
    import struct
    import torch
    from cofhe_amd import CofheHipError
    d = hx(params128["delta"])
    E = engine(d)
    good = P.serialize_ciphertext_tensor([2], _random_tensor(d, 2, 5, nbase=2))
    rec = torch.zeros(4 * 168, dtype=torch.int32, device="cuda")

    def unpack(data, kind=2, cap=4):
        t = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        return E.unpack_tensor_device(t.data_ptr(), len(data), kind, rec.data_ptr(), cap)

    assert unpack(good) == ([2], 4)
    with pytest.raises(CofheHipError, match="too short"):
        unpack(good[:40])
    with pytest.raises(CofheHipError, match="record buffer too small"):
        unpack(good, cap=3)
    bad = bytearray(good)
    struct.pack_into("<Q", bad, 8 + 8 * 3, 1 << 40)              # offset beyond the body
    with pytest.raises(CofheHipError, match="corrupt offset table"):
        unpack(bytes(bad))
    wide = P.serialize_ciphertext_tensor([1], [(P.Form(1 << 1300, 1, 1), P.Form(1, 1, 1))])
    with pytest.raises(CofheHipError, match="outside the supported range"):
        unpack(wide)
    zero_a = P.serialize_ciphertext_tensor([1], [(P.Form(0, 1, 1), P.Form(1, 1, 1))])
    with pytest.raises(CofheHipError, match="outside the supported range"):
        unpack(zero_a)
    # forms that pass the range checks but are not reduced forms of THIS discriminant must be refused before they reach
    # the arithmetic: wrong discriminant, not reduced (a > c), non-canonical sign (b = -a)
    g = _random_tensor(d, 1, 9, nbase=1)[0][0]
    for forged in (P.Form((1 << 900) + 12345, 7, 5), P.Form(g.c, -g.b, g.a) if g.a != g.c else P.Form(g.a + 1, g.b, g.c),
                   P.Form(g.a, g.b, g.c + 1)):
        data = P.serialize_ciphertext_tensor([1], [(forged, g)])
        with pytest.raises(CofheHipError, match="not a reduced form"):
            unpack(data)
        with pytest.raises(CofheHipError, match="not a reduced form"):
            E.add_ciphertext_tensors(data, P.serialize_ciphertext_tensor([1], [(g, g)]))
    assert E.device_status() == 0          # nothing reached a safety cap
    # 1M-element scan path: offsets of a large tensor agree with the host packer
    n = 3000
    cts = _random_tensor(d, n, 6)
    data = P.serialize_ciphertext_tensor([n], cts)
    t = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    big = torch.zeros(2 * n * 168, dtype=torch.int32, device="cuda")
    assert E.unpack_tensor_device(t.data_ptr(), len(data), 2, big.data_ptr(), 2 * n) == ([n], 2 * n)
    cap = E.packed_size_bound(2 * n, 2, 1)
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    ln = E.pack_tensor_device(big.data_ptr(), 2 * n, 2, [n], out.data_ptr(), cap)
    assert out.cpu().numpy()[:ln].tobytes() == data


def test_fundamental_odd_discriminant_vs_oracle():
    """Delta = -q, q = 7 (mod 8) prime of 2088 bits: forms with odd b; add, 1-D scal and the
    matrix product byte-compared with the oracle"""
    rng = P.SplitMix64(91)
    q = P.random_prime(2088, rng, 7)
    d = -q
    E = engine(d)
    x = P.serialize_ciphertext_tensor([8, 8], _random_tensor(d, 64, 51, nbase=12))
    y = P.serialize_ciphertext_tensor([8, 8], _random_tensor(d, 64, 52, nbase=12))
    assert O.check_tensor(d, x) == 1
    assert E.add_ciphertext_tensors(x, y) == O.add(d, x, y)
    ident = (P.identity(d), P.identity(d))
    z = P.serialize_ciphertext_tensor([2], [ident, _random_tensor(d, 1, 53, nbase=2)[0]])
    assert E.add_ciphertext_tensors(z, z) == O.add(d, z, z)
    exps = [rng.bits(128) for _ in range(8)] + [0, -1, (1 << 128) - 1, 5]
    s = _pt_bytes([12], exps)
    c = P.serialize_ciphertext_tensor([12], _random_tensor(d, 12, 54, nbase=6))
    assert E.scal_ciphertext_tensors(s, c) == O.scal_1d(d, s, c)
    n, m, p = 3, 4, 2
    s2 = _pt_bytes([m, p], [j * p + k + 1 for j in range(m) for k in range(p)])
    c2 = P.serialize_ciphertext_tensor([n, m], _random_tensor(d, n * m, 55, nbase=6))
    zero = P.serialize_ciphertext_tensor([1], _random_tensor(d, 1, 56, nbase=2))
    assert E.scal_ciphertext_tensors(s2, c2, zero) == O.scal_2d(d, s2, c2, zero)


def test_encrypt_fixed_base_golden(golden):
    """encrypt_tensor on the GPU (k_encrypt: fixed-base signed-digit product for f^m, one h^r / pk^r per
    tensor) reproduces the fixture ciphertexts of the Python restatement bit for bit, given the same r;
    plaintexts that are negative, zero, 2^k - 1 or wider than k bits reduce mod 2^k"""
    import numpy as np
    import torch
    prm, vec = golden
    d, k = hx(prm["delta"]), prm["k"]
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record
    fr = lambda o: form_record(hx(o["a"]), hx(o["b"]), hx(o["c"]))
    frec = fr(prm["f"])
    rng = P.SplitMix64(vec["seed"])
    r1 = rng.below(hx(prm["exponent_bound"]))            # first draw of make_vectors
    base = torch.from_numpy(np.concatenate([fr(prm["h"]), fr(prm["pk"])]).view(np.int32)).cuda()
    ex = torch.from_numpy(exp_records([r1, r1]).view(np.int32)).cuda()
    hp = torch.empty_like(base)
    E.pow_form_records(base.data_ptr(), ex.data_ptr(), hp.data_ptr(), 2)

    def enc(ms):
        pl = torch.from_numpy(exp_records(ms).view(np.int32)).cuda()
        out = torch.zeros(len(ms) * 336, dtype=torch.int32, device="cuda")
        E.encrypt_records(pl.data_ptr(), hp.data_ptr(), frec, out.data_ptr(), len(ms), k)
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.uint32)

    assert E.records_to_bytes(enc([1, 2, 3, 4]), [2, 2]) == bytes.fromhex(vec["add_valid"]["ct1"])
    # odd cases against the closed-form meaning: c2 = pk^r o f^(m mod 2^k)
    M = 1 << k
    f = P.Form(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    _, golden_cts = P.deserialize_ciphertext_tensor(bytes.fromhex(vec["add_valid"]["ct1"]))
    c1, c2_of_1 = golden_cts[0]
    pkr = P.compose(c2_of_1, P.inverse(f))
    ms = [0, M - 1, -1, -5, M + 7, (M << 3) + 9, rng.bits(k), -rng.bits(k // 2)]
    want = [(c1, P.compose(pkr, P.power(f, m % M, d))) for m in ms]
    assert E.records_to_bytes(enc(ms), [len(ms)]) == P.serialize_ciphertext_tensor([len(ms)], want)


def test_pow_fixed_base_equals_ladder(golden):
    """h^e / pk^e through the context's cached tables base^(2^j) + product tree == the generic powering ladder,
    for zero, +-1, powers of two, all-ones, negative and full-width exponents; several bases evict each other"""
    import numpy as np
    import torch
    prm, vec = golden
    d = hx(prm["delta"])
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record
    fr = lambda o: form_record(hx(o["a"]), hx(o["b"]), hx(o["c"]))
    rng = P.SplitMix64(77)
    bound = hx(prm["exponent_bound"])
    exps = [0, 1, -1, 2, 3, 1 << 64, (1 << 200) - 1, -((1 << 131) + 5), rng.below(bound), -rng.below(bound), (1 << 991) + 1, (1 << 992) - 1]
    bases = [fr(prm["h"]), fr(prm["pk"])]
    _, cts = P.deserialize_ciphertext_tensor(bytes.fromhex(vec["add_valid"]["ct1"]))
    for c1, c2 in cts[:2]:                      # 4 more bases: the context keeps 4 tables, so some are rebuilt
        bases += [form_record(c1.a, c1.b, c1.c), form_record(c2.a, c2.b, c2.c)]
    for rnd in range(2):
        for bi, b in enumerate(bases if rnd == 0 else bases[:2]):
            es = exps if bi < 2 else exps[:4] + exps[8:9]
            ex = exp_records(es)
            want = torch.empty(len(es) * 168, dtype=torch.int32, device="cuda")
            bb = torch.from_numpy(np.tile(b, len(es)).view(np.int32)).cuda()
            d_ex = torch.from_numpy(ex.view(np.int32)).cuda()
            E.pow_form_records(bb.data_ptr(), d_ex.data_ptr(), want.data_ptr(), len(es))
            got = torch.empty_like(want)
            for i in range(len(es)):
                E.pow_fixed_base_record(b, ex.reshape(len(es), 32)[i], got.data_ptr() + i * 168 * 4)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (rnd, bi)
    # four powers of different bases and lengths in one tree == the four ladders
    es = [exps[8], -exps[9], 3, 0]
    bb = np.concatenate(bases[:4])
    ex = exp_records(es)
    want = torch.empty(4 * 168, dtype=torch.int32, device="cuda")
    d_bb, d_ex = torch.from_numpy(bb.view(np.int32)).cuda(), torch.from_numpy(ex.view(np.int32)).cuda()      # kept alive over the launch
    E.pow_form_records(d_bb.data_ptr(), d_ex.data_ptr(), want.data_ptr(), 4)
    got = torch.zeros_like(want)
    E.pow_fixed_base_records(bb, ex, got.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert E.device_status() == 0


def _records_of(E, cts):
    import numpy as np
    _, recs = E.bytes_to_records(P.serialize_ciphertext_tensor([len(cts)], cts))
    return recs.view(np.int32)


def test_add_128x128_full_bytes_vs_oracle(params128):
    """BASELINE config C2 at full size: every one of the 16 384 output ciphertexts byte-compared
    with the oracle (distinct random group elements per slot, built on the GPU from a pool)"""
    import numpy as np
    import torch
    d = hx(params128["delta"])
    E = engine(d)
    n = 128 * 128
    pool = torch.from_numpy(_records_of(E, _random_tensor(d, 96, 21, nbase=40)).reshape(96, 336)).cuda()
    rng = P.SplitMix64(22)
    ia = torch.tensor([rng.below(96) for _ in range(n)], device="cuda")
    ib = torch.tensor([rng.below(96) for _ in range(n)], device="cuda")
    ic = torch.tensor([rng.below(96) for _ in range(n)], device="cuda")

    def add(a, b):
        o = torch.empty_like(a)
        E.compose_records(a.data_ptr(), b.data_ptr(), o.data_ptr(), a.numel() // 168)
        torch.cuda.synchronize()
        return o
    # x = pool[ia] + pool[ib] gives ~9000 distinct elements; then the measured op: x + pool[ic]
    x = add(pool[ia].reshape(-1).contiguous(), pool[ib].reshape(-1).contiguous())
    y = pool[ic].reshape(-1).contiguous()
    z = add(x, y)
    to_b = lambda t: E.records_to_bytes(t.cpu().numpy().view(np.uint32), [128, 128])
    assert to_b(z) == O.add(d, to_b(x), to_b(y))


@pytest.mark.parametrize("pset", ["s128_k128", "s128_k256"])
def test_add_1024x1024_sampled(pset):
    """BASELINE config C5 shape (1 048 576 ciphertexts, one GPU), at both parameter sets the survey maps it to (k = 128,
    |Delta| = 2088 bits; k = 256, |Delta| = 2344 bits): commutativity on the whole tensor and a byte comparison of 2048
    sampled elements with the oracle"""
    import numpy as np
    import torch
    d = hx(load_json("params_%s.json" % pset)["delta"])
    E = engine(d)
    n = 1024 * 1024
    pool = torch.from_numpy(_records_of(E, _random_tensor(d, 64, 31, nbase=32)).reshape(64, 336)).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    ia = torch.randint(0, 64, (n,), device="cuda", generator=g)
    ib = torch.randint(0, 64, (n,), device="cuda", generator=g)
    x = pool[ia].reshape(-1).contiguous()
    y = pool[ib].reshape(-1).contiguous()
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    E.compose_records(x.data_ptr(), y.data_ptr(), o1.data_ptr(), 2 * n)
    E.compose_records(y.data_ptr(), x.data_ptr(), o2.data_ptr(), 2 * n)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    sel = torch.randint(0, n, (2048,), device="cuda", generator=g)
    pick = lambda t: E.records_to_bytes(t.view(n, 336)[sel].cpu().numpy().view(np.uint32).reshape(-1), [2048])
    assert pick(o1) == O.add(d, pick(x), pick(y))


def test_scal_matmul_256_sampled(params128):
    """BASELINE config C3 at its full size (cts 256x256, s 256x256 ramp exponents as in the harness): the
    windowed kernel's outputs for 8 rows x 8 columns (64 outputs spread over the matrix) byte-compared with the oracle"""
    import numpy as np
    import torch
    d = hx(params128["delta"])
    E = engine(d)
    n, m, p = 256, 256, 256
    pool_cts = _random_tensor(d, 48, 41, nbase=24)
    pool = torch.from_numpy(_records_of(E, pool_cts).reshape(48, 336)).cuda()
    rng = P.SplitMix64(42)
    idx = [rng.below(48) for _ in range(n * m)]
    cts = pool[torch.tensor(idx, device="cuda")].reshape(-1).contiguous()
    svals = [j * p + k + 1 for j in range(m) for k in range(p)]
    sys.path.insert(0, ROOT)
    from bench import exp_records
    ex = torch.from_numpy(exp_records(svals).view(np.int32)).cuda()
    zero_ct = _random_tensor(d, 1, 43, nbase=2)
    zero = torch.from_numpy(_records_of(E, zero_ct)).cuda()
    out = torch.empty(n * p * 336, dtype=torch.int32, device="cuda")
    E.scal_matmul_records(cts.data_ptr(), ex.data_ptr(), zero.data_ptr(), out.data_ptr(), n, m, p)
    torch.cuda.synchronize()
    rows, cols = [0, 1, 37, 100, 128, 191, 254, 255], [0, 1, 63, 100, 101, 200, 254, 255]
    sub_cts = [pool_cts[idx[i * m + j]] for i in rows for j in range(m)]
    sub_s = _pt_bytes([m, len(cols)], [svals[j * p + k] for j in range(m) for k in cols])
    want = O.scal_2d(d, sub_s, P.serialize_ciphertext_tensor([len(rows), m], sub_cts),
                     P.serialize_ciphertext_tensor([1], zero_ct))
    o = out.view(n, p, 336)
    got_recs = torch.stack([o[i, k] for i in rows for k in cols]).cpu().numpy().view(np.uint32).reshape(-1)
    assert E.records_to_bytes(got_recs, [len(rows), len(cols)]) == want


def test_c4_row_sharded_scal_matmul_through_rccl(params128, tmp_path):
    """BASELINE config C4: bench.py's row-sharded plaintext-matrix x ciphertext-matrix path (256x256 block per rank,
    replicated s and Enc(0), result rows all-gathered over RCCL) run on this one GPU with the process group forced on
    (world size 1: the same code path, collective included); 64 outputs of the gathered result vs the oracle"""
    import subprocess
    env = dict(os.environ, COFHE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "scal_matmul", "--rows", "256", "--cols", "256",
                        "--steps", "1", "--warmup", "1", "--dump-dir", str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    meta = json.load(open(tmp_path / "meta.json"))
    assert meta["distributed"] is True and meta["rccl_nranks"] == 1
    assert line["device_status"] == 0 and line["config"]["rccl_nranks"] == 1
    # the dominant kernel of the 256^3 product is the tree's (round 4); the Horner chain is among the others
    assert line["roofline"]["kernel"] == "k_tree_level" and line["roofline"]["launch_ms"] > 0
    assert line["roofline"]["other_kernels_ms"]["k_scal_matmul_wnaf"] > 0
    n, m, p = meta["n"], meta["m"], meta["p"]
    d = hx(params128["delta"])
    _, cts = P.deserialize_ciphertext_tensor(open(tmp_path / "cts.bin", "rb").read())
    _, out = P.deserialize_ciphertext_tensor(open(tmp_path / "out.bin", "rb").read())
    zero = open(tmp_path / "zero.bin", "rb").read()
    rows, cols = [0, 1, 37, 100, 128, 191, 254, 255], [0, 1, 63, 100, 101, 200, 254, 255]
    sub_cts = [cts[i * m + j] for i in rows for j in range(m)]
    sub_s = _pt_bytes([m, len(cols)], [j * p + k + 1 for j in range(m) for k in cols])
    want = O.scal_2d(d, sub_s, P.serialize_ciphertext_tensor([len(rows), m], sub_cts), zero)
    got = P.serialize_ciphertext_tensor([len(rows), len(cols)], [out[i * p + k] for i in rows for k in cols])
    assert got == want


def test_c_abi_communicator_all_gather_world_of_one(params128):
    """the C ABI's RCCL path (cofhe_hip_comm_create / cofhe_hip_all_gather_rows) on the one GPU there is: a world of one
    rank gathers its own rows, through both the all-gather branch and -- 7 rows over 1 rank is never ragged, so the
    branch is forced by gathering twice with different row counts -- the same buffers bit for bit"""
    import torch
    E = engine(hx(params128["delta"]))
    comm = E.comm_create(E.comm_unique_id(), 1, 0)
    try:
        assert E.shard_rows(7, 1, 0) == (0, 7)
        g = torch.Generator(device="cuda").manual_seed(5)
        for rows, cols in ((7, 3), (128, 2)):
            local = torch.randint(-2**31, 2**31 - 1, (rows * cols * 336,), dtype=torch.int32, device="cuda", generator=g)
            out = torch.zeros_like(local)
            E.all_gather_rows(comm, local.data_ptr(), rows, cols * 336 * 4, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(out, local)
            # the ragged route (one ncclBroadcast per block inside a group), forced: a world of one never has ragged blocks
            E.comm_set_option(comm, "force_grouped_broadcast", 1)
            out2 = torch.zeros_like(local)
            E.all_gather_rows(comm, local.data_ptr(), rows, cols * 336 * 4, out2.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(out2, local)
            E.comm_set_option(comm, "force_grouped_broadcast", 0)
        assert E.comm_info(comm) == (1, 0, 1)          # world, rank, and what RCCL itself reports (ncclCommCount)
        from cofhe_amd import CofheHipError
        with pytest.raises(CofheHipError):
            E.comm_set_option(comm, "no_such_option", 1)
    finally:
        E.comm_destroy(comm)


def test_add_ciphertext_records_shared_c1_folding(golden):
    """cofhe_hip_add_ciphertext_records == cofhe_hip_compose_records over all 2 n records, whether both operands share
    their c1 (encrypt_tensor-made: n + 1 compositions run), only one does, or none; one ciphertext; out aliasing a"""
    import numpy as np
    import torch
    prm, vec = golden
    d = hx(prm["delta"])
    E = engine(d)
    n = 300                                               # not a multiple of the 32 groups of a workgroup
    rng = P.SplitMix64(9)
    _, g = P.deserialize_ciphertext_tensor(bytes.fromhex(vec["add_valid"]["ct1"]))
    _, g2 = P.deserialize_ciphertext_tensor(bytes.fromhex(vec["add_valid"]["ct2"]))
    pool = _random_tensor(d, 40, 17, nbase=12)
    shared_a = [(g[0][0], pool[rng.below(40)][1]) for _ in range(n)]          # one c1, arbitrary c2
    shared_b = [(g2[0][0], pool[rng.below(40)][0]) for _ in range(n)]
    mixed = [pool[rng.below(40)] for _ in range(n)]
    almost = list(shared_a)
    almost[n - 1] = (pool[3][0], almost[n - 1][1])                           # a single differing c1, in the last ciphertext
    dev = lambda cts: torch.from_numpy(_records_of(E, cts)).cuda()
    for A, B in ((shared_a, shared_b), (shared_a, mixed), (mixed, shared_b), (almost, shared_b), (mixed, mixed), (shared_a[:1], shared_b[:1])):
        a, b = dev(A), dev(B)
        want = torch.empty_like(a)
        E.compose_records(a.data_ptr(), b.data_ptr(), want.data_ptr(), 2 * len(A))
        got = torch.zeros_like(a)
        E.add_ciphertext_records(a.data_ptr(), b.data_ptr(), got.data_ptr(), len(A))
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        E.add_ciphertext_records(a.data_ptr(), b.data_ptr(), a.data_ptr(), len(A))       # in place
        torch.cuda.synchronize()
        assert torch.equal(a, want)
    assert E.device_status() == 0


def test_compose_with_powers_of_f_and_the_bench_ops_regression(golden):
    """forms with a short first coefficient (f^(+-2^j): a = 2^(2(k-j)), down to 4) composed with full-size forms, both
    orders: the remainder sequence starts lopsided, takes long-division steps and swaps the pair.  Includes the
    plaintext / randomness pair on which encrypt_tensor once produced a form of the wrong discriminant (the serving
    lane kept its top-limb hints per name across the client's swap; found by tools/bench_ops.py)"""
    import numpy as np
    import torch
    prm, vec = golden
    d, k = hx(prm["delta"]), prm["k"]
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record
    F = lambda o: P.Form(hx(o["a"]), hx(o["b"]), hx(o["c"]))
    f, pk = F(prm["f"]), F(prm["pk"])
    m = 0xe35425b964bdb6d05a03893b5c79a49c
    r = int("c82e101ee83d683efd4905a925cbbc24f11c50088370731d23689cedb7caca5532b1e56a5bd176f91893d737e90a739d12de7f4321468c73ea174c3"
            "76cae7cb7d7ea367748b2e6efe01e16b79d801488717264fc5d68823f9416023e7bab39b017539f8bea12672556de3b214a20b71dfc4cbbd4e9d2d02e", 16)
    pkr = P.power(pk, r, d)
    _, cts = P.deserialize_ciphertext_tensor(bytes.fromhex(vec["add_valid"]["ct1"]))
    xs = [pkr, cts[0][0], cts[1][1]]
    lhs, rhs = [], []
    fj = f
    for j in range(k):
        for x in xs:
            for y in (fj, P.inverse(fj)):
                lhs += [x, y]
                rhs += [y, x]
        fj = P.compose(fj, fj)
    # the chain pk^r o f^m of the old chain kernel, exact partial products as left operands: its 8th step is the
    # composition that failed (a 1042-bit first coefficient against 2^218)
    acc, x3 = pkr, 3 * m
    for j in range(k):
        dg = ((x3 >> (j + 1)) & 1) - ((m >> (j + 1)) & 1)
        if dg:
            y = P.power(f, 1 << j, d)
            y = P.inverse(y) if dg < 0 else y
            lhs.append(acc)
            rhs.append(y)
            acc = P.compose(acc, y)
    rec = lambda forms: torch.from_numpy(np.concatenate([form_record(t.a, t.b, t.c) for t in forms]).view(np.int32)).cuda()
    a, b = rec(lhs), rec(rhs)
    out = torch.zeros_like(a)
    E.compose_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), len(lhs))
    torch.cuda.synchronize()
    want = rec([P.compose(u, v) for u, v in zip(lhs, rhs)])
    assert torch.equal(out, want)
    # the encryption itself: c2 = pk^r o f^m
    hp = rec([P.power(F(prm["h"]), r, d), pkr])
    pl = torch.from_numpy(exp_records([m, m ^ 1, 4, (1 << k) - 4]).view(np.int32)).cuda()
    enc = torch.zeros(4 * 336, dtype=torch.int32, device="cuda")
    E.encrypt_records(pl.data_ptr(), hp.data_ptr(), form_record(f.a, f.b, f.c), enc.data_ptr(), 4, k)
    torch.cuda.synchronize()
    assert E.validate_records(enc.data_ptr(), 8)
    got_c2 = enc.view(4, 2, 168)[:, 1].contiguous().view(-1)
    assert torch.equal(got_c2, rec([P.compose(pkr, P.power(f, e, d)) for e in (m, m ^ 1, 4, (1 << k) - 4)]))
    assert E.device_status() == 0


def test_decrypt_shared_and_mixed_first_components(golden):
    """decrypt_tensor / part_decrypt_tensor on a tensor whose ciphertexts share c1 (encrypt_tensor's one r: ONE ladder runs,
    its result is copied) and on a tensor mixed from two encryptions (every ladder runs): same plaintexts, and the
    partial decryptions equal the ones computed 32 ciphertexts at a time (below the folding threshold)"""
    import numpy as np
    import torch
    prm, _vec = golden
    d, k = hx(prm["delta"]), prm["k"]
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import SplitMix64, exp_records, form_record
    from gpu_inputs import encrypt_tensor_gpu
    dev = torch.device("cuda", 0)
    rng = SplitMix64(31)
    n = 160
    ms1 = [rng.bits(k) for _ in range(n)]
    ms2 = [rng.bits(k) for _ in range(n)]
    t1 = encrypt_tensor_gpu(E, torch, prm, ms1, rng.bits(900), dev)
    t2 = encrypt_tensor_gpu(E, torch, prm, ms2, rng.bits(900), dev)
    mixed = torch.cat([t1[: 80 * 336], t2[80 * 336:]]).contiguous()
    want_mixed = ms1[:80] + ms2[80:]
    frec = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    dsk = torch.from_numpy(exp_records([hx(prm["sk"])]).view(np.int32)).to(dev)
    ow = (k + 31) // 32 + 1
    for cts, want in ((t1, ms1), (mixed, want_mixed)):
        pt = torch.zeros(n * ow, dtype=torch.int32, device=dev)
        E.decrypt_records(cts.data_ptr(), dsk.data_ptr(), frec, pt.data_ptr(), n, k)
        torch.cuda.synchronize()
        a = pt.cpu().numpy().view(np.uint32).reshape(n, ow)
        assert not a[:, -1].any()
        assert [int.from_bytes(a[i, :-1].tobytes(), "little") for i in range(n)] == want
        whole = torch.zeros(n * 168, dtype=torch.int32, device=dev)
        E.part_decrypt_records(cts.data_ptr(), dsk.data_ptr(), whole.data_ptr(), n)
        pieces = torch.zeros(n * 168, dtype=torch.int32, device=dev)
        for i0 in range(0, n, 32):
            E.part_decrypt_records(cts.data_ptr() + i0 * 336 * 4, dsk.data_ptr(), pieces.data_ptr() + i0 * 168 * 4, 32)
        torch.cuda.synchronize()
        assert torch.equal(whole, pieces)
    assert E.device_status() == 0


@pytest.mark.parametrize("name,n", [("s128_k128", 32768), ("s128_k256", 8192)])
def test_roundtrip_soak(name, n):
    """tools/gpu_soak_roundtrip.py: encrypt n plaintexts (all powers of two, their negatives, small and random values),
    decrypt, add the tensor to its reverse and decrypt again -- ~n * 2500 compositions, each checked by the value that comes
    back; encrypted forms valid, no error flags, status word clear"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_soak_roundtrip.py"), str(n), "11", name],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["encrypted_forms_valid"] is True
    assert (line["decrypt_flags"], line["decrypt_wrong"], line["sum_flags"], line["sum_wrong"], line["device_status"]) == (0, 0, 0, 0, 0)
    assert (line["mixed_decrypt_flags"], line["mixed_decrypt_wrong"], line["plain_add_equals_folded"]) == (0, 0, True)


@pytest.mark.parametrize("pset", ["s128_k128", "s128_k256"])
def test_lopsided_pairs_fuzz_vs_oracle(pset):
    """>= 20 000 compositions of operand pairs whose first coefficients differ in length by anything from 0 to ~1040
    bits, BOTH orders (the kernel orders the pair, the serving lane renames it after long-division steps -- the code
    path of round 2's stale-hint bug), squarings and inverse pairs included, byte-compared with the oracle"""
    import numpy as np
    import torch
    prm = load_json("params_%s.json" % pset)
    d, k = hx(prm["delta"]), prm["k"]
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import form_record
    f = P.Form(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    pool = _lopsided_pool(d, k, f)
    lens = sorted({x.a.bit_length() for x in pool})
    assert lens[0] == 1 and lens[1] <= 3 and lens[-1] >= (-d).bit_length() // 2 - 4 and len(lens) > 60
    n = len(pool)
    rng = P.SplitMix64(616)
    pairs = [(i, i) for i in range(n)]                                   # squarings
    pairs += [(i, n + i) for i in range(n)]                              # x o x^-1 (index >= n: the inverse)
    while len(pairs) < 10240:
        pairs.append((rng.below(2 * n), rng.below(2 * n)))
    pairs += [(j, i) for i, j in pairs]                                  # the other order of every pair
    assert len(pairs) >= 20000
    allf = pool + [P.inverse(x) for x in pool]
    recs = torch.from_numpy(np.stack([form_record(x.a, x.b, x.c) for x in allf]).view(np.int32)).cuda()
    ia = torch.tensor([i for i, _ in pairs], device="cuda")
    ib = torch.tensor([j for _, j in pairs], device="cuda")
    a, b = recs[ia].reshape(-1).contiguous(), recs[ib].reshape(-1).contiguous()
    out = torch.zeros_like(a)
    E.compose_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), len(pairs))
    torch.cuda.synchronize()
    assert E.device_status(clear=False) == 0
    assert E.validate_records(out.data_ptr(), len(pairs))
    m = len(pairs) // 2
    to_b = lambda t: E.records_to_bytes(t.cpu().numpy().view(np.uint32), [m])       # two forms per "ciphertext"
    assert to_b(out) == O.add(d, to_b(a), to_b(b))


@pytest.mark.parametrize("pset", ["s128_k128", "s128_k256"])
def test_wide_layout_composition_vs_oracle(pset):
    """the wavefront-wide composition of the latency kernels (csrc/wide.hpp, qfw.hpp; one composition per wavefront through
    cofhe_hip_compose_wide_records) byte-compared with the oracle: 4 096 pairs of independent random forms (the common
    route), their squarings, and the lopsided pool in both orders with inverse pairs and the identity -- most of which the wide
    route declines (common factors, a long third coefficient) and hands to qf_compose on 8 lanes of the same wavefront.
    Both routes must be taken, and the fallback must stay the exception on random forms."""
    import numpy as np
    import torch
    prm = load_json("params_%s.json" % pset)
    d, k = hx(prm["delta"]), prm["k"]
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record, SplitMix64
    f = P.Form(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    # (1) random forms made on the device (h^(e_i), 192-bit e_i)
    n = 4096
    hrec = form_record(hx(prm["h"]["a"]), hx(prm["h"]["b"]), hx(prm["h"]["c"]))
    hbase = torch.from_numpy(np.tile(hrec, n).view(np.int32)).cuda()

    def family(seed):
        r_ = SplitMix64(seed)
        ex = torch.from_numpy(exp_records([r_.bits(192) | 1 for _ in range(n)]).view(np.int32)).cuda()
        o = torch.empty(n * 168, dtype=torch.int32, device="cuda")
        E.pow_form_records(hbase.data_ptr(), ex.data_ptr(), o.data_ptr(), n)
        torch.cuda.synchronize()
        return o
    fa, fb = family(777), family(778)
    a = torch.cat([fa, fa[: 512 * 168]])                                        # + 512 squarings
    b = torch.cat([fb, fa[: 512 * 168]])
    tot = n + 512
    out_w, out_t = torch.zeros_like(a), torch.zeros_like(a)
    fallbacks = E.compose_wide_records(a.data_ptr(), b.data_ptr(), out_w.data_ptr(), tot, count_fallbacks=True)
    E.compose_records(a.data_ptr(), b.data_ptr(), out_t.data_ptr(), tot)
    torch.cuda.synchronize()
    assert E.device_status(clear=False) == 0
    assert torch.equal(out_w, out_t)                                              # the throughput kernel is checked against the oracle elsewhere
    to_b = lambda t, m: E.records_to_bytes(t.cpu().numpy().view(np.uint32), [m])
    assert to_b(out_w[: 2048 * 168], 1024) == O.add(d, to_b(a[: 2048 * 168], 1024), to_b(b[: 2048 * 168], 1024))
    assert 0 < fallbacks < tot // 10, fallbacks                                   # ~1-2 % of random pairs keep a common factor
    # (2) the lopsided pool: every structure the 8-lane route exists for
    pool = _lopsided_pool(d, k, f)
    allf = pool + [P.inverse(x) for x in pool]
    m = len(allf)
    rng = P.SplitMix64(919)
    pairs = [(i, i) for i in range(m)] + [(i, (i + len(pool)) % m) for i in range(m)]
    while len(pairs) < 2048:
        pairs.append((rng.below(m), rng.below(m)))
    pairs += [(j, i) for i, j in pairs]
    recs = torch.from_numpy(np.stack([form_record(x.a, x.b, x.c) for x in allf]).view(np.int32)).cuda()
    ia = torch.tensor([i for i, _ in pairs], device="cuda")
    ib = torch.tensor([j for _, j in pairs], device="cuda")
    a, b = recs[ia].reshape(-1).contiguous(), recs[ib].reshape(-1).contiguous()
    out = torch.zeros_like(a)
    fallbacks = E.compose_wide_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), len(pairs), count_fallbacks=True)
    torch.cuda.synchronize()
    assert E.device_status(clear=False) == 0
    assert E.validate_records(out.data_ptr(), len(pairs))
    h = len(pairs) // 2
    assert to_b(out, h) == O.add(d, to_b(a, h), to_b(b, h))
    assert fallbacks > len(pairs) // 10


def test_big_launches_with_long_common_factors(params128):
    """2^18 + 37 compositions in one launch: random forms (tiled, checked against a small launch of the same kernel, itself
    checked against the oracle elsewhere) with powers of f -- equal 256-bit first coefficients, a third coefficient too long
    for another representative, hence a common factor far beyond a word and the multi-limb branch of the general formula --
    planted at the first, the last and scattered positions, those byte-compared with the oracle; twice in a row on one
    stream, and through the ciphertext-level entry with distinct and with shared c1.  (Written for round 4's deferral of
    such pairs to a fix-up pass, experiments/deferred_big_factor/; kept as the parity test of that branch at scale.)"""
    import numpy as np
    import torch
    prm = params128
    d, k = hx(prm["delta"]), prm["k"]
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record, SplitMix64
    f = P.Form(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    n = (1 << 18) + 37
    pooln = 2048
    hrec = form_record(hx(prm["h"]["a"]), hx(prm["h"]["b"]), hx(prm["h"]["c"]))
    hbase = torch.from_numpy(np.tile(hrec, pooln).view(np.int32)).cuda()

    def family(seed):
        r_ = SplitMix64(seed)
        ex = torch.from_numpy(exp_records([r_.bits(192) | 1 for _ in range(pooln)]).view(np.int32)).cuda()
        o = torch.empty(pooln * 168, dtype=torch.int32, device="cuda")
        E.pow_form_records(hbase.data_ptr(), ex.data_ptr(), o.data_ptr(), pooln)
        torch.cuda.synchronize()
        return o
    pa, pb = family(4321), family(4322)
    ref = torch.zeros_like(pa)
    E.compose_records(pa.data_ptr(), pb.data_ptr(), ref.data_ptr(), pooln)            # the complete kernel (small launch)
    reps = (n + pooln - 1) // pooln
    a = pa.view(pooln, 168).repeat(reps, 1)[:n].contiguous()
    b = pb.view(pooln, 168).repeat(reps, 1)[:n].contiguous()
    want = ref.view(pooln, 168).repeat(reps, 1)[:n].contiguous()
    rng = P.SplitMix64(99)
    spots = [0, 1, 31, 32, 33, 4095, 70001, (1 << 17) + 5, (1 << 18) - 1, 1 << 18, n - 2, n - 1]
    for i, pos in enumerate(spots):
        m1, m2 = rng.bits(k - 1) | 1, rng.bits(k - 1) | 1
        x, y = P.power(f, m1), P.power(f, m2 if i % 3 else -m1)                      # every third one: a form and its inverse
        z = P.compose(x, y)
        a[pos] = torch.from_numpy(form_record(x.a, x.b, x.c).view(np.int32))
        b[pos] = torch.from_numpy(form_record(y.a, y.b, y.c).view(np.int32))
        want[pos] = torch.from_numpy(form_record(z.a, z.b, z.c).view(np.int32))
    assert P.power(f, 3).a.bit_length() > 64                                          # the planted pairs do share a long factor
    out = torch.zeros_like(a)
    for _ in range(2):
        out.zero_()
        E.compose_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        assert E.device_status(clear=False) == 0
        assert torch.equal(out, want)
    # ciphertext-level entry: n_ct = 2^18 ciphertexts = the same records taken in pairs (distinct c1), then with one shared c1
    n_ct = 1 << 18
    a2 = torch.cat([a, a[: 2 * n_ct - n]])
    b2 = torch.cat([b, b[: 2 * n_ct - n]])
    w2 = torch.cat([want, want[: 2 * n_ct - n]])
    out2 = torch.zeros_like(a2)
    E.add_ciphertext_records(a2.data_ptr(), b2.data_ptr(), out2.data_ptr(), n_ct)
    torch.cuda.synchronize()
    assert E.device_status(clear=False) == 0 and torch.equal(out2, w2)
    a3, b3, w3 = a2.clone(), b2.clone(), w2.clone()
    a3.view(n_ct, 2, 168)[:, 0, :] = a2[5]                                             # every c1 the same record
    b3.view(n_ct, 2, 168)[:, 0, :] = b2[5]
    w3.view(n_ct, 2, 168)[:, 0, :] = w2[5]
    # c2 slots keep the planted pairs that fell on odd record indices; plant two more there to be sure
    for pos in (3, 2 * n_ct - 1):
        a3[pos], b3[pos], w3[pos] = a[0], b[0], want[0]
    out3 = torch.zeros_like(a3)
    E.add_ciphertext_records(a3.data_ptr(), b3.data_ptr(), out3.data_ptr(), n_ct)
    torch.cuda.synchronize()
    assert E.device_status(clear=False) == 0 and torch.equal(out3, w3)


def test_decrypt_ladder_forms_agree(params128):
    """the four forms of the shared-exponent ladder (option "ladder_form": a pair of wavefronts per ladder in the wide layout
    -- one squares, one multiplies --, the 8-lane in-wave form, the throughput kernel, one wavefront per ladder with a table)
    give the same partial decryptions, for one ciphertext, a handful, and a tensor that shares its c1; short, zero and
    negative exponents through the pair as well"""
    import numpy as np
    import torch
    prm = params128
    d, k = hx(prm["delta"]), prm["k"]
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record
    from gpu_inputs import encrypt_tensor_gpu
    dev = torch.device("cuda", 0)
    rng = P.SplitMix64(1212)
    sk = rng.bits(960)
    dsk = torch.from_numpy(exp_records([sk]).view(np.int32)).cuda()
    shared = encrypt_tensor_gpu(E, torch, prm, [rng.bits(k) for _ in range(100)], rng.bits(900), dev)       # one c1 for all 100
    mixed = torch.cat([encrypt_tensor_gpu(E, torch, prm, [rng.bits(k)], rng.bits(900), dev) for _ in range(5)])
    for cts, n in ((shared[: 336], 1), (mixed, 5), (shared, 100)):
        outs = []
        for form in (1, 2, 3, 4):
            E.set_option("ladder_form", form)
            out = torch.zeros(n * 168, dtype=torch.int32, device="cuda")
            E.part_decrypt_records(cts.data_ptr(), dsk.data_ptr(), out.data_ptr(), n)
            torch.cuda.synchronize()
            outs.append(out.cpu())
        E.set_option("ladder_form", 0)
        assert E.device_status(clear=False) == 0
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3]), n
    for e in (0, 1, 2, 3, -1, -5, 0x55555555, (1 << 64) - 1, -(1 << 70) + 12345, rng.bits(990)):
        de = torch.from_numpy(exp_records([e]).view(np.int32)).cuda()
        outs = []
        for form in (1, 3):
            E.set_option("ladder_form", form)
            out = torch.zeros(5 * 168, dtype=torch.int32, device="cuda")
            E.part_decrypt_records(mixed.data_ptr(), de.data_ptr(), out.data_ptr(), 5)
            torch.cuda.synchronize()
            outs.append(out.cpu())
        E.set_option("ladder_form", 0)
        assert E.device_status(clear=False) == 0
        assert torch.equal(outs[0], outs[1]), e


def test_add_128x128_independent_forms_full_bytes(params128):
    """BASELINE config C2 at full size on input family (ii) of SURVEY 8(d): 2 x 32 768 INDEPENDENT random forms (h^(e_i),
    independent 192-bit e_i, made by the product's ladder and validated on the device), every one of the 16 384 output
    ciphertexts byte-compared with the oracle -- the family bench.py times as input_family_ii"""
    import numpy as np
    import torch
    d = hx(params128["delta"])
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record, SplitMix64
    n = 2 * 128 * 128
    h = form_record(hx(params128["h"]["a"]), hx(params128["h"]["b"]), hx(params128["h"]["c"]))
    base = torch.from_numpy(np.tile(h, n).view(np.int32)).cuda()

    def family(seed):
        rng = SplitMix64(seed)
        ex = torch.from_numpy(exp_records([rng.bits(192) | 1 for _ in range(n)]).view(np.int32)).cuda()
        out = torch.empty(n * 168, dtype=torch.int32, device="cuda")
        E.pow_form_records(base.data_ptr(), ex.data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        assert E.validate_records(out.data_ptr(), n)
        return out
    x, y = family(4242), family(4343)
    # independent: (almost) no two forms of a family coincide
    assert len({bytes(r) for r in x.view(n, 168)[:, :40].cpu().numpy()}) > n - 4
    z = torch.empty_like(x)
    E.compose_records(x.data_ptr(), y.data_ptr(), z.data_ptr(), n)
    torch.cuda.synchronize()
    to_b = lambda t: E.records_to_bytes(t.cpu().numpy().view(np.uint32), [128, 128])
    assert to_b(z) == O.add(d, to_b(x), to_b(y))


def test_fixed_base_tiny_exponents_fresh_context(params128):
    """2-4 tiny exponents of different non-adjacent-form weight on a NEW context (smallest possible workspace): the
    padding entries of the gather name no table and must not read one (a table pointer was once fetched for them,
    past the end of the index buffer)"""
    import numpy as np
    import torch
    torch.cuda.init()
    from cofhe_amd import Engine
    sys.path.insert(0, ROOT)
    from bench import exp_records, form_record
    d = hx(params128["delta"])
    fr = lambda o: form_record(hx(o["a"]), hx(o["b"]), hx(o["c"]))
    for es in ([1, 7], [3, 1, 21], [5, 1, 0, 85], [1, 2, 4]):
        E = Engine(d)
        try:
            bb = np.concatenate([fr(params128["h"]), fr(params128["pk"])] * 2)[: len(es) * 168]
            ex = exp_records(es)
            d_bb, d_ex = torch.from_numpy(bb.view(np.int32)).cuda(), torch.from_numpy(ex.view(np.int32)).cuda()
            want = torch.empty(len(es) * 168, dtype=torch.int32, device="cuda")
            E.pow_form_records(d_bb.data_ptr(), d_ex.data_ptr(), want.data_ptr(), len(es))
            got = torch.zeros_like(want)
            E.pow_fixed_base_records(bb, ex, got.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(got, want), es
            assert E.device_status() == 0
        finally:
            E.close()


def test_pow_records_in_place_equals_out_of_place(params128):
    """k_pow keeps the running power in the item's output record; an output that IS the base tensor (in-place use of
    cofhe_hip_pow_records / cofhe_hip_pow_form_records) must still see the original bases: the launcher copies them"""
    import numpy as np
    import torch
    d = hx(params128["delta"])
    E = engine(d)
    sys.path.insert(0, ROOT)
    from bench import exp_records
    n = 37                                   # 74 records: two full workgroups and a ragged third
    cts = _random_tensor(d, n, 311)
    data = P.serialize_ciphertext_tensor([n], cts)
    _, recs = E.bytes_to_records(data)
    rng = P.SplitMix64(312)
    exps = [0, 1, -1, 2, -3] + [rng.bits(96) * (-1 if i % 3 == 0 else 1) for i in range(n - 5)]
    de = torch.from_numpy(exp_records(exps).view(np.int32)).cuda()
    base = torch.from_numpy(recs.view(np.int32)).cuda()
    out = torch.zeros_like(base)
    E.pow_records(base.data_ptr(), de.data_ptr(), out.data_ptr(), n)
    inplace = base.clone()
    E.pow_records(inplace.data_ptr(), de.data_ptr(), inplace.data_ptr(), n)
    torch.cuda.synchronize()
    assert torch.equal(out, inplace)
    got = E.records_to_bytes(out.cpu().numpy().view(np.uint32), [n])
    assert got == O.scal_1d(d, _pt_bytes([n], exps), data)
    # per-form exponents (exp_mode 1), in place as well
    fe = [rng.bits(64) for _ in range(2 * n)]
    dfe = torch.from_numpy(exp_records(fe).view(np.int32)).cuda()
    out2 = torch.zeros_like(base)
    E.pow_form_records(base.data_ptr(), dfe.data_ptr(), out2.data_ptr(), 2 * n)
    inplace2 = base.clone()
    E.pow_form_records(inplace2.data_ptr(), dfe.data_ptr(), inplace2.data_ptr(), 2 * n)
    torch.cuda.synchronize()
    assert torch.equal(out2, inplace2)


@pytest.mark.parametrize("n,m,p", [(3, 9, 7), (16, 6, 3), (5, 4, 40), (40, 3, 2)])
def test_scal_matmul_column_major_chains(params128, n, m, p):
    """the chains of k_scal_matmul_wnaf are numbered column-major (16 rows x 2 forms of one column per workgroup when
    2 n is a multiple of 32): shapes whose workgroups hold several columns, part of a column, or a ragged tail, with
    exponents of very different lengths per column (so that neighbouring chains have different schedules)"""
    d = hx(params128["delta"])
    E = engine(d)
    rng = P.SplitMix64(1000 + n * 64 + m * 8 + p)
    exps = []
    for j in range(m):
        for k in range(p):
            bits = (1, 7, 33, 128, 64, 2, 90)[(k + 3 * j) % 7]
            e = rng.bits(bits) | (1 << (bits - 1))
            exps.append(0 if (j + k) % 11 == 5 else (-e if (j * p + k) % 4 == 1 else e))
    cts = _random_tensor(d, n * m, 2000 + n)
    zero = _random_tensor(d, 1, 2001, nbase=2)
    s = _pt_bytes([m, p], exps)
    ct = P.serialize_ciphertext_tensor([n, m], cts)
    z = P.serialize_ciphertext_tensor([1], zero)
    assert E.scal_ciphertext_tensors(s, ct, z) == O.scal_2d(d, s, ct, z)


@pytest.mark.parametrize("n,m,p,mode", [(2, 20, 3, "zeros"), (3, 9, 4, "zero_column"), (2, 33, 2, "zero_segment"), (16, 1, 2, "one_base"),
                                        (1, 64, 2, "zeros")])
def test_scal_matmul_schedules_with_nothing_to_multiply(params128, n, m, p, mode):
    """the per-column schedule (k_matmul_schedule) when a column, a segment of the inner dimension or the whole exponent
    matrix is zero: the chain then starts from Enc(0) (one segment) or from the principal form (a segment of a split
    product) without a single table entry"""
    d = hx(params128["delta"])
    E = engine(d)
    rng = P.SplitMix64(7000 + n * 100 + m * 10 + p)
    exps = []
    for j in range(m):
        for k in range(p):
            e = rng.bits(40) | 1
            if mode == "zeros" or (mode == "zero_column" and k == 1) or (mode == "zero_segment" and j < 12):
                e = 0
            exps.append(e)
    cts = _random_tensor(d, n * m, 7100 + m)
    zero = _random_tensor(d, 1, 7101, nbase=2)
    s = _pt_bytes([m, p], exps)
    ct = P.serialize_ciphertext_tensor([n, m], cts)
    z = P.serialize_ciphertext_tensor([1], zero)
    assert E.scal_ciphertext_tensors(s, ct, z) == O.scal_2d(d, s, ct, z)
