"""Per-operation timings of every entry point of the path, tensors resident in HBM (run through
gpurun; one JSON line per operation, copied to profiles/).  Sizes follow SURVEY.md section 8(d):
C2 matadd 128x128, C3 scal_matmul 256^3 (ramp exponents) and with 128-bit exponents (smaller n),
C5 matadd 1024x1024, 1-D scal with k-bit exponents, negation, decryption, threshold decryption,
accumulation of the ct x ct product."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import SplitMix64, exp_records, form_record, hx

sys.path.insert(0, os.path.join(ROOT, "tests"))

from gpu_inputs import encrypt_tensor_gpu
from cofhe_amd import Engine

prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
K = prm["k"]
eng = Engine(hx(prm["delta"]))
dev = torch.device("cuda", 0)
rng = SplitMix64(7)
QUICK = "--quick" in sys.argv
SKIP = set(os.environ.get("OPS_SKIP", "").split(","))       # diagnostics: leave out "big" (1024x1024) and / or "k256"


def timed(fn, reps=1):
    fn()                                  # warm-up (also builds cached tables)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def emit(op, shape, sec, units, unit, **extra):
    d = {"op": op, "shape": shape, "ms": round(sec * 1e3, 3), "rate": round(units / sec, 1), "unit": unit}
    d.update(extra)
    print(json.dumps(d), flush=True)


def dev_i32(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)


def fresh(n):
    return encrypt_tensor_gpu(eng, torch, prm, [rng.bits(K) for _ in range(n)], rng.bits(960), dev)


# ---- matadd C2 / C5 ---------------------------------------------------------------------------
for side in ((128,) if (QUICK or "big" in SKIP) else (128, 1024)):
    E = side * side
    a, b = fresh(E), fresh(E)
    out = torch.empty_like(a)
    sec = timed(lambda: eng.compose_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), 2 * E), reps=10)
    emit("add_ciphertext_tensors", [side, side], sec, E, "ciphertext-ops/s", kernel="k_compose_wg")
    # the ciphertext-level entry: both operands come from encrypt_tensor (one r each), so c1 o c1' is folded
    sec = timed(lambda: eng.add_ciphertext_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), E), reps=10)
    emit("add_ciphertext_tensors, shared c1 folded (cofhe_hip_add_ciphertext_records)", [side, side], sec, E, "ciphertext-ops/s",
         kernel="k_c1_distinct + k_add_ct + k_c1_spread")
    del a, b, out

# ---- C5's second parameter set: security 128, k = 256 (examples/node.cpp:33-34), |Delta| = 2344 bits ----
if not QUICK and "k256" not in SKIP:
    prm2 = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k256.json")))
    eng2 = Engine(hx(prm2["delta"]))
    E2 = 128 * 128
    a = encrypt_tensor_gpu(eng2, torch, prm2, [rng.bits(256) for _ in range(E2)], rng.bits(960), dev)
    b = encrypt_tensor_gpu(eng2, torch, prm2, [rng.bits(256) for _ in range(E2)], rng.bits(960), dev)
    out = torch.empty_like(a)
    sec = timed(lambda: eng2.compose_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), 2 * E2), reps=10)
    emit("add_ciphertext_tensors, k = 256 parameters", [128, 128], sec, E2, "ciphertext-ops/s", kernel="k_compose_wg",
         delta_bits=(-hx(prm2["delta"])).bit_length())
    del a, b, out
    if "big" not in SKIP:       # C5: 1024 x 1024 at the k = 256 parameter set
        E3 = 1024 * 1024
        a = encrypt_tensor_gpu(eng2, torch, prm2, [rng.bits(256) for _ in range(E3)], rng.bits(960), dev)
        b = encrypt_tensor_gpu(eng2, torch, prm2, [rng.bits(256) for _ in range(E3)], rng.bits(960), dev)
        out = torch.empty_like(a)
        sec = timed(lambda: eng2.compose_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), 2 * E3), reps=3)
        emit("add_ciphertext_tensors, k = 256 parameters (C5)", [1024, 1024], sec, E3, "ciphertext-ops/s", kernel="k_compose_wg")
        sec = timed(lambda: eng2.add_ciphertext_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), E3), reps=3)
        emit("add_ciphertext_tensors, k = 256 parameters (C5), shared c1 folded", [1024, 1024], sec, E3, "ciphertext-ops/s")
        del a, b, out
    del eng2

# ---- PCIe-inclusive: serialised host bytes in, serialised host bytes out -------------------------
E = 128 * 128
a, b = fresh(E), fresh(E)
ha = eng.records_to_bytes(a.cpu().numpy().view(np.uint32), [128, 128])
hb = eng.records_to_bytes(b.cpu().numpy().view(np.uint32), [128, 128])
t0 = time.perf_counter()
for _ in range(3):
    hc = eng.add_ciphertext_tensors(ha, hb)
sec = (time.perf_counter() - t0) / 3
emit("add_ciphertext_tensors, host bytes in/out (PCIe + GPU (de)serialisation)", [128, 128], sec, E, "ciphertext-ops/s",
     bytes_in=len(ha) + len(hb), bytes_out=len(hc))
del a, b

# ---- 1-D scal with k-bit exponents, negation (2^k - 1) ------------------------------------------
E = 128 * 128
cts = fresh(E)
out = torch.empty_like(cts)
ex = dev_i32(exp_records([rng.bits(K) for _ in range(E)]))
sec = timed(lambda: eng.pow_records(cts.data_ptr(), ex.data_ptr(), out.data_ptr(), E))
emit("scal_ciphertext_tensors 1-D, %d-bit exponents" % K, [E], sec, E, "ciphertexts/s", kernel="k_pow")
ex = dev_i32(exp_records([(1 << K) - 1] * E))
sec = timed(lambda: eng.pow_records(cts.data_ptr(), ex.data_ptr(), out.data_ptr(), E))
emit("negate_ciphertext_tensor (exponent 2^k - 1)", [E], sec, E, "ciphertexts/s", kernel="k_pow")

# ---- encryption with given randomness (fixed-base f^m) --------------------------------------------
pl = dev_i32(exp_records([rng.bits(K) for _ in range(E)]))
fr_ = lambda o: form_record(hx(o["a"]), hx(o["b"]), hx(o["c"]))
hp = dev_i32(np.concatenate([fr_(prm["h"]), fr_(prm["pk"])]))       # stand-ins for h^r, pk^r
enc = torch.empty(E * 336, dtype=torch.int32, device=dev)
sec = timed(lambda: eng.encrypt_records(pl.data_ptr(), hp.data_ptr(), fr_(prm["f"]), enc.data_ptr(), E, K))
emit("encrypt_tensor (h^r, pk^r given)", [E], sec, E, "ciphertexts/s", kernel="k_encrypt_select + k_gather_signed + k_compose_pairs tree + k_zip_ciphertexts")
# the whole call: h^r and pk^r through the fixed-base tables of the context (first use builds them), then k_encrypt
r_ex = exp_records([rng.bits(960)])
hp2 = torch.empty(2 * 168, dtype=torch.int32, device=dev)


def encrypt_whole():
    eng.pow_fixed_base_records(np.concatenate([fr_(prm["h"]), fr_(prm["pk"])]), np.concatenate([r_ex, r_ex]), hp2.data_ptr())
    eng.encrypt_records(pl.data_ptr(), hp2.data_ptr(), fr_(prm["f"]), enc.data_ptr(), E, K)


t0 = time.perf_counter()
encrypt_whole()
torch.cuda.synchronize()
first = time.perf_counter() - t0
sec = timed(encrypt_whole, reps=3)
emit("encrypt_tensor, whole call incl. h^r and pk^r (fixed-base tables)", [E], sec, E, "ciphertexts/s",
     kernel="fixed-base tree for h^r, pk^r + the element tree", first_call_ms=round(first * 1e3, 1),
     first_call_note="builds the tables h^(2^j), pk^(2^j): one chain of ~1000 squarings each (k_square_chain)")
for E1 in (1, 64):
    pl1 = dev_i32(exp_records([rng.bits(K) for _ in range(E1)]))
    enc1 = torch.empty(E1 * 336, dtype=torch.int32, device=dev)

    def enc_small():
        eng.pow_fixed_base_records(np.concatenate([fr_(prm["h"]), fr_(prm["pk"])]), np.concatenate([r_ex, r_ex]), hp2.data_ptr())
        eng.encrypt_records(pl1.data_ptr(), hp2.data_ptr(), fr_(prm["f"]), enc1.data_ptr(), E1, K)
    sec = timed(enc_small, reps=3)
    emit("encrypt_tensor, whole call incl. h^r and pk^r (fixed-base tables)", [E1], sec, E1, "ciphertexts/s")
del enc, pl

# ---- decryption / threshold decryption ---------------------------------------------------------
frec = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
sk = hx(prm["sk"])
dsk = dev_i32(exp_records([sk]))
ow = (K + 31) // 32 + 1
pt = torch.zeros(E * ow, dtype=torch.int32, device=dev)
sec = timed(lambda: eng.decrypt_records(cts.data_ptr(), dsk.data_ptr(), frec, pt.data_ptr(), E, K))
flags = pt.cpu().numpy().reshape(E, ow)[:, -1]
emit("decrypt_tensor", [E], sec, E, "ciphertexts/s", kernel="k_wnaf_digits + k_pow_shared + k_decrypt",
     error_flags=int(np.count_nonzero(flags)), first_flagged=[int(i) for i in np.nonzero(flags)[0][:8]], device_status=eng.device_status())
if np.count_nonzero(flags) and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
    bad = np.nonzero(flags)[0][:64]
    np.save(os.path.join(ROOT, "gpurun_out", "bad_cts.npy"), cts.cpu().numpy().view(np.uint32).reshape(E, 336)[bad])
    np.save(os.path.join(ROOT, "gpurun_out", "bad_idx.npy"), bad)
# 2-of-2 additive split of sk: s0 - s1 = sk
s1 = rng.bits(960)
s0 = sk + s1
parts = torch.zeros(2 * E * 168, dtype=torch.int32, device=dev)
d0, d1 = dev_i32(exp_records([s0])), dev_i32(exp_records([s1]))
sec = timed(lambda: eng.part_decrypt_records(cts.data_ptr(), d0.data_ptr(), parts.data_ptr(), E))
emit("part_decrypt_tensor", [E], sec, E, "ciphertexts/s", kernel="k_wnaf_digits + k_pow_shared")
eng.part_decrypt_records(cts.data_ptr(), d1.data_ptr(), parts.data_ptr() + E * 168 * 4, E)
pt2 = torch.zeros(E * ow, dtype=torch.int32, device=dev)
sec = timed(lambda: eng.combine_part_decryptions_records(cts.data_ptr(), parts.data_ptr(), [1, -1], frec, pt2.data_ptr(), E, K))
emit("combine_part_decryption_results_tensor (2 parts)", [E], sec, E, "ciphertexts/s", kernel="k_decrypt")
emit("threshold == plain decryption", [E], 1.0, 1, "check", equal=bool(torch.equal(pt, pt2)))
del parts, pt, pt2

# ---- accumulation of the ct x ct matrix product -------------------------------------------------
n, m, p = 16, 64, 16
x = fresh(n * m * p)
zero = fresh(1)
acc = torch.empty(n * p * 336, dtype=torch.int32, device=dev)
sec = timed(lambda: eng.accumulate_records(x.data_ptr(), zero.data_ptr(), acc.data_ptr(), n, m, p))
emit("accumulate (ct x ct matmul)", [n, m, p], sec, n * m * p, "ciphertext-ops/s", kernel="k_compose_pairs (tree; k_accumulate chains for large n*p)")
del x, acc

# ---- plaintext-matrix x ciphertext-matrix: C3 ---------------------------------------------------
# (8, 64, 64) is the reference's own default shape (benchmarks/local.cpp: scal_matmul 8 64 64)
shapes = [(8, 64, 64, "ramp"), (64, 64, 64, "ramp")] if QUICK else [(8, 64, 64, "ramp"), (64, 64, 64, "ramp"), (256, 256, 256, "ramp"), (32, 256, 256, "k-bit")]
for (n, m, p, kind) in shapes:
    cts = fresh(n * m)
    if kind == "ramp":
        evals = [j * p + k + 1 for j in range(m) for k in range(p)]       # benchmarks/local.cpp:171-174
    else:
        evals = [rng.bits(K) for _ in range(m * p)]
    ex = dev_i32(exp_records(evals))
    out = torch.empty(n * p * 336, dtype=torch.int32, device=dev)
    sec = timed(lambda: eng.scal_matmul_records(cts.data_ptr(), ex.data_ptr(), zero.data_ptr(), out.data_ptr(), n, m, p))
    emit("scal_ciphertext_tensors 2-D, %s exponents" % kind, [n, m, p], sec, n * p, "output-ciphertexts/s",
         macs_per_s=round(n * m * p / sec, 1), kernel="k_wnaf_digits + k_pow_table + k_scal_matmul_wnaf")
    del cts, out
