"""one-off robustness soak (through gpurun): random tensors at both parameter sets and a fundamental
discriminant, add / 1-D scal / matrix product / accumulate against the oracle"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
torch.cuda.init()
import pyref as P, oracle_lib as O
from cofhe_amd import Engine
from test_gpu_parity import _random_tensor, _pt_bytes, hx

def one(d, k, seed):
    E = Engine(d)
    rng = P.SplitMix64(seed)
    n = 2048
    x = P.serialize_ciphertext_tensor([n], _random_tensor(d, n, seed * 10 + 1, nbase=40))
    y = P.serialize_ciphertext_tensor([n], _random_tensor(d, n, seed * 10 + 2, nbase=40))
    assert E.add_ciphertext_tensors(x, y) == O.add(d, x, y)
    m = 256
    exps = [rng.bits(1 + rng.below(k)) * (-1 if rng.below(4) == 0 else 1) for _ in range(m)]
    s = _pt_bytes([m], exps)
    c = P.serialize_ciphertext_tensor([m], _random_tensor(d, m, seed * 10 + 3, nbase=20))
    assert E.scal_ciphertext_tensors(s, c) == O.scal_1d(d, s, c)
    nn, mm, pp = 3, 24, 5
    e2 = [rng.bits(1 + rng.below(40)) * (-1 if rng.below(5) == 0 else 1) if rng.below(6) else 0 for _ in range(mm * pp)]
    s2 = _pt_bytes([mm, pp], e2)
    c2 = P.serialize_ciphertext_tensor([nn, mm], _random_tensor(d, nn * mm, seed * 10 + 4, nbase=20))
    z = P.serialize_ciphertext_tensor([1], _random_tensor(d, 1, seed * 10 + 5, nbase=2))
    assert E.scal_ciphertext_tensors(s2, c2, z) == O.scal_2d(d, s2, c2, z)

t0 = time.time()
for name in ("s128_k128", "s128_k256"):
    prm = json.load(open(os.path.join(ROOT, "tests/golden/params_%s.json" % name)))
    for seed in (1, 2, 3):
        one(hx(prm["delta"]), prm["k"], seed)
        print(name, "seed", seed, "ok", round(time.time() - t0, 1), "s", flush=True)
q = P.random_prime(2300, P.SplitMix64(5), 7)
one(-q, 128, 9)
print("fundamental 2300-bit ok", round(time.time() - t0, 1), "s")
