"""one-off soak (through gpurun): larger plaintext-matrix x ciphertext-matrix products with mixed exponent widths
(0, small, negative, k-bit, wider than k) against the oracle, every output compared; both parameter sets"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
torch.cuda.init()
import pyref as P, oracle_lib as O
from cofhe_amd import Engine
from test_gpu_parity import _random_tensor, _pt_bytes, hx

t0 = time.time()
for name, shapes in (("s128_k128", [(16, 96, 24), (5, 200, 7), (40, 33, 40)]), ("s128_k256", [(8, 64, 16)])):
    prm = json.load(open(os.path.join(ROOT, "tests/golden/params_%s.json" % name)))
    d, k = hx(prm["delta"]), prm["k"]
    E = Engine(d)
    for si, (n, m, p) in enumerate(shapes):
        rng = P.SplitMix64(100 + si)
        exps = []
        for _ in range(m * p):
            r = rng.below(10)
            e = 0 if r == 0 else rng.bits(1 + rng.below(16)) if r < 5 else rng.bits(k) if r < 8 else rng.bits(k + 40)
            exps.append(-e if rng.below(4) == 0 else e)
        s = _pt_bytes([m, p], exps)
        c = P.serialize_ciphertext_tensor([n, m], _random_tensor(d, n * m, 7000 + si, nbase=32))
        z = P.serialize_ciphertext_tensor([1], _random_tensor(d, 1, 7100 + si, nbase=2))
        got = E.scal_ciphertext_tensors(s, c, z)
        want = O.scal_2d(d, s, c, z)
        print(json.dumps({"params": name, "shape": [n, m, p], "equal": got == want, "status": E.device_status(), "seconds": round(time.time() - t0, 1)}), flush=True)
        assert got == want
