"""k_compose_wg launch time versus tensor size (fixed launch overhead / tail versus slope)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cofhe_amd import Engine
from bench import hx, SplitMix64
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_inputs import encrypt_tensor_gpu
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
eng = Engine(hx(prm["delta"]))
dev = torch.device("cuda", 0)
rng = SplitMix64(3)
E = 16384 * 8
a = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(128) for _ in range(E)], rng.bits(900), dev)
b = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(128) for _ in range(E)], rng.bits(900), dev)
out = torch.empty_like(a)
for frac in (1 / 32, 1 / 16, 1 / 8, 3 / 16, 1 / 4, 3 / 8, 1 / 2, 1.0):
    n = int(2 * E * frac)
    eng.time_compose(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, 3)
    ms = eng.time_compose(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, 20)
    print("records %7d  workgroups %5d  %.4f ms  %.2f ns/composition" % (n, n // 32, ms, ms * 1e6 / n), flush=True)
