"""times k_scal_matmul / k_pow at a few shapes (tuning helper; run through gpurun)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cofhe_amd import Engine
from bench import hx, form_record, exp_records, SplitMix64
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_inputs import encrypt_tensor_gpu
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
eng = Engine(hx(prm["delta"]))
dev = torch.device("cuda", 0)
rng = SplitMix64(3)
for (n, m, p, ebits) in [(16, 16, 16, 8), (32, 64, 64, 12), (64, 64, 64, 12), (64, 256, 256, 16)]:
    cts = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(128) for _ in range(n * m)], rng.bits(900), dev)
    zero = encrypt_tensor_gpu(eng, torch, prm, [0], rng.bits(900), dev)
    ex = torch.from_numpy(exp_records([j * p + k + 1 for j in range(m) for k in range(p)]).view(np.int32)).to(dev)
    out = torch.empty(n * p * 336, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.scal_matmul_records(cts.data_ptr(), ex.data_ptr(), zero.data_ptr(), out.data_ptr(), n, m, p)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bits = (m * p).bit_length()
    comps = 2 * n * p * (bits + sum(bin(j * p + k + 1).count("1") for j in range(m) for k in range(p)) / p)
    print("scal_matmul n=%d m=%d p=%d: %.3f s, %.1f output-elements/s, ~%.2e compositions, %.2f Mcomp/s" % (n, m, p, dt, n * p / dt, comps, comps / dt / 1e6), flush=True)
