"""What a matrix-product round would cost as ONE compose launch over all chains (131072 records at 256^3), back to back
on one stream, against the persistent k_scal_matmul_wnaf's time per lockstep round (through gpurun)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from bench import SplitMix64, hx
from gpu_inputs import encrypt_tensor_gpu
from cofhe_amd import Engine
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
eng = Engine(hx(prm["delta"])); dev = torch.device("cuda", 0); rng = SplitMix64(11)
for nct in (16384, 65536):
    a = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(128) for _ in range(nct)], rng.bits(900), dev)
    b = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(128) for _ in range(nct)], rng.bits(900), dev)
    bufs = [torch.empty_like(a), torch.empty_like(a)]
    cur = a
    for i in range(5):
        eng.compose_records(cur.data_ptr(), b.data_ptr(), bufs[i & 1].data_ptr(), 2 * nct); cur = bufs[i & 1]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 100
    for i in range(R):
        eng.compose_records(cur.data_ptr(), b.data_ptr(), bufs[i & 1].data_ptr(), 2 * nct); cur = bufs[i & 1]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / R
    print(json.dumps({"records": 2 * nct, "ms_per_launch": round(ms, 4), "ns_per_composition": round(ms * 1e6 / (2 * nct), 2)}))
