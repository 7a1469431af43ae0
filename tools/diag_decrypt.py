"""diagnostic: decrypt E fresh ciphertexts, count error flags, print the device status word"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import SplitMix64, encrypt_tensor_gpu, exp_records, form_record, hx
from cofhe_amd import Engine
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
K = prm["k"]
eng = Engine(hx(prm["delta"]))
dev = torch.device("cuda", 0)
rng = SplitMix64(7)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ms = [rng.bits(K) for _ in range(E)]
cts = encrypt_tensor_gpu(eng, torch, prm, ms, rng.bits(960), dev)
print("status after encrypt", eng.device_status())
steps = sys.argv[2].split(",") if len(sys.argv) > 2 else []
fr_ = lambda o: form_record(hx(o["a"]), hx(o["b"]), hx(o["c"]))
dev_i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
if "k256" in steps:
    prm2 = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k256.json")))
    eng2 = Engine(hx(prm2["delta"]))
    a = encrypt_tensor_gpu(eng2, torch, prm2, [rng.bits(256) for _ in range(1024)], rng.bits(960), dev)
    o = torch.empty_like(a)
    eng2.compose_records(a.data_ptr(), a.data_ptr(), o.data_ptr(), 2048)
    torch.cuda.synchronize()
    del a, o, eng2
if "big" in steps:
    a = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(K) for _ in range(1 << 20)], rng.bits(960), dev)
    o = torch.empty_like(a)
    eng.compose_records(a.data_ptr(), a.data_ptr(), o.data_ptr(), 2 << 20)
    torch.cuda.synchronize()
    del a, o
if "bytes" in steps:
    a, b = cts, cts
    ha = eng.records_to_bytes(a.cpu().numpy().view(np.uint32), [128, 128])
    hc = eng.add_ciphertext_tensors(ha, ha)
    print("bytes ok", len(hc))
if "pow" in steps:
    out = torch.empty_like(cts)
    ex = dev_i32(exp_records([rng.bits(K) for _ in range(E)]))
    eng.pow_records(cts.data_ptr(), ex.data_ptr(), out.data_ptr(), E)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    eng.pow_records(cts.data_ptr(), ex.data_ptr(), out.data_ptr(), E)
    torch.cuda.synchronize()
    print("pow_records 16384 x 128-bit: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
if "enc" in steps:
    pl = dev_i32(exp_records([rng.bits(K) for _ in range(E)]))
    hp = dev_i32(np.concatenate([fr_(prm["h"]), fr_(prm["pk"])]))
    enc = torch.empty(E * 336, dtype=torch.int32, device=dev)
    eng.encrypt_records(pl.data_ptr(), hp.data_ptr(), fr_(prm["f"]), enc.data_ptr(), E, K)
    torch.cuda.synchronize()
if "fixed" in steps:
    r_ex = exp_records([rng.bits(960)])
    hp2 = torch.empty(2 * 168, dtype=torch.int32, device=dev)
    eng.pow_fixed_base_record(fr_(prm["h"]), r_ex, hp2.data_ptr())
    eng.pow_fixed_base_record(fr_(prm["pk"]), r_ex, hp2.data_ptr() + 168 * 4)
    torch.cuda.synchronize()
if "encsmall" in steps:
    r_ex = exp_records([rng.bits(960)])
    hp2 = torch.empty(2 * 168, dtype=torch.int32, device=dev)
    for E1 in (1, 64):
        pl1 = dev_i32(exp_records([rng.bits(K) for _ in range(E1)]))
        enc1 = torch.empty(E1 * 336, dtype=torch.int32, device=dev)
        eng.pow_fixed_base_record(fr_(prm["h"]), r_ex, hp2.data_ptr())
        eng.pow_fixed_base_record(fr_(prm["pk"]), r_ex, hp2.data_ptr() + 168 * 4)
        eng.encrypt_records(pl1.data_ptr(), hp2.data_ptr(), fr_(prm["f"]), enc1.data_ptr(), E1, K)
        torch.cuda.synchronize()
if "foldedbig" in steps:
    a = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(K) for _ in range(1 << 20)], rng.bits(960), dev)
    o = torch.empty_like(a)
    eng.add_ciphertext_records(a.data_ptr(), a.data_ptr(), o.data_ptr(), 1 << 20)
    torch.cuda.synchronize()
    del a, o
if "folded" in steps:
    out = torch.empty_like(cts)
    eng.add_ciphertext_records(cts.data_ptr(), cts.data_ptr(), out.data_ptr(), E)
    torch.cuda.synchronize()
print("status after steps", steps, eng.device_status())
frec = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
dsk = torch.from_numpy(exp_records([hx(prm["sk"])]).view(np.int32)).to(dev)
ow = (K + 31) // 32 + 1
for rep in range(3):
    pt = torch.zeros(E * ow, dtype=torch.int32, device=dev)
    eng.decrypt_records(cts.data_ptr(), dsk.data_ptr(), frec, pt.data_ptr(), E, K)
    torch.cuda.synchronize()
    a = pt.cpu().numpy().view(np.uint32).reshape(E, ow)
    bad = np.nonzero(a[:, -1])[0]
    got = [int.from_bytes(a[i, :-1].tobytes(), "little") for i in range(E)]
    wrong = [i for i in range(E) if got[i] != ms[i]]
    print("rep", rep, "flags", len(bad), list(bad[:10]), "wrong", len(wrong), wrong[:10], "status", eng.device_status())
