"""diagnostic: the fixed-base product of k_encrypt step by step through cofhe_hip_compose_records against pyref"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import pyref as P
from bench import form_record, hx
from cofhe_amd import Engine
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
d, K = hx(prm["delta"]), prm["k"]
eng = Engine(d)
m = 0xe35425b964bdb6d05a03893b5c79a49c
r = 0xc82e101ee83d683efd4905a925cbbc24f11c50088370731d23689cedb7caca5532b1e56a5bd176f91893d737e90a739d12de7f4321468c73ea174c376cae7cb7d7ea367748b2e6efe01e16b79d801488717264fc5d68823f9416023e7bab39b017539f8bea12672556de3b214a20b71dfc4cbbd4e9d2d02e
F = lambda o: P.Form(hx(o["a"]), hx(o["b"]), hx(o["c"]))
f, pk = F(prm["f"]), F(prm["pk"])
acc = P.power(pk, r, d)
dev = lambda fm: torch.from_numpy(form_record(fm.a, fm.b, fm.c).view(np.int32)).cuda()


def rec_form(t):
    a = t.cpu().numpy().view(np.uint32)
    g = lambda lo, hi: int.from_bytes(a[lo:hi].tobytes(), "little")
    b = g(40, 80)
    return g(0, 40), (-b if a[160] else b), g(80, 160)


x3 = 3 * m
step = 0
for j in range(K):
    dg = ((x3 >> (j + 1)) & 1) - ((m >> (j + 1)) & 1)
    if dg == 0:
        continue
    rhs = P.power(f, 1 << j, d)
    if dg < 0:
        rhs = P.inverse(rhs)
    want = P.compose(acc, rhs)
    out = torch.zeros(168, dtype=torch.int32, device="cuda")
    eng.compose_records(dev(acc).data_ptr(), dev(rhs).data_ptr(), out.data_ptr(), 1)
    torch.cuda.synchronize()
    got = rec_form(out)
    step += 1
    if got != (want.a, want.b, want.c):
        print("MISMATCH at digit", j, dg, "step", step, "status", eng.device_status())
        json.dump({"x": [hex(acc.a), hex(acc.b), hex(acc.c)], "y": [hex(rhs.a), hex(rhs.b), hex(rhs.c)],
                   "got": [hex(v) for v in got], "want": [hex(want.a), hex(want.b), hex(want.c)]},
                  open(os.path.join(ROOT, "gpurun_out", "bad_pair.json"), "w"))
        break
    acc = want
print("steps", step)
