"""diagnostic (through gpurun): every pk^r o f^(+-2^j) and, step by step, the fixed-base chain pk^r o f^m on which
tools/bench_ops.py once caught a wrong result (its 8th step), through cofhe_hip_compose_records of a chosen build of the
library; the first wrong pair goes to gpurun_out/chain_bad_pair.json.  usage: compose_probe.py [lib.so]
(Keep operand tensors alive across the launch: `f(x).cuda().data_ptr()` of a temporary hands the kernel freed memory.)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
torch.cuda.init()
import pyref as P
import cofhe_amd
from bench import form_record, hx
if len(sys.argv) > 1:
    cofhe_amd.load_library(os.path.abspath(sys.argv[1]))
from cofhe_amd import Engine
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
d, k = hx(prm["delta"]), prm["k"]
eng = Engine(d)
F = lambda o: P.Form(hx(o["a"]), hx(o["b"]), hx(o["c"]))
f, pk = F(prm["f"]), F(prm["pk"])
r = int("c82e101ee83d683efd4905a925cbbc24f11c50088370731d23689cedb7caca5532b1e56a5bd176f91893d737e90a739d12de7f4321468c73ea174c3"
        "76cae7cb7d7ea367748b2e6efe01e16b79d801488717264fc5d68823f9416023e7bab39b017539f8bea12672556de3b214a20b71dfc4cbbd4e9d2d02e", 16)
pkr = P.power(pk, r, d)
rec = lambda forms: torch.from_numpy(np.concatenate([form_record(t.a, t.b, t.c) for t in forms]).view(np.int32)).cuda()
lhs, rhs = [], []
fj = f
for j in range(k):
    for y in (fj, P.inverse(fj)):
        lhs += [pkr, y]
        rhs += [y, pkr]
    fj = P.compose(fj, fj)
want = rec([P.compose(u, v) for u, v in zip(lhs, rhs)])
bad = []
for reps in range(3):
    # one pair per launch (as the failing call was), then all at once
    for i in range(len(lhs)):
        a, b = rec([lhs[i]]), rec([rhs[i]])
        o = torch.zeros(168, dtype=torch.int32, device="cuda")
        eng.compose_records(a.data_ptr(), b.data_ptr(), o.data_ptr(), 1)
        torch.cuda.synchronize()
        if not torch.equal(o, want[i * 168:(i + 1) * 168]):
            bad.append(i)
a, b = rec(lhs), rec(rhs)
o = torch.zeros_like(a)
eng.compose_records(a.data_ptr(), b.data_ptr(), o.data_ptr(), len(lhs))
torch.cuda.synchronize()
allbad = [i for i in range(len(lhs)) if not torch.equal(o[i * 168:(i + 1) * 168], want[i * 168:(i + 1) * 168])]
# the fixed-base product of the old k_encrypt for the plaintext that failed, step by step (operands kept alive)
m = 0xe35425b964bdb6d05a03893b5c79a49c
acc = pkr
chain_bad = None
x3 = 3 * m
step = 0
for j in range(k):
    dg = ((x3 >> (j + 1)) & 1) - ((m >> (j + 1)) & 1)
    if dg == 0:
        continue
    y = P.power(f, 1 << j, d)
    if dg < 0:
        y = P.inverse(y)
    w = P.compose(acc, y)
    ta, tb = rec([acc]), rec([y])
    o = torch.zeros(168, dtype=torch.int32, device="cuda")
    eng.compose_records(ta.data_ptr(), tb.data_ptr(), o.data_ptr(), 1)
    torch.cuda.synchronize()
    step += 1
    if not torch.equal(o, rec([w])):
        chain_bad = {"step": step, "digit": j, "sign": dg, "x": [hex(acc.a), hex(acc.b), hex(acc.c)], "y": [hex(y.a), hex(y.b), hex(y.c)]}
        break
    acc = w
if chain_bad:
    json.dump(chain_bad, open(os.path.join(ROOT, "gpurun_out", "chain_bad_pair.json"), "w"))
print(json.dumps({"chain_first_wrong_step": (chain_bad or {}).get("step"), "chain_digit": (chain_bad or {}).get("digit"), "lib": sys.argv[1] if len(sys.argv) > 1 else "in-tree", "pairs": len(lhs), "wrong_single_launch": sorted(set(bad)), "wrong_batched": allbad,
                  "status": eng.device_status()}))
