"""matrix product, product tree against lockstep chains (option "matmul_tree"): same records out, time of each; run through gpurun"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cofhe_amd import Engine
from bench import hx, exp_records, SplitMix64
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_inputs import encrypt_tensor_gpu
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
eng = Engine(hx(prm["delta"]))
dev = torch.device("cuda", 0)
rng = SplitMix64(3)
shapes = [(8, 64, 64, 0), (64, 64, 64, 0), (32, 256, 256, 128), (256, 256, 256, 0)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (n, m, p, kbits) in shapes:
    cts = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(128) for _ in range(n * m)], rng.bits(900), dev)
    zero = encrypt_tensor_gpu(eng, torch, prm, [0], rng.bits(900), dev)
    exps = [rng.bits(kbits) for _ in range(m * p)] if kbits else [j * p + k + 1 for j in range(m) for k in range(p)]
    ex = torch.from_numpy(exp_records(exps).view(np.int32)).to(dev)
    outs = {}
    for tree in (1, 0, 1, 0):
        eng.set_option("matmul_tree", tree)
        out = torch.zeros(n * p * 336, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.scal_matmul_records(cts.data_ptr(), ex.data_ptr(), zero.data_ptr(), out.data_ptr(), n, m, p)
        eng.stream_sync(0)
        dt = time.perf_counter() - t0
        outs.setdefault(tree, out.cpu())
        print("scal_matmul %dx%d . %dx%d %s exponents, %s: %.4f s  status %d" % (n, m, m, p, ("%d-bit" % kbits) if kbits else "harness",
              "tree  " if tree else "chains", dt, eng.device_status(clear=True)), flush=True)
    print("   same records: %s" % bool(torch.equal(outs[0], outs[1])), flush=True)
    eng.set_option("matmul_tree", -1)
    del cts, out, outs
    torch.cuda.empty_cache()
