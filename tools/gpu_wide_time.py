"""latency of ONE composition in the wavefront-wide layout (k_compose_wide, reps inside the kernel) against the throughput
kernel alone on a CU, and the decryption ladder in its three forms; run through gpurun"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cofhe_amd import Engine
from bench import hx, exp_records, form_record, SplitMix64
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_inputs import encrypt_tensor_gpu
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
eng = Engine(hx(prm["delta"]))
dev = torch.device("cuda", 0)
rng = SplitMix64(11)
K = prm["k"]
hrec = form_record(hx(prm["h"]["a"]), hx(prm["h"]["b"]), hx(prm["h"]["c"]))
n = 1024
base = torch.from_numpy(np.tile(hrec, n).view(np.int32)).to(dev)
def fam(seed):
    r = SplitMix64(seed)
    ex = torch.from_numpy(exp_records([r.bits(192) | 1 for _ in range(n)]).view(np.int32)).to(dev)
    o = torch.empty(n * 168, dtype=torch.int32, device=dev)
    eng.pow_form_records(base.data_ptr(), ex.data_ptr(), o.data_ptr(), n)
    torch.cuda.synchronize()
    return o
a, b = fam(1), fam(2)
out = torch.zeros_like(a)
for cnt in (1, 8, 256, 1024):
    for reps in (1, 101):
        eng.compose_wide_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), cnt, reps)
        eng.stream_sync(0)
        ms = eng.time_stream(lambda: eng.compose_wide_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), cnt, reps), 3)
        if reps > 1:
            print("k_compose_wide  %4d wavefronts: %.1f us per composition (%d in a row)" % (cnt, (ms - ms1) * 1e3 / (reps - 1), reps), flush=True)
        else:
            ms1 = ms
fb = eng.compose_wide_records(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, 1, count_fallbacks=True)
print("fallbacks among %d random pairs: %d" % (n, fb))
for cnt in (1, 32):
    ms = eng.time_compose(a.data_ptr(), b.data_ptr(), out.data_ptr(), cnt, 5)
    print("k_compose_wg    %4d compositions (one workgroup): %.1f us per launch" % (cnt, ms * 1e3), flush=True)
# the decryption ladder
sk = rng.bits(960)
dsk = torch.from_numpy(exp_records([sk]).view(np.int32)).to(dev)
frec = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
for E_ in (1, 16384):
    cts = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(K) for _ in range(E_)], rng.bits(900), dev)
    ow = (K + 31) // 32 + 1
    res = {}
    for form, name in ((1, "wide"), (4, "wide-1"), (2, "solo"), (3, "throughput")):
        eng.set_option("ladder_form", form)
        o = torch.zeros(E_ * ow, dtype=torch.int32, device=dev)
        eng.decrypt_records(cts.data_ptr(), dsk.data_ptr(), frec, o.data_ptr(), E_, K)
        eng.stream_sync(0)
        t0 = time.perf_counter()
        eng.decrypt_records(cts.data_ptr(), dsk.data_ptr(), frec, o.data_ptr(), E_, K)
        eng.stream_sync(0)
        dt = time.perf_counter() - t0
        p = torch.zeros(E_ * 168, dtype=torch.int32, device=dev)
        t0 = time.perf_counter()
        eng.part_decrypt_records(cts.data_ptr(), dsk.data_ptr(), p.data_ptr(), E_)
        eng.stream_sync(0)
        dp = time.perf_counter() - t0
        res[name] = o.cpu()
        print("decrypt_tensor %6d ciphertexts, ladder %-10s: %.1f ms   (part_decrypt_tensor %.1f ms)  status %d" % (E_, name, dt * 1e3, dp * 1e3, eng.device_status(clear=True)), flush=True)
    eng.set_option("ladder_form", 0)
    print("   same plaintexts: %s" % (bool(torch.equal(res["wide"], res["solo"])) and bool(torch.equal(res["wide"], res["throughput"]))))
