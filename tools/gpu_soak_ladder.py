"""soak (through gpurun) of the two-wavefront decryption ladder (k_pow_shared_pair: two workgroups that meet through a ring
and two counts in the workspace): R rounds of random exponents of random lengths (0 .. 990 bits, either sign) over 1, 3,
64 and 256 ladders, every result compared with the throughput kernel's (ladder_form 3).  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import SplitMix64, exp_records, form_record, hx
from cofhe_amd import Engine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 60
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
eng = Engine(hx(prm["delta"]))
dev = torch.device("cuda", 0)
rng = SplitMix64(2026)
hrec = form_record(hx(prm["h"]["a"]), hx(prm["h"]["b"]), hx(prm["h"]["c"]))
nmax = 256
base = torch.from_numpy(np.tile(hrec, 2 * nmax).view(np.int32)).to(dev)
ex = torch.from_numpy(exp_records([rng.bits(192) | 1 for _ in range(2 * nmax)]).view(np.int32)).to(dev)
cts = torch.empty(2 * nmax * 168, dtype=torch.int32, device=dev)
eng.pow_form_records(base.data_ptr(), ex.data_ptr(), cts.data_ptr(), 2 * nmax)        # 256 "ciphertexts" of distinct random forms
torch.cuda.synchronize()
t0 = time.time()
bad, runs, bits_total = 0, 0, 0
for r in range(R):
    nb = [0, 1, 2, 3, 17, 64, 65, 127, 500, 959, 960, 990][r % 12] if r < 24 else rng.bits(10) % 991
    e = rng.bits(nb) if nb else 0
    if nb:
        e |= 1 << (nb - 1)
    if r % 3 == 2:
        e = -e
    de = torch.from_numpy(exp_records([e]).view(np.int32)).to(dev)
    for n in (1, 3, 64, 256):
        outs = []
        for form in (1, 3):
            eng.set_option("ladder_form", form)
            o = torch.zeros(n * 168, dtype=torch.int32, device=dev)
            eng.part_decrypt_records(cts.data_ptr(), de.data_ptr(), o.data_ptr(), n)
            eng.stream_sync(0)
            outs.append(o)
        runs += 1
        bits_total += nb
        if not torch.equal(outs[0], outs[1]):
            bad += 1
    if r % 10 == 9:
        print("round %d of %d, %d mismatches, %.0f s" % (r + 1, R, bad, time.time() - t0), file=sys.stderr, flush=True)
eng.set_option("ladder_form", 0)
print(json.dumps({"soak": "k_pow_shared_pair against k_pow_shared", "rounds": R, "comparisons": runs, "mismatches": bad,
                  "mean_exponent_bits": round(bits_total / max(1, runs), 1), "device_status": eng.device_status(clear=False),
                  "seconds": round(time.time() - t0, 1)}))
