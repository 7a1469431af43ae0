"""soak (through gpurun): encrypt N random plaintexts, decrypt them, compare -- every composition of ~N * 1200
(fixed-base products, the shared-exponent ladder, the discrete-log peeling) is checked by the value that comes back;
also add / negate identities on the same tensors.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import SplitMix64, exp_records, form_record, hx
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_inputs import encrypt_tensor_gpu
from cofhe_amd import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
name = sys.argv[3] if len(sys.argv) > 3 else "s128_k128"
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_%s.json" % name)))
K = prm["k"]
eng = Engine(hx(prm["delta"]))
dev = torch.device("cuda", 0)
rng = SplitMix64(seed)
M = 1 << K
special = [0, 1, 2, 3, 4, M - 1, M - 2, M - 4, M >> 1, (M >> 1) - 1, (M >> 1) + 1] + [1 << j for j in range(K)] + [M - (1 << j) for j in range(K)]
ms = (special + [rng.bits(K) for _ in range(N)])[:N]
t0 = time.time()
cts = encrypt_tensor_gpu(eng, torch, prm, ms, rng.bits(960), dev)
cts2 = encrypt_tensor_gpu(eng, torch, prm, ms[::-1], rng.bits(960), dev)
valid = eng.validate_records(cts.data_ptr(), 2 * N) and eng.validate_records(cts2.data_ptr(), 2 * N)
frec = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
dsk = torch.from_numpy(exp_records([hx(prm["sk"])]).view(np.int32)).to(dev)
ow = (K + 31) // 32 + 1


def decrypt(t):
    pt = torch.zeros(N * ow, dtype=torch.int32, device=dev)
    eng.decrypt_records(t.data_ptr(), dsk.data_ptr(), frec, pt.data_ptr(), N, K)
    torch.cuda.synchronize()
    a = pt.cpu().numpy().view(np.uint32).reshape(N, ow)
    vals = [int.from_bytes(a[i, :-1].tobytes(), "little") for i in range(N)]
    return vals, int(np.count_nonzero(a[:, -1]))


got, flags = decrypt(cts)
wrong = sum(1 for i in range(N) if got[i] != ms[i])
s = torch.empty_like(cts)
eng.add_ciphertext_records(cts.data_ptr(), cts2.data_ptr(), s.data_ptr(), N)
got2, flags2 = decrypt(s)
wrong2 = sum(1 for i in range(N) if got2[i] != (ms[i] + ms[N - 1 - i]) % M)
# the plain paths as well: a tensor mixed from the two encryptions has differing c1 (one ladder per ciphertext in
# decrypt, no folding in the add), and the form-level composition over all 2 N records must equal the folded sum
NM = min(N, 131072)
mix = cts.view(N, 336)[:NM].clone()
mix[1::2] = cts2.view(N, 336)[:NM][1::2]
mix = mix.reshape(-1).contiguous()
want_mix = [ms[i] if i % 2 == 0 else ms[N - 1 - i] for i in range(NM)]
N_keep, N = N, NM
got3, flags3 = decrypt(mix)
N = N_keep
wrong3 = sum(1 for i in range(NM) if got3[i] != want_mix[i])
plain = torch.empty_like(cts)
eng.compose_records(cts.data_ptr(), cts2.data_ptr(), plain.data_ptr(), 2 * N)
torch.cuda.synchronize()
plain_equals_folded = bool(torch.equal(plain, s))
print(json.dumps({"mixed_c1_n": NM, "mixed_decrypt_flags": flags3, "mixed_decrypt_wrong": wrong3, "plain_add_equals_folded": plain_equals_folded,
                  "params": name, "n": N, "seed": seed, "encrypted_forms_valid": bool(valid), "decrypt_flags": flags, "decrypt_wrong": wrong,
                  "sum_flags": flags2, "sum_wrong": wrong2, "device_status": eng.device_status(), "seconds": round(time.time() - t0, 1)}))
