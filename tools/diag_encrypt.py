"""diagnostic: encrypt one known-bad plaintext (found by tools/bench_ops.py) and validate the result"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import SplitMix64, encrypt_tensor_gpu, exp_records, form_record, hx
from cofhe_amd import Engine
prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
eng = Engine(hx(prm["delta"]))
dev = torch.device("cuda", 0)
m = 0xe35425b964bdb6d05a03893b5c79a49c
r = 0xc82e101ee83d683efd4905a925cbbc24f11c50088370731d23689cedb7caca5532b1e56a5bd176f91893d737e90a739d12de7f4321468c73ea174c376cae7cb7d7ea367748b2e6efe01e16b79d801488717264fc5d68823f9416023e7bab39b017539f8bea12672556de3b214a20b71dfc4cbbd4e9d2d02e
rng = SplitMix64(99)
for E, pos in ((1, 0), (32, 25), (64, 57), (16384, 16057)):
    ms = [rng.bits(128) for _ in range(E)]
    ms[pos] = m
    cts = encrypt_tensor_gpu(eng, torch, prm, ms, r, dev)
    ok_all = eng.validate_records(cts.data_ptr(), 2 * E)
    ok_one = eng.validate_records(cts.data_ptr() + pos * 336 * 4, 2)
    print("E", E, "pos", pos, "all valid", ok_all, "that one valid", ok_one, "status", eng.device_status())
