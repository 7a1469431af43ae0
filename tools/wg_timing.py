"""writes the inputs of tools/wg_timing.hip (|Delta| bytes, two 128x128 record tensors) and, given its
CSV, prints the per-XCC / per-CU summary"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "gen":
    import numpy as np, torch
    from cofhe_amd import Engine
    from bench import hx, SplitMix64
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_inputs import encrypt_tensor_gpu
    prm = json.load(open(os.path.join(ROOT, "tests/golden/params_s128_k128.json")))
    d = hx(prm["delta"]); eng = Engine(d); dev = torch.device("cuda", 0); rng = SplitMix64(5)
    out = sys.argv[2]
    m = -d
    open(os.path.join(out, "delta.bin"), "wb").write(m.to_bytes((m.bit_length() + 7) // 8, "little"))
    for name in ("a", "b"):
        t = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(128) for _ in range(16384)], rng.bits(900), dev)
        t.cpu().numpy().tofile(os.path.join(out, name + ".bin"))
else:
    import csv, statistics as st
    rows = list(csv.DictReader(open(sys.argv[2])))
    dur = [float(r["end_us"]) - float(r["start_us"]) for r in rows]
    end = [float(r["end_us"]) for r in rows]
    start = [float(r["start_us"]) for r in rows]
    print("workgroups %d  start: max %.1f us  duration: mean %.1f  min %.1f  max %.1f  stdev %.1f us  last end %.1f us" %
          (len(rows), max(start), st.mean(dur), min(dur), max(dur), st.pstdev(dur), max(end)))
    by = {}
    for r, du, e in zip(rows, dur, end):
        by.setdefault(int(r["xcc_id"]), []).append((du, e))
    for x in sorted(by):
        v = by[x]
        print("  XCC %d: %4d workgroups  mean duration %.1f us  last end %.1f us" % (x, len(v), st.mean(a for a, _ in v), max(b for _, b in v)))
    cu = {}
    for r, du in zip(rows, dur):
        h = int(r["hw_id"])
        key = (int(r["xcc_id"]), (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15)      # xcc, SE, SH, CU
        cu.setdefault(key, []).append(du)
    sizes = sorted(len(v) for v in cu.values())
    print("  distinct (xcc, se, sh, cu): %d; workgroups per CU: min %d max %d" % (len(cu), sizes[0], sizes[-1]))
    if "server_simd" in rows[0]:
        # which SIMD hosts the serving wavefront of each workgroup, per CU: co-resident servers on one SIMD share its issue slots
        srv = {}
        spread = {}
        for r in rows:
            h = int(r["hw_id"])
            key = (int(r["xcc_id"]), (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15)
            srv.setdefault(key, []).append(int(r["server_simd"]))
            spread.setdefault(len({int(r["simd%d" % w]) for w in range(4)}), 0)
            spread[len({int(r["simd%d" % w]) for w in range(4)})] += 1
        hist = {}
        for v in srv.values():
            hist[len(set(v))] = hist.get(len(set(v)), 0) + 1
        print("  distinct SIMDs among the 4 wavefronts of a workgroup -> workgroups:", dict(sorted(spread.items())))
        print("  distinct SIMDs among the serving wavefronts of a CU's workgroups -> CUs:", dict(sorted(hist.items())))
    means = sorted(st.mean(v) for v in cu.values())
    print("  per-CU mean duration: min %.1f  median %.1f  max %.1f us" % (means[0], means[len(means) // 2], means[-1]))
