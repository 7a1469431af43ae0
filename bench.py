#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X CoFHE engine.

Metric (BASELINE.json): ciphertext-ops/s and HBM GB/s (% of the 8 TB/s roofline) for the
128x128 `ciphertext_matadd` (config C2: E = 16 384 ciphertexts, security 128, k = 128).
A "step" is one add_ciphertext_tensors over the whole resident tensor, chained like the
reference harness (res = add(res, ct2), /root/reference/benchmarks/local.cpp:99-117).

Inputs are VALID ciphertexts of random plaintexts produced by the product path itself (GPU
powering + composition kernels: c1 = h^r, c2 = f^m o pk^r, one r per tensor as in
cpu_cryptosystem_tensor_ops.inl:7-15) from the committed public parameters; they are resident
in HBM before the timed region starts.  N > 1: one process per GPU, the (128 N) x 128 tensor is
row-sharded, no exchange between chained adds, one RCCL all-gather of the final result
(inside the timed region).

Extra objects on the JSON line: "roofline" (algorithmic bytes of one compose launch / its
HIP-event time vs 8 TB/s) and "cpu_baseline" (the C++/GMP oracle restating the reference loop,
timed on this box's host cores on a bounded sample, rank 0, N = 1 only).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def hx(s):
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


def form_record(a, b, c):
    import numpy as np
    r = np.zeros(168, dtype=np.uint32)
    r[0:40] = np.frombuffer(a.to_bytes(160, "little"), dtype="<u4")
    r[40:80] = np.frombuffer(abs(b).to_bytes(160, "little"), dtype="<u4")
    r[80:160] = np.frombuffer(c.to_bytes(320, "little"), dtype="<u4")
    r[160] = 1 if b < 0 else 0
    return r


def exp_records(vals):
    import numpy as np
    out = np.zeros((len(vals), 32), dtype=np.uint32)
    for i, v in enumerate(vals):
        out[i, :31] = np.frombuffer(abs(v).to_bytes(124, "little"), dtype="<u4")
        out[i, 31] = 1 if v < 0 else 0
    return out.reshape(-1)


class SplitMix64:
    M = (1 << 64) - 1

    def __init__(self, seed):
        self.s = seed & self.M

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
        return z ^ (z >> 31)

    def bits(self, n):
        v, sh = 0, 0
        while sh < n:
            v |= self.next() << sh
            sh += 64
        return v & ((1 << n) - 1)


def encrypt_tensor_gpu(eng, torch, prm, plaintexts, r, dev):
    """Product-path encryption: c1 = h^r (shared), c2_i = f^{m_i} o pk^r.  Returns a device
    int32 tensor of 2E records."""
    import numpy as np
    E = len(plaintexts)
    f = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    h = form_record(hx(prm["h"]["a"]), hx(prm["h"]["b"]), hx(prm["h"]["c"]))
    pk = form_record(hx(prm["pk"]["a"]), hx(prm["pk"]["b"]), hx(prm["pk"]["c"]))
    # (h^r, pk^r): one "ciphertext" whose two records are h and pk, one exponent
    base = torch.from_numpy(np.concatenate([h, pk]).view(np.int32)).to(dev)
    ex = torch.from_numpy(exp_records([r]).view(np.int32)).to(dev)
    hp = torch.empty_like(base)
    eng.pow_records(base.data_ptr(), ex.data_ptr(), hp.data_ptr(), 1)
    torch.cuda.synchronize()
    em = torch.from_numpy(exp_records(plaintexts).view(np.int32)).to(dev)
    out = torch.empty(E * 2 * 168, dtype=torch.int32, device=dev)
    eng.encrypt_records(em.data_ptr(), hp.data_ptr(), f, out.data_ptr(), E, prm["k"])
    torch.cuda.synchronize()
    return out


def host_cpu_share(limit):
    """threads worth giving the CPU baseline: the scheduler affinity and the cgroup CPU quota of this
    process (a GPU box hands each job a share of its host cores), capped by `limit`"""
    n = limit
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        n = min(n, max(1, int(round(q / int(fh2.read().split()[0])))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


class _StdoutToStderr:
    """RCCL prints a version banner on first use; keep stdout for the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--cols", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["matadd", "scal_matmul"], default="matadd",
                    help="matadd: the BASELINE.json metric (default).  scal_matmul: configs C3/C4, a rows x cols "
                         "ciphertext block per GPU times a cols x cols plaintext matrix, result rows all-gathered")
    args = ap.parse_args()
    if args.workload == "scal_matmul":
        return main_scal_matmul(args)

    import numpy as np
    import torch
    from cofhe_amd import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("COFHE_BENCH_FORCE_DIST") == "1":    # the flag rehearses the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        with _StdoutToStderr():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    with open(os.path.join(ROOT, "tests", "golden", "params_s128_k128.json")) as fh:
        prm = json.load(fh)
    delta = hx(prm["delta"])
    k = prm["k"]
    eng = Engine(delta, device=local_rank)

    E = args.rows * args.cols                     # ciphertexts per GPU (weak scaling)
    rng = SplitMix64(1000 + rank)
    bound_bits = hx(prm["exponent_bound"]).bit_length() - 1
    pts1 = [rng.bits(k) for _ in range(E)]
    pts2 = [rng.bits(k) for _ in range(E)]
    ct1 = encrypt_tensor_gpu(eng, torch, prm, pts1, rng.bits(bound_bits), dev)
    ct2 = encrypt_tensor_gpu(eng, torch, prm, pts2, rng.bits(bound_bits), dev)
    nrec = 2 * E
    stream = torch.cuda.current_stream().cuda_stream
    bufs = [torch.empty_like(ct1), torch.empty_like(ct1)]

    def step(src, dst):
        eng.compose_records(src.data_ptr(), ct2.data_ptr(), dst.data_ptr(), nrec, stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    cur = ct1
    for i in range(args.warmup):
        step(cur, bufs[i & 1])
        cur = bufs[i & 1]
    gathered = None
    from cofhe_amd import shard
    if dist is not None:
        gathered = shard.all_gather_rows(cur, args.rows * world, args.cols, dist, world, rank)   # warm the communicator
    barrier()
    t0 = time.perf_counter()
    cur = ct1
    for i in range(args.steps):
        step(cur, bufs[i & 1])
        cur = bufs[i & 1]
    if dist is not None:
        # reassemble the (128 N) x 128 result: rank r owns rows [128 r, 128 r + 128)
        gathered = shard.all_gather_rows(cur, args.rows * world, args.cols, dist, world, rank)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline: HIP events around back-to-back launches of the dominant kernel ----------
    iters = max(5, min(args.steps, 20))
    ms_launch = eng.time_compose(ct1.data_ptr(), ct2.data_ptr(), bufs[0].data_ptr(), nrec, iters, stream)
    # algorithmic bytes: S = payload bytes of one serialised ciphertext (no offset table),
    # measured on the inputs and the output actually used; matadd moves 3 S per ciphertext-op
    samp = min(E, 1024)

    def payload_per_ct(t):
        recs = t[: samp * 2 * 168].cpu().numpy().view(np.uint32)
        b = eng.records_to_bytes(recs, [samp])
        return (len(b) - 4 - 4 - 48 * samp) / samp

    S_in1, S_in2, S_out = payload_per_ct(ct1), payload_per_ct(ct2), payload_per_ct(bufs[0])
    alg_bytes = (S_in1 + S_in2 + S_out) * E
    achieved = alg_bytes / (ms_launch * 1e-3) / 1e9
    # HBM-side bytes per launch: PMC counters (FETCH_SIZE, WRITE_SIZE, separate rocprofv3 passes) corrected
    # with factors calibrated on a record-copy kernel of the same access pattern -- measured by
    # tools/gpu_traffic.sh and committed under profiles/ (a profiler cannot run inside this process)
    traffic, traffic_src = None, None
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")),
                       key=lambda q: (len(os.path.dirname(q)), q), reverse=True):      # newest: r01_v10 after r01_v9
        try:
            with open(cand) as fh:
                tj = json.load(fh)
            if tj.get("records_per_launch") == nrec and tj.get("traffic_bytes_per_launch"):
                traffic, traffic_src = int(tj["traffic_bytes_per_launch"]), os.path.relpath(cand, ROOT)
                break
        except (OSError, ValueError):
            pass
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "k_compose_wg", "launch_ms": round(ms_launch, 4),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "note": "class-group composition is integer-VALU bound (see DESIGN.md); HBM fraction is reported as the contract asks"}

    # ---- CPU baseline (oracle = checker, timed on a bounded sample) -------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        ns = min(E, 16384)
        a = eng.records_to_bytes(ct1[: ns * 336].cpu().numpy().view(np.uint32), [ns])
        b = eng.records_to_bytes(ct2[: ns * 336].cpu().numpy().view(np.uint32), [ns])
        cores = host_cpu_share(O.max_threads())
        chain = 5
        sec, want = O.time_matadd_chain(delta, a, b, chain, threads=cores, want_out=True)
        if sec < 8.0:       # size the sample to ~10-30 s of CPU work
            chain = int(min(400, max(chain, chain * 12.0 / max(sec, 1e-3))))
            sec, want = O.time_matadd_chain(delta, a, b, chain, threads=cores, want_out=True)
        # the same chain on the GPU must give the same bytes (the oracle is only the checker)
        x = ct1[: ns * 336].clone()
        y = ct2[: ns * 336].clone()
        o = torch.empty_like(x)
        for _ in range(chain):
            eng.compose_records(x.data_ptr(), y.data_ptr(), o.data_ptr(), 2 * ns, stream)
            torch.cuda.synchronize()
            x, o = o, x
        got = eng.records_to_bytes(x.cpu().numpy().view(np.uint32), [ns])
        cpu = {"value": round(ns * chain / sec, 2), "unit": "ciphertext-ops/s", "cores": cores, "kind": "port",
               "sample": "%d ciphertexts x %d chained adds of the same workload (%.1f s)" % (ns, chain, sec),
               "parity_with_gpu": bool(got == want)}

    if rank == 0:
        ops = E * world * args.steps
        line = {
            "metric": "ciphertext-ops/sec + HBM GB/s (% roofline), 128x128 matadd",
            "value": round(ops / elapsed, 2),
            "unit": "ciphertext-ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "ciphertext_matadd %dx%d per GPU (C2), security 128, k 128, |Delta| = %d bits, "
                                   "valid ciphertexts of random plaintexts" % (args.rows, args.cols, (-delta).bit_length()),
                       "elements_per_gpu": E, "parallelism": "row-shard x%d" % world,
                       "collective": "all_gather of the final result" if world > 1 else "none"},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def main_scal_matmul(args):
    """Configs C3 / C4: out = s (cols x cols plaintexts, harness ramp 1..cols^2, benchmarks/local.cpp:
    171-174) applied to a rows x cols ciphertext block per GPU (row shard of a (rows N) x cols matrix); the exponent
    matrix and Enc(0) are replicated, the result rows are all-gathered once per step.  One JSON line, own metric."""
    import numpy as np
    import torch
    from cofhe_amd import Engine, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("COFHE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _StdoutToStderr():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
    dev = torch.device("cuda", local_rank)
    with open(os.path.join(ROOT, "tests", "golden", "params_s128_k128.json")) as fh:
        prm = json.load(fh)
    eng = Engine(hx(prm["delta"]), device=local_rank)
    n, m, p = args.rows, args.cols, args.cols
    rng = SplitMix64(2000 + rank)
    bound_bits = hx(prm["exponent_bound"]).bit_length() - 1
    cts = encrypt_tensor_gpu(eng, torch, prm, [rng.bits(prm["k"]) for _ in range(n * m)], rng.bits(bound_bits), dev)
    zero = encrypt_tensor_gpu(eng, torch, prm, [0], SplitMix64(7).bits(bound_bits), dev)      # the same Enc(0) on every rank
    ex = torch.from_numpy(exp_records([j * p + k + 1 for j in range(m) for k in range(p)]).view(np.int32)).to(dev)
    out = torch.empty(n * p * 336, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.scal_matmul_records(cts.data_ptr(), ex.data_ptr(), zero.data_ptr(), out.data_ptr(), n, m, p, stream)
        if dist is not None:
            return shard.all_gather_rows(out, n * world, p, dist, world, rank)
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        outs = n * p * world * args.steps
        print(json.dumps({
            "metric": "output ciphertexts/sec, scal_matmul (plaintext matrix x ciphertext matrix)", "value": round(outs / elapsed, 2),
            "unit": "output-ciphertexts/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "scal_matmul %dx%d ciphertexts per GPU x %dx%d plaintexts (harness exponents 1..%d), "
                                   "security 128, k 128" % (n, m, m, p, m * p),
                       "ciphertext_macs_per_s": round(n * m * p * world * args.steps / elapsed, 1),
                       "parallelism": "row-shard x%d" % world,
                       "collective": "all_gather of the result rows" if world > 1 else "none"}}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
