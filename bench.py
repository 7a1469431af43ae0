#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X CoFHE engine.

Metric (BASELINE.json): ciphertext-ops/s and HBM GB/s (% of the 8 TB/s roofline) for the
128x128 `ciphertext_matadd` (config C2: E = 16 384 ciphertexts, security 128, k = 128).
A "step" is one add_ciphertext_tensors over the whole resident tensor, chained like the
reference harness (res = add(res, ct2), /root/reference/benchmarks/local.cpp:99-117).

Inputs are VALID ciphertexts of random plaintexts produced by the product path itself (GPU
powering + composition kernels: c1 = h^r, c2 = f^m o pk^r, one r per tensor as in
cpu_cryptosystem_tensor_ops.inl:7-15) from the committed public parameters; they are resident
in HBM before the timed region starts.

One HIP runtime.  Every device buffer, copy, launch, collective, event and fence of this program goes through
libcofhe_hip.so (cofhe_hip_malloc / upload / download / stream_sync / time_compose / all_gather_rows): the process
never initialises PyTorch's own HIP runtime, so nothing depends on how two runtimes in one process order their
work.  `torch` is imported for ONE thing, the host-side rendezvous of the ranks (torch.distributed over gloo/TCP:
barrier, MAX of the elapsed times, hand-over of the RCCL unique id).

More than one GPU: one process per GPU.
  * started by a launcher (torchrun: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment): this process is a rank;
  * `python bench.py --gpus N` with no WORLD_SIZE: this process makes NO GPU call, starts N fresh children of itself,
    one per GPU, with those variables set (`--launch-dry-run` prints them), and relays rank 0's JSON line.
The (128 N) x 128 tensor is row-sharded (reference loops cpu_cryptosystem_tensor_ops.inl:242-264, :396-417 are
independent per row), no exchange between chained adds, ONE all-gather of the final result inside the timed region
-- RCCL through the library's communicator, on the stream the kernels run on.

Extra objects on the JSON line: "roofline" (algorithmic bytes of one launch of the dominant kernel / its HIP-event
time vs 8 TB/s), "roofline_valu" (the bound that binds, from committed counter files) and "cpu_baseline" (the C++/GMP
oracle restating the reference loop, timed on this box's host cores on a bounded sample, rank 0, N = 1 only).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REC_WORDS = 168
REC_BYTES = REC_WORDS * 4
CT_BYTES = 2 * REC_BYTES


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


# ------------------------------------------------------------------------------------------ device buffers
class DBuf:
    """a block of device memory from the library's allocator (cofhe_hip_malloc); .ptr is the device address"""

    def __init__(self, eng, nbytes):
        self.eng, self.nbytes = eng, int(nbytes)
        self.ptr = eng.malloc(max(self.nbytes, 4))

    @classmethod
    def of(cls, eng, host):
        import numpy as np
        a = np.ascontiguousarray(host)
        b = cls(eng, a.nbytes)
        eng.upload(b.ptr, a)
        return b

    def host(self, nbytes=None, dtype="uint32", offset=0):
        return self.eng.download(self.ptr + offset, self.nbytes - offset if nbytes is None else nbytes, dtype)

    def free(self):
        if self.ptr:
            self.eng.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def encrypt_tensor_gpu(eng, prm, plaintexts, r):
    """Product-path encryption: c1 = h^r (shared), c2_i = f^{m_i} o pk^r.  Returns a DBuf of 2E records."""
    import numpy as np
    E = len(plaintexts)
    f = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    h = form_record(hx(prm["h"]["a"]), hx(prm["h"]["b"]), hx(prm["h"]["c"]))
    pk = form_record(hx(prm["pk"]["a"]), hx(prm["pk"]["b"]), hx(prm["pk"]["c"]))
    # (h^r, pk^r): one "ciphertext" whose two records are h and pk, one exponent
    base = DBuf.of(eng, np.concatenate([h, pk]))
    ex = DBuf.of(eng, exp_records([r]))
    hp = DBuf(eng, 2 * REC_BYTES)
    eng.pow_records(base.ptr, ex.ptr, hp.ptr, 1)
    em = DBuf.of(eng, exp_records(plaintexts))
    out = DBuf(eng, E * CT_BYTES)
    eng.encrypt_records(em.ptr, hp.ptr, f, out.ptr, E, prm["k"])
    eng.stream_sync()
    for b in (base, ex, hp, em):
        b.free()
    return out


def random_forms_gpu(eng, prm, n, seed, bits=192):
    """input family (ii) of SURVEY.md 8(d): n independent reduced forms, h^(e_i) with independent random e_i
    (a ladder per form; the exponent width only has to exceed what separates the forms)"""
    import numpy as np
    h = form_record(hx(prm["h"]["a"]), hx(prm["h"]["b"]), hx(prm["h"]["c"]))
    rng = SplitMix64(seed)
    base = DBuf.of(eng, np.tile(h, n))
    ex = DBuf.of(eng, exp_records([rng.bits(bits) | 1 for _ in range(n)]))
    out = DBuf(eng, n * REC_BYTES)
    eng.pow_form_records(base.ptr, ex.ptr, out.ptr, n)
    eng.stream_sync()
    base.free()
    ex.free()
    return out


# ------------------------------------------------------------------------------------------ counter files
def kernel_code_hash():
    """sha256 over the sources the device code is built from (+ the build flags of __graft_entry__): the key that
    ties a committed profiles/r*/traffic.json or valu.json to the kernel it was measured on"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "cofhe_amd", "csrc")
    for f in ("cofhe_hip.hip", "wire.hip", "ctx.hpp", "lane.hpp", "mp.hpp", "qf.hpp", "form_io.hpp", "layout.hpp"):
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(b"--offload-arch=gfx950 -O2 -std=c++17")
    return h.hexdigest()[:16]


def committed_counter_file(name, nrec, key):
    """newest profiles/r*/<name> measured on THIS kernel (code hash) at this launch size; a stale file is refused"""
    stale = None
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), key=os.path.getmtime, reverse=True):
        try:
            with open(cand) as fh:
                tj = json.load(fh)
        except (OSError, ValueError):
            continue
        if tj.get("records_per_launch") != nrec or not tj.get(key):
            continue
        if tj.get("kernel_code_hash") == kernel_code_hash():
            return tj, os.path.relpath(cand, ROOT)
        stale = stale or os.path.relpath(cand, ROOT)
    return None, ("stale (other kernel build): " + stale) if stale else None


def valu_roofline(vj, src, ms_launch, kernel):
    """The bound that binds: VALU issue.  Everything comes from ONE committed counter file (tools/gpu_counters.sh):
    achieved = SQ_INSTS_VALU per launch / launch time; peak = SIMDs x clock / (mix-weighted issue cycles per
    wave-instruction), with the clock MEASURED inside the kernel (delta s_memtime / delta s_memrealtime, diagnostic
    build) and the weights the tools/inst_bench.hip issue costs over the kernel's static instruction mix -- both stated
    in the object.  busy_frac_counters is the direct reading: SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (SIMDs x
    GRBM_GUI_ACTIVE / 8)."""
    if not vj:
        return {"bound": "valu", "achieved": None, "source": src}
    insts = float(vj["valu_wave_insts_per_launch"])
    clock = vj.get("clock_ghz_in_kernel") or vj.get("clock_ghz")
    cyc = vj.get("issue_cycles_per_valu_inst")
    simds = 1024
    out = {"bound": "valu", "unit": "G wave-instr/s", "kernel": kernel, "source": src,
           "valu_wave_insts_per_launch": int(insts), "valu_per_wave": vj.get("valu_per_wave"),
           "achieved": round(insts / (ms_launch * 1e-3) / 1e9, 2)}
    if clock and cyc:
        peak = simds * float(clock) / float(cyc)
        out.update({"peak": round(peak, 1), "frac": round(out["achieved"] / peak, 4), "clock_ghz": clock,
                    "clock_source": vj.get("clock_source"), "issue_cycles_per_valu_inst": cyc,
                    "issue_weights": vj.get("issue_weights_source")})
    else:
        out.update({"peak": None, "frac": None})
    for k in ("busy_frac_counters", "sq_active_inst_valu_quadcycles", "sq_wave_cycles_quadcycles", "sq_busy_cycles",
              "grbm_gui_active_sum_over_8_xcds", "clock_ghz_grbm"):
        if k in vj:
            out[k] = vj[k]
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


# ------------------------------------------------------------------------------------------ launcher (no GPU call)
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environments(n, port, base=None):
    """the environment of each of the n ranks `--gpus n` starts: what torchrun would set, plus the IPC mode RCCL needs
    on this pool"""
    envs = []
    for r in range(n):
        e = dict(base if base is not None else {})
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        envs.append(e)
    return envs


def launch_ranks(n, argv, cmd=None, poll_s=0.1, grace_s=5.0):
    """Parent of `python bench.py --gpus n`: starts n fresh interpreters of this file (one per GPU) BEFORE anything in
    this process has touched the GPU -- it never does -- relays rank 0's stdout and SUPERVISES the ranks: it polls all
    of them, and the first rank that exits non-zero (or by a signal) ends the run -- the others are terminated (they
    are ordinary child processes, no exec of a GPU-initialised process anywhere; left alone they would sit in the gloo /
    RCCL rendezvous until the driver's time limit), the failed rank is named with the tail of its stderr, and the
    parent returns non-zero.  A rank's stderr goes through a pipe to a temporary file so that it can be quoted."""
    import tempfile
    import time
    port = free_port()
    envs = rank_environments(n, port, os.environ)
    cmd = (cmd if cmd is not None else [sys.executable, os.path.abspath(__file__)]) + argv
    procs, errs = [], []
    for r, e in enumerate(envs):
        ef = tempfile.TemporaryFile()
        errs.append(ef)
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=ef))
    # rank 0's stdout is read by a thread so that polling never blocks on it
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def tail(r, nbytes=2000):
        errs[r].flush()
        errs[r].seek(0, os.SEEK_END)
        size = errs[r].tell()
        errs[r].seek(max(0, size - nbytes))
        return errs[r].read().decode(errors="replace")

    failed = None                    # (rank, returncode) of the first rank that failed
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c is not None and c != 0]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(poll_s)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + grace_s
        for p in procs:
            try:
                p.wait(timeout=max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=grace_s)
    for r in range(n):                # the ranks' diagnostics, in rank order, on the parent's stderr
        txt = tail(r, 20000 if failed is None else 4000)
        if txt and (failed is None or r == failed[0]):
            sys.stderr.write(txt if txt.endswith("\n") else txt + "\n")
    if failed is not None:
        r, c = failed
        how = "signal %d" % -c if c < 0 else "exit code %d" % c
        sys.stderr.write("bench.py: rank %d of %d failed (%s); the other ranks were terminated\n" % (r, n, how))
        sys.stderr.flush()
        return abs(c) if abs(c) < 256 else 1
    sys.stdout.write(out0[0].decode() if out0 else "")
    sys.stdout.flush()
    return 0


class Rendezvous:
    """Host-side meeting point of the ranks: torch.distributed over gloo / TCP, CPU tensors only (the device side is the
    library's RCCL communicator).  world == 1 needs no process group at all."""

    def __init__(self, world, rank):
        self.world, self.rank, self.dist = world, rank, None
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # finite timeouts everywhere: a rank that died before the rendezvous must not leave the others waiting for
            # the default half hour (the parent's supervisor ends the run first; this is the second line of defence)
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=float(os.environ.get("COFHE_RDV_TIMEOUT_S", "120"))))
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_float(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def or_int(self, x):
        """bitwise OR of a small non-negative integer over the ranks (the device status words)"""
        if self.dist is None:
            return int(x)
        import torch
        t = torch.tensor([int(x)], dtype=torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.BOR)
        return int(t.item())

    def broadcast_bytes(self, data, nbytes, root=0):
        """`nbytes` bytes from rank `root` to everybody"""
        if self.dist is None:
            return bytes(data)
        import torch
        t = torch.zeros(nbytes, dtype=torch.uint8)
        if self.rank == root:
            t.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        self.dist.broadcast(t, src=root)
        return bytes(t.numpy().tobytes())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def timed_region(steps, warmup, step, gather, fence, rdv):
    """The contract's timing protocol, one place for every workload and world size: `warmup` untimed steps (and one
    untimed gather to warm the communicator), then EXACTLY `steps` steps + ONE gather between two (rank barrier +
    device fence) pairs, MAX of the elapsed time over the ranks.  step(i) enqueues one pass, gather() enqueues the
    collective (or does nothing), fence() waits for the device."""
    for i in range(warmup):
        step(i)
    gather()
    rdv.barrier()
    fence()
    rdv.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    gather()
    fence()
    rdv.barrier()
    elapsed = time.perf_counter() - t0
    return rdv.max_float(elapsed)


def make_comm(eng, rdv, force):
    """the library's RCCL communicator over the ranks (unique id drawn by rank 0, handed over the gloo store); None for
    a single rank unless `force` (rehearsal of the collective on one GPU)"""
    if rdv.world == 1 and not force:
        return None, None
    with _StdoutToStderr():
        uid = eng.comm_unique_id() if rdv.rank == 0 else b""
        uid = rdv.broadcast_bytes(uid, 128)
        comm = eng.comm_create(uid, rdv.world, rdv.rank)
        _, _, nranks = eng.comm_info(comm)
    return comm, nranks


def rank_identity():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def device_of(local_rank, share):
    """the GPU a rank computes on: its local rank -- or, with --share-gpu (a rehearsal of the multi-rank path on a box with
    fewer GPUs than ranks; RCCL may refuse two ranks on one device), local rank modulo the visible devices"""
    if not share:
        return local_rank
    import torch
    return local_rank % max(1, torch.cuda.device_count())       # device_count does not initialise the GPU


# ------------------------------------------------------------------------------------------ matadd (the metric)
def main_matadd(args):
    import numpy as np
    from cofhe_amd import Engine, load_library, shard

    world, rank, local_rank = rank_identity()
    rdv = Rendezvous(world, rank)
    with open(os.path.join(ROOT, "tests", "golden", "params_s128_k128.json")) as fh:
        prm = json.load(fh)
    delta = hx(prm["delta"])
    k = prm["k"]
    if args.lib:
        load_library(os.path.abspath(args.lib))
    eng = Engine(delta, device=device_of(local_rank, args.share_gpu))
    for opt in getattr(args, "option", []):
        name, _, val = opt.partition("=")
        eng.set_option(name, int(val))
    comm, rccl_nranks = make_comm(eng, rdv, args.force_comm)

    # weak: a whole rows x cols tensor per GPU; strong: this rank's row block of ONE rows x cols tensor
    _, my_rows, total_rows = shard.rows_for_mode(args.rows, world, rank, args.scaling)
    E = my_rows * args.cols                       # ciphertexts on this GPU
    if E == 0:
        raise SystemExit("strong scaling: fewer rows than GPUs")
    rng = SplitMix64(1000 + rank)
    bound_bits = hx(prm["exponent_bound"]).bit_length() - 1
    pts1 = [rng.bits(k) for _ in range(E)]
    pts2 = [rng.bits(k) for _ in range(E)]
    ct1 = encrypt_tensor_gpu(eng, prm, pts1, rng.bits(bound_bits))
    ct2 = encrypt_tensor_gpu(eng, prm, pts2, rng.bits(bound_bits))
    nrec = 2 * E
    stream = 0                                    # the library's null stream: kernels, collective, events and fences
    bufs = [DBuf(eng, ct1.nbytes), DBuf(eng, ct1.nbytes)]
    gathered = DBuf(eng, total_rows * args.cols * CT_BYTES) if comm is not None else None
    state = {"cur": ct1}

    def step(i):
        dst = bufs[i & 1]
        eng.compose_records(state["cur"].ptr, ct2.ptr, dst.ptr, nrec, stream)
        state["cur"] = dst

    def restart(_=None):
        state["cur"] = ct1

    def gather():
        # reassemble the whole result: rank r owns a contiguous block of its rows; same stream as the kernels
        if comm is not None:
            eng.all_gather_rows(comm, state["cur"].ptr, total_rows, args.cols * CT_BYTES, gathered.ptr, stream)

    def fence():
        eng.stream_sync(stream)

    # warm-up and timed chain both start from ct1 (the reference harness's chain)
    def step_chain(i):
        if i == 0:
            restart()
        step(i)

    eng.device_status(clear=True)
    elapsed = timed_region(args.steps, args.warmup, step_chain, gather, fence, rdv)
    status = eng.device_status(clear=True)

    # ---- roofline: HIP events around back-to-back launches of the dominant kernel ----------
    iters = max(5, min(args.steps, 20))
    ms_launch = eng.time_compose(ct1.ptr, ct2.ptr, bufs[0].ptr, nrec, iters, stream)
    # algorithmic bytes: S = payload bytes of one serialised ciphertext (no offset table),
    # measured on the inputs and the output actually used; matadd moves 3 S per ciphertext-op
    samp = min(E, 1024)

    def payload_per_ct(buf):
        recs = buf.host(samp * CT_BYTES)
        b = eng.records_to_bytes(recs, [samp])
        return (len(b) - 4 - 4 - 48 * samp) / samp

    S_in1, S_in2, S_out = payload_per_ct(ct1), payload_per_ct(ct2), payload_per_ct(bufs[0])
    alg_bytes = (S_in1 + S_in2 + S_out) * E
    achieved = alg_bytes / (ms_launch * 1e-3) / 1e9
    # HBM-side bytes per launch: PMC counters (FETCH_SIZE, WRITE_SIZE, separate rocprofv3 passes) corrected
    # with factors calibrated on a record-copy kernel of the same access pattern -- measured by
    # tools/gpu_counters.sh and committed under profiles/ (a profiler cannot run inside this process)
    tj, traffic_src = committed_counter_file("traffic.json", nrec, "traffic_bytes_per_launch")
    traffic = int(tj["traffic_bytes_per_launch"]) if tj else None
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "k_compose_wg", "launch_ms": round(ms_launch, 4),
                "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_code_hash": kernel_code_hash(),
                "note": "class-group composition is integer-VALU bound (see DESIGN.md); HBM fraction is reported as the contract asks"}
    vj, valu_src = committed_counter_file("valu.json", nrec, "valu_wave_insts_per_launch")
    roofline_valu = valu_roofline(vj, valu_src, ms_launch, "k_compose_wg")

    # ---- the ciphertext-level entry point on the same inputs (what add_ciphertext_tensors calls): the operands were made
    # by encrypt_tensor, one r per tensor as in the reference, so their c1 are shared and that composition is done once.
    # Reported beside the headline, which stays on the plain 2E-composition kernel.
    folded = None
    if rank == 0 and world == 1:
        fo = DBuf(eng, ct1.nbytes)
        eng.add_ciphertext_records(ct1.ptr, ct2.ptr, fo.ptr, E, stream)
        eng.compose_records(ct1.ptr, ct2.ptr, bufs[1].ptr, nrec, stream)
        eng.stream_sync(stream)
        same = bool(np.array_equal(fo.host(), bufs[1].host()))
        # HIP events on the launch stream, 3 warm-up calls (the first use of k_c1_distinct / k_add_ct / k_c1_spread loads
        # their code objects: round 3 timed 20 calls by wall clock after ONE untimed call and the driver's fresh box
        # reported 1.52 ms where every other run had 0.33), then 7 repetitions of `iters` calls: median, min, max
        for _ in range(3):
            eng.add_ciphertext_records(ct1.ptr, ct2.ptr, fo.ptr, E, stream)
        eng.stream_sync(stream)
        reps = sorted(eng.time_stream(lambda: eng.add_ciphertext_records(ct1.ptr, ct2.ptr, fo.ptr, E, stream), iters, stream)
                      for _ in range(7))
        msf = reps[len(reps) // 2]
        folded = {"entry": "cofhe_hip_add_ciphertext_records (shared c1 folded: E + 1 compositions + a scan and a copy)",
                  "ms_per_add": round(msf, 4), "min_ms": round(reps[0], 4), "max_ms": round(reps[-1], 4),
                  "timing": "HIP events on the launch stream, median of %d x %d calls after 3 warm-up calls" % (len(reps), iters),
                  "ciphertext_ops_per_s": round(E / (msf * 1e-3), 1), "same_records_as_plain": same}
        fo.free()

    # ---- input family (ii): independent random forms, same launch size ------------------------
    fam2 = None
    if rank == 0 and world == 1 and not args.no_family2:
        fa = random_forms_gpu(eng, prm, nrec, 4242)
        fb = random_forms_gpu(eng, prm, nrec, 4343)
        fo = DBuf(eng, fa.nbytes)
        eng.compose_records(fa.ptr, fb.ptr, fo.ptr, nrec, stream)      # warm
        ms2 = eng.time_compose(fa.ptr, fb.ptr, fo.ptr, nrec, iters, stream)
        fam2 = {"inputs": "2 x %d independent random forms h^(e_i), 192-bit e_i (SURVEY 8(d) family ii)" % nrec,
                "launch_ms": round(ms2, 4), "ciphertext_ops_per_s": round(E / (ms2 * 1e-3), 1),
                "family_i_launch_ms": round(ms_launch, 4)}
        for b in (fa, fb, fo):
            b.free()

    # ---- CPU baseline (oracle = checker, timed on a bounded sample) -------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        ns = min(E, 16384)
        a = eng.records_to_bytes(ct1.host(ns * CT_BYTES), [ns])
        b = eng.records_to_bytes(ct2.host(ns * CT_BYTES), [ns])
        cores = host_cpu_share(O.max_threads())
        chain = 5
        sec, want = O.time_matadd_chain(delta, a, b, chain, threads=cores, want_out=True)
        if sec < 6.0:       # size the sample to ~10 s of CPU work at the full share
            chain = int(min(400, max(chain, chain * 10.0 / max(sec, 1e-3))))
            sec, want = O.time_matadd_chain(delta, a, b, chain, threads=cores, want_out=True)
        # one whole-tensor add at the full share, 11 times: the median is what one call of the reference's
        # add_ciphertext_tensors costs (the chain above hides the first-touch and thread start-up of a single call)
        singles = sorted(O.time_matadd_chain(delta, a, b, 1, threads=cores) for _ in range(11))
        # one thread: the scalar port itself
        ns1 = min(ns, 4096)
        a1 = eng.records_to_bytes(ct1.host(ns1 * CT_BYTES), [ns1])
        b1 = eng.records_to_bytes(ct2.host(ns1 * CT_BYTES), [ns1])
        sec1 = O.time_matadd_chain(delta, a1, b1, 4, threads=1)
        chain1 = int(min(400, max(4, 4 * 8.0 / max(sec1, 1e-3))))
        sec1 = O.time_matadd_chain(delta, a1, b1, chain1, threads=1)
        # the same chain on the GPU must give the same bytes (the oracle is only the checker)
        x, o = DBuf(eng, ns * CT_BYTES), DBuf(eng, ns * CT_BYTES)
        eng.upload(x.ptr, ct1.host(ns * CT_BYTES))
        for _ in range(chain):
            eng.compose_records(x.ptr, ct2.ptr, o.ptr, 2 * ns, stream)
            x, o = o, x
        eng.stream_sync(stream)
        got = eng.records_to_bytes(x.host(), [ns])
        x.free()
        o.free()
        cpu = {"value": round(ns * chain / sec, 2), "unit": "ciphertext-ops/s", "cores": cores, "kind": "port",
               "sample": "%d ciphertexts x %d chained adds of the same workload (%.1f s)" % (ns, chain, sec),
               "parity_with_gpu": bool(got == want),
               "single_thread": {"value": round(ns1 * chain1 / sec1, 2), "cores": 1,
                                 "sample": "%d ciphertexts x %d chained adds (%.1f s)" % (ns1, chain1, sec1)},
               "single_op": {"median_ms": round(singles[len(singles) // 2] * 1e3, 3), "min_ms": round(singles[0] * 1e3, 3),
                             "max_ms": round(singles[-1] * 1e3, 3), "runs": len(singles), "cores": cores,
                             "what": "one %d-ciphertext add_ciphertext_tensors call" % ns}}

    status |= eng.device_status(clear=True)
    status = rdv.or_int(status)          # every rank's status word reaches the line and the exit code, not only rank 0's
    rc = 0
    if rank == 0:
        ops = total_rows * args.cols * args.steps
        line = {
            "metric": "ciphertext-ops/sec + HBM GB/s (% roofline), 128x128 matadd",
            "value": round(ops / elapsed, 2),
            "unit": "ciphertext-ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "ciphertext_matadd %dx%d %s (C2), security 128, k 128, |Delta| = %d bits, "
                                   "valid ciphertexts of random plaintexts" % (args.rows, args.cols, "per GPU" if args.scaling == "weak"
                                                                               else "in total, row-sharded", (-delta).bit_length()),
                       "elements_per_gpu": E, "parallelism": "row-shard x%d" % world,
                       "collective": ("cofhe_hip_all_gather_rows (RCCL) of the final result, on the compute stream"
                                      if comm is not None else "none"),
                       "rccl_nranks": rccl_nranks,
                       "fence": "cofhe_hip_stream_sync (the runtime the kernels are launched on); ranks meet over gloo"},
            "device_status": status,
            "roofline": roofline,
            "roofline_valu": roofline_valu,
            "input_family_ii": fam2,
            "add_ciphertext_records": folded,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
        if status != 0:
            sys.stderr.write("bench.py: device status word %d after the run (a safety cap was hit: results are not trustworthy)\n" % status)
    if status != 0:
        rc = 3                           # on every rank (the word is OR-reduced): the launcher names whichever exits first
    if comm is not None:
        eng.comm_destroy(comm)
    rdv.close()
    return rc


# ------------------------------------------------------------------------------------------ scal_matmul (C3 / C4)
def main_scal_matmul(args):
    """Configs C3 / C4: out = s (cols x cols plaintexts, harness ramp 1..cols^2, benchmarks/local.cpp:
    171-174) applied to a rows x cols ciphertext block per GPU (row shard of a (rows N) x cols matrix); the exponent
    matrix and Enc(0) are replicated, the result rows are all-gathered once per step.  One JSON line, own metric, with
    the roofline of k_scal_matmul_wnaf (HIP events around that kernel inside the library) and the CPU baseline (the
    oracle's scal_2d on a bounded row sample)."""
    import numpy as np
    from cofhe_amd import Engine, load_library, shard

    world, rank, local_rank = rank_identity()
    rdv = Rendezvous(world, rank)
    with open(os.path.join(ROOT, "tests", "golden", "params_s128_k128.json")) as fh:
        prm = json.load(fh)
    delta = hx(prm["delta"])
    if args.lib:
        load_library(os.path.abspath(args.lib))
    eng = Engine(delta, device=device_of(local_rank, args.share_gpu))
    for opt in getattr(args, "option", []):
        name, _, val = opt.partition("=")
        eng.set_option(name, int(val))
    comm, rccl_nranks = make_comm(eng, rdv, args.force_comm)
    _, n, total_rows = shard.rows_for_mode(args.rows, world, rank, args.scaling)
    if n == 0:
        raise SystemExit("strong scaling: fewer rows than GPUs")
    m, p = args.cols, args.cols
    rng = SplitMix64(2000 + rank)
    bound_bits = hx(prm["exponent_bound"]).bit_length() - 1
    cts = encrypt_tensor_gpu(eng, prm, [rng.bits(prm["k"]) for _ in range(n * m)], rng.bits(bound_bits))
    zero = encrypt_tensor_gpu(eng, prm, [0], SplitMix64(7).bits(bound_bits))      # the same Enc(0) on every rank
    exps = [j * p + kk + 1 for j in range(m) for kk in range(p)]
    ex = DBuf.of(eng, exp_records(exps))
    out = DBuf(eng, n * p * CT_BYTES)
    gathered = DBuf(eng, total_rows * p * CT_BYTES) if comm is not None else None
    stream = 0

    def step(_):
        eng.scal_matmul_records(cts.ptr, ex.ptr, zero.ptr, out.ptr, n, m, p, stream)
        if comm is not None:
            eng.all_gather_rows(comm, out.ptr, total_rows, p * CT_BYTES, gathered.ptr, stream)

    eng.device_status(clear=True)
    elapsed = timed_region(args.steps, max(1, args.warmup), step, lambda: None, lambda: eng.stream_sync(stream), rdv)
    status = eng.device_status(clear=True)

    # ---- roofline of the dominant kernel: events around every kernel of the product on the launch stream
    eng.set_option("profile_kernels", 1)
    iters = max(1, min(args.steps, 3))
    for _ in range(iters):
        eng.scal_matmul_records(cts.ptr, ex.ptr, zero.ptr, out.ptr, n, m, p, stream)
    eng.stream_sync(stream)
    # per call: summed duration and launch count of every kernel of the product (the tree form launches k_tree_level once per
    # level and row chunk); the dominant one carries the roofline object
    kms, launches = {}, {}
    for kn in ("k_tree_level", "k_scal_matmul_wnaf", "k_pow_table", "k_wnaf_digits"):
        ms, cnt = eng.profile_read(kn)
        kms[kn] = round(ms / iters, 4) if cnt else None
        launches[kn] = cnt // iters if cnt else 0
    eng.profile_read("k_wnaf_digits", clear=True)
    eng.set_option("profile_kernels", 0)
    samp = min(n * m, 1024)
    S_in = (len(eng.records_to_bytes(cts.host(samp * CT_BYTES), [samp])) - 8 - 48 * samp) / samp
    so = min(n * p, 1024)
    S_out = (len(eng.records_to_bytes(out.host(so * CT_BYTES), [so])) - 8 - 48 * so) / so
    expbits = max(exps).bit_length()
    alg_bytes = n * m * S_in + n * p * S_out + m * p * ((expbits + 7) // 8)
    roofline = None
    timed = {kk: v for kk, v in kms.items() if v}
    if timed:
        dom = max(timed, key=timed.get)
        ach = alg_bytes / (timed[dom] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(ach, 4), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 8),
                    "traffic": None, "kernel": dom, "launch_ms": timed[dom], "launches_per_call": launches[dom],
                    "other_kernels_ms": {kk: v for kk, v in kms.items() if kk != dom},
                    "algorithmic_bytes_per_launch": int(alg_bytes),
                    "formula": "(n m + n p) S + m p ceil(expbits / 8), S = measured payload of a serialised ciphertext; launch_ms = the "
                               "dominant kernel's launches of ONE product summed (HIP events inside the library)",
                    "kernel_code_hash": kernel_code_hash(),
                    "note": "integer-VALU bound: one output coefficient is ~bits squarings + m bits/(w+1) compositions"}
        mj, msrc = committed_counter_file("valu_matmul.json", n * p * 2, "valu_wave_insts_per_launch")
        if mj and mj.get("shape") == [n, m, p] and mj.get("kernel", "k_scal_matmul_wnaf") == dom:
            roofline["valu"] = valu_roofline(mj, msrc, timed[dom], dom)
            roofline["traffic"] = mj.get("traffic_bytes_per_launch")

    # ---- CPU baseline: the oracle's scal_2d (reference loop structure, qfi.inl wNAF-7 tables) on a row sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        cores = host_cpu_share(O.max_threads())
        sb = eng.records_to_bytes(zero.host(), [1])
        # plaintext tensor in the reference's format: ndim, shape, offsets, little-endian magnitudes
        import struct
        body, offs = b"", []
        for v in exps:
            offs.append(len(body))
            body += v.to_bytes(v.bit_length() // 8 + 1, "little")
        s_bytes = struct.pack("<III", 2, m, p) + b"".join(struct.pack("<Q", o) for o in offs) + body
        nr = 1
        cb = eng.records_to_bytes(cts.host(nr * m * CT_BYTES), [nr, m])
        sec = O.time_scal_2d(delta, s_bytes, cb, sb, threads=cores)
        if sec < 6.0 and n > nr:
            nr = int(min(n, max(1, round(nr * 12.0 / max(sec, 1e-3)))))
            cb = eng.records_to_bytes(cts.host(nr * m * CT_BYTES), [nr, m])
            sec = O.time_scal_2d(delta, s_bytes, cb, sb, threads=cores)
        # the checker's verdict on the GPU result: as many leading rows as ~65 k multiply-accumulates allow (at least one)
        nc = max(1, min(nr, 65536 // max(1, m * p)))
        want = O.scal_2d(delta, s_bytes, eng.records_to_bytes(cts.host(nc * m * CT_BYTES), [nc, m]), sb)
        got = eng.records_to_bytes(out.host(nc * p * CT_BYTES), [nc, p])
        cpu = {"value": round(nr * p / sec, 3), "unit": "output-ciphertexts/s", "cores": cores, "kind": "port",
               "ciphertext_macs_per_s": round(nr * m * p / sec, 1),
               "sample": "%d of the %d rows of the same product (%d x %d . %d x %d, %.1f s)" % (nr, n, nr, m, m, p, sec),
               "parity_with_gpu": bool(got == want), "parity_rows_checked": nc}

    if rank == 0 and args.dump_dir:
        os.makedirs(args.dump_dir, exist_ok=True)
        step(0)
        eng.stream_sync(stream)
        g = gathered if comm is not None else out
        for name, blob in (("cts.bin", eng.records_to_bytes(cts.host(), [n, m])), ("zero.bin", eng.records_to_bytes(zero.host(), [1])),
                           ("out.bin", eng.records_to_bytes(g.host(), [total_rows, p]))):
            with open(os.path.join(args.dump_dir, name), "wb") as fh:
                fh.write(blob)
        with open(os.path.join(args.dump_dir, "meta.json"), "w") as fh:
            json.dump({"n": n, "m": m, "p": p, "world": world, "distributed": comm is not None, "rccl_nranks": rccl_nranks}, fh)
    status |= eng.device_status(clear=True)
    status = rdv.or_int(status)          # every rank's status word reaches the line and the exit code, not only rank 0's
    rc = 0
    if rank == 0:
        outs = total_rows * p * args.steps
        print(json.dumps({
            "metric": "output ciphertexts/sec, scal_matmul (plaintext matrix x ciphertext matrix)", "value": round(outs / elapsed, 2),
            "unit": "output-ciphertexts/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "scal_matmul %dx%d ciphertexts %s x %dx%d plaintexts (harness exponents 1..%d), "
                                   "security 128, k 128" % (args.rows, m, "per GPU" if args.scaling == "weak" else "in total, row-sharded",
                                                            m, p, m * p),
                       "ciphertext_macs_per_s": round(total_rows * m * p * args.steps / elapsed, 1),
                       "parallelism": "row-shard x%d" % world,
                       "collective": "cofhe_hip_all_gather_rows (RCCL) of the result rows, every step" if comm is not None else "none",
                       "rccl_nranks": rccl_nranks},
            "device_status": status, "roofline": roofline, "cpu_baseline": cpu}))
    if status != 0:
        rc = 3
    if comm is not None:
        eng.comm_destroy(comm)
    rdv.close()
    return rc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--cols", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): every GPU holds its own rows x cols tensor.  strong: ONE rows x cols "
                         "tensor is row-sharded over the GPUs (e.g. --rows 128 --cols 128, --rows 1024 --cols 1024)")
    ap.add_argument("--dump-dir", default=None, help="scal_matmul: rank 0 writes its inputs and the gathered result in the "
                                                     "wire format there (tests/test_gpu_parity.py checks them against the oracle)")
    ap.add_argument("--lib", default=None, help="kernel-tuning experiments: another build of libcofhe_hip.so (tools/build_variant.sh)")
    ap.add_argument("--no-family2", action="store_true", help="skip timing the independent-random-forms input family")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="tuning experiments: cofhe_hip_set_option on the rank's context before anything is timed (e.g. defer_min_n=0)")
    ap.add_argument("--workload", choices=["matadd", "scal_matmul"], default="matadd",
                    help="matadd: the BASELINE.json metric (default).  scal_matmul: configs C3/C4, a rows x cols "
                         "ciphertext block per GPU times a cols x cols plaintext matrix, result rows all-gathered")
    ap.add_argument("--force-comm", action="store_true",
                    help="one rank: still create the RCCL communicator and run the all-gather through it (rehearsal on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: ranks beyond the visible GPUs share them (local rank modulo the device count)")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="print the rank environments and the command `--gpus N` would start, as JSON, and exit (no GPU call)")
    args = ap.parse_args(argv)
    if os.environ.get("COFHE_BENCH_FORCE_DIST") == "1":
        args.force_comm = True
    return args


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")
    launched = "WORLD_SIZE" in os.environ          # a launcher (torchrun, or this file's own parent) made us a rank
    if args.launch_dry_run:
        port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
        print(json.dumps({"ranks": rank_environments(args.gpus, port), "command": [sys.executable, os.path.abspath(__file__)] +
                          [a for a in argv if a != "--launch-dry-run"],
                          "mode": "external launcher: this process is rank %s" % os.environ.get("RANK", "0") if launched
                          else ("spawn %d children" % args.gpus if args.gpus > 1 else "single rank, in process")}))
        return 0
    if not launched and args.gpus > 1:
        return launch_ranks(args.gpus, argv)
    if launched and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s; the launcher's world is used\n" % (args.gpus, os.environ["WORLD_SIZE"]))
    if args.workload == "scal_matmul":
        return main_scal_matmul(args)
    return main_matadd(args)


if __name__ == "__main__":
    sys.exit(main())
