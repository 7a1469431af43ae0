"""CPU, 2 and 3 ranks over gloo: the row-sharded paths (partition -> per-rank op -> gather) reassemble exactly the
unsharded result.  The per-rank operation is the ORACLE (the checker standing in for the GPU kernels, which need a
GPU).  Under test: the partition bench.py uses (cofhe_amd/shard.py) and the collective PLAN the product executes --
cofhe_hip_gather_plan, the host function cofhe_hip_all_gather_rows (csrc/shard.hip) calls: execute_gather_plan below
runs that plan's output step for step over gloo the way shard.hip runs it over RCCL (ncclAllGather when the blocks are
equal; otherwise one broadcast per non-empty block from its owning rank into out + offset, root reading its local block,
everybody else receiving in place)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_json

sys.path.insert(0, os.path.join(ROOT, "oracle"))


def hx(s):
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


def execute_gather_plan(local_bytes, n_rows, row_bytes, world, rank):
    """what cofhe_hip_all_gather_rows does with cofhe_hip_gather_plan's output (csrc/shard.hip), over gloo: local_bytes is
    this rank's row block as a uint8 tensor; returns the assembled uint8 tensor of n_rows * row_bytes bytes"""
    from cofhe_amd import engine
    plan, uniform = engine.gather_plan(n_rows, row_bytes, world)
    assert plan[rank][1] == local_bytes.numel(), "the rank's block is not what the plan expects of it"
    out = torch.empty(n_rows * row_bytes, dtype=torch.uint8)
    if uniform:
        dist.all_gather_into_tensor(out, local_bytes.contiguous())
        return out
    for r, (off, cnt) in enumerate(plan):
        if cnt == 0:
            continue
        dst = out[off:off + cnt]
        if r == rank:
            dst.copy_(local_bytes)                 # ncclBroadcast(src = d_local, dst = out + off) on the root
        dist.broadcast(dst, src=r)                 # in place (src == dst) on everybody else
    return out


def _gather(local_i32, n_rows, n_cols, world, rank):
    from cofhe_amd import shard
    lb = torch.from_numpy(np.ascontiguousarray(local_i32).view(np.uint8).copy())       # (a view of an EMPTY int32 tensor has stride 0)
    full = execute_gather_plan(lb, n_rows, n_cols * shard.CT_WORDS * 4, world, rank)
    return full.view(torch.int32)


def _worker(rank, world, port, n_rows, n_cols, recs1, recs2, delta, out_q, scaling="strong"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cofhe_amd import shard
    import oracle_lib as O
    import simlib as S
    import pyref as P
    # bench.py's bookkeeping: weak = every rank its own n_rows-row tensor (here: consecutive blocks of the test's
    # 2 n_rows-row input), strong = one n_rows-row tensor split over the ranks
    row0, rows, total = shard.rows_for_mode(n_rows, world, rank, scaling)
    w = n_cols * shard.CT_WORDS
    a = torch.from_numpy(recs1)[row0 * w: (row0 + rows) * w]
    b = torch.from_numpy(recs2)[row0 * w: (row0 + rows) * w]
    if scaling == "strong":
        assert torch.equal(a, shard.shard_records(torch.from_numpy(recs1), n_rows, n_cols, world, rank))
    n_rows = total

    def to_bytes(t):
        arr = t.numpy().view(np.uint32).reshape(-1, S.REC_WORDS)
        forms = [P.Form(*S.record_form(r)) for r in arr]
        cts = list(zip(forms[0::2], forms[1::2]))
        return P.serialize_ciphertext_tensor([rows, n_cols], cts)

    if rows:
        out = O.add(delta, to_bytes(a), to_bytes(b))
        _, cts = P.deserialize_ciphertext_tensor(out)
        local = np.concatenate([S.form_record(f.a, f.b, f.c) for ct in cts for f in ct]).view(np.int32)
    else:
        local = np.zeros(0, dtype=np.int32)
    full = _gather(local, n_rows, n_cols, world, rank)
    if rank == 0:
        out_q.put(full.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


# even, ragged (remainder row), per-rank tensors, fewer rows than ranks (an empty block)
@pytest.mark.parametrize("world,n_rows,scaling", [(2, 4, "strong"), (2, 3, "strong"), (2, 2, "weak"), (3, 3, "strong"), (3, 4, "strong"),
                                                  (3, 2, "strong")])
def test_row_sharded_add_executes_the_library_plan(world, n_rows, scaling):
    import simlib as S
    import pyref as P
    prm = load_json("params_tiny_k8.json")
    d = hx(prm["delta"])
    n_cols = 2
    rng = P.SplitMix64(77)
    total = n_rows * (world if scaling == "weak" else 1)
    cts1 = [(P.random_form(d, rng, 16, 12), P.random_form(d, rng, 16, 12)) for _ in range(total * n_cols)]
    cts2 = [(P.random_form(d, rng, 16, 12), P.random_form(d, rng, 16, 12)) for _ in range(total * n_cols)]
    pack = lambda cts: np.concatenate([S.form_record(f.a, f.b, f.c) for ct in cts for f in ct]).view(np.int32)
    want = pack(P.add_tensor(cts1, cts2))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_rows + 10 * world + (7 if scaling == "weak" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, n_cols, pack(cts1), pack(cts2), d, q, scaling)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got, want)


def _worker_scal(rank, world, port, n, m, p, recs, s_bytes, zero_bytes, delta, out_q, scaling="strong"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cofhe_amd import shard
    import oracle_lib as O
    import simlib as S
    import pyref as P
    row0, rows, n = shard.rows_for_mode(n, world, rank, scaling)                   # this rank's rows of the operand
    a = torch.from_numpy(recs)[row0 * m * shard.CT_WORDS: (row0 + rows) * m * shard.CT_WORDS]
    if rows:
        arr = a.numpy().view(np.uint32).reshape(-1, S.REC_WORDS)
        forms = [P.Form(*S.record_form(r)) for r in arr]
        block = P.serialize_ciphertext_tensor([rows, m], list(zip(forms[0::2], forms[1::2])))
        out = O.scal_2d(delta, s_bytes, block, zero_bytes)                       # replicated exponents and Enc(0)
        _, cts = P.deserialize_ciphertext_tensor(out)
        local = np.concatenate([S.form_record(f.a, f.b, f.c) for ct in cts for f in ct]).view(np.int32)
    else:
        local = np.zeros(0, dtype=np.int32)
    full = _gather(local, n, p, world, rank)                                       # rows of the n x p result
    if rank == 0:
        out_q.put(full.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows,scaling", [(2, 4, "strong"), (2, 3, "strong"), (2, 2, "weak"), (3, 4, "strong")])
def test_row_sharded_scal_matmul_executes_the_library_plan(world, n_rows, scaling):
    """config C4: the plaintext-matrix x ciphertext-matrix product row-sharded over 2 / 3 ranks (replicated exponent
    matrix and Enc(0), one gather of the result rows by the library's plan) reassembles the unsharded product"""
    import simlib as S
    import pyref as P
    prm = load_json("params_tiny_k8.json")
    d = hx(prm["delta"])
    m, p = 3, 2
    rng = P.SplitMix64(91)
    total = n_rows * (world if scaling == "weak" else 1)
    cts = [(P.random_form(d, rng, 16, 12), P.random_form(d, rng, 16, 12)) for _ in range(total * m)]
    zero = (P.random_form(d, rng, 16, 12), P.random_form(d, rng, 16, 12))
    svals = [rng.below(200) - 60 for _ in range(m * p)]
    want_cts = P.scal_tensor_2d(svals, cts, zero, total, m, p, d)
    pack = lambda cc: np.concatenate([S.form_record(f.a, f.b, f.c) for ct in cc for f in ct]).view(np.int32)
    # plaintext tensor in the reference's binary format (cpu_cryptosystem.inl:229-270)
    import struct
    offs, body = [], b""
    for v in svals:
        offs.append(len(body) | ((1 << 63) if v <= 0 else 0))
        body += abs(v).to_bytes(abs(v).bit_length() // 8 + 1 if v else 1, "little")
    s_bytes = struct.pack("<III", 2, m, p) + b"".join(struct.pack("<Q", o) for o in offs) + body
    zero_bytes = P.serialize_ciphertext_tensor([1], [zero])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + n_rows + 10 * world + (7 if scaling == "weak" else 0)
    procs = [ctx.Process(target=_worker_scal, args=(r, world, port, n_rows, m, p, pack(cts), s_bytes, zero_bytes, d, q, scaling)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=180)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert np.array_equal(got, pack(want_cts))


def test_row_partition():
    from cofhe_amd.shard import row_partition, rows_for_mode
    assert rows_for_mode(128, 8, 3, "weak") == (384, 128, 1024)
    assert rows_for_mode(128, 8, 3, "strong") == (48, 16, 128)
    assert rows_for_mode(10, 4, 3, "strong") == (8, 2, 10)
    assert row_partition(128, 8) == [(16 * i, 16 * i + 16) for i in range(8)]
    assert row_partition(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert row_partition(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
