"""bench.py's multi-GPU plumbing on the CPU tier: the launcher (`--gpus N` without a launcher's environment starts N
fresh ranks; nothing in the parent touches the GPU), the host-side rendezvous and the contract's timing protocol driven
by two real processes over gloo, and the collective plan of the library (cofhe_hip_gather_plan, host only) against the
row partition bench.py and the tests use."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def test_launch_dry_run_prints_two_rank_environments():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "7", "--launch-dry-run"], env=_clean_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["mode"] == "spawn 2 children"
    ranks = plan["ranks"]
    assert [e["RANK"] for e in ranks] == ["0", "1"] and [e["LOCAL_RANK"] for e in ranks] == ["0", "1"]
    assert all(e["WORLD_SIZE"] == "2" and e["MASTER_ADDR"] == "127.0.0.1" for e in ranks)
    assert len({e["MASTER_PORT"] for e in ranks}) == 1 and 1024 < int(ranks[0]["MASTER_PORT"]) < 65536
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in ranks)
    # the children run this same file with the same arguments (and become ranks because WORLD_SIZE is set)
    assert plan["command"][1] == BENCH and plan["command"][2:] == ["--gpus", "2", "--steps", "7"]


def test_dry_run_under_a_launcher_reports_the_rank():
    env = dict(_clean_env(), WORLD_SIZE="8", RANK="3", LOCAL_RANK="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--launch-dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["mode"].startswith("external launcher") and plan["mode"].endswith("rank 3")
    assert len(plan["ranks"]) == 8 and plan["ranks"][0]["MASTER_PORT"] == "29999"


def test_single_rank_runs_in_process_and_parent_makes_no_gpu_call():
    plan = subprocess.run([sys.executable, BENCH, "--launch-dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert json.loads(plan.stdout.strip().splitlines()[-1])["mode"] == "single rank, in process"
    # the launcher's code path imports nothing that could initialise a GPU runtime
    code = ("import sys; sys.argv=['bench.py','--gpus','2','--launch-dry-run']; import bench; bench.main(); "
            "bad=[m for m in ('torch','cofhe_amd','cofhe_amd.engine') if m in sys.modules]; assert not bad, bad")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]


def test_gpus_2_without_gpus_fails_loudly_in_the_children():
    """no GPU in this container: both children must die with the engine's error (there is no CPU path) and the parent
    must report it, not hang"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "no such HIP device" in r.stderr or "hipGetDeviceCount" in r.stderr or "HIP" in r.stderr


def _rank_worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    assert bench.rank_identity() == (world, rank, rank)
    rdv = bench.Rendezvous(world, rank)
    # the RCCL id travels as 128 opaque bytes from rank 0
    uid = bytes(range(128)) if rank == 0 else b""
    got = rdv.broadcast_bytes(uid, 128)
    log = []
    data = np.full(4, rank + 1, dtype=np.int64)
    state = {"x": data.copy(), "gathers": 0, "fences": 0}

    def step(i):
        log.append(i)
        state["x"] = state["x"] + 1
        if rank == 1:
            import time
            time.sleep(0.01)              # the slow rank sets the time everybody reports

    def gather():
        state["gathers"] += 1

    def fence():
        state["fences"] += 1

    elapsed = bench.timed_region(5, 2, step, gather, fence, rdv)
    mx = rdv.max_float(rank * 10.0)
    rdv.close()
    q.put((rank, got, log, state["gathers"], state["fences"], elapsed, mx, state["x"].tolist()))


def test_rendezvous_and_timing_protocol_two_ranks_gloo():
    port = bench.free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, got, log, gathers, fences, elapsed, mx, x in res:
        assert got == bytes(range(128))
        assert log == [0, 1, 0, 1, 2, 3, 4]           # 2 warm-up steps, then exactly 5 timed ones
        assert gathers == 2 and fences == 2           # one untimed (warms the communicator), one inside the timed region
        assert mx == 10.0
        assert x == [rank + 1 + 7] * 4
    # MAX over ranks: both report the slow rank's time (>= 5 x 10 ms)
    assert res[0][5] == res[1][5] and res[0][5] >= 0.05


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_gather_plan_matches_row_partition(world):
    """the library's collective plan (what cofhe_hip_all_gather_rows will send where, and by which RCCL call) against the
    Python row partition, ragged and even"""
    import cofhe_amd
    from cofhe_amd import shard
    import ctypes as C
    L = cofhe_amd.load_library()
    L.cofhe_hip_shard_rows.restype = None
    for n_rows in (0, 1, 2, 7, 8, 128, 129, 1000, 1024):
        for row_bytes in (1344, 128 * 1344):
            plan, uniform = cofhe_amd.gather_plan(n_rows, row_bytes, world)
            parts = shard.row_partition(n_rows, world)
            assert plan == [(a * row_bytes, (b - a) * row_bytes) for a, b in parts]
            assert uniform == (n_rows % world == 0)
            # blocks tile the assembled tensor exactly
            assert sum(c for _, c in plan) == n_rows * row_bytes
            assert all(plan[r][0] + plan[r][1] == (plan[r + 1][0] if r + 1 < world else n_rows * row_bytes) for r in range(world))
            for r in range(world):
                r0, nl = C.c_uint64(), C.c_uint64()
                L.cofhe_hip_shard_rows(C.c_uint64(n_rows), C.c_uint32(world), C.c_uint32(r), C.byref(r0), C.byref(nl))
                assert (r0.value, r0.value + nl.value) == parts[r]
    with pytest.raises(cofhe_amd.CofheHipError):
        cofhe_amd.gather_plan(1 << 62, 1 << 10, world)      # byte count would wrap


def _rank_script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text("import os, sys, time\nrank = int(os.environ['RANK'])\n" + body)
    return [sys.executable, str(p)]


def test_supervisor_ends_the_run_when_a_rank_dies_before_the_rendezvous(tmp_path, capfd):
    """rank 1 exits 7 before it ever reaches the rendezvous while rank 0 waits in it (here: sleeps for ten minutes): the
    parent names rank 1, terminates rank 0 and returns non-zero within seconds -- round 3's launcher blocked in rank 0's
    communicate() until the driver's limit"""
    import time
    cmd = _rank_script(tmp_path, "sys.stderr.write('rank %d up\\n' % rank)\n"
                                 "if rank == 1:\n    sys.stderr.write('rank 1: no GPU of mine\\n'); sys.exit(7)\n"
                                 "time.sleep(600)\nprint('{\"never\": 1}')\n")
    t0 = time.time()
    rc = bench.launch_ranks(2, [], cmd=cmd)
    took = time.time() - t0
    out, err = capfd.readouterr()
    assert rc == 7 and took < 20.0, (rc, took)
    assert "rank 1 of 2 failed (exit code 7)" in err and "rank 1: no GPU of mine" in err
    assert "never" not in out


def test_supervisor_reports_a_rank_killed_by_a_signal_even_when_the_others_exit_0(tmp_path, capfd):
    """rank 0 dies by SIGABRT (what a GPU fault does to a process) while rank 1 exits 0: round 3 computed
    max(rc, abs(...)) over the OTHER ranks only and could return 0 with no JSON line"""
    cmd = _rank_script(tmp_path, "if rank == 0:\n    time.sleep(0.5); os.abort()\nsys.exit(0)\n")
    rc = bench.launch_ranks(2, [], cmd=cmd)
    out, err = capfd.readouterr()
    assert rc == 6 and "rank 0 of 2 failed (signal 6)" in err


def test_supervisor_relays_rank_0_when_all_ranks_succeed(tmp_path, capfd):
    cmd = _rank_script(tmp_path, "time.sleep(0.2 * rank)\nif rank == 0:\n    print('{\"value\": 42}')\n"
                                 "sys.stderr.write('note from rank %d\\n' % rank)\n")
    rc = bench.launch_ranks(3, [], cmd=cmd)
    out, err = capfd.readouterr()
    assert rc == 0 and out.strip() == '{"value": 42}'
    assert all("note from rank %d" % r in err for r in range(3))
