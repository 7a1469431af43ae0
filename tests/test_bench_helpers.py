"""bench.py's bookkeeping that needs no GPU: the counter files under profiles/ are tied to the kernel sources, a file
measured on another build is refused, and the row bookkeeping of the two scaling modes."""
import json
import os
import sys

import pytest
from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_committed_counters_match_the_current_kernel_sources():
    """profiles/r*/traffic.json and valu.json that bench.py would report were measured on THIS kernel (same source hash)"""
    h = bench.kernel_code_hash()
    for name, key in (("traffic.json", "traffic_bytes_per_launch"), ("valu.json", "valu_wave_insts_per_launch")):
        tj, src = bench.committed_counter_file(name, 32768, key)
        if tj is None:      # kernels edited since the last profile: bench.py reports null until tools/gpu_traffic.sh has run
            pytest.skip("no current %s under profiles/ (%s): re-run tools/gpu_traffic.sh" % (name, src))
        assert tj["kernel_code_hash"] == h and tj[key] > 0


def test_stale_counter_file_is_refused(tmp_path, monkeypatch):
    d = tmp_path / "profiles" / "r99_x"
    d.mkdir(parents=True)
    (d / "traffic.json").write_text(json.dumps({"records_per_launch": 32768, "traffic_bytes_per_launch": 1, "kernel_code_hash": "0000000000000000"}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_code_hash", lambda: "1111111111111111")
    tj, src = bench.committed_counter_file("traffic.json", 32768, "traffic_bytes_per_launch")
    assert tj is None and src.startswith("stale")
    (d / "traffic.json").write_text(json.dumps({"records_per_launch": 32768, "traffic_bytes_per_launch": 7, "kernel_code_hash": "1111111111111111"}))
    tj, src = bench.committed_counter_file("traffic.json", 32768, "traffic_bytes_per_launch")
    assert tj["traffic_bytes_per_launch"] == 7 and src.endswith("traffic.json")
    tj, _ = bench.committed_counter_file("traffic.json", 999, "traffic_bytes_per_launch")        # other launch size
    assert tj is None


def test_scaling_modes_bookkeeping():
    from cofhe_amd import shard
    assert shard.rows_for_mode(128, 8, 0, "weak") == (0, 128, 1024)
    assert [shard.rows_for_mode(128, 8, r, "strong")[1] for r in range(8)] == [16] * 8
    assert sum(shard.rows_for_mode(1000, 8, r, "strong")[1] for r in range(8)) == 1000
