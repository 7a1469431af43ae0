"""GPU: the C++ host interface (cofhe_amd/host/hip_cryptosystem.hpp) driven by the harness that
mirrors the reference's benchmarks/local.cpp; its serialised result is checked against the oracle."""
import json
import os
import subprocess
import sys

import pytest

import oracle_lib as O
from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()          # no-op when libcofhe_hip.so and local_bench are up to date


def test_local_bench_matadd_chain_matches_oracle(tmp_path):
    exe = os.path.join(ROOT, "cofhe_amd", "host", "local_bench")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    r = subprocess.run([exe, "ciphertext_matadd", "4", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "agree: yes" in r.stdout
    delta = -int(open(tmp_path / "local_bench_absdelta.txt").read().strip())
    ct1 = open(tmp_path / "local_bench_matadd_ct1.bin", "rb").read()
    ct2 = open(tmp_path / "local_bench_matadd_ct2.bin", "rb").read()
    out = open(tmp_path / "local_bench_matadd_out.bin", "rb").read()
    assert O.check_tensor(delta, ct1) == 1 and O.check_tensor(delta, out) == 1
    _, want = O.time_matadd_chain(delta, ct1, ct2, 50, want_out=True)
    assert out == want


def test_local_bench_scal_matmul_chain_matches_oracle(tmp_path):
    """the C++ 2-D path (exponent packing, upload, Enc(0) handling, chained products on device blocks, serialiser):
    the reference protocol of 1 + 49 chained products at a small shape, final tensor byte-compared with the oracle"""
    exe = os.path.join(ROOT, "cofhe_amd", "host", "local_bench")
    r = subprocess.run([exe, "scal_matmul", "2", "4", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    assert "chain: 50" in r.stdout
    delta = -int(open(tmp_path / "local_bench_absdelta.txt").read().strip())
    s = open(tmp_path / "local_bench_scal_s.bin", "rb").read()
    cts = open(tmp_path / "local_bench_scal_cts.bin", "rb").read()
    zero = open(tmp_path / "local_bench_scal_zero.bin", "rb").read()
    out = open(tmp_path / "local_bench_scal_out.bin", "rb").read()
    want = cts
    for _ in range(50):
        want = O.scal_2d(delta, s, want, zero)
    assert out == want


def test_local_bench_encrypt_decrypt_roundtrip(tmp_path):
    """keygen -> encrypt_tensor -> decrypt_tensor on the GPU returns the plaintexts, and
    Dec(Enc a + Enc b) = a + b, Dec(3 Enc a) = 3a, Dec(-Enc a) = -a"""
    exe = os.path.join(ROOT, "cofhe_amd", "host", "local_bench")
    r = subprocess.run([exe, "encrypt_decrypt", "4", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checks: ok" in r.stdout


def test_local_bench_zero_degree_branch_matches_oracle(tmp_path):
    """0-D tensors take the scalar forms (tensor_ops.inl:199-202, 275-278): randomised like the reference's (checked
    through decryption inside the harness), and with re-randomisation off byte-equal to the oracle's composition / power"""
    exe = os.path.join(ROOT, "cofhe_amd", "host", "local_bench")
    r = subprocess.run([exe, "encrypt_decrypt", "2", "2"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checks: ok" in r.stdout
    delta = -int(open(tmp_path / "local_bench_absdelta.txt").read().strip())
    a = open(tmp_path / "local_bench_scalar_a.bin", "rb").read()
    b = open(tmp_path / "local_bench_scalar_b.bin", "rb").read()
    assert open(tmp_path / "local_bench_scalar_sum.bin", "rb").read() == O.add(delta, a, b)
    import struct
    three = struct.pack("<II", 1, 1) + struct.pack("<Q", 0) + b"\x03"          # plaintext tensor [3] (cpu_cryptosystem.inl:229-270)
    assert open(tmp_path / "local_bench_scalar_tri.bin", "rb").read() == O.scal_1d(delta, three, a)


def test_local_bench_formats(tmp_path):
    """text formats of single values and the binary plaintext-tensor format: round trips inside the harness, the
    files against an independent packing here"""
    import struct
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyref as P
    exe = os.path.join(ROOT, "cofhe_amd", "host", "local_bench")
    r = subprocess.run([exe, "formats"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "formats: ok" in r.stdout
    _, cts = P.deserialize_ciphertext_tensor(open(tmp_path / "local_bench_fmt_ct.bin", "rb").read())
    (c1, c2), = cts
    txt = open(tmp_path / "local_bench_fmt_ct.txt").read()
    assert txt == "%d %d %d %d %d %d" % (c1.a, c1.b, c1.c, c2.a, c2.b, c2.c)          # cpu_cryptosystem.inl:199-204
    vals = [int(v) for v in open(tmp_path / "local_bench_fmt_pt.txt").read().split()]
    assert vals == [0, 1, (1 << 128) - 1, 255, (1 << 128) - 65536, 123456]
    offs, body = [], b""
    for v in vals:
        offs.append(len(body) | ((1 << 63) if v <= 0 else 0))
        body += v.to_bytes(max(v.bit_length(), 1) // 8 + 1, "little")
    want = struct.pack("<III", 2, 2, 3) + b"".join(struct.pack("<Q", o) for o in offs) + body
    assert open(tmp_path / "local_bench_fmt_pt.bin", "rb").read() == want


def test_local_bench_two_threads_one_cryptosystem(tmp_path):
    exe = os.path.join(ROOT, "cofhe_amd", "host", "local_bench")
    r = subprocess.run([exe, "threads", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rounds each: ok" in r.stdout


def test_local_bench_threshold_decrypt(tmp_path):
    """keygen(sk, 2, 3) -> part_decrypt_tensor per party -> combine_part_decryption_results_tensor
    returns the plaintexts; 3-of-3 as well (two inverted partial decryptions in the product)"""
    exe = os.path.join(ROOT, "cofhe_amd", "host", "local_bench")
    for args in (["4", "4", "2", "3"], ["2", "2", "3", "3"]):
        r = subprocess.run([exe, "threshold"] + args, cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ": ok" in r.stdout


def test_local_bench_ciphertext_matmul_beaver(tmp_path):
    """ciphertext x ciphertext matrix product through the Beaver-triplet protocol with the in-process
    client (smpc_local.hpp): decrypts to the integer matrix product, with secret-key decryption and
    with 2-of-3 threshold decryption inside the protocol"""
    exe = os.path.join(ROOT, "cofhe_amd", "host", "local_bench")
    for args in (["2", "3", "2"], ["2", "2", "2", "2", "3"]):
        r = subprocess.run([exe, "ciphertext_matmul"] + args, cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ": ok" in r.stdout


def test_product_make_plaintext_against_golden(tmp_path):
    """HIPCryptoSystem::make_plaintext / get_float_from_plaintext (the PRODUCT functions, not the oracle's) on the
    committed cases: fractional values (truncation), negative fractions (the reference's mpf sum), magnitudes from
    2^-20 to the float range"""
    import struct
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "plaintext_k128.json")))["cases"]
    exe = os.path.join(ROOT, "cofhe_amd", "host", "local_bench")
    with open(tmp_path / "in.txt", "w") as fh:
        for c in cases:
            fh.write("%08x\n" % struct.unpack("<I", struct.pack("<f", c["x"]))[0])
    r = subprocess.run([exe, "plaintexts", "in.txt", "out.txt"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = open(tmp_path / "out.txt").read().split("\n")[:-1]
    assert len(lines) == len(cases) >= 28
    for c, ln in zip(cases, lines):
        pt, back = ln.split()
        assert int(pt) == int(c["pt"], 16), c
        got = struct.unpack("<f", struct.pack("<I", int(back, 16)))[0]
        want = struct.unpack("<f", struct.pack("<f", c["back"]))[0]
        assert got == want, (c, got)
