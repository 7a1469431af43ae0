"""ctypes binding of tests/hostsim/libhostsim.so (device arithmetic run on host threads) and
the limb/record packing helpers shared by the CPU and GPU tests.  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_SO = os.path.join(HERE, "hostsim", "libhostsim.so")
_lib = None

REC_WORDS = 168


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "hostsim", "sim.cpp")
        deps = [src] + [os.path.join(ROOT, "cofhe_amd", "csrc", f) for f in
                        ("lane.hpp", "mp.hpp", "qf.hpp", "form_io.hpp", "layout.hpp")]
        if (not os.path.exists(_SO)) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
            # COFHE_SIM_FLAGS: extra defines for experiments on the device headers
            # the simulated workgroup has 8 groups = 64 threads = one wavefront (the kernels' 32 groups would be 256 threads)
            extra = ["-DCOFHE_WG_GROUPS=8"] + os.environ.get("COFHE_SIM_FLAGS", "").split()
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread"] + extra + ["-o", _SO, src])
        _lib = C.CDLL(_SO)
    return _lib


_SO32 = os.path.join(HERE, "hostsim", "libhostsim_wg32.so")
_lib32 = None


def lib_wg32():
    """the same simulator built with the kernels' workgroup geometry: 32 groups = 256 host threads = FOUR wavefronts, so that
    the serving wavefront serves groups of other wavefronts across the workgroup barrier as it does on the GPU"""
    global _lib32
    if _lib32 is None:
        src = os.path.join(HERE, "hostsim", "sim.cpp")
        deps = [src] + [os.path.join(ROOT, "cofhe_amd", "csrc", f) for f in
                        ("lane.hpp", "mp.hpp", "qf.hpp", "form_io.hpp", "layout.hpp")]
        if (not os.path.exists(_SO32)) or any(os.path.getmtime(d) > os.path.getmtime(_SO32) for d in deps):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-DCOFHE_WG_GROUPS=32", "-o", _SO32, src])
        _lib32 = C.CDLL(_SO32)
    return _lib32


def to_limbs(x: int, n: int) -> np.ndarray:
    assert 0 <= x < (1 << (32 * n)), "value does not fit"
    return np.frombuffer(x.to_bytes(4 * n, "little"), dtype="<u4").copy()


def from_limbs(a) -> int:
    return int.from_bytes(np.asarray(a, dtype="<u4").tobytes(), "little")


def pack(vals, n):
    return np.concatenate([to_limbs(v, n) for v in vals]) if len(vals) else np.zeros(0, dtype=np.uint32)


def unpack(arr, n):
    arr = np.asarray(arr, dtype=np.uint32).reshape(-1, n)
    return [from_limbs(r) for r in arr]


def P(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def form_record(a: int, b: int, c: int) -> np.ndarray:
    r = np.zeros(REC_WORDS, dtype=np.uint32)
    r[0:40] = to_limbs(a, 40)
    r[40:80] = to_limbs(abs(b), 40)
    r[80:160] = to_limbs(c, 80)
    r[160] = 1 if b < 0 else 0
    return r


def record_form(r):
    a = from_limbs(r[0:40])
    b = from_limbs(r[40:80])
    c = from_limbs(r[80:160])
    return a, (-b if r[160] else b), c


def compose(forms1, forms2, half_dbits, delta=None):
    """delta: the (negative) discriminant; when omitted it is derived from the first form"""
    n = len(forms1)
    if delta is None:
        a, b, c = forms1[0]
        delta = b * b - 4 * a * c
    ad = to_limbs(-delta, 80)
    f1 = np.concatenate([form_record(*f) for f in forms1])
    f2 = np.concatenate([form_record(*f) for f in forms2])
    out = np.zeros(n * REC_WORDS, dtype=np.uint32)
    lib().sim_compose(P(f1), P(f2), P(out), n, half_dbits, P(ad))
    return [record_form(out[i * REC_WORDS:(i + 1) * REC_WORDS]) for i in range(n)]


def compose_wg32(forms1, forms2, half_dbits, delta):
    """compose_wg on the 32-group (four-wavefront) build of the simulator; up to 32 pairs per call"""
    n = len(forms1)
    L = lib_wg32()
    assert 1 <= n <= L.sim_wg_groups()
    ad = to_limbs(-delta, 80)
    f1 = np.concatenate([form_record(*f) for f in forms1])
    f2 = np.concatenate([form_record(*f) for f in forms2])
    out = np.zeros(n * REC_WORDS, dtype=np.uint32)
    L.sim_compose_wg(P(f1), P(f2), P(out), n, half_dbits, P(ad))
    return [record_form(out[i * REC_WORDS:(i + 1) * REC_WORDS]) for i in range(n)], L.sim_status()


def compose_wg(forms1, forms2, half_dbits, delta):
    """the kernels' form of the composition: a simulated workgroup whose remainder sequences are served by one
    wavefront (mp.hpp: euclid_run_wg); up to lib().sim_wg_groups() pairs per call"""
    n = len(forms1)
    assert 1 <= n <= lib().sim_wg_groups()
    ad = to_limbs(-delta, 80)
    f1 = np.concatenate([form_record(*f) for f in forms1])
    f2 = np.concatenate([form_record(*f) for f in forms2])
    out = np.zeros(n * REC_WORDS, dtype=np.uint32)
    lib().sim_compose_wg(P(f1), P(f2), P(out), n, half_dbits, P(ad))
    return [record_form(out[i * REC_WORDS:(i + 1) * REC_WORDS]) for i in range(n)]


def power(forms, exps, delta):
    """forms[i] ^ exps[i] through the device ladder (qf_pow) on the host simulator"""
    n = len(forms)
    half = ((-delta).bit_length() + 1) // 2
    ad = to_limbs(-delta, 80)
    base = np.concatenate([form_record(*f) for f in forms])
    ex = np.zeros((n, 32), dtype=np.uint32)
    for i, e in enumerate(exps):
        ex[i, :31] = to_limbs(abs(e), 31)
        ex[i, 31] = 1 if e < 0 else 0
    b0 = (-delta) & 1 if False else (delta & 1)
    one = form_record(1, delta & 1, ((delta & 1) - delta) // 4)
    out = np.zeros(n * REC_WORDS, dtype=np.uint32)
    ex = ex.reshape(-1)
    lib().sim_pow(P(base), P(ex), P(one), P(out), n, half, P(ad))
    return [record_form(out[i * REC_WORDS:(i + 1) * REC_WORDS]) for i in range(n)]
