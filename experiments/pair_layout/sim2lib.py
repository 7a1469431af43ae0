"""ctypes binding of libhostsim2_N*.so, built next to this file from sim2.cpp (lane-PAIR device arithmetic on host threads).
Part of the rejected pair-layout experiment: not collected by the product's test tiers (run: python -m pytest experiments/pair_layout)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
_libs = {}


def lib(n=17):
    if n not in _libs:
        so = os.path.join(HERE, "libhostsim2_N%d.so" % n)
        src = os.path.join(HERE, "sim2.cpp")
        deps = [src] + [os.path.join(ROOT, d, f) for d, f in (("experiments/pair_layout", "pair.hpp"), ("experiments/pair_layout", "qf2.hpp"), ("cofhe_amd/csrc", "mp.hpp"), ("cofhe_amd/csrc", "lane.hpp"), ("experiments/lehmer_variants", "lehmer_variants.hpp")) if
                        os.path.exists(os.path.join(ROOT, d, f))]
        if (not os.path.exists(so)) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-DSIM2_N=%d" % n, "-o", so, src])
        _libs[n] = C.CDLL(so)
    return _libs[n]


def to_limbs(x, n):
    assert 0 <= x < (1 << (32 * n)), "value does not fit"
    return np.frombuffer(x.to_bytes(4 * n, "little"), dtype="<u4").copy()


def pack(vals, n):
    return np.concatenate([to_limbs(v, n) for v in vals]) if len(vals) else np.zeros(0, dtype=np.uint32)


def unpack(arr, n):
    arr = np.asarray(arr, dtype=np.uint32).reshape(-1, n)
    return [int.from_bytes(r.astype("<u4").tobytes(), "little") for r in arr]


def P(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))
