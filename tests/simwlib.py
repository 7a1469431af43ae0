"""ctypes binding of tests/hostsim/libhostsimw.so: the wavefront-wide layout of the latency kernels (cofhe_amd/csrc/wide.hpp,
qfw.hpp) emulated on the host -- lane values as 64-element vectors, cross-lane primitives as loops, the device source
otherwise unchanged.  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_SO = os.path.join(HERE, "hostsim", "libhostsimw.so")
_lib = None
CAP_LIMBS = 128
M = 1 << (32 * CAP_LIMBS)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "hostsim", "simw.cpp")
        deps = [src] + [os.path.join(ROOT, "cofhe_amd", "csrc", f) for f in ("wide.hpp", "qfw.hpp", "lane.hpp", "mp.hpp", "qf.hpp", "layout.hpp")]
        if (not os.path.exists(_SO)) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", _SO, src])
        _lib = C.CDLL(_SO)
    return _lib


def pack(vals):
    return np.concatenate([np.frombuffer(int(v).to_bytes(4 * CAP_LIMBS, "little"), dtype="<u4") for v in vals]).astype(np.uint32)


def unpack(arr):
    arr = np.asarray(arr, dtype=np.uint32).reshape(-1, CAP_LIMBS)
    return [int.from_bytes(r.tobytes(), "little") for r in arr]


def P(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))
