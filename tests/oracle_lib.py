"""ctypes binding of oracle/libcofhe_oracle.so -- TEST INFRASTRUCTURE (checker only)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libcofhe_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(_SO)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_time_matadd_chain.restype = C.c_double
        L.oracle_time_scal_2d.restype = C.c_double
        L.oracle_get_float_from_plaintext.restype = C.c_float
        L.oracle_get_float_from_plaintext.argtypes = [C.c_char_p, C.c_uint32]
        L.oracle_make_plaintext.argtypes = [C.c_float, C.c_uint32, C.c_char_p, C.c_size_t]
        # the checker runs on the CPU share this job really has (cgroup quota / affinity), not on one
        # thread per host core
        n = L.oracle_max_threads()
        try:
            n = min(n, len(os.sched_getaffinity(0)))
            with open("/sys/fs/cgroup/cpu.max") as fh:
                q = fh.read().split()
            if q[0] != "max":
                n = min(n, max(1, round(int(q[0]) / int(q[1]))))
        except (AttributeError, OSError, ValueError, IndexError):
            pass
        L.oracle_set_threads(C.c_int(max(1, int(n))))
        _lib = L
    return _lib


def _delta_bytes(delta: int) -> bytes:
    assert delta < 0
    m = -delta
    return m.to_bytes((m.bit_length() + 7) // 8, "little")


def _call(fn, *bufs, extra=()):
    L = lib()
    out = C.POINTER(C.c_uint8)()
    outlen = C.c_size_t()
    args = []
    for b in bufs:
        args += [C.c_char_p(b), C.c_size_t(len(b))]
    args += list(extra)
    rc = fn(*args, C.byref(out), C.byref(outlen))
    if rc != 0:
        raise ValueError(L.oracle_last_error().decode())
    data = C.string_at(out, outlen.value)
    L.oracle_free(out)
    return data


def add(delta, t1, t2, mode=0):
    return _call(lib().oracle_add_ciphertext_tensors, _delta_bytes(delta), t1, t2, extra=(C.c_int(mode),))


def scal_1d(delta, s, cts):
    return _call(lib().oracle_scal_ciphertext_tensors_1d, _delta_bytes(delta), s, cts)


def scal_2d(delta, s, cts, zero):
    return _call(lib().oracle_scal_ciphertext_tensors_2d, _delta_bytes(delta), s, cts, zero)


def qfi_nupow(delta, s, base_ct):
    return _call(lib().oracle_qfi_nupow, _delta_bytes(delta), s, base_ct)


def check_tensor(delta, t):
    d = _delta_bytes(delta)
    return lib().oracle_check_tensor(C.c_char_p(d), C.c_size_t(len(d)), C.c_char_p(t), C.c_size_t(len(t)))


def make_plaintext(x, k):
    buf = C.create_string_buffer(4096)
    assert lib().oracle_make_plaintext(C.c_float(x), C.c_uint32(k), buf, 4096) == 0
    return int(buf.value.decode())


def get_float(z, k):
    return float(lib().oracle_get_float_from_plaintext(str(z).encode(), C.c_uint32(k)))


def time_matadd_chain(delta, t1, t2, chain, threads=None, want_out=False):
    L = lib()
    if threads:
        L.oracle_set_threads(int(threads))
    d = _delta_bytes(delta)
    out = C.POINTER(C.c_uint8)()
    outlen = C.c_size_t()
    sec = L.oracle_time_matadd_chain(C.c_char_p(d), C.c_size_t(len(d)), C.c_char_p(t1), C.c_size_t(len(t1)),
                                     C.c_char_p(t2), C.c_size_t(len(t2)), C.c_int(chain),
                                     C.byref(out) if want_out else None, C.byref(outlen) if want_out else None)
    if sec < 0:
        raise ValueError(L.oracle_last_error().decode())
    if want_out:
        data = C.string_at(out, outlen.value)
        L.oracle_free(out)
        return sec, data
    return sec


def time_scal_2d(delta, s, cts, zero, threads=None):
    L = lib()
    if threads:
        L.oracle_set_threads(int(threads))
    d = _delta_bytes(delta)
    sec = L.oracle_time_scal_2d(C.c_char_p(d), C.c_size_t(len(d)), C.c_char_p(s), C.c_size_t(len(s)),
                                C.c_char_p(cts), C.c_size_t(len(cts)), C.c_char_p(zero), C.c_size_t(len(zero)))
    if sec < 0:
        raise ValueError(L.oracle_last_error().decode())
    return sec


def max_threads():
    return lib().oracle_max_threads()
