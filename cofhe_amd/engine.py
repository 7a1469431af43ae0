"""ctypes loader of libcofhe_hip.so (C ABI: include/cofhe_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

REC_WORDS = 168
EXP_REC_WORDS = 32


class CofheHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cofhe_hip error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


def lib_path():
    return os.path.join(_HERE, "libcofhe_hip.so")


def load_library(path=None):
    """Loads the HIP extension; raises when it has not been built (no fallback exists).  `path`: another build of the
    SAME extension (tools/build_variant.sh; bench.py --lib), honoured by the first call only."""
    global _LIB
    if _LIB is None:
        p = path or lib_path()
        if not os.path.exists(p):
            raise FileNotFoundError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
        # A process that also uses PyTorch holds two HIP runtimes (the wheel bundles its own libamdhip64.so, this
        # library binds /opt/rocm's libamdhip64.so.7), and the pair only works when PyTorch's has initialised the
        # device first.  If the host application has already imported torch, do that here.
        import sys
        torch = sys.modules.get("torch")
        if torch is not None:
            try:
                if torch.cuda.is_available() and not torch.cuda.is_initialized():
                    torch.cuda.init()
            except Exception:       # a CPU-only torch build: nothing to order
                pass
        L = C.CDLL(p)
        L.cofhe_hip_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def _chk(rc):
    if rc != 0:
        raise CofheHipError(rc, load_library().cofhe_hip_last_error().decode())


def _absdelta(delta):
    assert delta < 0
    m = -delta
    return m.to_bytes((m.bit_length() + 7) // 8, "little")


class Engine:
    """One GPU + one discriminant.  All tensors cross as bytes in the reference's binary
    formats (or as device pointers to records for the resident-tensor entry points)."""

    def __init__(self, delta: int, device: int = 0):
        self.L = load_library()
        self.ctx = C.c_void_p()
        d = _absdelta(delta)
        _chk(self.L.cofhe_hip_ctx_create(C.c_int(device), C.c_char_p(d), C.c_size_t(len(d)), C.byref(self.ctx)))
        self.delta = delta
        self.device = device

    def close(self):
        if self.ctx:
            self.L.cofhe_hip_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- whole ops on host byte buffers -------------------------------------------------
    def _take(self, out, outlen):
        data = C.string_at(out, outlen.value)
        self.L.cofhe_hip_host_free(out)
        return data

    def add_ciphertext_tensors(self, t1: bytes, t2: bytes) -> bytes:
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        _chk(self.L.cofhe_hip_add_ciphertext_tensors_bytes(self.ctx, C.c_char_p(t1), C.c_size_t(len(t1)),
                                                           C.c_char_p(t2), C.c_size_t(len(t2)), C.byref(out), C.byref(n)))
        return self._take(out, n)

    def scal_ciphertext_tensors(self, s: bytes, cts: bytes, zero: bytes = None) -> bytes:
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        z = C.c_char_p(zero) if zero is not None else None
        _chk(self.L.cofhe_hip_scal_ciphertext_tensors_bytes(self.ctx, C.c_char_p(s), C.c_size_t(len(s)), C.c_char_p(cts),
                                                            C.c_size_t(len(cts)), z, C.c_size_t(len(zero) if zero else 0),
                                                            C.byref(out), C.byref(n)))
        return self._take(out, n)

    # ---- format conversion (host) ----------------------------------------------------------
    def bytes_to_records(self, t: bytes):
        import numpy as np
        ndim = C.c_uint32()
        shape = (C.c_uint32 * 8)()
        recs = C.POINTER(C.c_uint32)()
        n = C.c_uint64()
        _chk(self.L.cofhe_hip_bytes_to_records(C.c_char_p(t), C.c_size_t(len(t)), C.byref(ndim), shape, C.byref(recs), C.byref(n)))
        arr = np.ctypeslib.as_array(recs, shape=(n.value * REC_WORDS,)).copy() if n.value else np.zeros(0, dtype=np.uint32)
        self.L.cofhe_hip_host_free(recs)
        return list(shape[:ndim.value]), arr

    def records_to_bytes(self, recs, shape) -> bytes:
        import numpy as np
        recs = np.ascontiguousarray(recs, dtype=np.uint32)
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        sh = (C.c_uint32 * len(shape))(*shape)
        _chk(self.L.cofhe_hip_records_to_bytes(recs.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(recs.size // REC_WORDS),
                                               C.c_uint32(len(shape)), sh, C.byref(out), C.byref(n)))
        return self._take(out, n)

    def pdr_bytes_to_records(self, t: bytes):
        """partial-decryption tensor (one form per element) -> records"""
        import numpy as np
        ndim = C.c_uint32()
        shape = (C.c_uint32 * 8)()
        recs = C.POINTER(C.c_uint32)()
        n = C.c_uint64()
        _chk(self.L.cofhe_hip_pdr_bytes_to_records(C.c_char_p(t), C.c_size_t(len(t)), C.byref(ndim), shape, C.byref(recs), C.byref(n)))
        arr = np.ctypeslib.as_array(recs, shape=(n.value * REC_WORDS,)).copy() if n.value else np.zeros(0, dtype=np.uint32)
        self.L.cofhe_hip_host_free(recs)
        return list(shape[:ndim.value]), arr

    def pdr_records_to_bytes(self, recs, shape) -> bytes:
        import numpy as np
        recs = np.ascontiguousarray(recs, dtype=np.uint32)
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        sh = (C.c_uint32 * len(shape))(*shape)
        _chk(self.L.cofhe_hip_pdr_records_to_bytes(recs.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(recs.size // REC_WORDS),
                                                   C.c_uint32(len(shape)), sh, C.byref(out), C.byref(n)))
        return self._take(out, n)

    # ---- format conversion on the GPU (device pointers) ------------------------------------
    def unpack_tensor_device(self, d_bytes, nbytes, kind, d_records, capacity_records, stream=0):
        """kind 2 ciphertexts / 1 partial decryptions / 0 plaintexts; returns (shape, n_records)"""
        ndim = C.c_uint32()
        shape = (C.c_uint32 * 8)()
        n = C.c_uint64()
        _chk(self.L.cofhe_hip_unpack_tensor_device(self.ctx, C.c_void_p(d_bytes), C.c_size_t(nbytes), C.c_int(kind),
                                                   C.c_void_p(d_records), C.c_uint64(capacity_records), C.byref(ndim), shape,
                                                   C.byref(n), C.c_void_p(stream)))
        return list(shape[:ndim.value]), n.value

    def pack_tensor_device(self, d_records, n_records, kind, shape, d_bytes, capacity, stream=0) -> int:
        """returns the serialised length written to d_bytes"""
        sh = (C.c_uint32 * max(len(shape), 1))(*shape)
        n = C.c_size_t()
        _chk(self.L.cofhe_hip_pack_tensor_device(self.ctx, C.c_void_p(d_records), C.c_uint64(n_records), C.c_int(kind),
                                                 C.c_uint32(len(shape)), sh, C.c_void_p(d_bytes), C.c_size_t(capacity),
                                                 C.byref(n), C.c_void_p(stream)))
        return n.value

    def packed_size_bound(self, n_records, kind, ndim) -> int:
        self.L.cofhe_hip_packed_size_bound.restype = C.c_size_t
        return self.L.cofhe_hip_packed_size_bound(C.c_uint64(n_records), C.c_int(kind), C.c_uint32(ndim))

    def bytes_to_exponents(self, t: bytes):
        import numpy as np
        ndim = C.c_uint32()
        shape = (C.c_uint32 * 8)()
        ex = C.POINTER(C.c_uint32)()
        n = C.c_uint64()
        _chk(self.L.cofhe_hip_bytes_to_exponents(C.c_char_p(t), C.c_size_t(len(t)), C.byref(ndim), shape, C.byref(ex), C.byref(n)))
        arr = np.ctypeslib.as_array(ex, shape=(n.value * EXP_REC_WORDS,)).copy() if n.value else np.zeros(0, dtype=np.uint32)
        self.L.cofhe_hip_host_free(ex)
        return list(shape[:ndim.value]), arr

    # ---- kernels on device-resident records (pointers are ints, e.g. torch .data_ptr()) -----
    def compose_records(self, d_a, d_b, d_out, n_records, stream=0):
        _chk(self.L.cofhe_hip_compose_records(self.ctx, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_out),
                                              C.c_uint64(n_records), C.c_void_p(stream)))

    def compose_wide_records(self, d_a, d_b, d_out, n_records, reps=1, count_fallbacks=False, stream=0):
        """one composition per wavefront in the wavefront-wide layout (the latency kernels' composition); returns the number
        of pairs that took the 8-lane fallback when count_fallbacks"""
        fb = C.c_uint32(0)
        _chk(self.L.cofhe_hip_compose_wide_records(self.ctx, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_out), C.c_uint64(n_records),
                                                   C.c_uint32(reps), C.byref(fb) if count_fallbacks else None, C.c_void_p(stream)))
        return int(fb.value)

    def add_ciphertext_records(self, d_a, d_b, d_out, n_ciphertexts, stream=0):
        """ciphertext-level add: folds the shared c1 of encrypt_tensor-made operands (n + 1 compositions, not 2 n)"""
        _chk(self.L.cofhe_hip_add_ciphertext_records(self.ctx, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_out),
                                                     C.c_uint64(n_ciphertexts), C.c_void_p(stream)))

    def pow_records(self, d_base, d_exp, d_out, n_ciphertexts, stream=0):
        _chk(self.L.cofhe_hip_pow_records(self.ctx, C.c_void_p(d_base), C.c_void_p(d_exp), C.c_void_p(d_out),
                                          C.c_uint64(n_ciphertexts), C.c_void_p(stream)))

    def scal_matmul_records(self, d_cts, d_exp, d_zero, d_out, n, m, p, stream=0):
        _chk(self.L.cofhe_hip_scal_matmul_records(self.ctx, C.c_void_p(d_cts), C.c_void_p(d_exp), C.c_void_p(d_zero),
                                                  C.c_void_p(d_out), C.c_uint32(n), C.c_uint32(m), C.c_uint32(p),
                                                  C.c_void_p(stream)))

    def decrypt_records(self, d_cts, d_sk, f_record, d_out, n_ciphertexts, kbits, stream=0):
        """f_record: host numpy uint32[168]; d_out: n * (ceil(k/32) + 1) words"""
        import numpy as np
        f = np.ascontiguousarray(f_record, dtype=np.uint32)
        _chk(self.L.cofhe_hip_decrypt_records(self.ctx, C.c_void_p(d_cts), C.c_void_p(d_sk), f.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              C.c_void_p(d_out), C.c_uint64(n_ciphertexts), C.c_uint32(kbits), C.c_void_p(stream)))

    def accumulate_records(self, d_x, d_zero, d_out, n, m, p, stream=0):
        """out[i,k] = zero o prod_j x[i,j,k] (x: n*m*p ciphertexts, out: n*p)"""
        _chk(self.L.cofhe_hip_accumulate_records(self.ctx, C.c_void_p(d_x), C.c_void_p(d_zero), C.c_void_p(d_out),
                                                 C.c_uint32(n), C.c_uint32(m), C.c_uint32(p), C.c_void_p(stream)))

    def pow_form_records(self, d_base, d_exp, d_out, n_forms, stream=0):
        _chk(self.L.cofhe_hip_pow_form_records(self.ctx, C.c_void_p(d_base), C.c_void_p(d_exp), C.c_void_p(d_out),
                                               C.c_uint64(n_forms), C.c_void_p(stream)))

    def pow_fixed_base_record(self, base_record, exp_record, d_out, stream=0):
        """base_record (168 u32) and exp_record (32 u32) are host numpy arrays; d_out one device record.  The
        context caches the table base^(2^j) of the last few bases."""
        import numpy as np
        b = np.ascontiguousarray(base_record, dtype=np.uint32)
        e = np.ascontiguousarray(exp_record, dtype=np.uint32)
        assert b.size == 168 and e.size == 32
        _chk(self.L.cofhe_hip_pow_fixed_base_record(self.ctx, b.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                                    C.c_void_p(d_out), C.c_void_p(stream)))

    def pow_fixed_base_records(self, base_records, exp_records, d_out, stream=0):
        """n <= 4 fixed-base powers in one product tree; base_records n x 168 u32, exp_records n x 32 u32 (host)"""
        import numpy as np
        b = np.ascontiguousarray(base_records, dtype=np.uint32).reshape(-1)
        e = np.ascontiguousarray(exp_records, dtype=np.uint32).reshape(-1)
        n = b.size // 168
        assert b.size == n * 168 and e.size == n * 32
        _chk(self.L.cofhe_hip_pow_fixed_base_records(self.ctx, C.c_uint32(n), b.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                                     C.c_void_p(d_out), C.c_void_p(stream)))

    def encrypt_records(self, d_plain, d_c1_pkr, f_record, d_out, n_ciphertexts, kbits, stream=0):
        """d_plain: n exponent records; d_c1_pkr: records of h^r and pk^r; d_out: 2n records"""
        import numpy as np
        f = np.ascontiguousarray(f_record, dtype=np.uint32)
        _chk(self.L.cofhe_hip_encrypt_records(self.ctx, C.c_void_p(d_plain), C.c_void_p(d_c1_pkr),
                                              f.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_void_p(d_out),
                                              C.c_uint64(n_ciphertexts), C.c_uint32(kbits), C.c_void_p(stream)))

    def part_decrypt_records(self, d_cts, d_share, d_out, n_ciphertexts, stream=0):
        """d_out: n form records = c1^share"""
        _chk(self.L.cofhe_hip_part_decrypt_records(self.ctx, C.c_void_p(d_cts), C.c_void_p(d_share), C.c_void_p(d_out),
                                                   C.c_uint64(n_ciphertexts), C.c_void_p(stream)))

    def combine_part_decryptions_records(self, d_cts, d_parts, lambdas, f_record, d_out, n_ciphertexts, kbits, stream=0):
        """d_parts: len(lambdas) x n form records (party-major); lambdas: +1 / -1 per party"""
        import numpy as np
        f = np.ascontiguousarray(f_record, dtype=np.uint32)
        lam = (C.c_int32 * len(lambdas))(*lambdas)
        _chk(self.L.cofhe_hip_combine_part_decryptions_records(
            self.ctx, C.c_void_p(d_cts), C.c_void_p(d_parts), C.c_uint32(len(lambdas)), lam,
            f.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_void_p(d_out), C.c_uint64(n_ciphertexts), C.c_uint32(kbits),
            C.c_void_p(stream)))

    # ---- device memory and fences of the library's own HIP runtime (bench.py uses no other) --------------------
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _chk(self.L.cofhe_hip_malloc(self.ctx, C.c_size_t(nbytes), C.byref(p)))
        return p.value

    def free(self, dptr: int, stream=0):
        _chk(self.L.cofhe_hip_free_on_stream(self.ctx, C.c_void_p(dptr), C.c_void_p(stream)))

    def upload(self, dptr: int, host, stream=0):
        """host: a contiguous numpy array (or bytes); copies all of it to dptr"""
        import numpy as np
        a = np.frombuffer(host, dtype=np.uint8) if isinstance(host, (bytes, bytearray)) else np.ascontiguousarray(host)
        _chk(self.L.cofhe_hip_upload(self.ctx, C.c_void_p(dptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), C.c_void_p(stream)))
        _chk(self.L.cofhe_hip_stream_sync(self.ctx, C.c_void_p(stream)))       # the host array may go away after the call

    def download(self, dptr: int, nbytes: int, dtype="uint32", stream=0):
        """device -> a new numpy array of `dtype` (synchronises the stream)"""
        import numpy as np
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _chk(self.L.cofhe_hip_download(self.ctx, out.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), C.c_size_t(out.nbytes), C.c_void_p(stream)))
        return out

    def stream_sync(self, stream=0):
        _chk(self.L.cofhe_hip_stream_sync(self.ctx, C.c_void_p(stream)))

    def profile_read(self, kernel: str, clear=False):
        """(summed ms, launches) of the spans the "profile_kernels" option recorded for `kernel`"""
        ms, n = C.c_float(), C.c_uint32()
        _chk(self.L.cofhe_hip_profile_read(self.ctx, C.c_char_p(kernel.encode()), C.byref(ms), C.byref(n), C.c_int(1 if clear else 0)))
        return float(ms.value), int(n.value)

    # ---- more than one GPU (cofhe_amd/csrc/shard.hip) -------------------------------------------------------
    def shard_rows(self, n_rows, world, rank):
        """(row0, n_local) of this rank's contiguous row block"""
        r0, nl = C.c_uint64(), C.c_uint64()
        self.L.cofhe_hip_shard_rows.restype = None
        self.L.cofhe_hip_shard_rows(C.c_uint64(n_rows), C.c_uint32(world), C.c_uint32(rank), C.byref(r0), C.byref(nl))
        return r0.value, nl.value

    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        _chk(self.L.cofhe_hip_comm_unique_id(buf))
        return bytes(buf)

    def comm_create(self, unique_id: bytes, world: int, rank: int):
        comm = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _chk(self.L.cofhe_hip_comm_create(self.ctx, buf, C.c_uint32(world), C.c_uint32(rank), C.byref(comm)))
        return comm

    def comm_info(self, comm):
        """(world, rank, rank count RCCL reports for the communicator)"""
        w, r, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _chk(self.L.cofhe_hip_comm_info(comm, C.byref(w), C.byref(r), C.byref(n)))
        return w.value, r.value, n.value

    def comm_set_option(self, comm, name: str, value: int):
        _chk(self.L.cofhe_hip_comm_set_option(comm, C.c_char_p(name.encode()), C.c_int64(value)))

    def comm_destroy(self, comm):
        self.L.cofhe_hip_comm_destroy.restype = None
        self.L.cofhe_hip_comm_destroy(comm)

    def all_gather_rows(self, comm, d_local, n_rows, row_bytes, d_out, stream=0):
        _chk(self.L.cofhe_hip_all_gather_rows(self.ctx, comm, C.c_void_p(d_local), C.c_uint64(n_rows), C.c_uint64(row_bytes),
                                              C.c_void_p(d_out), C.c_void_p(stream)))

    def set_option(self, name: str, value: int):
        """pins a launcher decision of the matrix product ("wnaf_width", "matmul_segments"; 0 = automatic)"""
        _chk(self.L.cofhe_hip_ctx_set_option(self.ctx, C.c_char_p(name.encode()), C.c_int64(value)))

    def device_status(self, clear=True, stream=0) -> int:
        """status word of the kernels (bit 1: Euclid cap, 2: reduction cap, 4: division): 0 unless a record was not a form"""
        w = C.c_uint32()
        _chk(self.L.cofhe_hip_device_status(self.ctx, C.byref(w), C.c_int(1 if clear else 0), C.c_void_p(stream)))
        return w.value

    def validate_records(self, d_records, n_records, stream=0) -> bool:
        """True when every record is a reduced form of the context's discriminant"""
        ok = C.c_int()
        _chk(self.L.cofhe_hip_validate_records(self.ctx, C.c_void_p(d_records), C.c_uint64(n_records), C.byref(ok), C.c_void_p(stream)))
        return bool(ok.value)

    def time_compose(self, d_a, d_b, d_out, n_records, iters, stream=0) -> float:
        ms = C.c_float()
        _chk(self.L.cofhe_hip_time_compose(self.ctx, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_out),
                                           C.c_uint64(n_records), C.c_int(iters), C.c_void_p(stream), C.byref(ms)))
        return float(ms.value)


    def time_stream(self, fn, iters=1, stream=0) -> float:
        """milliseconds per call of `iters` calls of fn() between two HIP events on `stream` (cofhe_hip_timer_*): the
        stream the library launches on"""
        h = C.c_void_p()
        _chk(self.L.cofhe_hip_timer_start(self.ctx, C.c_void_p(stream), C.byref(h)))
        try:
            for _ in range(iters):
                fn()
        finally:
            ms = C.c_float()
            rc = self.L.cofhe_hip_timer_stop(self.ctx, h, C.c_void_p(stream), C.byref(ms))
        _chk(rc)
        return float(ms.value) / iters


def workspace_plan(op: str, *args):
    """cofhe_hip_workspace_plan (host only): ([(name, byte offset, byte count)], total bytes) -- how the entry point `op`
    carves the context's workspace for these operand counts (the launchers use the same plan functions)"""
    L = load_library()

    class Region(C.Structure):
        _fields_ = [("name", C.c_char * 24), ("offset", C.c_uint64), ("bytes", C.c_uint64)]

    regs = (Region * 8)()
    n, total = C.c_uint32(), C.c_uint64()
    a = (C.c_uint64 * max(1, len(args)))(*[int(x) for x in args])
    _chk(L.cofhe_hip_workspace_plan(op.encode(), a, C.c_uint32(len(args)), regs, C.c_uint32(8), C.byref(n), C.byref(total)))
    return [(regs[i].name.decode(), int(regs[i].offset), int(regs[i].bytes)) for i in range(n.value)], int(total.value)


def gather_plan(n_rows: int, row_bytes: int, world: int):
    """cofhe_hip_gather_plan (host only): ([(byte offset, byte count)] per rank, uniform) -- the collective the
    library will run for a row-sharded tensor"""
    L = load_library()
    offs = (C.c_uint64 * world)()
    cnts = (C.c_uint64 * world)()
    uni = C.c_int()
    _chk(L.cofhe_hip_gather_plan(C.c_uint64(n_rows), C.c_uint64(row_bytes), C.c_uint32(world), offs, cnts, C.byref(uni)))
    return [(int(offs[r]), int(cnts[r])) for r in range(world)], bool(uni.value)
