// wire.hip -- the reference's binary tensor format produced and consumed ON the GPU.
//
// Format (serialize_ciphertext_tensor, include/x86_64/cpu_cryptosystem.inl:320-392; the
// partial-decryption variant :510-559 and the plaintext variant :229-270 differ only in the number
// of integers per element):
//     u32 ndim; u32 shape[ndim]; u64 off[I * E]; bytes...
// off[j] = running byte offset of integer j in the data area, bit 63 set when sgn() != 1 (negative
// OR zero); slot width = bits/8 + 1 bytes (bits = mpz_sizeinbase(x, 2), 1 for zero), little-endian
// magnitude.  The reader takes each length from the next offset and the last one from the buffer
// size (:429-434, :472-477).
//
// Unpacking is one thread per destination word of the record array (coalesced stores, byte loads
// served by L2); packing is bit lengths -> exclusive prefix sum of the slot widths (three small
// kernels) -> one thread per 4 output bytes.  All HBM-bound byte shuffling: no LDS, no MFMA.
#include <cstring>

#include "ctx.hpp"

using namespace cofhe;

namespace {

constexpr uint64_t OFFMASK = ~(1ull << 63);
constexpr int EXP_MAG_WORDS_W = 31, EXP_REC_WORDS_W = 32;      // exponent records (qf.hpp)

// kind 2 / 1: forms, 2 or 1 per element (3 integers each); kind 0: plaintext exponents (1 integer)
struct Geometry {
    int ints_per_rec;      // integers per destination record: 3 (a, b, c) or 1
    int rec_words;         // 168 or 32
};
__host__ __device__ inline Geometry geometry(int kind) { return kind == 0 ? Geometry{1, EXP_REC_WORDS_W} : Geometry{3, REC_WORDS}; }

// destination of integer `which` of a record: first word, capacity in words
__device__ inline void slot_of(int kind, int which, int &first, int &cap) {
    if (kind == 0) {
        first = 0;
        cap = EXP_MAG_WORDS_W;
    } else if (which == 0) {
        first = REC_A;
        cap = 40;
    } else if (which == 1) {
        first = REC_B;
        cap = 40;
    } else {
        first = REC_C;
        cap = 80;
    }
}

__device__ inline void int_span(const uint64_t *__restrict__ off, uint64_t idx, uint64_t count, uint64_t body_len,
                                uint64_t &st, uint64_t &en, bool &flag) {
    const uint64_t o = off[idx];
    st = o & OFFMASK;
    flag = (o >> 63) != 0;
    en = idx + 1 < count ? (off[idx + 1] & OFFMASK) : body_len;
}

// err bits: 1 corrupt offset table, 2 value wider than its limb plane, 4 a or c of a form is zero
__global__ void k_unpack(const uint8_t *__restrict__ body, const uint64_t *__restrict__ off, uint64_t body_len,
                         uint64_t n_rec, int kind, uint32_t *__restrict__ rec, uint32_t *__restrict__ err) {
    const Geometry gm = geometry(kind);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = tid / gm.rec_words;
    const int w = (int)(tid % gm.rec_words);
    if (r >= n_rec) return;
    const uint64_t count = n_rec * gm.ints_per_rec;
    const int sign_word = kind == 0 ? EXP_MAG_WORDS_W : REC_SIGN;
    uint32_t val = 0;
    if (w == sign_word) {
        // sign of b (forms) / of the exponent: the flag also marks zero, which is not negative
        const uint64_t idx = r * gm.ints_per_rec + (kind == 0 ? 0 : 1);
        uint64_t st, en;
        bool flag;
        int_span(off, idx, count, body_len, st, en, flag);
        bool nonzero = false;
        if (flag && st <= en && en <= body_len)
            for (uint64_t i = st; i < en; i++) nonzero |= body[i] != 0;
        val = (flag && nonzero) ? 1u : 0u;
    } else if (w < sign_word) {
        int which = 0, first, cap;
        if (kind != 0) which = w < REC_B ? 0 : w < REC_C ? 1 : 2;
        slot_of(kind, which, first, cap);
        const uint64_t idx = r * gm.ints_per_rec + which;
        uint64_t st, en;
        bool flag;
        int_span(off, idx, count, body_len, st, en, flag);
        if (st > en || en > body_len) {
            atomicOr(err, 1u);
            return;
        }
        const uint64_t len = en - st;
        const uint64_t b0 = (uint64_t)(w - first) * 4;
        for (int i = 0; i < 4; i++)
            if (b0 + i < len) val |= (uint32_t)body[st + b0 + i] << (8 * i);
        if (w - first == cap - 1) {                  // top word of the plane: nothing may lie above it
            bool over = false;
            for (uint64_t i = (uint64_t)cap * 4; i < len; i++) over |= body[st + i] != 0;
            if (over) atomicOr(err, 2u);
        }
        if (w == first && kind != 0 && which != 1) { // a and c are positive
            bool nonzero = false;
            for (uint64_t i = 0; i < len; i++) nonzero |= body[st + i] != 0;
            if (!nonzero) atomicOr(err, 4u);
        }
    }
    rec[r * gm.rec_words + w] = val;
}

// width[j] = slot bytes of integer j, flag bit in bit 63 (same packing as the offset table)
__global__ void k_widths(const uint32_t *__restrict__ rec, uint64_t n_rec, int kind, uint64_t *__restrict__ width) {
    const Geometry gm = geometry(kind);
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rec * gm.ints_per_rec) return;
    const uint64_t r = j / gm.ints_per_rec;
    const int which = (int)(j % gm.ints_per_rec);
    int first, cap;
    slot_of(kind, which, first, cap);
    const uint32_t *p = rec + r * gm.rec_words + first;
    uint32_t bits = 0;
    for (int i = cap - 1; i >= 0; i--) {
        const uint32_t v = p[i];
        if (v) {
            bits = (uint32_t)i * 32 + 32 - __clz(v);
            break;
        }
    }
    const uint32_t sign = rec[r * gm.rec_words + (kind == 0 ? EXP_MAG_WORDS_W : REC_SIGN)];
    const bool signed_slot = kind == 0 || which == 1;
    const bool flag = bits == 0 || (signed_slot && sign != 0);
    width[j] = (uint64_t)((bits ? bits : 1) / 8 + 1) | (flag ? (1ull << 63) : 0ull);
}

// exclusive prefix sum of the low 63 bits, flag bit carried through: off[j] = sum_{i<j} width[i] | flag[j]
constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;
__device__ inline uint64_t block_exclusive_scan(uint64_t v, uint64_t *sh, uint64_t &total) {
    const int t = (int)threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int d = 1; d < SCAN_BLOCK; d <<= 1) {
        const uint64_t x = t >= d ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    total = sh[SCAN_BLOCK - 1];
    const uint64_t excl = sh[t] - v;
    __syncthreads();
    return excl;
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_tile_sums(const uint64_t *__restrict__ width, uint64_t n, uint64_t *__restrict__ tile_sum) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) s += width[base + i] & OFFMASK;
    uint64_t total;
    block_exclusive_scan(s, sh, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
// one block: tile_sum[] -> exclusive prefix in place, grand total to *total_out
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(uint64_t *__restrict__ tile_sum, uint64_t n_tiles, uint64_t *__restrict__ total_out) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < n_tiles; b0 += SCAN_BLOCK) {
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t v = i < n_tiles ? tile_sum[i] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive_scan(v, sh, total);
        if (i < n_tiles) tile_sum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_offsets(const uint64_t *__restrict__ width, uint64_t n, const uint64_t *__restrict__ tile_base,
                                                       uint64_t *__restrict__ off) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t w[SCAN_ITEMS], s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        w[i] = base + i < n ? width[base + i] : 0;
        s += w[i] & OFFMASK;
    }
    uint64_t total;
    uint64_t run = tile_base[blockIdx.x] + block_exclusive_scan(s, sh, total);
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) off[base + i] = run | (w[i] & ~OFFMASK);
        run += w[i] & OFFMASK;
    }
}

// one thread per (integer, group of 4 output bytes)
__global__ void k_pack(const uint32_t *__restrict__ rec, const uint64_t *__restrict__ off, const uint64_t *__restrict__ width,
                       uint64_t n_rec, int kind, int max_words, uint8_t *__restrict__ body) {
    const Geometry gm = geometry(kind);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j = tid / max_words;
    const int w = (int)(tid % max_words);
    if (j >= n_rec * gm.ints_per_rec) return;
    const uint64_t r = j / gm.ints_per_rec;
    const int which = (int)(j % gm.ints_per_rec);
    int first, cap;
    slot_of(kind, which, first, cap);
    const uint64_t len = width[j] & OFFMASK;
    const uint64_t b0 = (uint64_t)w * 4;
    if (b0 >= len) return;
    const uint32_t v = w < cap ? rec[r * gm.rec_words + first + w] : 0u;      // the slot may be one byte longer than the plane
    uint8_t *dst = body + (off[j] & OFFMASK) + b0;
    for (int i = 0; i < 4; i++)
        if (b0 + i < len) dst[i] = (uint8_t)(v >> (8 * i));
}

// scratch from the context's block cache (cofhe_hip_malloc): hipMalloc / hipFree would synchronise the device several
// times per call
struct Tmp {
    cofhe_hip_ctx *ctx = nullptr;
    void *p = nullptr;
    int get(cofhe_hip_ctx *c, size_t bytes) {
        ctx = c;
        return cofhe_hip_malloc(c, bytes, &p);
    }
    ~Tmp() {
        if (p) (void)cofhe_hip_free(ctx, p);
    }
};

int header_of(int kind, const uint8_t *hdr, size_t len, uint32_t *ndim, uint32_t shape[8], uint64_t *count) {
    if (len < 4) return fail(COFHE_HIP_EINVAL, "tensor buffer too short");
    uint32_t nd;
    memcpy(&nd, hdr, 4);
    if (nd > 8) return fail(COFHE_HIP_EINVAL, "tensor rank above 8");
    if (len < 4 + 4ull * nd) return fail(COFHE_HIP_EINVAL, "tensor buffer too short");
    uint64_t ne = 1;
    for (uint32_t i = 0; i < nd; i++) {
        memcpy(&shape[i], hdr + 4 + 4 * i, 4);
        if (shape[i] != 0 && ne > (1ull << 40) / shape[i]) return fail(COFHE_HIP_EINVAL, "tensor too large");     // before the product can wrap
        ne *= shape[i];
    }
    *ndim = nd;
    *count = ne * (kind == 0 ? 1 : 3 * (uint64_t)kind);
    if (len < 4 + 4ull * nd + 8ull * *count) return fail(COFHE_HIP_EINVAL, "tensor buffer too short");
    return COFHE_HIP_OK;
}

}  // namespace

extern "C" {

int cofhe_hip_unpack_tensor_device(cofhe_hip_ctx *ctx, const void *d_bytes, size_t len, int kind, void *d_records,
                                   uint64_t capacity_records, uint32_t *ndim, uint32_t shape[8], uint64_t *n_records,
                                   void *stream) {
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (kind < 0 || kind > 2) return fail(COFHE_HIP_EINVAL, "kind must be 0 (plaintexts), 1 (forms) or 2 (ciphertexts)");
    HIPCHK(hipSetDevice(ctx->device));
    uint8_t hdr[36] = {0};
    const size_t hl = len < sizeof hdr ? len : sizeof hdr;
    HIPCHK(hipMemcpyAsync(hdr, d_bytes, hl, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    uint64_t count;
    if (int rc = header_of(kind, hdr, len, ndim, shape, &count)) return rc;
    // the offset table must be 8-byte aligned for the kernels; d_bytes + 4 + 4*ndim is when ndim is odd
    const Geometry gm = geometry(kind);
    const uint64_t nrec = count / gm.ints_per_rec;
    *n_records = nrec;
    if (nrec > capacity_records) return fail(COFHE_HIP_EINVAL, "record buffer too small for this tensor");
    if (nrec == 0) return COFHE_HIP_OK;
    const size_t tab = 4 + 4ull * *ndim, hdrlen = tab + 8ull * count;
    const uint8_t *src = (const uint8_t *)d_bytes;
    Tmp aligned;
    const uint64_t *off = (const uint64_t *)(src + tab);
    if (((uintptr_t)off & 7) != 0) {
        if (int rc = aligned.get(ctx, 8ull * count)) return rc;
        HIPCHK(hipMemcpyAsync(aligned.p, src + tab, 8ull * count, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        off = (const uint64_t *)aligned.p;
    }
    Tmp err;
    if (int rc = err.get(ctx, 4)) return rc;
    HIPCHK(hipMemsetAsync(err.p, 0, 4, (hipStream_t)stream));
    const uint64_t threads = nrec * gm.rec_words;
    hipLaunchKernelGGL(k_unpack, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src + hdrlen, off,
                       (uint64_t)(len - hdrlen), nrec, kind, (uint32_t *)d_records, (uint32_t *)err.p);
    HIPCHK(hipGetLastError());
    uint32_t e = 0;
    HIPCHK(hipMemcpyAsync(&e, err.p, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (e & 1) return fail(COFHE_HIP_EINVAL, "corrupt offset table");
    if (e & 2) return fail(COFHE_HIP_EINVAL, kind == 0 ? "exponent wider than 992 bits" : "form coefficient outside the supported range");
    if (e & 4) return fail(COFHE_HIP_EINVAL, "form coefficient outside the supported range");
    if (kind != 0) {
        // data from outside: every form must be a reduced form of the context's discriminant -- the kernels assume it
        int valid = 1;
        if (int rc = cofhe_hip_validate_records(ctx, d_records, nrec, &valid, stream)) return rc;
        if (!valid) return fail(COFHE_HIP_EINVAL, "not a reduced form of the context's discriminant");
    }
    return COFHE_HIP_OK;
}

int cofhe_hip_pack_tensor_device(cofhe_hip_ctx *ctx, const void *d_records, uint64_t n_records, int kind, uint32_t ndim,
                                 const uint32_t *shape, void *d_bytes, size_t capacity, size_t *len, void *stream) {
    if (kind < 0 || kind > 2) return fail(COFHE_HIP_EINVAL, "kind must be 0 (plaintexts), 1 (forms) or 2 (ciphertexts)");
    if (ndim > 8) return fail(COFHE_HIP_EINVAL, "tensor rank above 8");
    uint64_t ne = 1;
    for (uint32_t i = 0; i < ndim; i++) {
        if (shape[i] != 0 && ne > (1ull << 40) / shape[i]) return fail(COFHE_HIP_EINVAL, "tensor too large");
        ne *= shape[i];
    }
    const uint64_t per_elem = kind == 0 ? 1 : (uint64_t)kind;
    if (ne * per_elem != n_records) return fail(COFHE_HIP_EINVAL, "shape does not match the record count");
    HIPCHK(hipSetDevice(ctx->device));
    const Geometry gm = geometry(kind);
    const uint64_t count = n_records * gm.ints_per_rec;
    const size_t tab = 4 + 4ull * ndim, hdrlen = tab + 8ull * count;
    if (capacity < hdrlen) return fail(COFHE_HIP_EINVAL, "output buffer too small");
    uint8_t hdr[36];
    memcpy(hdr, &ndim, 4);
    for (uint32_t i = 0; i < ndim; i++) memcpy(hdr + 4 + 4 * i, &shape[i], 4);
    uint8_t *dst = (uint8_t *)d_bytes;
    HIPCHK(hipMemcpyAsync(dst, hdr, tab, hipMemcpyHostToDevice, (hipStream_t)stream));
    if (count == 0) {
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        *len = hdrlen;
        return COFHE_HIP_OK;
    }
    const uint64_t tiles = (count + SCAN_TILE - 1) / SCAN_TILE;
    Tmp width, offs, tile, total;
    if (int rc = width.get(ctx, 8ull * count)) return rc;
    if (int rc = offs.get(ctx, 8ull * count)) return rc;
    if (int rc = tile.get(ctx, 8ull * tiles)) return rc;
    if (int rc = total.get(ctx, 8)) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_widths, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, (const uint32_t *)d_records, n_records,
                       kind, (uint64_t *)width.p);
    hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, (const uint64_t *)width.p, count,
                       (uint64_t *)tile.p);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, st, (uint64_t *)tile.p, tiles, (uint64_t *)total.p);
    hipLaunchKernelGGL(k_offsets, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, (const uint64_t *)width.p, count,
                       (const uint64_t *)tile.p, (uint64_t *)offs.p);
    HIPCHK(hipGetLastError());
    uint64_t body_len = 0;
    HIPCHK(hipMemcpyAsync(&body_len, total.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *len = hdrlen + body_len;
    if (capacity < hdrlen + body_len) return fail(COFHE_HIP_EINVAL, "output buffer too small");
    HIPCHK(hipMemcpyAsync(dst + tab, offs.p, 8ull * count, hipMemcpyDeviceToDevice, st));
    const int max_words = kind == 0 ? EXP_MAG_WORDS_W + 1 : 81;       // 4-byte groups per slot, incl. the extra byte
    const uint64_t threads = count * (uint64_t)max_words;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const uint32_t *)d_records,
                       (const uint64_t *)offs.p, (const uint64_t *)width.p, n_records, kind, max_words, dst + hdrlen);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return COFHE_HIP_OK;
}

// upper bound of the serialised size of n_records records of this kind (for sizing d_bytes)
size_t cofhe_hip_packed_size_bound(uint64_t n_records, int kind, uint32_t ndim) {
    const uint64_t per_rec = kind == 0 ? (8 + 125) : 3 * 8 + 161 + 161 + 321;
    return 4 + 4ull * ndim + n_records * per_rec;
}

}  // extern "C"
