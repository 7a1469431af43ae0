// wide.hip -- the latency kernels: ONE ladder per wavefront in the wavefront-wide layout (wide.hpp, qfw.hpp).
// Reference: the c1^sk of CPUCryptoSystem::decrypt_tensor / decrypt (include/x86_64/cpu_cryptosystem_tensor_ops.inl:21-33,
// cpu_cryptosystem.inl:16-19) and the c1^share of part_decrypt_tensor (cpu_cryptosystem_distributed.inl:244-254) when the
// tensor carries one c1 -- a single chain of ~1100 dependent compositions whatever the tensor's size.
#include <hip/hip_runtime.h>

#include "form_io.hpp"
#include "qfw.hpp"

using namespace cofhe;
using namespace cofhe::wide;

namespace cofhe_k {

constexpr int WIDE_LDS_WORDS = 3 * REC_WORDS + SCRATCH_WORDS;

// out = reduced(fa o fb): the wide composition, or -- for the pairs it declines -- qf_compose on lanes 0..7 of this very
// wavefront (operands and result travel through three LDS records; the 8-lane code is the in-group form, no workgroup
// protocol).  The bytes are the same either way: both end in the unique reduced form.
__device__ __forceinline__ void compose_wide_or_fallback(uint32_t *lds, WForm &out, const WForm &fa, const WForm &fb, const QDisc &dd,
                                                         uint32_t *status) {
    if (wf_compose(out, fa, fb, dd)) return;
    uint32_t *ra = lds, *rb = lds + REC_WORDS, *ro = lds + 2 * REC_WORDS;
    wf_store(fa, ra);
    wf_store(fb, rb);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if ((threadIdx.x & 63) < G) {
        Ctx c;
        c.gl = (int)(threadIdx.x & (G - 1));
        c.base4 = 0;
        c.scr = lds + 3 * REC_WORDS;
        c.rank = -1;
        c.status = status;
        QForm a, b, r;
        qf_load(c, a, ra);
        qf_load(c, b, rb);
        qf_compose<0, false>(c, r, a, b, dd);
        qf_store(c, r, ro);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    out = wf_load(ro);
}

__device__ __forceinline__ void wf_inverse(WForm &f) {      // (a, -b, c) kept inside the reduced domain (qf.hpp: qf_inverse)
    if (w_is_zero(f.bm)) return;
    if (w_cmp(f.bm, f.a) == 0) return;
    if (w_cmp(f.a, f.c) == 0) return;
    f.bneg ^= 1;
}

// out[g] = base[g * base_stride]^e for the one exponent whose width-w non-adjacent digits are in `digits` (k_wnaf_digits):
// the schedule of k_pow_shared (cofhe_hip.hip), one ladder per WAVEFRONT.  table: (tw + 2) records per ladder.
__global__ void __launch_bounds__(64) k_pow_shared_wide(const uint32_t *__restrict__ base, const int8_t *__restrict__ digits,
                                                        const uint32_t *__restrict__ maxlen, uint32_t *__restrict__ table,
                                                        uint32_t *__restrict__ out, uint64_t n_items, uint32_t base_stride, uint32_t tw,
                                                        const uint32_t *__restrict__ one_rec, const uint32_t *__restrict__ absdelta,
                                                        int half_dbits, uint32_t *__restrict__ status) {
    __shared__ uint32_t lds[WIDE_LDS_WORDS];
    const QDisc dd{absdelta, half_dbits};
    const uint64_t g = blockIdx.x;
    if (g >= n_items) return;
    uint32_t *tab = table + g * (tw + 2) * REC_WORDS;
    uint32_t *accp = tab + (uint64_t)(tw + 1) * REC_WORDS;
    {
        const WForm x = wf_load(base + g * base_stride * REC_WORDS);
        wf_store(x, tab);
    }
    const int len = (int)*maxlen;
    const uint32_t table_steps = tw > 1 ? tw : 0;
    uint32_t ts = 0;
    int t = len - 1;
    bool have = false, mul_pending = false;
    // the running power stays in REGISTERS (a form is six of them in this layout): the ~bits squarings of the ladder touch no
    // memory at all; only the table entries (slots 0 .. tw: odd powers and x^2) live in HBM
    WForm acc = wf_load(one_rec);
    (void)accp;
    while (true) {
        const uint32_t *lsrc = nullptr, *rsrc = nullptr;  // lsrc == nullptr: the running power; rsrc == nullptr: a squaring
        uint32_t *dst = nullptr;                           // nullptr: the running power
        bool rinv = false;
        if (ts < table_steps) {
            if (ts == 0) {
                lsrc = tab;
                dst = tab + (uint64_t)tw * REC_WORDS;
            } else {
                lsrc = tab + (uint64_t)(ts - 1) * REC_WORDS;
                rsrc = tab + (uint64_t)tw * REC_WORDS;
                dst = tab + (uint64_t)ts * REC_WORDS;
            }
        } else if (t < 0) {
            break;
        } else if (!have) {
            const int dg = digits[t];
            acc = wf_load(tab + (uint64_t)((dg < 0 ? -dg : dg) >> 1) * REC_WORDS);
            if (dg < 0) wf_inverse(acc);
            have = true;
            t--;
            continue;
        } else if (!mul_pending) {
            mul_pending = digits[t] != 0;
            if (!mul_pending) t--;
        } else {
            const int dg = digits[t];
            rsrc = tab + (uint64_t)((dg < 0 ? -dg : dg) >> 1) * REC_WORDS;
            rinv = dg < 0;
            mul_pending = false;
            t--;
        }
        const WForm l_ = lsrc ? wf_load(lsrc) : acc;
        WForm r_ = l_;
        if (rsrc) {
            r_ = wf_load(rsrc);
            if (rinv) wf_inverse(r_);
        }
        WForm r;
        compose_wide_or_fallback(lds, r, l_, r_, dd, status);
        if (dst) wf_store(r, dst); else acc = r;
        if (ts < table_steps) ts++;
    }
    if (len == 0) acc = wf_load(one_rec);
    wf_store(acc, out + g * REC_WORDS);
}

// out[g] = base[g * base_stride]^e on TWO wavefronts: the exponent in non-adjacent form (digits 0, +-1 from k_wnaf_digits
// with w = 2) is read right to left; the squaring wavefront computes S_t = base^(2^t) -- the only chain the ladder cannot do
// without -- and hands S_t over for every non-zero digit; the multiplying wavefront multiplies the S_t (or their inverses: a
// sign flip) into the result while the squaring goes on.  One multiplication per >= 2 squarings at the most, so the
// multiplier never falls behind, and the latency of the ladder is that of its len - 1 squarings plus one product, instead
// of the squarings, the products (one per w + 1 digits) and the table of the left-to-right ladder (k_pow_shared_wide:
// ~19 % more compositions in a row at w = 5).
// The two wavefronts are two WORKGROUPS (blocks 2g and 2g + 1), so that they sit on different CUs: as two wavefronts of one
// workgroup they shared an instruction cache that neither's ~50 KB of straight-line code fits, and the squarer ran 6 %
// slower for the company (138.6 against 130.7 us per squaring).  The forms travel through a ring of PAIR_RING records in
// the workspace; `ctl[0]` / `ctl[1]` count the forms published / taken (release / acquire at agent scope).  Both blocks
// are resident by construction -- the launcher uses this kernel for at most 256 ladders, 512 single-wavefront blocks -- and
// each spins only on a count the other one is bound to advance: the squarer publishes exactly the non-zero digits, the
// multiplier takes exactly those.  ctl must be zero at launch.
constexpr int PAIR_RING = 4;
constexpr int PAIR_CTL_WORDS = 4;
__global__ void __launch_bounds__(64) k_pow_shared_pair(const uint32_t *__restrict__ base, const int8_t *__restrict__ digits,
                                                        const uint32_t *__restrict__ maxlen, uint32_t *__restrict__ ring_all,
                                                        uint32_t *__restrict__ ctl_all, uint32_t *__restrict__ out, uint64_t n_items,
                                                        uint32_t base_stride, const uint32_t *__restrict__ one_rec,
                                                        const uint32_t *__restrict__ absdelta, int half_dbits, uint32_t *__restrict__ status) {
    __shared__ uint32_t lds[WIDE_LDS_WORDS];
    const QDisc dd{absdelta, half_dbits};
    const uint64_t g = blockIdx.x >> 1;
    if (g >= n_items) return;
    uint32_t *ring = ring_all + g * (uint64_t)(PAIR_RING * REC_WORDS);
    uint32_t *ctl = ctl_all + g * PAIR_CTL_WORDS;
    const int len = (int)*maxlen;
    const bool lane0 = (threadIdx.x & 63) == 0;
    if ((blockIdx.x & 1) == 0) {
        WForm s = wf_load(base + g * base_stride * REC_WORDS);
        uint32_t k = 0;
        for (int t = 0; t < len; t++) {
            if (digits[t] != 0) {
                while (k >= PAIR_RING + __hip_atomic_load(&ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) __builtin_amdgcn_s_sleep(8);
                wf_store(s, ring + (k % PAIR_RING) * REC_WORDS);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                k++;
                if (lane0) __hip_atomic_store(&ctl[0], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (t + 1 < len) {
                WForm r;
                compose_wide_or_fallback(lds, r, s, s, dd, status);
                s = r;
            }
        }
    } else {
        WForm acc = wf_load(one_rec);
        bool have = false;
        uint32_t k = 0;
        for (int t = 0; t < len; t++) {
            const int dg = digits[t];
            if (dg == 0) continue;
            while (__hip_atomic_load(&ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= k) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            WForm x = wf_load(ring + (k % PAIR_RING) * REC_WORDS);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // the loads are done before the slot is given back
            k++;
            if (lane0) __hip_atomic_store(&ctl[1], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (dg < 0) wf_inverse(x);
            if (!have) {
                acc = x;
                have = true;
            } else {
                WForm r;
                compose_wide_or_fallback(lds, r, acc, x, dd, status);
                acc = r;
            }
        }
        wf_store(acc, out + g * REC_WORDS);
    }
}

// table[j] = base^(2^j), j < len: the chain of squarings behind a fixed-base table (h of the cryptosystem, a public key:
// cofhe_hip_pow_fixed_base_records), one wavefront, the running square in registers.  Until round 4 this was a workgroup of
// the throughput layout squaring in lockstep for the sake of ONE chain (k_square_chain: ~0.29 ms per squaring).
__global__ void __launch_bounds__(64) k_square_chain_wide(const uint32_t *__restrict__ base, uint32_t *__restrict__ table, uint32_t len,
                                                          const uint32_t *__restrict__ absdelta, int half_dbits, uint32_t *__restrict__ status) {
    __shared__ uint32_t lds[WIDE_LDS_WORDS];
    const QDisc dd{absdelta, half_dbits};
    WForm acc = wf_load(base);
    wf_store(acc, table);
    for (uint32_t j = 1; j < len; j++) {
        WForm r;
        compose_wide_or_fallback(lds, r, acc, acc, dd, status);
        acc = r;
        wf_store(acc, table + (uint64_t)j * REC_WORDS);
    }
}

// out[i] = a[i] o b[i], one composition per wavefront: the wide composition as a plain kernel (parity tests, and the
// latency of one composition measured by itself)
__global__ void __launch_bounds__(64) k_compose_wide(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint32_t *__restrict__ out,
                                                     uint64_t n, uint32_t reps, const uint32_t *__restrict__ absdelta, int half_dbits,
                                                     uint32_t *__restrict__ status, uint32_t *__restrict__ fallbacks) {
    __shared__ uint32_t lds[WIDE_LDS_WORDS];
    const QDisc dd{absdelta, half_dbits};
    const uint64_t g = blockIdx.x;
    if (g >= n) return;
    const WForm x = wf_load(a + g * REC_WORDS), y = wf_load(b + g * REC_WORDS);
    WForm r;
    for (uint32_t i = 0; i < (reps ? reps : 1u); i++) {
        if (fallbacks) {
            WForm probe;
            if (!wf_compose(probe, x, y, dd) && (threadIdx.x & 63) == 0 && i == 0) atomicAdd(fallbacks, 1u);
        }
        compose_wide_or_fallback(lds, r, x, y, dd, status);
    }
    wf_store(r, out + g * REC_WORDS);
}

}  // namespace cofhe_k
