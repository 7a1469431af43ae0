// pair.hpp -- multi-precision integers spread over a lane PAIR (2 lanes of a wave64, 32 instances per
// wavefront), the arithmetic layer of the latency-oriented composition kernel k_compose_pair.
//
// Why a second layout.  The 8-lane groups of mp.hpp fill the GPU at the size of the headline workload
// (32 768 compositions = 4 wavefronts on each of the 1024 SIMDs), but their Euclid rounds are serialised on
// one wavefront per workgroup that runs the scalar Lehmer batches (tools/wg_timing.hip: 4.1 us of a 5.1 us
// round is spent waiting for it), and every primitive pays ~25-50 instructions of cross-lane overhead for 5
// limbs of work per lane.  With TWO lanes per integer there is one lane boundary, the batch runs inside the
// lane (no workgroup barrier, no mailbox), a wavefront holds its whole working set in registers (one
// wavefront per SIMD, 512 registers) and a single wavefront streams straight-line code at the full single-wave
// issue rate (tools/icache_bench.hip: 4.0 -> 4.2 cycles per instruction from an 8 KB to a 192 KB body).
//
// Layout.  BN<N> is a non-negative integer of 2N limbs (radix 2^32): the LOW lane of the pair (even lane)
// holds limbs [0, N), the HIGH lane (odd lane) limbs [N, 2N), all in registers with compile-time indices.
// Control flow is PAIR-uniform: both lanes of a pair take the same branches; pairs of a wave may diverge.
// Everything is templated on N (17: |Delta| <= 2140 bits, 19: <= 2400 bits).
//
// COFHE_HOSTSIM (tests/hostsim only): the two lanes are two host threads with a spin barrier.
#pragma once
#include <stdint.h>

#include "../../cofhe_amd/csrc/mp.hpp"        // scalar helpers shared with the 8-lane layout: WordDiv (division by an invariant word)
#include "../lehmer_variants/lehmer_variants.hpp"   // the 64-bit integer batch this experiment was built on (the product's is f64 now)

#if defined(COFHE_HOSTSIM)
#include <atomic>
#include <cstdio>
#include <cstdlib>
#define P2_DEV inline
#define P2_UNROLL _Pragma("GCC unroll 64")
#else
#include <hip/hip_runtime.h>
#define P2_DEV __device__ __forceinline__
#define P2_UNROLL _Pragma("unroll")
#endif

namespace cofhe2 {

#if defined(COFHE_HOSTSIM)
struct PairShared {
    alignas(64) uint32_t xchg[2];
    std::atomic<int> count{0};
    std::atomic<int> sense{0};
    void wait(int &local_sense) {
        local_sense ^= 1;
        if (count.fetch_add(1, std::memory_order_acq_rel) == 1) {
            count.store(0, std::memory_order_relaxed);
            sense.store(local_sense, std::memory_order_release);
        } else {
            long spins = 0;
            while (sense.load(std::memory_order_acquire) != local_sense)
                if (++spins > 4000000000L) {
                    fprintf(stderr, "hostsim: pair barrier timeout (pair-divergent control flow?)\n");
                    abort();
                }
        }
    }
};
struct PCtx {
    int hi;                 // 0: low lane (limbs [0, N)), 1: high lane (limbs [N, 2N))
    PairShared *ps;
    int sense = 0;
};
P2_DEV uint32_t p2_exchange(PCtx &c, uint32_t v, int src) {
    c.ps->xchg[c.hi] = v;
    c.ps->wait(c.sense);
    const uint32_t r = c.ps->xchg[src];
    c.ps->wait(c.sense);
    return r;
}
P2_DEV uint32_t xl(PCtx &c, uint32_t v) { return p2_exchange(c, v, c.hi ^ 1); }     // the partner's value
P2_DEV uint32_t flo(PCtx &c, uint32_t v) { return p2_exchange(c, v, 0); }           // the low lane's value in both
P2_DEV uint32_t fhi(PCtx &c, uint32_t v) { return p2_exchange(c, v, 1); }           // the high lane's value in both
P2_DEV bool wave_any(PCtx &c, bool p) { return (p2_exchange(c, p ? 1u : 0u, 0) | p2_exchange(c, p ? 1u : 0u, 1)) != 0; }
#else
struct PCtx {
    int hi;
};
// DPP quad permutes: one VALU instruction each, no LDS crossbar
P2_DEV uint32_t xl(PCtx &, uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }    // [1,0,3,2]
P2_DEV uint32_t flo(PCtx &, uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xA0, 0xF, 0xF, true); }   // [0,0,2,2]
P2_DEV uint32_t fhi(PCtx &, uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xF5, 0xF, 0xF, true); }   // [1,1,3,3]
// true when p holds in some active lane of the wavefront (a scalar branch); code guarded by it must be a no-op
// for the lanes where p is false
P2_DEV bool wave_any(PCtx &, bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
#endif
P2_DEV bool pair_any(PCtx &c, bool p) { return (xl(c, p ? 1u : 0u) | (p ? 1u : 0u)) != 0; }
// max of v (0 <= v < 64) over the lanes with enable set: six ballots (0 when no lane is enabled)
P2_DEV int wave_max_small(PCtx &c, int v, bool enable) {
    int r = 0;
    for (int b = 32; b; b >>= 1)
        if (wave_any(c, enable && v >= r + b)) r += b;
    return r;
}

// phase stamps of the diagnostic build tools/pair_timing.hip (lane 0 of every wavefront, shader clock)
#if defined(COFHE_PAIR_TIMING) && !defined(COFHE_HOSTSIM)
#define P2_PHASE(id) do { if ((threadIdx.x & 63) == 0) g_pair_phase[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define P2_PHASE(id) do { } while (0)
#endif
// why a pair left the fast path: recorded by the host simulator only (diagnostics of the tests)
#if defined(COFHE_HOSTSIM)
inline const char *g_p2_last_reason = "";
#define P2_LEAVE(okvar, why) do { (okvar) = false; if (c.hi == 0) g_p2_last_reason = (why); } while (0)
#else
#define P2_LEAVE(okvar, why) do { (okvar) = false; } while (0)
#endif
P2_DEV int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }
P2_DEV int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }

template <int N>
struct BN {
    uint32_t v[N];
};
template <int N>
struct SBN {            // sign-magnitude; neg is pair-uniform; a zero magnitude may carry either flag
    BN<N> m;
    int neg;
};

// ---------------------------------------------------------------------------------------- basics
template <int N>
P2_DEV void bn_zero(BN<N> &x) {
    P2_UNROLL for (int j = 0; j < N; j++) x.v[j] = 0u;
}
template <int N>
P2_DEV void bn_set_word(PCtx &c, BN<N> &x, uint32_t w) {
    bn_zero(x);
    x.v[0] = c.hi ? 0u : w;
}
template <int N>
P2_DEV void bn_select(BN<N> &r, bool take_y, const BN<N> &x, const BN<N> &y) {
    P2_UNROLL for (int j = 0; j < N; j++) r.v[j] = take_y ? y.v[j] : x.v[j];
}
template <int N>
P2_DEV bool bn_is_zero(PCtx &c, const BN<N> &x) {
    uint32_t o = 0;
    P2_UNROLL for (int j = 0; j < N; j++) o |= x.v[j];
    return (o | xl(c, o)) == 0;
}
// limb i (compile-time) of the full number, in both lanes
template <int I, int N>
P2_DEV uint32_t bn_limb(PCtx &c, const BN<N> &x) {
    static_assert(I >= 0 && I < 2 * N, "limb index");
    if (I < N) return flo(c, x.v[I < N ? I : 0]);
    return fhi(c, x.v[I >= N ? I - N : 0]);
}
// x == w ?
template <int N>
P2_DEV bool bn_is_word(PCtx &c, const BN<N> &x, uint32_t w) {
    uint32_t o = x.v[0] ^ (c.hi ? 0u : w);
    P2_UNROLL for (int j = 1; j < N; j++) o |= x.v[j];
    return (o | xl(c, o)) == 0;
}
// number of significant bits (0 for zero); pair-uniform
template <int N>
P2_DEV int bn_bitlen(PCtx &c, const BN<N> &x) {
    uint32_t top = x.v[0], idx = 0;
    P2_UNROLL for (int j = 1; j < N; j++) {
        const uint32_t w = x.v[j];
        top = w ? w : top;
        idx = w ? (uint32_t)j : idx;
    }
    const uint32_t mine = top ? (idx + (c.hi ? (uint32_t)N : 0u)) * 32u + 32u - (uint32_t)clz32(top) : 0u;
    const uint32_t other = xl(c, mine);
    return (int)(mine > other ? mine : other);
}
// -1 / 0 / +1; pair-uniform
template <int N>
P2_DEV int bn_cmp(PCtx &c, const BN<N> &x, const BN<N> &y) {
    uint32_t key = 0;        // 0: equal; else 2 + (x > y) of the most significant differing limb of this lane
    P2_UNROLL for (int j = 0; j < N; j++) {
        const uint32_t a = x.v[j], b = y.v[j];
        key = (a != b) ? (2u | (a > b ? 1u : 0u)) : key;
    }
    const uint32_t kh = fhi(c, key), kl = flo(c, key);
    const uint32_t k = kh ? kh : kl;
    return k == 0 ? 0 : ((k & 1u) ? 1 : -1);
}

// ---------------------------------------------------------------------------------------- carries
// adds the word w into r from limb 0 upwards (a no-op for w == 0).  The carry almost never travels beyond
// limb 1; the rest of the chain runs only when some lane of the wave needs it.  Returns the carry out of r.
template <int N>
P2_DEV uint32_t bn_lane_add_word(PCtx &c, BN<N> &r, uint32_t w) {
    uint32_t s = r.v[0] + w;
    uint32_t cy = s < w ? 1u : 0u;
    r.v[0] = s;
    s = r.v[1] + cy;
    cy = s < cy ? 1u : 0u;
    r.v[1] = s;
    if (wave_any(c, cy != 0)) {
        P2_UNROLL for (int j = 2; j < N; j++) {
            s = r.v[j] + cy;
            cy = s < cy ? 1u : 0u;
            r.v[j] = s;
        }
    }
    return cy;
}
// subtracts the word w from r (borrow chain, as above); returns the borrow out of r
template <int N>
P2_DEV uint32_t bn_lane_sub_word(PCtx &c, BN<N> &r, uint32_t w) {
    uint32_t a = r.v[0];
    uint32_t bw = a < w ? 1u : 0u;
    r.v[0] = a - w;
    a = r.v[1];
    r.v[1] = a - bw;
    bw = a < bw ? 1u : 0u;
    if (wave_any(c, bw != 0)) {
        P2_UNROLL for (int j = 2; j < N; j++) {
            a = r.v[j];
            r.v[j] = a - bw;
            bw = a < bw ? 1u : 0u;
        }
    }
    return bw;
}

// r = x + y; returns the carry out of the top limb (pair-uniform)
template <int N>
P2_DEV uint32_t bn_add(PCtx &c, BN<N> &r, const BN<N> &x, const BN<N> &y) {
    uint32_t cy = 0;
    P2_UNROLL for (int j = 0; j < N; j++) {
        const uint64_t t = (uint64_t)x.v[j] + y.v[j] + cy;
        r.v[j] = (uint32_t)t;
        cy = (uint32_t)(t >> 32);
    }
    const uint32_t o = xl(c, cy);
    const uint32_t c2 = bn_lane_add_word(c, r, c.hi ? o : 0u);
    return fhi(c, cy + c2);
}
// r = x - y (mod 2^(64N)); returns the borrow out of the top limb (1 when x < y), pair-uniform
template <int N>
P2_DEV uint32_t bn_sub(PCtx &c, BN<N> &r, const BN<N> &x, const BN<N> &y) {
    uint32_t bw = 0;
    P2_UNROLL for (int j = 0; j < N; j++) {
        const uint64_t t = (uint64_t)x.v[j] - y.v[j] - bw;
        r.v[j] = (uint32_t)t;
        bw = (uint32_t)(t >> 63);
    }
    const uint32_t o = xl(c, bw);
    const uint32_t b2 = bn_lane_sub_word(c, r, c.hi ? o : 0u);
    return fhi(c, bw + b2);
}

// ---------------------------------------------------------------------------------------- shifts
// x >> n, 0 <= n < 32 (n pair-uniform)
template <int N>
P2_DEV BN<N> bn_shr_small(PCtx &c, const BN<N> &x, int n) {
    BN<N> y;
    const uint32_t nxt0 = xl(c, x.v[0]);
    const uint32_t nxt = c.hi ? 0u : nxt0;           // the limb that follows this lane's top limb
    const uint32_t ls = (uint32_t)(32 - n) & 31u, keep = n ? 0xFFFFFFFFu : 0u;
    P2_UNROLL for (int j = 0; j < N; j++) {
        const uint32_t up = (j + 1 < N) ? x.v[j + 1 < N ? j + 1 : 0] : nxt;
        y.v[j] = (x.v[j] >> n) | ((up << ls) & keep);
    }
    return y;
}
// x << n, 0 <= n < 32 (bits shifted past limb 2N-1 are dropped)
template <int N>
P2_DEV BN<N> bn_shl_small(PCtx &c, const BN<N> &x, int n) {
    BN<N> y;
    const uint32_t prv0 = xl(c, x.v[N - 1]);
    const uint32_t prv = c.hi ? prv0 : 0u;           // the limb below this lane's limb 0
    const uint32_t rs = (uint32_t)(32 - n) & 31u, keep = n ? 0xFFFFFFFFu : 0u;
    P2_UNROLL for (int j = 0; j < N; j++) {
        const uint32_t dn = (j >= 1) ? x.v[j >= 1 ? j - 1 : 0] : prv;
        y.v[j] = (x.v[j] << n) | ((dn >> rs) & keep);
    }
    return y;
}
// x mod 2^bits (bits pair-uniform, 0 <= bits <= 64N)
template <int N>
P2_DEV BN<N> bn_mask_bits(PCtx &c, const BN<N> &x, int bits) {
    BN<N> y;
    P2_UNROLL for (int j = 0; j < N; j++) {
        const int lo = 32 * (c.hi * N + j);                 // bit position of this limb
        const int keep = bits - lo;                          // number of low bits of the limb that stay
        const uint32_t mask = keep >= 32 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : ((1u << keep) - 1u));
        y.v[j] = x.v[j] & mask;
    }
    return y;
}
// x << 32 (one limb up; the top limb is dropped) and x >> 32
template <int N>
P2_DEV BN<N> bn_shl_limb(PCtx &c, const BN<N> &x) {
    BN<N> y;
    const uint32_t prv0 = xl(c, x.v[N - 1]);
    y.v[0] = c.hi ? prv0 : 0u;
    P2_UNROLL for (int j = 1; j < N; j++) y.v[j] = x.v[j - 1];
    return y;
}
template <int N>
P2_DEV BN<N> bn_shr_limb(PCtx &c, const BN<N> &x) {
    BN<N> y;
    const uint32_t nxt0 = xl(c, x.v[0]);
    P2_UNROLL for (int j = 0; j + 1 < N; j++) y.v[j] = x.v[j + 1];
    y.v[N - 1] = c.hi ? 0u : nxt0;
    return y;
}

// ---------------------------------------------------------------------------------------- word multipliers
// r = A*x + B*(NOTY ? ~y : y) + cin over this lane's limbs, as a flag-free 64-bit multiply-add chain
// (A + B <= 2^32 keeps every partial sum below 2^64); returns the word that leaves the lane
template <bool NOTY, int N>
P2_DEV uint32_t lincomb_lane(BN<N> &r, uint32_t A, const BN<N> &x, uint32_t B, const BN<N> &y, uint32_t cin) {
    uint64_t cy = cin;
    P2_UNROLL for (int j = 0; j < N; j++) {
#if defined(COFHE_HOSTSIM)
        const uint64_t t = (uint64_t)A * x.v[j] + ((uint64_t)B * (NOTY ? ~y.v[j] : y.v[j]) + cy);
        r.v[j] = (uint32_t)t;
        cy = t >> 32;
#else
        // two v_mad_u64_u32 on one aligned pair, then ONE 64-bit shift for the carry (the compiler's own code for
        // "t >> 32" is two moves per limb)
        uint64_t t;
        const uint32_t yy = NOTY ? ~y.v[j] : y.v[j];
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"(A), "v"(x.v[j]), "v"(cy) : "vcc");
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(B), "v"(yy) : "vcc");
        r.v[j] = (uint32_t)t;
        asm("v_lshrrev_b64 %0, 32, %1" : "=v"(cy) : "v"(t));
#endif
    }
    return (uint32_t)cy;
}
// r = A*x + B*y; returns the word leaving the top limb (pair-uniform)
template <int N>
P2_DEV uint32_t bn_lincomb_add(PCtx &c, BN<N> &r, uint32_t A, const BN<N> &x, uint32_t B, const BN<N> &y) {
    const uint32_t w = lincomb_lane<false>(r, A, x, B, y, 0u);
    const uint32_t in0 = xl(c, w);
    const uint32_t c2 = bn_lane_add_word(c, r, c.hi ? in0 : 0u);
    return fhi(c, w + c2);
}
// r = A*x - B*y mod 2^(64N) (the caller guarantees 0 <= A*x - B*y): -B*y == B*~y + B over the full width
template <int N>
P2_DEV void bn_lincomb_sub(PCtx &c, BN<N> &r, uint32_t A, const BN<N> &x, uint32_t B, const BN<N> &y) {
    const uint32_t w = lincomb_lane<true>(r, A, x, B, y, c.hi ? 0u : B);
    const uint32_t in0 = xl(c, w);
    (void)bn_lane_add_word(c, r, c.hi ? in0 : 0u);
}
// same, returning the word that leaves the top limb: A*x - B*y == r + (word - B) * 2^(64N)
template <int N>
P2_DEV uint32_t bn_lincomb_sub_carry(PCtx &c, BN<N> &r, uint32_t A, const BN<N> &x, uint32_t B, const BN<N> &y) {
    const uint32_t w = lincomb_lane<true>(r, A, x, B, y, c.hi ? 0u : B);
    const uint32_t in0 = xl(c, w);
    const uint32_t c2 = bn_lane_add_word(c, r, c.hi ? in0 : 0u);
    return fhi(c, w + c2);
}

// ---------------------------------------------------------------------------------------- multiplication
// (L, H) = x * y: L the low 2N limbs, H the high 2N limbs.  Every lane multiplies its OWN N limbs of x by
// ALL 2N limbs of y (fetched with 2N quad permutes) -- N * 2N products per lane and no idle lane -- column
// by column into a three-word accumulator; the low lane's partial product P0 (weight 0) and the high lane's
// P1 (weight N) are then summed across the lane boundary.
template <int N>
P2_DEV void bn_mul(PCtx &c, BN<N> &L, BN<N> &H, const BN<N> &x, const BN<N> &y) {
    uint32_t yf[2 * N];
    P2_UNROLL for (int j = 0; j < N; j++) {
        yf[j] = flo(c, y.v[j]);
        yf[N + j] = fhi(c, y.v[j]);
    }
    uint32_t p[3 * N];
#if defined(COFHE_HOSTSIM)
    {
        unsigned __int128 acc = 0;
        P2_UNROLL for (int k = 0; k < 3 * N; k++) {
            P2_UNROLL for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 0 && j < 2 * N) acc += (uint64_t)x.v[i] * yf[j];
            }
            p[k] = (uint32_t)acc;
            acc >>= 32;
        }
    }
#else
    {
        // acc = (a1:a0) in an aligned pair, a2 the overflow count of the column.  Every product is one
        // v_mad_u64_u32 whose carry-out goes to an SGPR pair of its own and one v_addc_co_u32 that collects it
        // two or more instructions later (2 wait states between a VALU write of an SGPR and its VALU use).
        uint64_t acc = 0;
        uint32_t a2 = 0;
        P2_UNROLL for (int k = 0; k < 3 * N - 1; k++) {
            const int i0 = k - (2 * N - 1) > 0 ? k - (2 * N - 1) : 0, i1 = k < N - 1 ? k : N - 1;
            const int cnt = i1 - i0 + 1;
            uint64_t co[N];
            P2_UNROLL for (int i = i0; i <= i1; i++)
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(co[i - i0]) : "v"(x.v[i]), "v"(yf[k - i]));
            if (cnt < 3) asm volatile("s_nop 1");
            P2_UNROLL for (int i = i0; i <= i1; i++) {
                uint64_t dummy;
                asm volatile("v_addc_co_u32 %0, %1, 0, %0, %2" : "+v"(a2), "=s"(dummy) : "s"(co[i - i0]));
            }
            p[k] = (uint32_t)acc;
            asm volatile("v_lshrrev_b64 %0, 32, %0" : "+v"(acc));
            acc |= (uint64_t)a2 << 32;
            a2 = 0;
        }
        p[3 * N - 1] = (uint32_t)acc;
    }
#endif
    // x*y = P0 + P1 * B^N (P0 from the low lane, P1 from the high lane, 3N limbs each):
    //   limbs [0, N)   = P0[0, N)                                  -> L, low lane
    //   limbs [N, 2N)  = P0[N, 2N) + P1[0, N)                      -> L, high lane
    //   limbs [2N, 3N) = P0[2N, 3N) + P1[N, 2N) + carry            -> H, low lane
    //   limbs [3N, 4N) = P1[2N, 3N) + carry                        -> H, high lane
    // each lane adds "mine" and "the partner's" block of the same position
    BN<N> mid_a, mid_b;     // high lane: P0[N,2N) (partner) + P1[0,N) (mine); low lane: P0[2N,3N) (mine) + P1[N,2N) (partner)
    // the low lane needs P1[N, 2N) = the partner's p[N + j]; the high lane needs P0[N, 2N) = the partner's p[N + j]:
    // the same exchanged register serves both
    P2_UNROLL for (int j = 0; j < N; j++) {
        const uint32_t part = xl(c, p[N + j]);
        mid_a.v[j] = c.hi ? part : p[2 * N + j];
        mid_b.v[j] = c.hi ? p[j] : part;
    }
    BN<N> mid;
    uint32_t cy = 0;
    P2_UNROLL for (int j = 0; j < N; j++) {
        const uint64_t t = (uint64_t)mid_a.v[j] + mid_b.v[j] + cy;
        mid.v[j] = (uint32_t)t;
        cy = (uint32_t)(t >> 32);
    }
    // high lane: mid = limbs [N, 2N), carry cy into limb 2N (the low lane's block); low lane: mid = limbs [2N, 3N)
    // before that carry, its own carry cy goes into limb 3N (the high lane's top block)
    const uint32_t cy_hi_to_lo = xl(c, cy);                 // low lane receives the high lane's carry
    uint32_t c_lo = 0;
    {
        BN<N> t = mid;
        const uint32_t w = c.hi ? 0u : cy_hi_to_lo;
        c_lo = bn_lane_add_word(c, t, w);
        mid = t;
    }
    const uint32_t lo_total = cy + c_lo;                    // low lane: carry out of limbs [2N, 3N)
    const uint32_t to_top = xl(c, lo_total);                // high lane receives it
    BN<N> top;
    P2_UNROLL for (int j = 0; j < N; j++) top.v[j] = p[2 * N + j];      // high lane: P1[2N, 3N)
    (void)bn_lane_add_word(c, top, c.hi ? to_top : 0u);
    P2_UNROLL for (int j = 0; j < N; j++) {
        L.v[j] = c.hi ? mid.v[j] : p[j];
        H.v[j] = c.hi ? top.v[j] : mid.v[j];
    }
}

// ---------------------------------------------------------------------------------------- indexed limbs
// limb i of x for a RUN-TIME index that is uniform over the wavefront (a scalar jump, then one quad permute);
// i outside [0, 2N) gives 0
#define P2_CASE(K) case K: if (K < 2 * N) return bn_limb<(K < 2 * N ? K : 0)>(c, x); break;
template <int N>
P2_DEV uint32_t bn_limb_at(PCtx &c, const BN<N> &x, int i) {
    static_assert(2 * N <= 40, "limb switch covers 40 limbs");
    switch (i) {
        P2_CASE(0) P2_CASE(1) P2_CASE(2) P2_CASE(3) P2_CASE(4) P2_CASE(5) P2_CASE(6) P2_CASE(7) P2_CASE(8) P2_CASE(9)
        P2_CASE(10) P2_CASE(11) P2_CASE(12) P2_CASE(13) P2_CASE(14) P2_CASE(15) P2_CASE(16) P2_CASE(17) P2_CASE(18) P2_CASE(19)
        P2_CASE(20) P2_CASE(21) P2_CASE(22) P2_CASE(23) P2_CASE(24) P2_CASE(25) P2_CASE(26) P2_CASE(27) P2_CASE(28) P2_CASE(29)
        P2_CASE(30) P2_CASE(31) P2_CASE(32) P2_CASE(33) P2_CASE(34) P2_CASE(35) P2_CASE(36) P2_CASE(37) P2_CASE(38) P2_CASE(39)
        default: break;
    }
    return 0u;
}
#undef P2_CASE
// x.limb[i] = w for a wave-uniform run-time index (pairs that must not store pass enable == false)
#define P2_CASE(K) case K: if (K < 2 * N) { const int jj = K % N; if (enable && c.hi == (K / N)) x.v[jj] = w; } break;
template <int N>
P2_DEV void bn_set_limb_at(PCtx &c, BN<N> &x, int i, uint32_t w, bool enable) {
    switch (i) {
        P2_CASE(0) P2_CASE(1) P2_CASE(2) P2_CASE(3) P2_CASE(4) P2_CASE(5) P2_CASE(6) P2_CASE(7) P2_CASE(8) P2_CASE(9)
        P2_CASE(10) P2_CASE(11) P2_CASE(12) P2_CASE(13) P2_CASE(14) P2_CASE(15) P2_CASE(16) P2_CASE(17) P2_CASE(18) P2_CASE(19)
        P2_CASE(20) P2_CASE(21) P2_CASE(22) P2_CASE(23) P2_CASE(24) P2_CASE(25) P2_CASE(26) P2_CASE(27) P2_CASE(28) P2_CASE(29)
        P2_CASE(30) P2_CASE(31) P2_CASE(32) P2_CASE(33) P2_CASE(34) P2_CASE(35) P2_CASE(36) P2_CASE(37) P2_CASE(38) P2_CASE(39)
        default: break;
    }
}
#undef P2_CASE

// r = s + q * (NOTD ? ~d : d) + cin over this lane's limbs (flag-free): q*d + s + carry < 2^64; returns the word
// leaving the lane.  The product and the limb of s are both folded in by v_mad_u64_u32 (the second one multiplies
// by the literal 1), so a limb costs three instructions.
template <bool NOTD, int N>
P2_DEV uint32_t mulacc_lane(BN<N> &r, const BN<N> &s, uint32_t q, const BN<N> &d, uint32_t cin) {
    uint64_t cy = cin;
    P2_UNROLL for (int j = 0; j < N; j++) {
        const uint32_t dd = NOTD ? ~d.v[j] : d.v[j];
#if defined(COFHE_HOSTSIM)
        const uint64_t t = (uint64_t)q * dd + s.v[j] + cy;
        r.v[j] = (uint32_t)t;
        cy = t >> 32;
#else
        uint64_t t;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"(q), "v"(dd), "v"(cy) : "vcc");
        asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(t) : "v"(s.v[j]) : "vcc");
        r.v[j] = (uint32_t)t;
        asm("v_lshrrev_b64 %0, 32, %1" : "=v"(cy) : "v"(t));
#endif
    }
    return (uint32_t)cy;
}
// r = s - q*d mod 2^(64N); returns the word leaving the top limb: s - q*d == r + (word - q - 1) * 2^(64N)
// (s - q d == s + q ~d + q - q 2^(64N); the "+ q" enters as the low lane's carry-in ... plus the 1 of the
// two's complement folded differently: see bn_rem)
template <int N>
P2_DEV uint32_t bn_submul_word(PCtx &c, BN<N> &r, const BN<N> &s, uint32_t q, const BN<N> &d) {
    // s - q d = s + q (2^(64N) - 1 - d) + q - q 2^(64N) = [s + q ~d + q] - q 2^(64N)
    const uint32_t w = mulacc_lane<true>(r, s, q, d, c.hi ? 0u : q);
    const uint32_t in0 = xl(c, w);
    const uint32_t c2 = bn_lane_add_word(c, r, c.hi ? in0 : 0u);
    return fhi(c, w + c2);              // s - q d == r + (word - q) * 2^(64N)
}

// ---------------------------------------------------------------------------------------- remainder (Knuth D)
// R = (L + H * 2^(64N)) mod D for H < D.  D is normalised per pair (bit shift below 32 per lane; the LIMB part of
// the shift must be the same for every participating pair of the wavefront -- generic operands have their top bit
// in the same limb -- otherwise `ok` is cleared for that pair and its result is meaningless).  Per quotient
// digit: an f64 estimate from the three leading limbs (never below the digit, one above with probability
// ~2^-17 -> add-back), S -= q D over 2N limbs, S shifted up one limb taking the next limb of L.
template <int N>
P2_DEV void bn_rem(PCtx &c, BN<N> &R, const BN<N> &L, const BN<N> &H, const BN<N> &D, bool active, bool &ok) {
    const int db = bn_bitlen(c, D);
    int s = 64 * N - db;                                     // total left shift that normalises D
    if (db == 0) { P2_LEAVE(ok, "pair.hpp:512"); s = 0; }
    // uniform limb shift: the smallest over the participating pairs (pairs that would need more are dropped)
    int ls = s >> 5;
    {
        int best = 0;                                        // find max k such that all active pairs have ls >= k (k <= 2)
        if (!wave_any(c, active && ok && ls < 1)) best = 1;
        if (best == 1 && !wave_any(c, active && ok && ls < 2)) best = 2;
        if (active && ok && ls > best) P2_LEAVE(ok, "pair.hpp:519");           // unusually short divisor: not on the fast path
        ls = best;
    }
    const int bs = ok ? (s - 32 * ls) : 0;                   // per-pair bit shift, 0 .. 31
    // shift D, H, L left by 32 ls + bs: (H:L) as a 4N-limb number
    BN<N> Dn = D, Hn = H, Ln = L;
    for (int k = 0; k < ls; k++) {
        Dn = bn_shl_limb(c, Dn);
        const uint32_t up = bn_limb<2 * N - 1>(c, Ln);        // top limb of L moves into H
        Hn = bn_shl_limb(c, Hn);
        Hn.v[0] = c.hi ? Hn.v[0] : up;
        Ln = bn_shl_limb(c, Ln);
    }
    {
        const uint32_t ltop = bn_limb<2 * N - 1>(c, Ln);
        Dn = bn_shl_small(c, Dn, bs);
        Hn = bn_shl_small(c, Hn, bs);
        Ln = bn_shl_small(c, Ln, bs);
        const uint32_t carry_in = bs ? (ltop >> ((32 - bs) & 31)) : 0u;
        Hn.v[0] |= c.hi ? 0u : carry_in;
    }
    const uint32_t d1 = bn_limb<2 * N - 1>(c, Dn), d0 = bn_limb<2 * N - 2>(c, Dn);
    const double rd = 1.0 / ((double)d1 * 4294967296.0 + (double)d0);
    BN<N> S = Hn;
    uint32_t top = 0;
    for (int i = 2 * N - 1; i >= 0; i--) {
        // S = S * 2^32 + L[i]
        top = bn_limb<2 * N - 1>(c, S);
        const uint32_t nxt = bn_limb_at(c, Ln, i);
        S = bn_shl_limb(c, S);
        S.v[0] = c.hi ? S.v[0] : nxt;
        const uint32_t s1 = bn_limb<2 * N - 1>(c, S), s0 = bn_limb<2 * N - 2>(c, S);
        double x = (((double)top * 4294967296.0 + (double)s1) * 4294967296.0 + (double)s0) * rd;
        x += x * 1.7763568394002505e-15;                     // (1 + 2^-49): never below the true digit
        uint64_t qd = (uint64_t)x;
        if (qd > 0xFFFFFFFFull) qd = 0xFFFFFFFFull;
        BN<N> T;
        const uint32_t cw = bn_submul_word(c, T, S, (uint32_t)qd, Dn);
        int64_t nt = (int64_t)top + (int64_t)cw - (int64_t)qd;        // top word of S - q D: 0, or negative if q is too large
        S = T;
        for (int fix = 0; fix < 4 && wave_any(c, nt < 0); fix++) {     // add-back (rare)
            BN<N> U;
            const uint32_t cy = bn_add(c, U, S, Dn);
            if (nt < 0) {
                S = U;
                nt += (int64_t)cy;
            }
        }
        if (nt != 0) P2_LEAVE(ok, "pair.hpp:567");                               // estimate off by more than the add-backs: not on the fast path
    }
    // remainder = S >> (32 ls + bs)
    BN<N> Rr = bn_shr_small(c, S, bs);
    for (int k = 0; k < ls; k++) Rr = bn_shr_limb(c, Rr);
    R = Rr;
}

// ---------------------------------------------------------------------------------------- exact division (2-adic)
// Q = (numerator >> pre) / D mod 2^(32 nq) for an EXACT division (D > 0 divides it; W = the low 2N limbs of the
// numerator, wtop its limb 2N; nq <= 2N wave-uniform).  Q = W D^-1 mod 2^(32 nq) needs only the low limbs:
// with D odd (trailing zero bits are shifted out of both first) q_i = S[0] * D^-1 mod 2^32, S <- (S - q_i D) / 2^32.
template <int N>
P2_DEV void bn_divexact(PCtx &c, BN<N> &Q, const BN<N> &W, uint32_t wtop, int pre, const BN<N> &D, int nq, bool &ok) {
    // numerator = (W + wtop * 2^(64N) + ...) >> pre; the shift by pre + (trailing zeros of D) pulls its top bits
    // from wtop, so all 2N limbs of the shifted window are exact as long as the total shift stays within a limb
    const uint32_t d0raw = bn_limb<0>(c, D);
    if (d0raw == 0) P2_LEAVE(ok, "divexact: 32 trailing zero bits");
    const int tz = d0raw ? __builtin_ctz(d0raw) : 0;
    const int sh = pre + tz;
    if (sh > 31) P2_LEAVE(ok, "divexact: shift beyond a limb");
    const BN<N> Dn = bn_shr_small(c, D, tz);
    BN<N> S = bn_shr_small(c, W, sh & 31);
    S.v[N - 1] |= (c.hi && (sh & 31)) ? (wtop << ((32 - sh) & 31)) : 0u;
    const uint32_t d0 = bn_limb<0>(c, Dn);
    uint32_t dinv = d0;                                      // d0 * d0 == 1 (mod 8); each Newton step doubles the valid bits
    P2_UNROLL for (int i = 0; i < 4; i++) dinv *= 2u - d0 * dinv;
    bn_zero(Q);
    for (int i = 0; i < nq; i++) {
        const uint32_t q = bn_limb<0>(c, S) * dinv;
        BN<N> T;
        (void)bn_submul_word(c, T, S, q, Dn);                // low limb of T is 0
        S = bn_shr_limb(c, T);
        bn_set_limb_at(c, Q, i, q, true);
    }
}

// ---------------------------------------------------------------------------------------- word helpers
// x mod m for a 28-bit modulus m: sum of limb * (2^(32 j) mod m), two 64-bit accumulators (even / odd limbs
// cannot overflow: 10 terms below 2^60 each), the high lane's sum weighted by 2^(32 N) mod m.  pw[j] = 2^(32 j) mod m
// for j <= N (a table the caller keeps in constant memory or registers).  Pair-uniform result.
template <int N>
P2_DEV uint32_t bn_mod_small(PCtx &c, const BN<N> &x, uint32_t m, const uint32_t *pw) {
    uint64_t e = 0, o = 0;
    P2_UNROLL for (int j = 0; j < N; j++) {
        if (j & 1) o += (uint64_t)x.v[j] * pw[j];
        else e += (uint64_t)x.v[j] * pw[j];
    }
    const uint32_t mine = (uint32_t)((e % m + o % m) % m);
    const uint32_t lo = flo(c, mine), hi = fhi(c, mine);
    return (uint32_t)((lo + (uint64_t)hi * pw[N]) % m);
}
// x mod w for an arbitrary 32-bit w > 0 (Horner with a reciprocal of w: rare paths only); pair-uniform
template <int N>
P2_DEV uint32_t bn_mod_word(PCtx &c, const BN<N> &x, uint32_t w) {
    const cofhe::WordDiv wd = cofhe::worddiv_make(w);
    uint32_t r = 0, rem;
    // high lane first, then the low lane continues from its remainder
    P2_UNROLL for (int j = N - 1; j >= 0; j--) {
        (void)cofhe::worddiv_divmod(wd, ((uint64_t)r << 32) | x.v[j], rem);
        r = rem;
    }
    const uint32_t rh = fhi(c, r);
    uint32_t r2 = rh;
    P2_UNROLL for (int j = N - 1; j >= 0; j--) {
        (void)cofhe::worddiv_divmod(wd, ((uint64_t)r2 << 32) | x.v[j], rem);
        r2 = rem;
    }
    return flo(c, r2);
}
// q = x / w (exact or not), returns x mod w; w > 0 (rare paths)
template <int N>
P2_DEV uint32_t bn_divrem_word(PCtx &c, BN<N> &q, const BN<N> &x, uint32_t w) {
    const cofhe::WordDiv wd = cofhe::worddiv_make(w);
    uint32_t r = 0, rem;
    P2_UNROLL for (int j = N - 1; j >= 0; j--) {          // meaningful in the high lane
        (void)cofhe::worddiv_divmod(wd, ((uint64_t)r << 32) | x.v[j], rem);
        r = rem;
    }
    const uint32_t rh = fhi(c, r);                         // remainder of the high half enters the low half
    uint32_t r2 = c.hi ? 0u : rh;
    P2_UNROLL for (int j = N - 1; j >= 0; j--) {
        const uint64_t qq = cofhe::worddiv_divmod(wd, ((uint64_t)r2 << 32) | x.v[j], rem);
        q.v[j] = (uint32_t)qq;
        r2 = rem;
    }
    return flo(c, r2);
}
// r = x * w (word); returns the word leaving the top limb
template <int N>
P2_DEV uint32_t bn_mul_word(PCtx &c, BN<N> &r, const BN<N> &x, uint32_t w) {
    BN<N> z;
    bn_zero(z);
    return bn_lincomb_add(c, r, w, x, 0u, z);
}

}  // namespace cofhe2
