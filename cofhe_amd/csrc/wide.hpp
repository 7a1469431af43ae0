// wide.hpp -- multi-precision integers spread over a WHOLE wavefront: the latency layout.
//
// The throughput layout (lane.hpp / mp.hpp) gives a form-operation 8 lanes and 5 limbs per lane, eight operations per
// wavefront: every multi-limb primitive is ~5-10x the instructions a wavefront-wide layout needs, which is the right trade
// when 32 768 compositions are in flight and the wrong one when a decryption is ONE ladder of ~1100 dependent compositions
// (cofhe_hip_decrypt_records of a tensor that shares its c1, scalar decrypt(): 0.3 s at any size until round 4).  Here one
// number is spread over the 64 lanes of a wavefront, TWO 32-bit limbs per lane (lane L holds limbs 2L and 2L+1: 128 limbs =
// 4096 bits of capacity, one type for every width the composition needs), and a wavefront runs one composition:
//   * neighbour hand-offs are DPP wave shifts (wave_shr:1 / wave_shl:1), broadcasts of a limb are v_readlane into SGPRs;
//   * carries are resolved for the whole number at once from two 64-bit ballots (generate / propagate) and one 64-bit
//     scalar add -- the same trick as mp_resolve, but over 64 lanes it is the only carry logic there is;
//   * all control flow is WAVE-uniform (bit lengths, comparisons and digit estimates come out of ballots and readlanes),
//     so there is no divergence, no mailbox, no s_barrier; the Lehmer batch (mp.hpp: lehmer_batch, the same function) runs
//     uniformly on the wavefront itself.
//
// COFHE_HOSTSIM: the lane values become 64-element vectors with element-wise operators and the cross-lane primitives become
// loops, so the very same source runs single-threaded on the host (tests/hostsim/simw.cpp) -- possible because nothing
// here branches on a per-lane value.
#pragma once
#include "mp.hpp"

namespace cofhe {
namespace wide {

constexpr int WL = 64;                 // lanes
constexpr int WLIMBS = 2 * WL;         // limbs of capacity

#if defined(COFHE_HOSTSIM)
// ---------------------------------------------------------------------------- host emulation: lane vectors
template <typename T>
struct LV {
    T v[WL];
    LV() {
        for (int i = 0; i < WL; i++) v[i] = T();
    }
    LV(T s) {                          // broadcast
        for (int i = 0; i < WL; i++) v[i] = s;
    }
};
#define CF_LV_BIN(op)                                                                 \
    template <typename T>                                                             \
    inline LV<T> operator op(const LV<T> &x, const LV<T> &y) {                        \
        LV<T> r;                                                                      \
        for (int i = 0; i < WL; i++) r.v[i] = (T)(x.v[i] op y.v[i]);                  \
        return r;                                                                     \
    }                                                                                 \
    template <typename T>                                                             \
    inline LV<T> operator op(const LV<T> &x, T y) { return x op LV<T>(y); }           \
    template <typename T>                                                             \
    inline LV<T> operator op(T x, const LV<T> &y) { return LV<T>(x) op y; }
CF_LV_BIN(+) CF_LV_BIN(-) CF_LV_BIN(*) CF_LV_BIN(&) CF_LV_BIN(|) CF_LV_BIN(^)
#undef CF_LV_BIN
#define CF_LV_CMP(op)                                                                 \
    template <typename T>                                                             \
    inline LV<bool> operator op(const LV<T> &x, const LV<T> &y) {                     \
        LV<bool> r;                                                                   \
        for (int i = 0; i < WL; i++) r.v[i] = x.v[i] op y.v[i];                       \
        return r;                                                                     \
    }                                                                                 \
    template <typename T>                                                             \
    inline LV<bool> operator op(const LV<T> &x, T y) { return x op LV<T>(y); }
CF_LV_CMP(==) CF_LV_CMP(!=) CF_LV_CMP(<) CF_LV_CMP(>) CF_LV_CMP(<=) CF_LV_CMP(>=)
#undef CF_LV_CMP
template <typename T>
inline LV<T> operator~(const LV<T> &x) {
    LV<T> r;
    for (int i = 0; i < WL; i++) r.v[i] = (T)~x.v[i];
    return r;
}
inline LV<bool> operator!(const LV<bool> &x) {
    LV<bool> r;
    for (int i = 0; i < WL; i++) r.v[i] = !x.v[i];
    return r;
}
inline LV<bool> operator&&(const LV<bool> &x, const LV<bool> &y) {
    LV<bool> r;
    for (int i = 0; i < WL; i++) r.v[i] = x.v[i] && y.v[i];
    return r;
}
inline LV<bool> operator||(const LV<bool> &x, const LV<bool> &y) {
    LV<bool> r;
    for (int i = 0; i < WL; i++) r.v[i] = x.v[i] || y.v[i];
    return r;
}
template <typename T>
inline LV<T> operator<<(const LV<T> &x, int n) {      // uniform shift amounts only
    LV<T> r;
    for (int i = 0; i < WL; i++) r.v[i] = (T)(x.v[i] << n);
    return r;
}
template <typename T>
inline LV<T> operator>>(const LV<T> &x, int n) {
    LV<T> r;
    for (int i = 0; i < WL; i++) r.v[i] = (T)(x.v[i] >> n);
    return r;
}
using V32 = LV<uint32_t>;
using V64 = LV<uint64_t>;
using VM = LV<bool>;
#define CF_W inline
inline V32 lane_id() {
    V32 r;
    for (int i = 0; i < WL; i++) r.v[i] = (uint32_t)i;
    return r;
}
inline V32 sel(const VM &m, const V32 &x, const V32 &y) {
    V32 r;
    for (int i = 0; i < WL; i++) r.v[i] = m.v[i] ? x.v[i] : y.v[i];
    return r;
}
inline V64 mad(const V32 &x, const V32 &y, const V64 &c) {
    V64 r;
    for (int i = 0; i < WL; i++) r.v[i] = (uint64_t)x.v[i] * y.v[i] + c.v[i];
    return r;
}
inline V32 lo(const V64 &x) {
    V32 r;
    for (int i = 0; i < WL; i++) r.v[i] = (uint32_t)x.v[i];
    return r;
}
inline V32 hi(const V64 &x) {
    V32 r;
    for (int i = 0; i < WL; i++) r.v[i] = (uint32_t)(x.v[i] >> 32);
    return r;
}
inline V64 mk64(const V32 &l, const V32 &h) {
    V64 r;
    for (int i = 0; i < WL; i++) r.v[i] = ((uint64_t)h.v[i] << 32) | l.v[i];
    return r;
}
inline V64 zext(const V32 &x) { return mk64(x, V32(0u)); }
inline V32 up1(const V32 &x, uint32_t fill) {           // lane i <- lane i - 1
    V32 r;
    r.v[0] = fill;
    for (int i = 1; i < WL; i++) r.v[i] = x.v[i - 1];
    return r;
}
inline V32 down1(const V32 &x, uint32_t fill) {         // lane i <- lane i + 1
    V32 r;
    r.v[WL - 1] = fill;
    for (int i = 0; i + 1 < WL; i++) r.v[i] = x.v[i + 1];
    return r;
}
inline uint32_t rdlane(const V32 &x, int i) { return x.v[i & (WL - 1)]; }
inline uint64_t ballot(const VM &m) {
    uint64_t r = 0;
    for (int i = 0; i < WL; i++) r |= (uint64_t)(m.v[i] ? 1 : 0) << i;
    return r;
}
inline V32 bperm(const V32 &x, const V32 &src) {        // lane i <- lane src[i] (mod 64)
    V32 r;
    for (int i = 0; i < WL; i++) r.v[i] = x.v[src.v[i] & (WL - 1)];
    return r;
}
inline VM lane_bit(uint64_t mask) {                     // bit i of a wave mask, per lane
    VM r;
    for (int i = 0; i < WL; i++) r.v[i] = (mask >> i) & 1;
    return r;
}
#else
// ---------------------------------------------------------------------------- gfx950
using V32 = uint32_t;
using V64 = uint64_t;
using VM = bool;
#define CF_W __device__ __forceinline__
CF_W V32 lane_id() { return (uint32_t)(threadIdx.x & (WL - 1)); }
CF_W V32 sel(VM m, V32 x, V32 y) { return m ? x : y; }
CF_W V64 mad(V32 x, V32 y, V64 c) { return (uint64_t)x * y + c; }
CF_W V32 lo(V64 x) { return (uint32_t)x; }
CF_W V32 hi(V64 x) { return (uint32_t)(x >> 32); }
CF_W V64 mk64(V32 l, V32 h) { return ((uint64_t)h << 32) | l; }
CF_W V64 zext(V32 x) { return (uint64_t)x; }
CF_W V32 up1(V32 x, uint32_t fill) {                    // wave_shr:1 -- lane i reads lane i - 1, lane 0 keeps `fill`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x138, 0xF, 0xF, false);
}
CF_W V32 down1(V32 x, uint32_t fill) {                  // wave_shl:1 -- lane i reads lane i + 1, lane 63 keeps `fill`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x130, 0xF, 0xF, false);
}
CF_W uint32_t rdlane(V32 x, int i) { return (uint32_t)__builtin_amdgcn_readlane((int)x, i); }
CF_W uint64_t ballot(VM m) { return __builtin_amdgcn_ballot_w64(m); }
CF_W V32 bperm(V32 x, V32 src) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src & (WL - 1)) << 2), (int)x); }
CF_W VM lane_bit(uint64_t mask) { return (mask >> (threadIdx.x & (WL - 1))) & 1ull; }
#endif

// uniform (scalar) helpers
CF_W int clz64u(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
CF_W int clz32u(uint32_t x) { return x ? __builtin_clz(x) : 32; }

// ---------------------------------------------------------------------------- numbers
struct WN {            // limbs 2L (a) and 2L+1 (b) of lane L; non-negative
    V32 a, b;
};
struct SW {            // sign-magnitude; neg is wave-uniform; a zero magnitude may carry either flag
    WN m;
    int neg;
};

CF_W WN w_zero() { return WN{V32(0u), V32(0u)}; }
CF_W WN w_word(uint32_t w) { return WN{sel(lane_id() == V32(0u), V32(w), V32(0u)), V32(0u)}; }
CF_W WN w_word64(uint64_t w) {
    const VM l0 = lane_id() == V32(0u);
    return WN{sel(l0, V32((uint32_t)w), V32(0u)), sel(l0, V32((uint32_t)(w >> 32)), V32(0u))};
}
CF_W uint64_t w_nonzero_lanes(const WN &x) { return ballot((x.a | x.b) != V32(0u)); }
CF_W bool w_is_zero(const WN &x) { return w_nonzero_lanes(x) == 0; }
CF_W bool w_is_word(const WN &x, uint32_t w) {
    const VM l0 = lane_id() == V32(0u);
    return ballot(((x.a ^ sel(l0, V32(w), V32(0u))) | x.b) != V32(0u)) == 0;
}
// limb k (uniform k, 0 <= k < 128)
CF_W uint32_t w_limb(const WN &x, int k) { return (k & 1) ? rdlane(x.b, k >> 1) : rdlane(x.a, k >> 1); }
// number of significant bits (0 for zero)
CF_W int w_bitlen(const WN &x) {
    const uint64_t nz = w_nonzero_lanes(x);
    if (nz == 0) return 0;
    const int t = 63 - clz64u(nz);
    const uint32_t hb = rdlane(x.b, t), ha = rdlane(x.a, t);
    return 64 * t + (hb ? 64 - clz32u(hb) : 32 - clz32u(ha));
}
// -1 / 0 / +1
CF_W int w_cmp(const WN &x, const WN &y) {
    const uint64_t d = ballot((x.a != y.a) || (x.b != y.b));
    if (d == 0) return 0;
    const int t = 63 - clz64u(d);
    const uint64_t xv = ((uint64_t)rdlane(x.b, t) << 32) | rdlane(x.a, t), yv = ((uint64_t)rdlane(y.b, t) << 32) | rdlane(y.a, t);
    return xv > yv ? 1 : -1;
}
// bits [pos, pos + 64) of x (uniform pos >= 0; beyond the capacity: zero)
CF_W uint64_t w_bits64(const WN &x, int pos) {
    const int k = pos >> 5, o = pos & 31;
    const uint32_t l0 = k < WLIMBS ? w_limb(x, k) : 0u, l1 = k + 1 < WLIMBS ? w_limb(x, k + 1) : 0u, l2 = k + 2 < WLIMBS ? w_limb(x, k + 2) : 0u;
    const uint64_t low = ((uint64_t)l1 << 32) | l0;
    return o ? ((low >> o) | ((uint64_t)l2 << (64 - o))) : low;
}

// The leading limbs of a PAIR in one go (the remainder sequence asks for both bit lengths and both 64-bit windows every
// round: as two w_bitlen and two w_bits64 that was two ballots and ~16 readlanes with the scalar logic in between, a third
// of a round): one ballot finds the top lane L of the longer number, eight readlanes fetch lanes L and L - 1 of both, the
// rest is scalar.  The view holds bits [base, base + 128) of each number.  A number whose view is empty may still be
// non-zero below it: its length is then asked for the long way (w_bitlen) -- the pair is far apart and takes a
// long-division step anyway.
struct WTop {
    uint64_t xh, xl, yh, yl;      // bits [base + 64, base + 128) and [base, base + 64) of x and of y
    int base;
};
CF_W WTop w_top_pair(const WN &x, const WN &y) {
    WTop t{0, 0, 0, 0, 0};
    const uint64_t nz = ballot(((x.a | x.b) | (y.a | y.b)) != V32(0u));
    if (nz == 0) return t;
    const int L = 63 - clz64u(nz);
    const uint64_t xt = ((uint64_t)rdlane(x.b, L) << 32) | rdlane(x.a, L), yt = ((uint64_t)rdlane(y.b, L) << 32) | rdlane(y.a, L);
    if (L == 0) {
        t.xl = xt;
        t.yl = yt;
        return t;
    }
    t.xh = xt;
    t.yh = yt;
    t.xl = ((uint64_t)rdlane(x.b, L - 1) << 32) | rdlane(x.a, L - 1);
    t.yl = ((uint64_t)rdlane(y.b, L - 1) << 32) | rdlane(y.a, L - 1);
    t.base = 64 * (L - 1);
    return t;
}
// significant bits of the 128-bit view (h, l), 0 for an empty one
CF_W int top_bitlen(uint64_t h, uint64_t l) { return h ? 128 - clz64u(h) : 64 - clz64u(l); }
// bits [s, s + 64) of the view (0 <= s < 128)
CF_W uint64_t top_bits64(uint64_t h, uint64_t l, int s) {
    if (s >= 64) return h >> (s - 64);
    return s ? ((l >> s) | (h << (64 - s))) : l;
}

// ---------------------------------------------------------------------------- carries
// Every lane holds the 64-bit value (b:a) and hands the word `outw` to the lane above it.  One add per lane, then the
// single-bit ripples of all 64 lanes at once: G = lanes whose add overflowed, P = lanes left all ones; the carry INTO each
// lane is (G << 1) rippled through runs of P, which a 64-bit integer add computes.  Returns the bit leaving lane 63.
CF_W uint32_t w_resolve(WN &r, const V32 &outw) {
    const V64 inc = zext(up1(outw, 0u));
    const V64 s = mk64(r.a, r.b) + inc;
    const uint64_t G = ballot(s < inc), P = ballot(s == V64(~0ull));
    const uint64_t y = G << 1;
    const uint64_t sum = P + y;
    const uint64_t C = y | ((sum ^ P) ^ y);
    const V64 t = s + zext(sel(lane_bit(C), V32(1u), V32(0u)));
    r.a = lo(t);
    r.b = hi(t);
    return (uint32_t)((G >> 63) | ((sum < P) ? 1u : 0u));
}

// r = A x + B y (A + B <= 2^32); returns what leaves the top (word of lane 63 + ripple)
CF_W uint32_t w_lincomb_add(WN &r, uint32_t A, const WN &x, uint32_t B, const WN &y) {
    const V64 u0 = mad(V32(A), x.a, mad(V32(B), y.a, V64(0ull)));
    const V64 u1 = mad(V32(A), x.b, mad(V32(B), y.b, zext(hi(u0))));
    r.a = lo(u0);
    r.b = lo(u1);
    const V32 ow = hi(u1);
    const uint32_t top = rdlane(ow, WL - 1);
    return top + w_resolve(r, ow);
}
// r = A x - B y modulo 2^4096 (A + B <= 2^32; the caller guarantees A x >= B y): two's complement, -B y == B ~y + B.
// Returns the word that leaves the top: A x - B y == r + (word - B) 2^4096.
CF_W uint32_t w_lincomb_sub(WN &r, uint32_t A, const WN &x, uint32_t B, const WN &y) {
    const V64 cin = zext(sel(lane_id() == V32(0u), V32(B), V32(0u)));
    const V64 u0 = mad(V32(A), x.a, mad(V32(B), ~y.a, cin));
    const V64 u1 = mad(V32(A), x.b, mad(V32(B), ~y.b, zext(hi(u0))));
    r.a = lo(u0);
    r.b = lo(u1);
    const V32 ow = hi(u1);
    const uint32_t top = rdlane(ow, WL - 1);
    return top + w_resolve(r, ow);
}
CF_W uint32_t w_add(WN &r, const WN &x, const WN &y) { return w_lincomb_add(r, 1u, x, 1u, y); }
CF_W void w_sub(WN &r, const WN &x, const WN &y) { (void)w_lincomb_sub(r, 1u, x, 1u, y); }        // x >= y

// ---------------------------------------------------------------------------- shifts
// x >> n / x << n for uniform n >= 0 (bits shifted past either end are dropped)
CF_W WN w_shr(const WN &x, int n) {
    const int k = n >> 5, o = n & 31;
    WN y;
    if (k == 0) {
        y = x;
    } else {
        // limb m of y = limb m + k of x: lane L wants limbs 2L + k and 2L + 1 + k
        const V32 L = lane_id();
        const int h = k >> 1;
        const V32 s0 = L + V32((uint32_t)h), s1 = L + V32((uint32_t)(h + (k & 1)));
        const V32 p0 = bperm((k & 1) ? x.b : x.a, s0), p1 = bperm((k & 1) ? x.a : x.b, s1);
        y.a = sel(s0 < V32((uint32_t)WL), p0, V32(0u));
        y.b = sel(s1 < V32((uint32_t)WL), p1, V32(0u));
    }
    if (o) {
        const V32 na = down1(y.a, 0u);
        WN z;
        z.a = (y.a >> o) | (y.b << (32 - o));
        z.b = (y.b >> o) | (na << (32 - o));
        return z;
    }
    return y;
}
CF_W WN w_shl(const WN &x, int n) {
    const int k = n >> 5, o = n & 31;
    WN y;
    if (k == 0) {
        y = x;
    } else {
        // limb m of y = limb m - k of x: lane L wants limbs 2L - k and 2L + 1 - k
        const V32 L = lane_id();
        const int h = (k + 1) >> 1;                        // lanes down for the a limb: ceil(k / 2)
        const V32 s0 = L - V32((uint32_t)h), s1 = L - V32((uint32_t)(k >> 1));
        const V32 p0 = bperm((k & 1) ? x.b : x.a, s0), p1 = bperm((k & 1) ? x.a : x.b, s1);
        y.a = sel(L >= V32((uint32_t)h), p0, V32(0u));
        y.b = sel(L >= V32((uint32_t)(k >> 1)), p1, V32(0u));
    }
    if (o) {
        const V32 pb = up1(y.b, 0u);
        WN z;
        z.a = (y.a << o) | (pb >> (32 - o));
        z.b = (y.b << o) | (y.a >> (32 - o));
        return z;
    }
    return y;
}

// ---------------------------------------------------------------------------- multiplication
// r = x * y; nx = lanes of x in use (uniform, >= its length: the loop runs over them).  Schoolbook over the lanes of x:
// the two limbs of lane i are broadcast (v_readlane), y is shifted up one lane per step (DPP), and every lane accumulates
// its four partial products lazily in three 64-bit sums with overflow counters; the sums are folded and the carries
// resolved once at the end.  The product must fit the capacity (the composition's products do: <= 120 limbs).
CF_W WN w_mul(const WN &x, int nx, const WN &y) {
    V64 acc0(0ull), acc1(0ull), acc2(0ull);           // limbs (2L, 2L+1), (2L+1, 2L+2), (2L+2, 2L+3)
    V32 ov0(0u), ov1(0u), ov2(0u);
    V32 ya = y.a, yb = y.b;
    for (int i = 0; i < nx; i++) {
        const V32 xa(rdlane(x.a, i)), xb(rdlane(x.b, i));
        V64 t = mad(xa, ya, acc0);
        ov0 = ov0 + sel(t < acc0, V32(1u), V32(0u));
        acc0 = t;
        t = mad(xa, yb, acc1);
        ov1 = ov1 + sel(t < acc1, V32(1u), V32(0u));
        acc1 = t;
        t = mad(xb, ya, acc1);
        ov1 = ov1 + sel(t < acc1, V32(1u), V32(0u));
        acc1 = t;
        t = mad(xb, yb, acc2);
        ov2 = ov2 + sel(t < acc2, V32(1u), V32(0u));
        acc2 = t;
        ya = up1(ya, 0u);
        yb = up1(yb, 0u);
    }
    // local value of a lane = acc0 + ov0 2^64 + acc1 2^32 + ov1 2^96 + acc2 2^64 + ov2 2^128  ->  w0 + w1 2^64 + w2 2^128
    V64 w0 = acc0 + mk64(V32(0u), lo(acc1));
    V64 c = zext(sel(w0 < acc0, V32(1u), V32(0u)));
    V64 w1 = zext(hi(acc1)) + mk64(V32(0u), ov1) + zext(ov0);          // < 2^33 + 2^32 2^6: no overflow
    V64 w2 = zext(ov2);
    V64 t1 = w1 + c;
    V64 t2 = t1 + acc2;
    w2 = w2 + zext(sel(t2 < t1, V32(1u), V32(0u)));
    w1 = t2;
    // digit of lane L = w0[L] + w1[L - 1] + w2[L - 2] (+ carries)
    const V64 u1 = mk64(up1(lo(w1), 0u), up1(hi(w1), 0u));
    const V64 u2 = mk64(up1(up1(lo(w2), 0u), 0u), up1(up1(hi(w2), 0u), 0u));
    V64 d = w0 + u1;
    V32 cy = sel(d < w0, V32(1u), V32(0u));
    const V64 d2 = d + u2;
    cy = cy + sel(d2 < d, V32(1u), V32(0u));
    WN r{lo(d2), hi(d2)};
    (void)w_resolve(r, cy);
    return r;
}
CF_W int w_lanes(const WN &x) {                        // lanes in use (0 for zero)
    const uint64_t nz = w_nonzero_lanes(x);
    return nz ? 64 - clz64u(nz) : 0;
}
// r = x * y with the shorter operand driving the loop
CF_W WN w_mul(const WN &x, const WN &y) {
    const int nx = w_lanes(x), ny = w_lanes(y);
    return nx <= ny ? w_mul(x, nx, y) : w_mul(y, ny, x);
}

// ---------------------------------------------------------------------------- signed helpers
CF_W SW sw_add(const SW &x, const SW &y) {
    SW r;
    if (x.neg == y.neg) {
        (void)w_add(r.m, x.m, y.m);
        r.neg = x.neg;
    } else {
        const int cm = w_cmp(x.m, y.m);
        if (cm >= 0) {
            w_sub(r.m, x.m, y.m);
            r.neg = cm == 0 ? 0 : x.neg;
        } else {
            w_sub(r.m, y.m, x.m);
            r.neg = y.neg;
        }
    }
    return r;
}
CF_W SW sw_sub(const SW &x, const SW &y) { return sw_add(x, SW{y.m, y.neg ^ 1}); }
CF_W SW sw_mul(const SW &x, const SW &y) { return SW{w_mul(x.m, y.m), x.neg ^ y.neg}; }

// ---------------------------------------------------------------------------- division
// num mod den (den > 0), Knuth D with 32-bit digits.  The divisor is shifted so that its leading bit is the top bit of a
// limb; the running remainder S stays aligned with it and moves up one limb per digit (b <- a, a <- the lane below's b,
// lane 0 takes the next numerator limb).  Per digit: one f64 estimate from the three leading limbs (never below the true
// digit, one above with probability ~2^-17 -> add-back), one linear combination S - q D.  ok = false: something the
// composition never produces (zero divisor) -- the caller falls back.
CF_W WN w_mod(const WN &num, const WN &den, bool &ok) {
    const int nb = w_bitlen(num), db = w_bitlen(den);
    if (db == 0) {
        ok = false;
        return num;
    }
    if (nb < db) return num;
    const int s = (32 - (db & 31)) & 31;                   // leading bit of the divisor to the top of its limb
    const WN D = s ? w_shl(den, s) : den;
    const WN N = s ? w_shl(num, s) : num;                  // nb + s <= capacity (the composition's numerators: <= 2 x 1100 bits)
    const int dl = (db + s) >> 5;                          // limbs of D (its top limb: dl - 1, leading bit set)
    const int nl = (nb + s + 31) >> 5;                     // limbs of N
    const uint32_t d1 = w_limb(D, dl - 1), d0 = dl >= 2 ? w_limb(D, dl - 2) : 0u;
    const double rd = 1.0 / ((double)d1 * 4294967296.0 + (double)d0);
    // S = the top dl limbs of N, aligned with D (limb j of S against limb j of D); `top` = the limb above them
    WN S = w_shr(N, 32 * (nl - dl));
    uint32_t top = 0;
    for (int k = nl - dl;; k--) {                           // quotient digit k
        const uint32_t l1 = w_limb(S, dl - 1), l0 = dl >= 2 ? w_limb(S, dl - 2) : 0u;
        double xq = (((double)top * 4294967296.0 + (double)l1) * 4294967296.0 + (double)l0) * rd;
        xq += xq * 1.7763568394002505e-15;                 // (1 + 2^-49): never below the true digit
        uint64_t qd = xq >= 4294967295.0 ? 0xFFFFFFFFull : (uint64_t)xq;
        if (qd != 0) {
            WN T;
            // S - q D over the dl limbs in use: the two's complement runs over the whole capacity, so what leaves limb dl - 1
            // is read off the limb above it: T's limb dl holds (top-part) -- take the exact view instead:
            //   S + top 2^(32 dl) - q D  =  T (mod 2^4096) with T >= 0 iff the digit was not too large
            WN Sx = S;                                      // S with `top` placed in limb dl
            {
                const V32 L = lane_id();
                const VM at = L == V32((uint32_t)(dl >> 1));
                if (dl & 1) Sx.b = sel(at, V32(top), Sx.b); else Sx.a = sel(at, V32(top), Sx.a);
            }
            (void)w_lincomb_sub(T, 1u, Sx, (uint32_t)qd, D);
            // negative (digit one too large)?  then the two's complement left all ones at the top of the capacity
            for (int fix = 0; fix < 4 && rdlane(T.b, WL - 1) != 0u; fix++) {
                WN U;
                (void)w_add(U, T, D);
                T = U;
                qd--;
            }
            S = T;                                          // < D: limb dl is zero again
        }
        if (k == 0) break;
        // up one limb; the limb that leaves the top of the dl in use becomes `top`
        top = w_limb(S, dl - 1);
        const uint32_t next = w_limb(N, k - 1);
        WN Z;
        Z.b = S.a;
        Z.a = up1(S.b, next);
        // clear what moved above limb dl - 1 (the old top limb now sits in limb dl)
        {
            const V32 L = lane_id();
            const VM at = L == V32((uint32_t)(dl >> 1));
            if (dl & 1) Z.b = sel(at, V32(0u), Z.b); else Z.a = sel(at, V32(0u), Z.a);
        }
        S = Z;
    }
    return s ? w_shr(S, s) : S;
}

// quot = num / den for an EXACT division (den > 0 divides num), 2-adic with 64-bit digits: one digit per lane of the
// quotient, q = N[0] D^-1 mod 2^64, N <- (N - q D) / 2^64.  nq = lanes of quotient to produce.  ok = false when the low 64
// bits of the divisor are zero (never for form coefficients).
CF_W WN w_divexact(const WN &num, const WN &den, int nq, bool &ok) {
    WN Q = w_zero();
    if (nq <= 0) return Q;
    const uint64_t dlow = ((uint64_t)rdlane(den.b, 0) << 32) | rdlane(den.a, 0);
    if (dlow == 0) {
        ok = false;
        return Q;
    }
    const int tz = __builtin_ctzll(dlow);
    const WN D = tz ? w_shr(den, tz) : den;
    WN N = tz ? w_shr(num, tz) : num;
    const uint64_t d0 = ((uint64_t)rdlane(D.b, 0) << 32) | rdlane(D.a, 0);
    uint64_t dinv = d0;                                     // d0 * d0 == 1 (mod 8); each Newton step doubles the valid bits
    for (int i = 0; i < 5; i++) dinv *= 2ull - d0 * dinv;
    const V32 L = lane_id();
    for (int j = 0; j < nq && j < WL; j++) {
        const uint64_t n0 = ((uint64_t)rdlane(N.b, 0) << 32) | rdlane(N.a, 0);
        const uint64_t q = n0 * dinv;
        const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32);
        // N - q D in two's complement: N + q ~D + q.  Per lane the 128-bit product q ~D_L plus N_L and, from the lane below, the
        // upper half of its product.
        const V32 na = ~D.a, nbb = ~D.b;
        const V64 p00 = mad(V32(q0), na, V64(0ull)), p01 = mad(V32(q0), nbb, V64(0ull));
        const V64 p10 = mad(V32(q1), na, V64(0ull)), p11 = mad(V32(q1), nbb, V64(0ull));
        // low 64 bits: p00 + ((p01 + p10) << 32); high 64 bits: p11 + ((p01 + p10) >> 32) + carries
        const V64 mid = p01 + p10;
        const V64 midc = zext(sel(mid < p01, V32(1u), V32(0u)));            // 2^64 of the middle sum
        const V64 lo64 = p00 + mk64(V32(0u), lo(mid));
        const V64 c0 = zext(sel(lo64 < p00, V32(1u), V32(0u)));
        const V64 hi64 = p11 + zext(hi(mid)) + mk64(V32(0u), lo(midc)) + c0;        // < 2^64: the product is below 2^128
        // digit = N_L + lo64 (+ q in lane 0) + hi64 of the lane below
        const V64 nL = mk64(N.a, N.b);
        V64 d = nL + lo64;
        V32 cy = sel(d < nL, V32(1u), V32(0u));
        const V64 qin = mk64(sel(L == V32(0u), V32(q0), V32(0u)), sel(L == V32(0u), V32(q1), V32(0u)));
        V64 d1 = d + qin;
        cy = cy + sel(d1 < d, V32(1u), V32(0u));
        const V64 below = mk64(up1(lo(hi64), 0u), up1(hi(hi64), 0u));
        const V64 d2 = d1 + below;
        cy = cy + sel(d2 < d1, V32(1u), V32(0u));
        WN T{lo(d2), hi(d2)};
        (void)w_resolve(T, cy);
        // lane 0 is now zero; down one lane (the all-ones that the two's complement leaves above the number shift in from
        // lane 63's neighbour as zero: the number itself never reaches the top lanes)
        N.a = down1(T.a, 0u);
        N.b = down1(T.b, 0u);
        // the two's complement of an exact step leaves q 2^4096 above the capacity and nothing inside it but the true
        // difference, whose top lanes are zero
        const VM at = L == V32((uint32_t)j);
        Q.a = sel(at, V32(q0), Q.a);
        Q.b = sel(at, V32(q1), Q.b);
    }
    return Q;
}

}  // namespace wide
}  // namespace cofhe
