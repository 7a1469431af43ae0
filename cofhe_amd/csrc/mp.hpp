// mp.hpp -- multi-precision integers spread over the 8 lanes of a limb group (lane.hpp).
//
// Layout.  Mp<P> is a non-negative integer of P planes; plane p holds limbs
// [40p, 40p+40) and lane gl of the group holds the 5 consecutive limbs
// [40p + 5*gl, 40p + 5*gl + 5) of every plane in registers (v[p][0..4], radix 2^32).
// P = 1 (1280 bits) carries form coefficients a, b and every Euclid variable of the
// reference parameters (|Delta| <= ~2400 bits); P = 2 (2560 bits) carries c and products.
// Widening / narrowing between plane counts moves no data.  Signs live beside the magnitude
// as group-uniform flags (SMp).
//
// Carries.  Each lane runs its 5-limb carry chain in registers and hands ONE word to the next
// lane (DPP row_shr:1); the single-bit ripples that remain are resolved for the whole group
// at once from two ballots (generate / propagate) with an integer add -- no lane-serial loop.
// Subtractions are done in two's complement over the full width, so a linear combination
// A*x - B*y costs one pass and one resolve.
//
// Everything here is group-cooperative: all 8 lanes of a group call every function together.
#pragma once
#include "lane.hpp"

namespace cofhe {

#if defined(COFHE_HOSTSIM)
struct SimStats { long batches, batch_steps, divsteps, muls, divrems, resolves, bits_gained; };
inline SimStats g_stats;
#define CF_STAT(x) do { if (c.gl == 0) { x; } } while (0)
#else
#define CF_STAT(x) do { } while (0)
#endif

// the value, made opaque to the optimiser: it is computed here, unconditionally, and nothing is folded through it
CF_DEV uint32_t opaque(uint32_t v) {
#if defined(COFHE_HOSTSIM)
    asm volatile("" : "+r"(v));
#else
    asm volatile("" : "+v"(v));
#endif
    return v;
}

template <int P>
struct Mp {
    uint32_t v[P][CH];
};

template <int P>
struct SMp {            // sign-magnitude; neg is group-uniform; a zero magnitude may carry either flag
    Mp<P> m;
    int neg;
};

// ---------------------------------------------------------------------------- basics
template <int P>
CF_DEV void mp_zero(Mp<P> &x) {
    CF_UNROLL for (int p = 0; p < P; p++) CF_UNROLL for (int j = 0; j < CH; j++) x.v[p][j] = 0;
}

template <int P>
CF_DEV void mp_set_word(Ctx &c, Mp<P> &x, uint32_t w) {
    mp_zero(x);
    x.v[0][0] = (c.gl == 0) ? w : 0u;
}

template <int Q, int P>
CF_DEV Mp<Q> mp_resize(const Mp<P> &x) {      // zero-extend or truncate (no data movement)
    Mp<Q> r;
    CF_UNROLL for (int p = 0; p < Q; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) r.v[p][j] = (p < P) ? x.v[p < P ? p : 0][j] : 0u;
    return r;
}

template <int P>
CF_DEV void mp_select(Mp<P> &r, bool take_y, const Mp<P> &x, const Mp<P> &y) {
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) r.v[p][j] = take_y ? y.v[p][j] : x.v[p][j];
}

template <int P>
CF_DEV void mp_swap(Mp<P> &x, Mp<P> &y) {
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) {
            uint32_t t = x.v[p][j];
            x.v[p][j] = y.v[p][j];
            y.v[p][j] = t;
        }
}

template <int P>
CF_DEV bool mp_is_zero(Ctx &c, const Mp<P> &x) {
    uint32_t o = 0;
    CF_UNROLL for (int p = 0; p < P; p++) CF_UNROLL for (int j = 0; j < CH; j++) o |= x.v[p][j];
    return ballot8(c, o != 0) == 0;
}

// true when planes >= from are all zero
template <int P>
CF_DEV bool mp_high_planes_zero(Ctx &c, const Mp<P> &x, int from) {
    uint32_t o = 0;
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) o |= (p >= from) ? x.v[p][j] : 0u;
    return ballot8(c, o != 0) == 0;
}

template <int P>
CF_DEV bool mp_is_word(Ctx &c, const Mp<P> &x, uint32_t w) {   // x == w ?
    uint32_t o = 0;
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) {
            uint32_t want = (p == 0 && j == 0 && c.gl == 0) ? w : 0u;
            o |= x.v[p][j] ^ want;
        }
    return ballot8(c, o != 0) == 0;
}

// number of significant bits (0 for zero); group-uniform
template <int P>
CF_DEV int mp_bitlen(Ctx &c, const Mp<P> &x) {
    // most significant non-zero limb of this lane by selects, then ONE count-leading-zeros
    uint32_t top = x.v[0][0], idx = 0;
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) {
            if (p == 0 && j == 0) continue;
            const uint32_t w = x.v[p][j];
            top = w ? w : top;           // later (p, j) are more significant inside a lane
            idx = w ? (uint32_t)(p * PLIMBS + j) : idx;
        }
    const uint32_t best = top ? (idx + (uint32_t)c.gl * CH) * 32u + 32u - (uint32_t)clz32(top) : 0u;
    return (int)group_max(c, best);
}

// -1 / 0 / +1; group-uniform
template <int P>
CF_DEV int mp_cmp(Ctx &c, const Mp<P> &x, const Mp<P> &y) {
    uint32_t key = 0;
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) {
            uint32_t a = x.v[p][j], b = y.v[p][j];
            uint32_t k = ((uint32_t)(p * PLIMBS + c.gl * CH + j + 1) << 1) | (a > b ? 1u : 0u);
            key = (a != b) ? k : key;
        }
    uint32_t m = group_max(c, key);
    return m == 0 ? 0 : ((m & 1) ? 1 : -1);
}

// limb idx of x (0 when idx is outside [0, 40P)); idx group-uniform
template <int P>
CF_DEV uint32_t mp_get_limb(Ctx &c, const Mp<P> &x, int idx) {
    if (idx < 0 || idx >= P * PLIMBS) return 0;
    int p = idx / PLIMBS, w = idx % PLIMBS;
    int lane = w / CH, j = w % CH;
    uint32_t cand = 0;
    CF_UNROLL for (int pp = 0; pp < P; pp++)
        CF_UNROLL for (int jj = 0; jj < CH; jj++) cand = (pp == p && jj == j) ? x.v[pp][jj] : cand;
    return bcast(c, cand, lane);
}

// bits [pos, pos+64) of x, pos >= 0 group-uniform
template <int P>
CF_DEV uint64_t mp_bits64(Ctx &c, const Mp<P> &x, int pos) {
    int i0 = pos >> 5, o = pos & 31;
    uint32_t l0 = mp_get_limb(c, x, i0), l1 = mp_get_limb(c, x, i0 + 1), l2 = mp_get_limb(c, x, i0 + 2);
    uint64_t lo = ((uint64_t)l1 << 32) | l0;
    return o ? ((lo >> o) | ((uint64_t)l2 << (64 - o))) : lo;
}
template <int P>
CF_DEV uint32_t mp_bits32(Ctx &c, const Mp<P> &x, int pos) {
    int i0 = pos >> 5, o = pos & 31;
    uint32_t l0 = mp_get_limb(c, x, i0), l1 = mp_get_limb(c, x, i0 + 1);
    return o ? ((l0 >> o) | (l1 << (32 - o))) : l0;
}

// ---------------------------------------------------------------------------- carry resolve
// r holds per-lane chunk sums, hi[p] the word each lane hands to the next chunk of plane p.
// Returns the word leaving the top plane.
template <int P>
CF_DEV uint32_t mp_resolve(Ctx &c, Mp<P> &r, const uint32_t (&hi)[P]) {
    uint32_t plane_word = 0, plane_bit = 0;     // what enters lane 0 of the next plane
    CF_STAT(g_stats.resolves += P);
    CF_UNROLL for (int p = 0; p < P; p++) {
        uint32_t inc = shfl_up1(c, hi[p], plane_word);
        uint64_t t = (uint64_t)r.v[p][0] + inc + ((c.gl == 0) ? plane_bit : 0u);
        r.v[p][0] = (uint32_t)t;
        uint32_t cy = (uint32_t)(t >> 32);
        uint32_t all = r.v[p][0];
        CF_UNROLL for (int j = 1; j < CH; j++) {
            t = (uint64_t)r.v[p][j] + cy;
            r.v[p][j] = (uint32_t)t;
            cy = (uint32_t)(t >> 32);
            all &= r.v[p][j];
        }
        // cy <= 2 only in lane 0 (word + bit); it is folded into the generate mask as 1 and the
        // (impossible for our operand ranges) value 2 is excluded by the callers' bounds
        uint32_t gm = ballot8(c, cy != 0);
        uint32_t pm = ballot8(c, all == 0xFFFFFFFFu && cy == 0);
        uint32_t y = gm << 1;
        uint32_t cin = y | (((pm + y) ^ pm) ^ y);
        // the incoming bit almost never travels beyond the first limb of a chunk (that needs limb 0 == 2^32 - 1):
        // one add, and the rest of the chain only when some lane needs it
        uint32_t mine = (cin >> c.gl) & 1u;
        {
            const uint32_t s0 = r.v[p][0] + mine;
            mine = (s0 < mine) ? 1u : 0u;
            r.v[p][0] = s0;
        }
        if (CF_UNLIKELY(any_lane(c, mine != 0))) {
            CF_UNROLL for (int j = 1; j < CH; j++) {
                uint32_t s = r.v[p][j] + mine;
                mine = (s < mine) ? 1u : 0u;
                r.v[p][j] = s;
            }
        }
        plane_word = shfl_mirror(c, hi[p]);      // lane 0 <- lane 7 (only lane 0 uses it)
        plane_bit = (cin >> G) & 1u;
    }
    return bcast_first(c, plane_word) + plane_bit;
}

// The same for sums of NON-NEGATIVE terms (products, A*x + B*y): after a lane has added the word its neighbour hands over,
// a carry can only travel beyond limb 1 of the chunk -- or out of the chunk -- through a limb that is all ones, which random
// limbs are with probability 2^-32.  (The generate / propagate machinery above is for the two's-complement differences,
// whose cancelled high limbs are all ones by construction: there the single-bit ripple is the rule.)  So: one add, one
// add-with-carry, and the general resolve only when some lane of the wavefront has an all-ones limb 1: ~7 instructions per
// plane instead of ~37, twice per Euclid round (the cofactor updates) and once per plane of every product.
template <int P>
CF_DEV uint32_t mp_resolve_sparse(Ctx &c, Mp<P> &r, const uint32_t (&hi)[P]) {
    uint32_t ones = 0;
    CF_UNROLL for (int p = 0; p < P; p++) ones |= (r.v[p][1] == 0xFFFFFFFFu) ? 1u : 0u;
    if (CF_UNLIKELY(any_lane(c, ones != 0))) return mp_resolve(c, r, hi);
    CF_STAT(g_stats.resolves += P);
    uint32_t plane_word = 0;
    CF_UNROLL for (int p = 0; p < P; p++) {
        const uint32_t inc = shfl_up1(c, hi[p], plane_word);
        const uint32_t s0 = r.v[p][0] + inc;
        r.v[p][1] += (s0 < inc) ? 1u : 0u;          // cannot wrap: limb 1 is not all ones
        r.v[p][0] = s0;
        plane_word = shfl_mirror(c, hi[p]);         // lane 0 <- lane 7 (only lane 0 uses it)
    }
    return bcast_first(c, plane_word);
}

// r = x + y ; returns the carry out of the top plane
template <int P>
CF_DEV uint32_t mp_add(Ctx &c, Mp<P> &r, const Mp<P> &x, const Mp<P> &y) {
    uint32_t hi[P];
    CF_UNROLL for (int p = 0; p < P; p++) {
        uint32_t cy = 0;
        CF_UNROLL for (int j = 0; j < CH; j++) {
            uint64_t t = (uint64_t)x.v[p][j] + y.v[p][j] + cy;
            r.v[p][j] = (uint32_t)t;
            cy = (uint32_t)(t >> 32);
        }
        hi[p] = cy;
    }
    return mp_resolve(c, r, hi);
}

// One plane of A*x + B*y' (+ cin at the bottom): the CH products pairs are independent 64-bit values
// t_j = A x_j + B y'_j (fits: A + B <= 2^32), and the limbs are lo(t_j) + hi(t_j-1) + carry -- a plain
// add-with-carry chain, 2 multiply-adds + 1 add per limb and no 64-bit repacking.  Returns the word
// that leaves the lane (< 2^32 because the whole sum fits CH + 1 words).
template <bool NOTY>
CF_DEV uint32_t lincomb_plane(uint32_t (&r)[CH], uint32_t A, const uint32_t (&x)[CH], uint32_t B, const uint32_t (&y)[CH],
                              uint32_t cin) {
    uint64_t t[CH];
    CF_UNROLL for (int j = 0; j < CH; j++)
        t[j] = (uint64_t)A * x[j] + (uint64_t)B * (NOTY ? (uint32_t)~y[j] : y[j]);
    uint32_t prev = cin;
#if defined(COFHE_HOSTSIM)
    uint32_t carry = 0;
    CF_UNROLL for (int j = 0; j < CH; j++) {
        const uint64_t s = (uint64_t)(uint32_t)t[j] + prev + carry;
        r[j] = (uint32_t)s;
        carry = (uint32_t)(s >> 32);
        prev = (uint32_t)(t[j] >> 32);
    }
    return prev + carry;
#else
    // the compiler would re-pack this chain into 64-bit adds (a move and a 64-bit add per limb); keep it
    // as v_add_co / v_addc_co.  s_nop 1: two wait states between a VALU carry-out and its VALU consumer.
    static_assert(CH == 5, "carry chain written for 5 limbs per lane");
    uint32_t l0 = (uint32_t)t[0], l1 = (uint32_t)t[1], l2 = (uint32_t)t[2], l3 = (uint32_t)t[3], l4 = (uint32_t)t[4];
    const uint32_t h0 = (uint32_t)(t[0] >> 32), h1 = (uint32_t)(t[1] >> 32), h2 = (uint32_t)(t[2] >> 32),
                   h3 = (uint32_t)(t[3] >> 32);
    uint32_t h4 = (uint32_t)(t[4] >> 32);
    asm("v_add_co_u32 %0, vcc, %0, %6\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, %1, %7, vcc\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %3, vcc, %3, %9, vcc\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %4, vcc, %4, %10, vcc\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %5, vcc, 0, %5, vcc"
        : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(l4), "+v"(h4)
        : "v"(prev), "v"(h0), "v"(h1), "v"(h2), "v"(h3)
        : "vcc");
    r[0] = l0; r[1] = l1; r[2] = l2; r[3] = l3; r[4] = l4;
    return h4;
#endif
}


// r = A*x + B*y  (A + B <= 2^32); returns the word leaving the top plane
template <int P>
CF_DEV uint32_t mp_lincomb_add(Ctx &c, Mp<P> &r, uint32_t A, const Mp<P> &x, uint32_t B, const Mp<P> &y) {
    uint32_t hi[P];
    CF_UNROLL for (int p = 0; p < P; p++) hi[p] = lincomb_plane<false>(r.v[p], A, x.v[p], B, y.v[p], 0u);
    return mp_resolve_sparse(c, r, hi);
}

// r = A*x - B*y modulo 2^(1280 P)  (A + B <= 2^32).  The caller guarantees 0 <= A*x - B*y.
// Two's complement: -B*y == B*~y + B over the full width.
template <int P>
CF_DEV void mp_lincomb_sub(Ctx &c, Mp<P> &r, uint32_t A, const Mp<P> &x, uint32_t B, const Mp<P> &y) {
    uint32_t hi[P];
    CF_UNROLL for (int p = 0; p < P; p++)
        hi[p] = lincomb_plane<true>(r.v[p], A, x.v[p], B, y.v[p], (p == 0 && c.gl == 0) ? B : 0u);
    (void)mp_resolve(c, r, hi);
}

// same, returning the word that leaves the top plane: A*x - B*y == r + (word - B) * 2^(1280 P)
template <int P>
CF_DEV uint32_t mp_lincomb_sub_carry(Ctx &c, Mp<P> &r, uint32_t A, const Mp<P> &x, uint32_t B, const Mp<P> &y) {
    uint32_t hi[P];
    CF_UNROLL for (int p = 0; p < P; p++)
        hi[p] = lincomb_plane<true>(r.v[p], A, x.v[p], B, y.v[p], (p == 0 && c.gl == 0) ? B : 0u);
    return mp_resolve(c, r, hi);
}

template <int P>
CF_DEV void mp_sub(Ctx &c, Mp<P> &r, const Mp<P> &x, const Mp<P> &y) {   // x >= y
    mp_lincomb_sub(c, r, 1u, x, 1u, y);
}

// ---------------------------------------------------------------------------- shifts (LDS)
// y = x << n (bits shifted past the top plane are dropped), n >= 0 group-uniform
template <int P>
CF_DEV Mp<P> mp_shl(Ctx &c, const Mp<P> &x, int n) {
    uint32_t *s = c.scratch();
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) s[p * PLIMBS + c.gl * CH + j] = x.v[p][j];
    group_sync(c);
    int w = n >> 5, o = n & 31;
    Mp<P> y;
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) {
            int i = p * PLIMBS + c.gl * CH + j - w;
            uint32_t a = (i >= 0 && i < P * PLIMBS) ? s[i >= 0 && i < P * PLIMBS ? i : 0] : 0u;
            uint32_t b = (i - 1 >= 0 && i - 1 < P * PLIMBS) ? s[i - 1 >= 0 && i - 1 < P * PLIMBS ? i - 1 : 0] : 0u;
            y.v[p][j] = o ? ((a << o) | (b >> (32 - o))) : a;
        }
    group_sync(c);
    return y;
}

template <int P>
CF_DEV Mp<P> mp_shr(Ctx &c, const Mp<P> &x, int n) {
    uint32_t *s = c.scratch();
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) s[p * PLIMBS + c.gl * CH + j] = x.v[p][j];
    group_sync(c);
    int w = n >> 5, o = n & 31;
    Mp<P> y;
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) {
            int i = p * PLIMBS + c.gl * CH + j + w;
            uint32_t a = (i < P * PLIMBS) ? s[i < P * PLIMBS ? i : 0] : 0u;
            uint32_t b = (i + 1 < P * PLIMBS) ? s[i + 1 < P * PLIMBS ? i + 1 : 0] : 0u;
            y.v[p][j] = o ? ((a >> o) | (b << (32 - o))) : a;
        }
    group_sync(c);
    return y;
}

// x >> n for 0 <= n < 32 (n group-uniform) without LDS: one DPP per plane
template <int P>
CF_DEV Mp<P> mp_shr_small(Ctx &c, const Mp<P> &x, int n) {
    Mp<P> y;
    uint32_t above = 0;      // limb following the current plane's top chunk
    const uint32_t ls = (uint32_t)(32 - n) & 31u;
    const uint32_t keep = n ? 0xFFFFFFFFu : 0u;
    CF_UNROLL for (int p = P - 1; p >= 0; p--) {
        uint32_t nxt = shfl_down1(c, x.v[p][0], above);
        CF_UNROLL for (int j = 0; j < CH; j++) {
            uint32_t up = (j + 1 < CH) ? x.v[p][j + 1 < CH ? j + 1 : 0] : nxt;
            y.v[p][j] = (x.v[p][j] >> n) | ((up << ls) & keep);
        }
        above = bcast_first(c, x.v[p][0]);
    }
    return y;
}
template <int P>
CF_DEV Mp<P> mp_shr1(Ctx &c, const Mp<P> &x) { return mp_shr_small(c, x, 1); }

// ---------------------------------------------------------------------------- multiplication
// w += x * y for 5-limb chunks, operand scanning straight into the 10-limb window (+ overflow
// word), each row's carry rippled to the top
CF_DEV void chunk_mac(uint32_t (&w)[2 * CH + 1], const uint32_t (&x)[CH], const uint32_t (&y)[CH]) {
    CF_UNROLL for (int i = 0; i < CH; i++) {
        uint32_t cy = 0;
        CF_UNROLL for (int j = 0; j < CH; j++) {
            uint64_t m = (uint64_t)x[i] * y[j] + w[i + j] + cy;
            w[i + j] = (uint32_t)m;
            cy = (uint32_t)(m >> 32);
        }
        CF_UNROLL for (int k = i + CH; k < 2 * CH + 1; k++) {
            uint64_t m = (uint64_t)w[k] + cy;
            w[k] = (uint32_t)m;
            cy = (uint32_t)(m >> 32);
        }
    }
}
// r = x * y, operands staged in the group's LDS slice; output chunk 8*po + gl owned by lane gl
template <int P, int Q>
CF_DEV Mp<P + Q> mp_mul(Ctx &c, const Mp<P> &x, const Mp<Q> &y) {
    constexpr int R = P + Q;
    CF_STAT(g_stats.muls += P * Q);
    static_assert(R * PLIMBS <= SCRATCH_WORDS && R * G * (CH + 1) <= SCRATCH_WORDS, "scratch too small");
    uint32_t *s = c.scratch();
    CF_UNROLL for (int p = 0; p < P; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) s[p * PLIMBS + c.gl * CH + j] = x.v[p][j];
    CF_UNROLL for (int p = 0; p < Q; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) s[(P + p) * PLIMBS + c.gl * CH + j] = y.v[p][j];
    group_sync(c);
    uint32_t w[R][2 * CH + 1];
    CF_UNROLL for (int p = 0; p < R; p++) CF_UNROLL for (int i = 0; i < 2 * CH + 1; i++) w[p][i] = 0;
    // y is usually the short operand (Euclid remainders, cofactors, M1): only its non-zero
    // chunks are visited
    const int ybits = mp_bitlen(c, y);
    CF_UNROLL for (int px = 0; px < P; px++) {
        CF_UNROLL for (int py = 0; py < Q; py++) {
            // x chunk lx = gl - k (mod 8) times y chunk k lands in output chunk gl of plane
            // px+py while k <= gl and of plane px+py+1 afterwards: one running accumulator
            // that every lane re-targets exactly once (at k == gl + 1)
            int ky = (ybits - py * PLIMBS * 32 + CH * 32 - 1) / (CH * 32);
            ky = ky < 0 ? 0 : (ky > G ? G : ky);
            uint32_t cur[2 * CH + 1];
            CF_UNROLL for (int i = 0; i < 2 * CH + 1; i++) cur[i] = w[px + py][i];
            for (int k = 0; k < ky; k++) {
                const bool sw = (k == c.gl + 1);
                CF_UNROLL for (int i = 0; i < 2 * CH + 1; i++) {
                    w[px + py][i] = sw ? cur[i] : w[px + py][i];
                    cur[i] = sw ? w[px + py + 1][i] : cur[i];
                }
                const int lx = (c.gl - k) & (G - 1);
                uint32_t xc[CH], yc[CH];
                CF_UNROLL for (int j = 0; j < CH; j++) {
                    xc[j] = s[(px * G + lx) * CH + j];
                    yc[j] = s[P * PLIMBS + (py * G + k) * CH + j];
                }
                chunk_mac(cur, xc, yc);
            }
            const bool never = (c.gl + 1 >= ky);     // lanes that never reached their switch point
            CF_UNROLL for (int i = 0; i < 2 * CH + 1; i++) {
                w[px + py][i] = never ? cur[i] : w[px + py][i];
                w[px + py + 1][i] = never ? w[px + py + 1][i] : cur[i];
            }
        }
    }
    group_sync(c);
    // exchange the upper halves: chunk cc receives limbs 5..9 of chunk cc-1 and the overflow
    // word of chunk cc-2
    CF_UNROLL for (int p = 0; p < R; p++)
        CF_UNROLL for (int i = 0; i < CH + 1; i++) s[(p * G + c.gl) * (CH + 1) + i] = w[p][CH + i];
    group_sync(c);
    Mp<R> r;
    uint32_t hi[R];
    CF_UNROLL for (int p = 0; p < R; p++) {
        int cc = p * G + c.gl;
        uint32_t cy = 0;
        CF_UNROLL for (int j = 0; j < CH; j++) {
            uint32_t up = (cc >= 1) ? s[(cc >= 1 ? cc - 1 : 0) * (CH + 1) + j] : 0u;
            uint32_t ov = (j == 0 && cc >= 2) ? s[(cc >= 2 ? cc - 2 : 0) * (CH + 1) + CH] : 0u;
            uint64_t t = (uint64_t)w[p][j] + up + ov + cy;
            r.v[p][j] = (uint32_t)t;
            cy = (uint32_t)(t >> 32);
        }
        hi[p] = cy;
    }
    group_sync(c);
    (void)mp_resolve_sparse(c, r, hi);
    return r;
}

// ---------------------------------------------------------------------------- division
// conservative floor(n / d) for d <= 2^32: never above the true quotient, at most 2 below
CF_DEV uint64_t div64_lower(uint64_t n, uint64_t d) {
    double q = (double)(n & ~0x7FFull) / (double)d;
    uint64_t t = (uint64_t)q;
    return t > 0 ? t - 1 : 0;
}

// one conservative quotient digit for num / den:  returns qd < 2^31 and sh >= 0 with
// (qd << sh) * den <= num, (qd << sh) within ~2^-29 of the true quotient.  nb / db are the bit
// lengths; requires num >= den > 0.
template <int PN, int PD>
CF_DEV uint32_t mp_quot_digit(Ctx &c, const Mp<PN> &num, int nb, const Mp<PD> &den, int db, int &sh) {
    int npos = nb > 64 ? nb - 64 : 0;
    int dpos = db > 32 ? db - 32 : 0;
    uint64_t nt = mp_bits64(c, num, npos);
    uint64_t dt = (uint64_t)mp_bits32(c, den, dpos) + (dpos > 0 ? 1u : 0u);   // exact when den fits
    int e = npos - dpos;
    uint64_t t;
    if (CF_UNLIKELY(dt == 0)) {      // zero divisor (garbage input): flag it, take a harmless digit
        CF_STATUS(c, CF_ST_DIV_CAP);
        sh = 0;
        return 1;
    }
    if (dpos == 0 && npos == 0) {
        t = nt / dt;                 // both fit in a machine word: exact
    } else {
        t = div64_lower(nt, dt);
    }
    if (e < 0) {
        t = (-e >= 64) ? 0 : (t >> (-e));
        e = 0;
    }
    int extra = 33 - __builtin_clzll(t | 1);        // bits above 31
    if (extra > 0) {
        t >>= extra;
        e += extra;
    }
    sh = e;
    if (t == 0) {            // estimate too coarse but num >= den: take one den
        sh = 0;
        return 1;
    }
    return (uint32_t)t;
}

// num <- num mod den, quot <- floor(num / den); den > 0.  Schoolbook with ~30-bit conservative
// digits (no add-back: the running remainder never goes negative).
template <int PN, int PD>
CF_DEV void mp_divrem_cons(Ctx &c, Mp<PN> &num, const Mp<PD> &den, Mp<PN> &quot) {
    static_assert(PN >= PD, "numerator must be at least as wide as the divisor");
    mp_zero(quot);
    const Mp<PN> dw = mp_resize<PN>(den);
    const int db = mp_bitlen(c, den);
    if (db == 0) {                               // division by zero: flag it, leave num as it is
        CF_STATUS(c, CF_ST_DIV_CAP);
        return;
    }
    // every step removes >= 28 bits of num (one conservative ~30-bit digit): PN * 1280 / 28 + 2 steps at most
    for (int guard = 0;; guard++) {
        if (guard > PN * PLIMBS * 32 / 28 + 2) {
            CF_STATUS(c, CF_ST_DIV_CAP);
            break;
        }
        int nb = mp_bitlen(c, num);
        if (nb < db) break;
        if (nb == db && mp_cmp(c, num, dw) < 0) break;
        int sh;
        CF_STAT(g_stats.divsteps++);
        uint32_t qd = mp_quot_digit(c, num, nb, den, db, sh);
        Mp<PN> ds = sh ? mp_shl(c, dw, sh) : dw;
        mp_lincomb_sub(c, num, 1u, num, qd, ds);
        // quot += qd << sh
        Mp<PN> qa;
        int i0 = sh >> 5, o = sh & 31;
        uint32_t lo = qd << o, hiw = o ? (qd >> (32 - o)) : 0u;
        CF_UNROLL for (int p = 0; p < PN; p++)
            CF_UNROLL for (int j = 0; j < CH; j++) {
                int i = p * PLIMBS + c.gl * CH + j;
                qa.v[p][j] = (i == i0) ? lo : ((i == i0 + 1) ? hiw : 0u);
            }
        (void)mp_add(c, quot, quot, qa);
    }
}

// ---- division by a 32-bit word -------------------------------------------------------------
struct WordDiv {            // floor(t / w), t mod w for 64-bit t with t / w < 2^64
    uint32_t w;
    uint64_t m;             // floor((2^64 - 1) / w)
};
CF_DEV uint64_t umulhi64(uint64_t a, uint64_t b) {
#if defined(COFHE_HOSTSIM)
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#else
    return __umul64hi(a, b);
#endif
}
CF_DEV WordDiv worddiv_make(uint32_t w) {
    WordDiv d;
    d.w = w;
    d.m = ~0ull / w;
    return d;
}
CF_DEV uint64_t worddiv_divmod(const WordDiv &d, uint64_t t, uint32_t &rem) {
    uint64_t q = umulhi64(t, d.m);
    uint64_t r = t - q * d.w;
    while (r >= d.w) {       // at most 2 rounds
        r -= d.w;
        q++;
    }
    rem = (uint32_t)r;
    return q;
}
CF_DEV uint32_t worddiv_mulmod(const WordDiv &d, uint32_t a, uint32_t b) {
    uint32_t r;
    (void)worddiv_divmod(d, (uint64_t)a * b, r);
    return r;
}
CF_DEV uint32_t worddiv_addmod(const WordDiv &d, uint32_t a, uint32_t b) {   // a, b < w
    uint64_t t = (uint64_t)a + b;
    return (uint32_t)(t >= d.w ? t - d.w : t);
}

// x mod w (w > 0): every lane reduces its own chunks, the group combines them with a
// 3-step scan of (residue, weight) pairs.  Group-uniform result.
template <int P>
CF_DEV uint32_t mp_mod_word(Ctx &c, const Mp<P> &x, const WordDiv &d) {
    uint32_t rem;
    uint32_t two32 = (uint32_t)(0x100000000ull % d.w);
    // beta = 2^160 mod w
    uint32_t beta = 1;
    CF_UNROLL for (int j = 0; j < CH; j++) beta = worddiv_mulmod(d, beta, two32);
    uint32_t total = 0;       // Horner over planes, top plane first
    CF_UNROLL for (int p = P - 1; p >= 0; p--) {
        uint32_t r = 0;
        CF_UNROLL for (int j = CH - 1; j >= 0; j--) {
            (void)worddiv_divmod(d, ((uint64_t)r << 32) | x.v[p][j], rem);
            r = rem;
        }
        // value of the plane = sum r_i * beta^i: suffix-combine (hi * beta^len + lo)
        uint32_t val = r, wgt = beta;             // this lane covers 1 chunk
        CF_UNROLL for (int st = 1; st < G; st <<= 1) {
            uint32_t ov = shfl(c, val, c.gl + st), ow = shfl(c, wgt, c.gl + st);
            bool has = (c.gl + st) < G;
            // combined = other(higher lanes) * wgt + val ; weight = wgt * ow
            uint32_t nv = worddiv_addmod(d, worddiv_mulmod(d, ov, wgt), val);
            uint32_t nw = worddiv_mulmod(d, wgt, ow);
            val = has ? nv : val;
            wgt = has ? nw : wgt;
        }
        uint32_t plane_val = bcast_first(c, val), plane_w = bcast_first(c, wgt);   // lane 0 holds all 8 chunks
        total = worddiv_addmod(d, worddiv_mulmod(d, total, plane_w), plane_val);
    }
    return total;
}

// x mod 223092870 (= 2*3*5*7*11*13*17*19*23), the modulus of the composition's coprime-representative test (qf.hpp): a
// compile-time modulus, so limb i has the tabulated weight 2^(32 i) mod M (< 2^28) -- five multiply-adds per lane into a
// 63-bit sum, one reduction by the constant, a three-step group sum: ~45 instructions against the ~800 of the general
// mp_mod_word, four times per composition.  Group-uniform result.
constexpr uint32_t PRIMORIAL23 = 223092870u;
struct PrimorialWeights { uint32_t w[PLIMBS]; };
constexpr PrimorialWeights primorial_weights() {
    PrimorialWeights t{};
    uint64_t v = 1;
    for (int i = 0; i < PLIMBS; i++) {
        t.w[i] = (uint32_t)v;
        v = (v << 32) % PRIMORIAL23;
    }
    return t;
}
#if defined(COFHE_HOSTSIM)
static constexpr PrimorialWeights PRIMORIAL_W = primorial_weights();
#else
__device__ static constexpr PrimorialWeights PRIMORIAL_W = primorial_weights();
#endif
CF_DEV uint32_t mp_mod_primorial(Ctx &c, const Mp<1> &x) {
    uint64_t acc = 0;
    CF_UNROLL for (int j = 0; j < CH; j++) acc += (uint64_t)x.v[0][j] * PRIMORIAL_W.w[c.gl * CH + j];     // < 5 * 2^60
    uint32_t v = (uint32_t)(acc % PRIMORIAL23);
    v += shfl_xor1(c, v);
    v += shfl_xor2(c, v);
    v += shfl_mirror(c, v);                       // < 8 * 2^28
    return v % PRIMORIAL23;
}

// ---- residues modulo a word W built from a 16-bit factor (W = d^2, d < 2^16) ------------------------------------------
// The composition's rare route (a common word-sized factor d of the first coefficients, qf.hpp) needs the residues of
// six multi-limb numbers modulo d and d^2.  mp_mod_word above is general (any 32-bit w) and pays a 64-bit Barrett
// reduction per limb and per scan step: ~800 instructions per plane.  Here every limb is split into 16-bit halves and
// multiplied by tabulated 2^(16 i) mod W: ten 48-bit products per lane whose sum fits 52 bits, ONE reduction of that sum,
// one multiplication by the lane's weight 2^(160 gl) mod W, and a three-step group sum -- ~100 instructions per plane
// after a set-up of ~15 reductions per modulus.
struct ModW {
    WordDiv dv;                 // W
    uint32_t w16[2 * CH];       // 2^(16 i) mod W, i < 2 CH
    uint32_t lanew;             // 2^(32 CH gl) mod W: weight of this lane's chunk
    uint32_t planew;            // 2^(32 PLIMBS) mod W: weight of the next plane
};
CF_DEV ModW modw_make(Ctx &c, uint32_t W) {
    ModW mw;
    mw.dv = worddiv_make(W);
    uint32_t rem;
    mw.w16[0] = 1u % W;
    CF_UNROLL for (int i = 1; i < 2 * CH; i++) {
        (void)worddiv_divmod(mw.dv, (uint64_t)mw.w16[i - 1] << 16, rem);
        mw.w16[i] = rem;
    }
    (void)worddiv_divmod(mw.dv, (uint64_t)mw.w16[2 * CH - 1] << 16, rem);
    const uint32_t p1 = rem;                                    // 2^(32 CH) mod W
    const uint32_t p2 = worddiv_mulmod(mw.dv, p1, p1), p4 = worddiv_mulmod(mw.dv, p2, p2);
    static_assert(G == 8, "lane weights written for 8 lanes per group");
    uint32_t lw = (c.gl & 1) ? p1 : mw.w16[0];
    lw = (c.gl & 2) ? worddiv_mulmod(mw.dv, lw, p2) : lw;
    lw = (c.gl & 4) ? worddiv_mulmod(mw.dv, lw, p4) : lw;
    mw.lanew = lw;
    mw.planew = worddiv_mulmod(mw.dv, p4, p4);
    return mw;
}
// x mod W; group-uniform result
template <int P>
CF_DEV uint32_t mp_mod_word_fast(Ctx &c, const Mp<P> &x, const ModW &mw) {
    uint32_t total = 0, rem;
    CF_UNROLL for (int p = P - 1; p >= 0; p--) {                // Horner over the planes, top plane first
        uint64_t acc = 0;
        CF_UNROLL for (int j = 0; j < CH; j++)
            acc += (uint64_t)(x.v[p][j] & 0xFFFFu) * mw.w16[2 * j] + (uint64_t)(x.v[p][j] >> 16) * mw.w16[2 * j + 1];
        (void)worddiv_divmod(mw.dv, acc, rem);                  // acc < 10 * 2^48
        uint32_t v = worddiv_mulmod(mw.dv, rem, mw.lanew);
        v = worddiv_addmod(mw.dv, v, shfl_xor1(c, v));
        v = worddiv_addmod(mw.dv, v, shfl_xor2(c, v));
        v = worddiv_addmod(mw.dv, v, shfl_mirror(c, v));
        total = worddiv_addmod(mw.dv, worddiv_mulmod(mw.dv, total, mw.planew), v);
    }
    return total;
}
// num <- floor(num / w), returns num mod w
template <int P>
CF_DEV uint32_t mp_divrem_word(Ctx &c, Mp<P> &num, const WordDiv &d) {
    uint32_t rem;
    uint32_t two32 = (uint32_t)(0x100000000ull % d.w);
    uint32_t beta = 1;
    CF_UNROLL for (int j = 0; j < CH; j++) beta = worddiv_mulmod(d, beta, two32);
    uint32_t above = 0;        // residue of everything above the current plane
    CF_UNROLL for (int p = P - 1; p >= 0; p--) {
        uint32_t r = 0;
        CF_UNROLL for (int j = CH - 1; j >= 0; j--) {
            (void)worddiv_divmod(d, ((uint64_t)r << 32) | num.v[p][j], rem);
            r = rem;
        }
        // H = residue of all chunks above this lane (within the plane) combined with `above`:
        // exclusive suffix scan of (val, wgt)
        uint32_t val = r, wgt = beta;
        CF_UNROLL for (int st = 1; st < G; st <<= 1) {
            uint32_t ov = shfl(c, val, c.gl + st), ow = shfl(c, wgt, c.gl + st);
            bool has = (c.gl + st) < G;
            uint32_t nv = worddiv_addmod(d, worddiv_mulmod(d, ov, wgt), val);
            uint32_t nw = worddiv_mulmod(d, wgt, ow);
            val = has ? nv : val;
            wgt = has ? nw : wgt;
        }
        // inclusive suffix of lane gl+1 = value of chunks above lane gl (0 for the top lane)
        uint32_t hv = shfl_down1(c, val, 0u), hw = shfl_down1(c, wgt, 1u);
        uint32_t h = worddiv_addmod(d, worddiv_mulmod(d, above, hw), hv);   // above * beta^(#chunks above) + hv
        // long division of this lane's chunk with incoming remainder h
        uint32_t rr = h;
        CF_UNROLL for (int j = CH - 1; j >= 0; j--) {
            uint64_t q = worddiv_divmod(d, ((uint64_t)rr << 32) | num.v[p][j], rem);
            num.v[p][j] = (uint32_t)q;
            rr = rem;
        }
        uint32_t pv = bcast_first(c, val), pw = bcast_first(c, wgt);
        above = worddiv_addmod(d, worddiv_mulmod(d, above, pw), pv);
    }
    return above;
}

// Division by a single-plane divisor, Knuth D in a systolic layout.  The divisor is shifted so
// that its leading bit is the top bit of the plane; the running remainder S (one plane plus a
// top word) stays aligned with it, so the three words a digit estimate needs always sit in the
// same registers of lane 7.  Per 32-bit quotient digit: one f64 estimate (never below the true
// digit, one above with probability ~2^-17 -> add-back), one linear combination S - q*D over a
// single plane, one limb shift of S across the lanes (DPP) pulling the next numerator limb
// from the LDS slice, one LDS word for the digit.  No per-digit window extraction, no shifted
// copy of the divisor, no quotient carries.
template <int PN>
CF_DEV void mp_divrem_norm(Ctx &c, Mp<PN> &num, const Mp<1> &den, int db, Mp<PN> &quot) {
    static_assert(2 * PN * PLIMBS <= SCRATCH_WORDS, "scratch too small");
    mp_zero(quot);
    const int nb = mp_bitlen(c, num);
    if (nb < db) return;
    const int s = PLIMBS * 32 - db;
    const Mp<1> D = s ? mp_shl(c, den, s) : den;
    const uint32_t d39 = bcast_last(c, D.v[0][CH - 1]), d38 = bcast_last(c, D.v[0][CH - 2]);
    const double rd = 1.0 / ((double)d39 * 4294967296.0 + (double)d38);
    uint32_t *sn = c.scratch(), *sq = sn + PN * PLIMBS;
    CF_UNROLL for (int p = 0; p < PN; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) {
            sn[p * PLIMBS + c.gl * CH + j] = num.v[p][j];
            sq[p * PLIMBS + c.gl * CH + j] = 0u;
        }
    group_sync(c);
    const int sw = s >> 5, sb = s & 31;
    auto limb = [&](int k) -> uint32_t {          // limb k of num << s
        const int i = k - sw;
        const uint32_t a = (i >= 0 && i < PN * PLIMBS) ? sn[(i >= 0 && i < PN * PLIMBS) ? i : 0] : 0u;
        const uint32_t b = (i >= 1 && i <= PN * PLIMBS) ? sn[(i >= 1 && i <= PN * PLIMBS) ? i - 1 : 0] : 0u;
        return sb ? ((a << sb) | (b >> (32 - sb))) : a;
    };
    const int K = (nb + s - 1) >> 5;              // top limb of num << s  (K >= 39 because nb >= db)
    Mp<1> S;
    CF_UNROLL for (int j = 0; j < CH; j++) S.v[0][j] = limb(K - (PLIMBS - 1) + c.gl * CH + j);
    uint32_t top = 0;
    for (int k = K - (PLIMBS - 1);; k--) {
        CF_STAT(g_stats.divsteps++);
        const uint32_t l39 = bcast_last(c, S.v[0][CH - 1]), l38 = bcast_last(c, S.v[0][CH - 2]);
        double x = (((double)top * 4294967296.0 + (double)l39) * 4294967296.0 + (double)l38) * rd;
        x += x * 1.7763568394002505e-15;          // (1 + 2^-49): never below the true digit
        uint64_t qd = (uint64_t)x;
        if (qd > 0xFFFFFFFFull) qd = 0xFFFFFFFFull;
        if (CF_LIKELY(qd != 0)) {
            Mp<1> T;
            const uint32_t cw = mp_lincomb_sub_carry(c, T, 1u, S, (uint32_t)qd, D);
            int64_t nt = (int64_t)top + (int64_t)cw - (int64_t)qd;   // top word of S - q D: 0, or -1 if q is one too large
            S = T;
            for (int fix = 0; CF_UNLIKELY(nt < 0) && fix < 4; fix++) {       // the estimate is at most one too large
                CF_FLAG(8u);
                nt += (int64_t)mp_add(c, S, S, D);
                qd--;
            }
        }
        if (c.gl == 0) sq[k] = (uint32_t)qd;
        if (k == 0) break;
        top = bcast_last(c, S.v[0][CH - 1]);
        const uint32_t up = shfl_up1(c, S.v[0][CH - 1], limb(k - 1));
        CF_UNROLL for (int j = CH - 1; j >= 1; j--) S.v[0][j] = S.v[0][j - 1];
        S.v[0][0] = up;
    }
    group_sync(c);
    CF_UNROLL for (int p = 0; p < PN; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) quot.v[p][j] = sq[p * PLIMBS + c.gl * CH + j];
    group_sync(c);
    const Mp<1> R = s ? mp_shr(c, S, s) : S;
    num = mp_resize<PN>(R);
}

// num <- num mod den, quot <- floor(num / den); den > 0.  Knuth D with exact 32-bit digits:
// the digit comes from an f64 quotient of a 96-bit remainder window by the leading 64 bits of
// the divisor (never below the true digit, one above with probability ~2^-18 -> add-back), the
// divisor is staged once in the LDS slice and read back limb-shifted for each digit, and the
// quotient limbs are written in place (no carries).  Capacity: the operands must leave the top
// bit of the PN-plane window clear (all callers keep >= 100 bits of headroom).
template <int PN, int PD>
CF_DEV void mp_divrem(Ctx &c, Mp<PN> &num, const Mp<PD> &den, Mp<PN> &quot) {
    static_assert(PN >= PD, "numerator must be at least as wide as the divisor");
    CF_STAT(g_stats.divrems++);
    const int db = mp_bitlen(c, den);
    if (CF_UNLIKELY(db == 0)) {                  // division by zero (never for valid forms): flag it, quotient 0
        CF_STATUS(c, CF_ST_DIV_CAP);
        mp_zero(quot);
        return;
    }
    if (db <= 32) {
        WordDiv d = worddiv_make(mp_get_limb(c, den, 0));
        quot = num;
        uint32_t r = mp_divrem_word(c, quot, d);
        mp_set_word(c, num, r);
        return;
    }
    if constexpr (PD == 1) {
        mp_divrem_norm(c, num, den, db, quot);
        return;
    }
    if (db < 64) {
        mp_divrem_cons(c, num, den, quot);
        return;
    }
    mp_zero(quot);
    const int nb = mp_bitlen(c, num);
    if (nb < db) return;
    uint32_t *s = c.scratch();
    CF_UNROLL for (int p = 0; p < PD; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) s[p * PLIMBS + c.gl * CH + j] = den.v[p][j];
    group_sync(c);
    const uint64_t d64 = mp_bits64(c, den, db - 64);
    const double rd = 1.0 / (double)d64;        // one f64 division per call, one multiply per digit
    for (int jq = (nb - db) / 32; jq >= 0; jq--) {
        CF_STAT(g_stats.divsteps++);
        // 96-bit window of the remainder aligned with the divisor's leading 64 bits
        const int pos = db - 64 + 32 * jq;
        const uint64_t hi64 = mp_bits64(c, num, pos + 32);
        const uint32_t lo32 = mp_bits32(c, num, pos);
        double x = ((double)hi64 * 4294967296.0 + (double)lo32) * rd;
        x += x * 1.7763568394002505e-15;           // (1 + 2^-49): never below the true digit
        uint64_t qd = (uint64_t)x;
        if (qd > 0xFFFFFFFFull) qd = 0xFFFFFFFFull;
        if (qd != 0) {
            Mp<PN> ds;
            CF_UNROLL for (int p = 0; p < PN; p++)
                CF_UNROLL for (int j = 0; j < CH; j++) {
                    int i = p * PLIMBS + c.gl * CH + j - jq;
                    bool ok = i >= 0 && i < PD * PLIMBS;
                    ds.v[p][j] = ok ? s[ok ? i : 0] : 0u;
                }
            mp_lincomb_sub(c, num, 1u, num, (uint32_t)qd, ds);
            // negative (digit one too large)?  top bit of the window is set only then
            for (int fix = 0; fix < 4 && ballot8(c, c.gl == G - 1 && (num.v[PN - 1][CH - 1] >> 31)) != 0; fix++) {
                (void)mp_add(c, num, num, ds);
                qd--;
            }
        }
        CF_UNROLL for (int p = 0; p < PN; p++)
            CF_UNROLL for (int j = 0; j < CH; j++)
                quot.v[p][j] = (p * PLIMBS + c.gl * CH + j == jq) ? (uint32_t)qd : quot.v[p][j];
    }
    group_sync(c);
}

// quot = num / den for an EXACT division (den > 0 divides num), 2-adic from the low end (Hensel / Jebelean):
// with den odd (its trailing zero bits are shifted out of both operands first) every quotient digit is
// q = S[0] * den^-1 mod 2^32 -- no estimate, no normalisation of the divisor, no add-back -- followed by
// S <- (S - q*den) / 2^32 over ONE plane: the running value R = S + (H + adj) * 2^1280 keeps its not yet
// visited high limbs H in the LDS slice and the borrow of the plane in the small signed word adj
// (R >= 0 throughout because the partial quotients never exceed the quotient).  nq = number of quotient
// limbs to produce (group-uniform upper bound on the length of the quotient).  ~70 issue slots per digit
// against ~130 for the normalised long division.  A num that den does not divide gives a meaningless
// quotient (never a hang: the trip count is fixed).
template <int PN, int PQ>
CF_DEV void mp_divexact(Ctx &c, const Mp<PN> &num, const Mp<1> &den, Mp<PQ> &quot, int nq) {
    static_assert((PN + PQ) * PLIMBS <= SCRATCH_WORDS, "scratch too small");
    mp_zero(quot);
    if (nq <= 0) return;
    if (nq > PQ * PLIMBS) nq = PQ * PLIMBS;
    const uint32_t d0raw = bcast_first(c, den.v[0][0]);
    if (CF_UNLIKELY(d0raw == 0)) {     // 32 or more trailing zero bits (never for form coefficients): long division
        Mp<PN> rem = num, q;
        mp_divrem(c, rem, den, q);
        quot = mp_resize<PQ>(q);
        return;
    }
    const int tz = __builtin_ctz(d0raw);
    const Mp<1> D = mp_shr_small(c, den, tz);
    const Mp<PN> Nn = mp_shr_small(c, num, tz);
    const uint32_t d0 = d0raw >> tz | (tz ? bcast_first(c, den.v[0][1]) << (32 - tz) : 0u);
    uint32_t dinv = d0;                // d0 * d0 == 1 (mod 8); each Newton step doubles the valid bits
    CF_UNROLL for (int i = 0; i < 4; i++) dinv *= 2u - d0 * dinv;
    uint32_t *sn = c.scratch(), *sq = sn + PN * PLIMBS;
    CF_UNROLL for (int p = 0; p < PN; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) sn[p * PLIMBS + c.gl * CH + j] = Nn.v[p][j];
    CF_UNROLL for (int p = 0; p < PQ; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) sq[p * PLIMBS + c.gl * CH + j] = 0u;
    group_sync(c);
    Mp<1> S;
    CF_UNROLL for (int j = 0; j < CH; j++) S.v[0][j] = Nn.v[0][j];
    int64_t adj = 0;
    auto high = [&](int k) -> uint32_t { return (PLIMBS + k < PN * PLIMBS) ? sn[(PLIMBS + k < PN * PLIMBS) ? PLIMBS + k : 0] : 0u; };
    // Two digits per pass (round 4): q = (S[0] + S[1] 2^32) D^-1 mod 2^64 gives both at once (they depend on the two lowest
    // limbs only), and S - q0 D is NOT resolved before q1 D is taken off.  After the first chain every lane adds the WORD its
    // lower neighbour hands over to its limb 0 (one add; no ripple) and keeps the carry bit of that add as the carry-in of
    // its second chain -- the limb shift moves that bit's place, limb 1, to limb 0.  One carry resolve per pair instead of one
    // per digit.  Everything is linear in the running value
    //     R = sum_l (chunk_l + pending_l 2^160) 2^(160 l) + (H + adj) 2^1280,
    // so where a pending word or bit is added does not matter as long as its weight is right; and a lane's second chain
    // still hands over a single word (its sum stays below (q1 + 1) 2^160: the carry-in is at most 1 where q1 is not added).
    const uint64_t d64 = ((uint64_t)bcast_first(c, D.v[0][1]) << 32) | d0;
    uint64_t dinv64 = dinv;
    dinv64 *= 2ull - d64 * dinv64;                                          // 32 -> 64 valid bits
    int i = 0;
    for (; i + 1 < nq; i += 2) {
        CF_STAT(g_stats.divsteps += 2);
        const uint64_t s01 = ((uint64_t)bcast_first(c, S.v[0][1]) << 32) | bcast_first(c, S.v[0][0]);
        const uint64_t q = s01 * dinv64;
        const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32);
        // first digit: lane-local chains, then the neighbours' words (the top lane's goes into the high-limb bookkeeping)
        uint32_t r1[CH];
        const uint32_t hi1 = lincomb_plane<true>(r1, 1u, S.v[0], q0, D.v[0], c.gl == 0 ? q0 : 0u);
        const int64_t v1 = (int64_t)high(i) + adj + (int64_t)bcast_last(c, hi1) - (int64_t)q0;
        const int64_t adj1 = v1 >> 32;                                      // floor
        const uint32_t inc = shfl_up1(c, hi1, 0u);
        const uint32_t r0 = r1[0] + inc;
        const uint32_t bit = r0 < inc ? 1u : 0u;                            // belongs to limb 1 of this lane: limb 0 after the shift
        uint32_t s2[CH];
        s2[CH - 1] = shfl_down1(c, r0, (uint32_t)v1);                       // the limb shift; the top lane takes the new high limb
        CF_UNROLL for (int j = 0; j + 1 < CH; j++) s2[j] = r1[j + 1];
        // second digit
        Mp<1> T;
        uint32_t hi2[1];
        hi2[0] = lincomb_plane<true>(T.v[0], 1u, s2, q1, D.v[0], c.gl == 0 ? q1 : bit);
        const uint32_t cw = mp_resolve(c, T, hi2);
        const int64_t v2 = (int64_t)high(i + 1) + adj1 + (int64_t)cw - (int64_t)q1;
        adj = v2 >> 32;
        const uint32_t up = shfl_down1(c, T.v[0][0], (uint32_t)v2);
        CF_UNROLL for (int j = 0; j + 1 < CH; j++) S.v[0][j] = T.v[0][j + 1];
        S.v[0][CH - 1] = up;
        if (c.gl == 0) {
            sq[i] = q0;
            sq[i + 1] = q1;
        }
    }
    for (; i < nq; i++) {                                                   // an odd last digit
        CF_STAT(g_stats.divsteps++);
        const uint32_t q = bcast_first(c, S.v[0][0]) * dinv;
        Mp<1> T;
        const uint32_t cw = mp_lincomb_sub_carry(c, T, 1u, S, q, D);      // S - q D == T + (cw - q) * 2^1280, T[0] == 0
        const int64_t v = (int64_t)high(i) + adj + (int64_t)cw - (int64_t)q;
        adj = v >> 32;                                                       // floor
        const uint32_t up = shfl_down1(c, T.v[0][0], (uint32_t)v);
        CF_UNROLL for (int j = 0; j + 1 < CH; j++) S.v[0][j] = T.v[0][j + 1];
        S.v[0][CH - 1] = up;
        if (c.gl == 0) sq[i] = q;
    }
    group_sync(c);
    CF_UNROLL for (int p = 0; p < PQ; p++)
        CF_UNROLL for (int j = 0; j < CH; j++) quot.v[p][j] = sq[p * PLIMBS + c.gl * CH + j];
    group_sync(c);
}

// ---------------------------------------------------------------------------- signed helpers
template <int P>
CF_DEV void smp_add(Ctx &c, SMp<P> &r, const SMp<P> &x, const SMp<P> &y) {
    if (x.neg == y.neg) {
        (void)mp_add(c, r.m, x.m, y.m);
        r.neg = x.neg;
    } else {
        int cm = mp_cmp(c, x.m, y.m);
        if (cm >= 0) {
            Mp<P> t;
            mp_sub(c, t, x.m, y.m);
            r.m = t;
            r.neg = cm == 0 ? 0 : x.neg;
        } else {
            Mp<P> t;
            mp_sub(c, t, y.m, x.m);
            r.m = t;
            r.neg = y.neg;
        }
    }
}
template <int P>
CF_DEV void smp_sub(Ctx &c, SMp<P> &r, const SMp<P> &x, const SMp<P> &y) {
    SMp<P> ny = y;
    ny.neg ^= 1;
    smp_add(c, r, x, ny);
}
template <int P, int Q>
CF_DEV SMp<P + Q> smp_mul(Ctx &c, const SMp<P> &x, const SMp<Q> &y) {
    SMp<P + Q> r;
    r.m = mp_mul(c, x.m, y.m);
    r.neg = x.neg ^ y.neg;
    return r;
}

// bits [pos, pos+64) of x and of y through the LDS slice: 10 stores + 6 broadcast loads instead
// of six 5-way register selects (dynamic VGPR indexing does not exist; the VALU is the busy unit)
CF_DEV void mp_bits64_pair(Ctx &c, const Mp<1> &x, const Mp<1> &y, int pos, uint64_t &xh, uint64_t &yh) {
    uint32_t *s = c.scratch();
    CF_UNROLL for (int j = 0; j < CH; j++) {
        s[c.gl * CH + j] = x.v[0][j];
        s[PLIMBS + c.gl * CH + j] = y.v[0][j];
    }
    s[2 * PLIMBS + c.gl] = 0u;                     // two guard words per number would do; keep it simple
    group_sync(c);
    const int i0 = pos >> 5, o = pos & 31;
    const int i1 = i0 + 1 < PLIMBS ? i0 + 1 : 2 * PLIMBS, i2 = i0 + 2 < PLIMBS ? i0 + 2 : 2 * PLIMBS;
    const uint32_t x0 = s[i0], x1 = s[i1 < PLIMBS ? i1 : 2 * PLIMBS], x2 = s[i2 < PLIMBS ? i2 : 2 * PLIMBS];
    const uint32_t y0 = s[PLIMBS + i0], y1 = s[i1 < PLIMBS ? PLIMBS + i1 : 2 * PLIMBS],
                   y2 = s[i2 < PLIMBS ? PLIMBS + i2 : 2 * PLIMBS];
    group_sync(c);
    const uint64_t xl = ((uint64_t)x1 << 32) | x0, yl = ((uint64_t)y1 << 32) | y0;
    xh = o ? ((xl >> o) | ((uint64_t)x2 << (64 - o))) : xl;
    yh = o ? ((yl >> o) | ((uint64_t)y2 << (64 - o))) : yl;
}

// ---------------------------------------------------------------------------- Euclid (Lehmer)
// State of a remainder sequence with one cofactor column:  x >= 0, y >= 0 and
//   x == sx * ux * w,  y == sy * uy * w   (mod modulus)   for the tracked quantity w,
// ux, uy magnitudes, sx, sy in {+1, -1} always opposite (or the magnitude is zero).
template <int P>
struct Euclid {
    Mp<P> x, y, ux, uy;
    int sx, sy;
};

// float image of a 64-bit value (relative error <= 2^-23)
CF_DEV float u64_to_float(uint64_t v) {
    uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    // keep the two halves opaque: otherwise LLVM folds this back into a (10-instruction,
    // correctly rounded) u64 -> f32 conversion; 2 cvt + 1 fma is all the estimate needs
#if defined(COFHE_HOSTSIM)
    asm volatile("" : "+r"(hi));
#else
    asm volatile("" : "+v"(hi));
#endif
    return (float)hi * 4294967296.0f + (float)lo;
}
// float -> u32 as v_cvt_u32_f32 does it: saturating, NaN -> 0 (plain C++ conversion is undefined out of range)
CF_DEV uint32_t f32_to_u32_sat(float x) {
#if defined(COFHE_HOSTSIM)
    if (!(x >= 1.0f)) return 0u;
    return x >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)x;
#else
    // the instruction itself, not a C++ conversion: lanes that have stopped run on dead values, and a float -> integer
    // conversion out of range is undefined in the language (poison in LLVM) although the hardware saturates
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#endif
}
CF_DEV float fast_rcp(float x) {
#if defined(COFHE_HOSTSIM)
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}

// ---- the Lehmer batch, in double precision ----------------------------------------------------------------------------
// One batch runs the remainder sequence on the leading LEHMER_WINDOW = 53 bits of the pair (xh >= yh, exact integers in an
// f64) and returns the 2x2 matrix of the steps taken, cofactors below 2^26:
//     x' = A x - B y >= 0,   y' = D y - C x >= 0    for EVERY pair the truncated windows can stand for.
// The batch is the critical path of the whole composition: one lane of the serving wavefront per limb group, the other
// three wavefronts of the workgroup waiting at the barrier (tools/wg_timing.hip, round 3: 2.8 of the 3.9 us of a round
// at one workgroup per CU, 163 of 321 us per workgroup, were the serving lane's ~780 dependent instructions).  Until
// round 3 the windows were 64-bit integers with 31-bit cofactors: every half-step paid a two-word subtraction, a
// 64 x 32-bit multiply in three instructions, two-word compares and two u64 -> f32 images, ~33 VALU instructions plus the
// scalar mask logic (experiments/lehmer_variants/lehmer_variants.hpp: lehmer_batch_u64).  In f64 the remainder update
// p - t q is ONE fused multiply-add (exact: all values are integers below 2^53), each cofactor update one more, the
// compares single instructions: ~18 VALU per half-step and a dependent chain of six operations.  The price is the
// window: 53 bits carry 26-bit cofactors, so a sequence takes ~19 % more rounds of ~45 % of the serving time each.
//
// Quotient: lehmer_quotient below (an f32 estimate never above the true quotient).  A step is kept iff it is
// non-negative for every value the windows can stand for: with P in (p - b, p + a), Q in (q - c, q + d),
//     x-step:  P - t Q > (p - t q) - (b + t d)  -> keep iff  p' >= b'      y-step:  keep iff  q' >= c'
// (exact windows -- the numbers themselves, sh == 0: keep iff the new remainder is >= 0; p >= thr is tested after the
// snapshot: the step that crosses the threshold is the last one kept).  The cofactor columns are
// continuants (a <= b, c <= d after the first step), so the 2^26 bound is tested on the larger one.  A lane that has
// stopped runs on with dead values (possibly inf / NaN: every comparison with them is false) -- no exec-mask regions;
// the last valid matrix is kept in a snapshot; there is no branch at all (the double-steps are unrolled; a wave-uniform
// "everybody has stopped" exit cost more than it saved, see below).  thr: stop once a remainder drops below it
// (partial sequence; a power of two, handed over as a double).  tests/test_hostsim_device_code.py checks every matrix against the window intervals.
constexpr int LEHMER_WINDOW = 53;
// Double-steps per batch: the serving wavefront runs until its slowest lane has finished; the average lane fills its 26
// cofactor bits in 7-8 double-steps, a run of small quotients needs more.  Capped, such a lane hands back a smaller
// matrix and catches up in a later round; the round gets shorter for the whole workgroup.  Measured on the 128x128
// composition (ms per launch, profiles/r03_a/variants3.txt, variants4.txt): 12: 0.520, 10: 0.509, 8: 0.479, 7: 0.500,
// 6: 0.529, 5: 0.574 (the integer batch it replaces: 0.499).
#ifndef COFHE_LEHMER_CAP
#define COFHE_LEHMER_CAP 8
#endif
#if defined(COFHE_HOSTSIM)
#define CF_WAVE_ANY(x) (x)
CF_DEV double cf_fma(double a, double b, double c) { return std::fma(a, b, c); }
CF_DEV float cf_truncf(float x) { return std::trunc(x); }
#else
#define CF_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0)
CF_DEV double cf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
CF_DEV float cf_truncf(float x) { return __builtin_truncf(x); }
#endif
// keep ? v : old, bit-wise through a 0 / ~0 lane mask (v_bfi_b32 on both halves).  NOT a select: the compiler turns a
// select into a VOP2 v_cndmask reading VCC, and here that VCC comes out of scalar ANDs -- the combination that costs
// 12-16 cycles instead of ~3 on gfx950 (tools/inst_bench.hip: "s_mov vcc + cndmask vcc"), four times per half-step on the
// critical path.  One v_cndmask (VOP3, mask from an SGPR pair) makes the lane mask, the rest is plain ALU.
CF_DEV double keep_if(uint32_t mask, double v, double old) {
    uint64_t a, b;
    static_assert(sizeof(a) == sizeof(v), "f64 image");
    memcpy(&a, &v, 8);
    memcpy(&b, &old, 8);
    const uint64_t m = ((uint64_t)mask << 32) | mask;
    const uint64_t r = (a & m) | (b & ~m);
    double out;
    memcpy(&out, &r, 8);
    return out;
}
// The quotient estimate of a half-step: trunc(num * (rcp(den) * (1 - 2^-20))) on the f32 images -- never above
// floor(num / den) (the margin covers the two conversions, the reciprocal and the product) and at most one below it for
// quotients < 2^20 (such a step leaves a remainder >= the divisor; the next step of the other kind has quotient 0 and the one
// after takes what was left).  Round 4 tried an absolute bias instead, trunc(fma(num, rcp(den), -2^-14)), one instruction
// less on the critical path: a pair with num / den in [1, 1 + 2^-14) then makes no step at all, the round falls back to a
// long-division step on the whole workgroup's time, and that happened in 70 % of the workgroups of the 128x128 addition
// instead of 3 % (tools/gpu_wg_ab.sh: 0.366 against 0.341 ms per launch, profiles/r04_a/variants_quot.txt).
CF_DEV float lehmer_quotient(float num, float rden) { return cf_truncf(num * (rden * 0.99999905f)); }
CF_DEV bool lehmer_batch(uint64_t xh, uint64_t yh, bool exact, double thrd, uint32_t &A, uint32_t &B, uint32_t &C, uint32_t &D) {
    const double LIMIT = 67108864.0;                       // 2^26
    double p = (double)xh, q = (double)yh;                  // exact: below 2^53
    double a = 1.0, b = 0.0, cc = 0.0, d = 1.0;             // working state: runs on, meaningless once the lane has stopped
    double ra = 1.0, rb = 0.0, rc = 0.0, rd = 1.0;          // state after the last valid half-step
    const double eb = exact ? 0.0 : 1.0;
    float pf = (float)p, qf = (float)q;                     // f32 images: numerator of the coming quotient / the reciprocal's input
    float rq = fast_rcp(qf), rp;
    bool alive = true;
#ifdef COFHE_LEHMER_EARLY_EXIT
    bool any_prev = true;
#endif
    // Fully unrolled, no early exit: the serving wavefront runs until its slowest lane has finished and that lane almost
    // always needs 7-8 of the 8 double-steps, so the wave-uniform "everybody has stopped" test (a select, a compare, a
    // ballot and the loop counter: 10 of the 52 instructions of a double-step) saved less than it cost.
    CF_UNROLL for (int it = 0; it < COFHE_LEHMER_CAP; it++) {
#ifdef COFHE_LEHMER_EARLY_EXIT
        if (!any_prev) break;
#endif
        {   // x -= t y.  t == 0 (the previous quotient came out one short: p < q here) is a step that changes nothing and
            // passes the test below; the following y-step takes what was left
            const double t = (double)lehmer_quotient(pf, rq);   // q == 0: inf / NaN, fails below
            p = cf_fma(-t, q, p);
            b = cf_fma(t, d, b);
            a = cf_fma(t, cc, a);
            alive = alive & (b < LIMIT) & (p >= b * eb);
            const uint32_t m = opaque(alive ? 0xFFFFFFFFu : 0u);
            ra = keep_if(m, a, ra); rb = keep_if(m, b, rb);
            alive = alive & (p >= thrd);
            pf = (float)p;
            rp = fast_rcp(pf);
        }
        {   // y -= t x
            const double t = (double)lehmer_quotient(qf, rp);
            q = cf_fma(-t, p, q);
            d = cf_fma(t, b, d);
            cc = cf_fma(t, a, cc);
            alive = alive & (d < LIMIT) & (q >= cc * eb);
            const uint32_t m = opaque(alive ? 0xFFFFFFFFu : 0u);
            rd = keep_if(m, d, rd); rc = keep_if(m, cc, rc);
            alive = alive & (q >= thrd);
            qf = (float)q;
            rq = fast_rcp(qf);
        }
#ifdef COFHE_LEHMER_EARLY_EXIT
        any_prev = CF_WAVE_ANY(alive);
#endif
    }
    A = (uint32_t)ra; B = (uint32_t)rb; C = (uint32_t)rc; D = (uint32_t)rd;     // snapshots: valid integers below 2^26
    return (B | C) != 0;
}

// The same batch for ONE chain (qfw.hpp: the wavefront-wide layout, where every lane runs the batch on the same windows, so
// every test is a scalar branch and a lone wavefront pays ~5 cycles per instruction, whatever it is): no run-on lanes, no
// snapshots, no lane masks, no copies.  A double-step is computed straight through from one state into a second one --
// two quotients, six fused multiply-adds, the two reciprocals -- and ONE test (three maxima, three compares) says whether
// both halves stand and the sequence goes on; the next double-step runs back into the first state.  When the test fails,
// the state the double-step started from is still there, and the exit path redoes the tests of the two halves one by
// one and keeps what lehmer_batch would have kept.  Same quotients, same tests, same order: for equal CAP the two return
// the same matrix (tests/test_hostsim_device_code.py).  The cap may be higher here, nobody waits for a slow lane.  Exact
// windows (the numbers themselves, the last two rounds of a full sequence) take lehmer_batch: their tests differ
// (eb == 0) and a remainder may legitimately become zero there.
struct LehmerState {
    double p, q, a, b, c, d;      // remainders, cofactors: x' = a x - b y, y' = d y - c x
    float pf, qf, rq;             // f32 images of p and q, 1 / qf
};
CF_DEV double cf_fmax(double x, double y) {
#if defined(COFHE_HOSTSIM)
    return std::fmax(x, y);
#else
    return __builtin_fmax(x, y);
#endif
}
// s -> n; true: both halves are valid, both remainders >= thr1 (>= 1), go on.  Every value is finite as long as the
// divisors are non-zero, which the test of the previous double-step (and the caller, for the first) guarantees.
CF_DEV bool lehmer_double_step(const LehmerState &s, LehmerState &n, double thr1) {
    const double LIMIT = 67108864.0;                       // 2^26
    const double tx = (double)lehmer_quotient(s.pf, s.rq);
    n.p = cf_fma(-tx, s.q, s.p);
    n.b = cf_fma(tx, s.d, s.b);
    n.a = cf_fma(tx, s.c, s.a);
    n.pf = (float)n.p;
    const float rp = fast_rcp(n.pf);
    const double ty = (double)lehmer_quotient(s.qf, rp);
    n.q = cf_fma(-ty, n.p, s.q);
    n.d = cf_fma(ty, n.b, s.d);
    n.c = cf_fma(ty, n.a, s.c);
    n.qf = (float)n.q;
    n.rq = fast_rcp(n.qf);
    return (cf_fmax(n.b, n.d) < LIMIT) & (n.p >= cf_fmax(n.b, thr1)) & (n.q >= cf_fmax(n.c, thr1));
}
// the double-step s -> n failed its test: the half-steps one by one, as lehmer_batch tests them
CF_DEV bool lehmer_finish(const LehmerState &s, const LehmerState &n, double thrd, uint32_t &A, uint32_t &B, uint32_t &C, uint32_t &D) {
    const double LIMIT = 67108864.0;
    double a = s.a, b = s.b, c = s.c, d = s.d;
    if ((n.b < LIMIT) & (n.p >= n.b)) {
        a = n.a;
        b = n.b;
        if ((n.p >= thrd) & (n.d < LIMIT) & (n.q >= n.c)) {
            c = n.c;
            d = n.d;
        }
    }
    A = (uint32_t)a; B = (uint32_t)b; C = (uint32_t)c; D = (uint32_t)d;
    return (B | C) != 0;
}
template <int CAP>
CF_DEV bool lehmer_batch_uniform(uint64_t xh, uint64_t yh, bool exact, double thrd, uint32_t &A, uint32_t &B, uint32_t &C, uint32_t &D) {
    static_assert(CAP % 2 == 0, "double-steps come in pairs (two states, no copies)");
    if (exact || yh == 0) return lehmer_batch(xh, yh, exact, thrd, A, B, C, D);
    const double thr1 = thrd > 1.0 ? thrd : 1.0;
    LehmerState s0, s1;
    s0.p = (double)xh; s0.q = (double)yh;                   // exact: below 2^53
    s0.a = 1.0; s0.b = 0.0; s0.c = 0.0; s0.d = 1.0;
    s0.pf = (float)s0.p; s0.qf = (float)s0.q;
    s0.rq = fast_rcp(s0.qf);
    CF_NOUNROLL for (int it = 0; it < CAP; it += 2) {
        if (CF_UNLIKELY(!lehmer_double_step(s0, s1, thr1))) return lehmer_finish(s0, s1, thrd, A, B, C, D);
        if (CF_UNLIKELY(!lehmer_double_step(s1, s0, thr1))) return lehmer_finish(s1, s0, thrd, A, B, C, D);
    }
    A = (uint32_t)s0.a; B = (uint32_t)s0.b; C = (uint32_t)s0.c; D = (uint32_t)s0.d;
    return (B | C) != 0;
}
template <int CAP>
CF_DEV bool lehmer_batch_uniform_unordered(uint64_t xh, uint64_t yh, bool exact, double thr, uint32_t &A, uint32_t &B, uint32_t &C, uint32_t &D) {
    const bool sw = xh < yh;
    uint32_t a, b, cc, d;
    const bool ok = lehmer_batch_uniform<CAP>(sw ? yh : xh, sw ? xh : yh, exact, thr, a, b, cc, d);
    A = sw ? d : a;
    B = sw ? cc : b;
    C = sw ? b : cc;
    D = sw ? a : d;
    return ok;
}

// One Lehmer batch for a pair whose order is unknown: the batch runs on (larger, smaller) and the
// matrix comes back in the caller's naming, x' = A x - B y, y' = D y - C x.  Equal windows make
// the batch fail (its first quotient estimate is below 1) and the caller falls back to a
// long-division step.  Not ordering the multi-precision pair every round saves a full compare
// and a 4-operand swap per batch.
CF_DEV bool lehmer_batch(uint64_t xh, uint64_t yh, bool exact, uint64_t thr, uint32_t &A, uint32_t &B, uint32_t &C, uint32_t &D) {
    return lehmer_batch(xh, yh, exact, (double)thr, A, B, C, D);
}
// 2^tb as the batch's threshold (0 for tb <= 0): one v_ldexp_f64 instead of a 64-bit shift and a u64 -> f64 conversion
CF_DEV double lehmer_threshold(int tb) {
#if defined(COFHE_HOSTSIM)
    return tb <= 0 ? 0.0 : std::ldexp(1.0, tb);
#else
    return tb <= 0 ? 0.0 : __builtin_ldexp(1.0, tb);
#endif
}
CF_DEV bool lehmer_batch_unordered(uint64_t xh, uint64_t yh, bool exact, double thr, uint32_t &A, uint32_t &B,
                                   uint32_t &C, uint32_t &D) {
    const bool sw = xh < yh;
    uint32_t a, b, cc, d;
    const bool ok = lehmer_batch(sw ? yh : xh, sw ? xh : yh, exact, thr, a, b, cc, d);
    A = sw ? d : a;
    B = sw ? cc : b;
    C = sw ? b : cc;
    D = sw ? a : d;
    return ok;
}

template <int P>
CF_DEV void euclid_order(Ctx &c, Euclid<P> &s) {
    if (mp_cmp(c, s.x, s.y) < 0) {
        mp_swap(s.x, s.y);
        mp_swap(s.ux, s.uy);
        int t = s.sx; s.sx = s.sy; s.sy = t;
    }
}

// Runs the remainder sequence until the smaller of the pair has bitlen <= stop_bits (stop_bits < 0:
// until it is 0).  On return x >= y.
template <int P>
CF_DEV void euclid_run(Ctx &c, Euclid<P> &s, int stop_bits) {
    // every round removes >= 1 bit from the pair (a batch or a long-division step with a digit >= 1):
    // 2 * 1280 * P rounds bound any input; valid operands need ~40 per 1000 bits
    for (int guard = 0;; guard++) {
        if (guard > 2 * P * PLIMBS * 32) {
            CF_STATUS(c, CF_ST_EUCLID_CAP);
            break;
        }
        const int xb0 = mp_bitlen(c, s.x), yb0 = mp_bitlen(c, s.y);
        const int lo = xb0 < yb0 ? xb0 : yb0, hi = xb0 < yb0 ? yb0 : xb0;
        if (lo == 0 || lo <= stop_bits) break;
        bool done = false;
        if (hi - lo < LEHMER_WINDOW / 2) {
            int sh = hi > LEHMER_WINDOW ? hi - LEHMER_WINDOW : 0;
            uint64_t xh, yh;
            mp_bits64_pair(c, s.x, s.y, sh, xh, yh);
            const double thr = stop_bits >= 0 ? lehmer_threshold(stop_bits - sh) : 0.0;
            uint32_t A, B, C, D;
            if (lehmer_batch_unordered(xh, yh, sh == 0, thr, A, B, C, D)) {
                CF_STAT(g_stats.batches++);
                Mp<P> nx, ny;
                mp_lincomb_sub(c, nx, A, s.x, B, s.y);
                mp_lincomb_sub(c, ny, D, s.y, C, s.x);
                s.x = nx; s.y = ny;
                (void)mp_lincomb_add(c, nx, A, s.ux, B, s.uy);
                (void)mp_lincomb_add(c, ny, D, s.uy, C, s.ux);
                s.ux = nx; s.uy = ny;
                done = true;
            }
        }
        if (!done) {
            // long-division step on the ordered pair: x -= (qd << sh) * y, cofactor follows
            CF_STAT(g_stats.batch_steps++);
            euclid_order(c, s);
            int sh;
            uint32_t qd = mp_quot_digit(c, s.x, hi, s.y, lo, sh);
            Mp<P> ys = sh ? mp_shl(c, s.y, sh) : s.y;
            mp_lincomb_sub(c, s.x, 1u, s.x, qd, ys);
            Mp<P> us = sh ? mp_shl(c, s.uy, sh) : s.uy;
            (void)mp_lincomb_add(c, s.ux, 1u, s.ux, qd, us);
        }
    }
    euclid_order(c, s);
}

// ---------------------------------------------------------------------------- Euclid, workgroup form
// Same remainder sequence as euclid_run, but the group-uniform scalar work (the Lehmer batch)
// of all 64 limb groups of a 512-thread workgroup is done by ONE wavefront, one group per lane:
//   phase A  every group orders its pair, takes the leading 64 bits and posts a request in LDS
//   barrier
//   phase S  wavefront 0 runs lehmer_batch for the 64 requests (lane = group) and posts the
//            2x2 matrices; it also tells everybody whether any group is still running
//   barrier
//   phase M  every group applies its matrix (or takes a long-division step)
// The 8-fold redundant scalar loop of the in-wave version becomes 1 execution per 64 groups.
// Every thread of the workgroup must call this (uniform trip count by construction: the exit
// flag comes from LDS); the host simulator has no workgroups and uses euclid_run.
// What the serving lane does for ONE request: x and y are the 40-limb images of the group's pair in its LDS
// slice (words [0, 40) and [40, 80)); tx / ty: last known top limb index of each (remainders only shrink, so
// the scan starts at the higher of the two).  Returns the reply words (matrix in the group's naming) and updates sdone.
//   w0 = A | ok << 31, w1 = B | done << 31, w2 = C, w3 = D
// ok == 0 and not done: the group takes a long-division step (quotient beyond a batch, or sizes >= 31 bits apart).
constexpr int SERVE_WORDS = 4;      // reply words per request
CF_DEV void euclid_serve(const uint32_t *xs, int stop_bits, int &tx, int &ty, bool &sdone, uint32_t (&w)[SERVE_WORDS]) {
    const uint32_t *ys = xs + PLIMBS;
    // A long-division step orders the pair first (euclid_order renames x and y on the client), so each hint only
    // bounds the LARGER of the two: start both scans from the higher one.  (With the hints kept per name, a pair
    // that was swapped while its lengths differed by a limb or more got windows cut below its top limb -- wrong
    // matrices; found by tools/bench_ops.py in the 8th step of a fixed-base product, a 1042-bit first coefficient
    // against 2^218, and reproduced by tests/test_hostsim_device_code.py::test_compose_through_the_workgroup_protocol.)
    // LDS round trips are the serving lane's longest stalls (~110 cycles each, and it runs alone on its SIMD's issue port
    // while three wavefronts wait for it): the words it needs are fetched in two batches of independent loads -- the two
    // top candidates of each number (a round removes at most ~26 bits, so the top limb index drops by at most one), then
    // the six window words -- instead of one conditional load after the other (seven serialized round trips before).
    const int t = tx > ty ? tx : ty, t1 = t > 0 ? t - 1 : 0;
    const uint32_t xa = xs[t], xl = xs[t1], ya = ys[t], yl = ys[t1];
    uint32_t xt = xa, yt = ya;
    tx = ty = t;
    if (xt == 0u && t > 0) { tx = t1; xt = xl; }
    if (yt == 0u && t > 0) { ty = t1; yt = yl; }
    while (CF_UNLIKELY(tx > 0 && xt == 0u)) { tx--; xt = xs[tx]; }      // a number that is limbs shorter than its partner
    while (CF_UNLIKELY(ty > 0 && yt == 0u)) { ty--; yt = ys[ty]; }
    const int xb0 = xt ? tx * 32 + 32 - __builtin_clz(xt) : 0, yb0 = yt ? ty * 32 + 32 - __builtin_clz(yt) : 0;
    const int xb = xb0 < yb0 ? yb0 : xb0, yb = xb0 < yb0 ? xb0 : yb0;
    uint32_t A = 1, B = 0, C = 0, D = 1, ok = 0;
    if (yb == 0 || yb <= stop_bits) {
        sdone = true;
    } else if (xb - yb < LEHMER_WINDOW / 2) {
        const int sh = xb > LEHMER_WINDOW ? xb - LEHMER_WINDOW : 0, i0 = sh >> 5, o = sh & 31;
        // all six loads unconditionally (indices clamped to the last limb -- not to i0: the compiler turns a load from
        // `in range ? i0 + k : i0` back into `in range ? load : x0`), the out-of-range words masked afterwards: a conditional
        // load is compiled into an exec-mask region with its own wait
        const int i1 = i0 + 1 < PLIMBS - 1 ? i0 + 1 : PLIMBS - 1, i2 = i0 + 2 < PLIMBS - 1 ? i0 + 2 : PLIMBS - 1;
        const uint32_t x0 = xs[i0], x1r = xs[i1], x2r = xs[i2], y0 = ys[i0], y1r = ys[i1], y2r = ys[i2];
        // (opaque masks: `load & (in range ? ~0 : 0)` would be folded back into a conditional load as well)
        const uint32_t in1 = opaque(i0 + 1 < PLIMBS ? 0xFFFFFFFFu : 0u), in2 = opaque(i0 + 2 < PLIMBS ? 0xFFFFFFFFu : 0u);
        const uint32_t x1 = x1r & in1, x2 = x2r & in2, y1 = y1r & in1, y2 = y2r & in2;
        const uint64_t xl = ((uint64_t)x1 << 32) | x0, yl = ((uint64_t)y1 << 32) | y0;
        const uint64_t xh = o ? ((xl >> o) | ((uint64_t)x2 << (64 - o))) : xl;
        const uint64_t yh = o ? ((yl >> o) | ((uint64_t)y2 << (64 - o))) : yl;
        const double thr = stop_bits >= 0 ? lehmer_threshold(stop_bits - sh) : 0.0;
        if (sh == 0 && xh == yh) {
            // x == y: every full sequence ends here, because the batch's quotient is biased low and an exact last division
            // k g / g comes out as k - 1 (leaving g, g); the batch cannot step on equal windows, and the group used to take
            // the long-division route for what is one subtraction:  x' = x - y = 0, y' = y
            A = 1; B = 1; C = 0; D = 1;
            ok = 1;
        } else {
            ok = lehmer_batch_unordered(xh, yh, sh == 0, thr, A, B, C, D) ? 1u : 0u;
        }
    }
    w[0] = A | (ok << 31);
    w[1] = B | (sdone ? 0x80000000u : 0u);
    w[2] = C;
    w[3] = D;
}

// What the protocol needs from the machine; the host simulator maps it to thread barriers (lane.hpp: WgShared), so the
// code below is the same on both.
#if defined(COFHE_HOSTSIM)
#define CF_WG_TID(c) ((unsigned)(c).tid)
#define CF_WG_BARRIER(c) (c).wg->bar.wait((c).wg_sense)
#define CF_SETPRIO(n) do { } while (0)
inline bool cf_server_any(Ctx &c, bool p) {          // ballot of the serving wavefront != 0
    c.wg->vote[c.tid & 63] = p ? 1u : 0u;
    c.wg->wave_bar.wait(c.wave_sense);
    uint32_t m = 0;
    for (int i = 0; i < c.wg->wave_threads; i++) m |= c.wg->vote[i];
    c.wg->wave_bar.wait(c.wave_sense);
    return m != 0;
}
#else
#define CF_WG_TID(c) (threadIdx.x)
#define CF_WG_BARRIER(c) __syncthreads()
#define CF_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
CF_DEV bool cf_server_any(Ctx &, bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
#endif
// One round of service by the serving wavefront: lane l answers request l (the pair in group l's LDS slice) with its reply
// record in the mailbox.  "Somebody is still running" travels in bit 30 of every group's second reply word (B < 2^26): the
// clients read ONE 16-byte record per round instead of a flag word and then, behind a branch, the record.  Returns that flag.
CF_DEV bool euclid_serve_round(Ctx &c, int l, const uint32_t *stopw, int &tx, int &ty, bool &sdone) {
    uint32_t w[SERVE_WORDS] = {1u, 0x80000000u, 0u, 1u};
    if (l < WG_GROUPS && !sdone) euclid_serve(c.wg_scr0 + l * SCRATCH_WORDS, (int)stopw[l], tx, ty, sdone, w);
    const bool any = cf_server_any(c, l < WG_GROUPS && !sdone);
    if (l < WG_GROUPS) {
        w[1] |= any ? 0x40000000u : 0u;
        uint32_t *o = c.wg_mail + l * SERVE_WORDS;
        CF_UNROLL for (int k = 0; k < SERVE_WORDS; k++) o[k] = w[k];
    }
    return any;
}
template <int P>
CF_DEV void euclid_run_wg(Ctx &c, Euclid<P> &s, int stop_bits) {
    static_assert(P == 1, "the serving lane reads single-plane images");
    // Per round a group only stashes its pair in its LDS slice; the serving lane finds the bit lengths (scanning
    // down from the last top limb), cuts the 64-bit windows, decides "done" / "long step" and runs the batch:
    // that scalar work costs the server ~50 instructions per round and used to cost every client wavefront
    // ~95 (two bit lengths over 8 lanes, window reads and funnel shifts, threshold, request record).
    uint32_t *mail = c.wg_mail;
    uint32_t *res = mail + c.gi * SERVE_WORDS;
    uint32_t *stopw = mail + WG_GROUPS * SERVE_WORDS + 4;          // per group: where its partial sequence stops
    uint32_t *stash = c.scratch();
    bool done = false;
    if (c.gl == 0) stopw[c.gi] = (uint32_t)stop_bits;
    int tx = PLIMBS - 1, ty = PLIMBS - 1;       // serving lane: top limb indices of its group's pair
    bool sdone = false;
    // the slices are LDS scratch of the arithmetic in between: the last reader of the previous user is this
    // group itself, so no barrier is needed before the first stash
    // Round cap: a round removes >= 28 bits from the pair of every running group unless it takes the long-step
    // route (>= 1 bit); 1024 rounds cover the worst all-single-digit sequence of 1280-bit operands many times
    // over (valid operands need ~55).  Hitting it is reported, not silent.
    bool capped = true;
    for (int round = 0; round < 1024; round++) {
        // The SIMD arbiter issues the oldest wavefront first, so of the four workgroups that start
        // together on a CU the first to arrive finished ~20 % before the last (tools/wg_timing.hip) and the CU
        // drained at falling occupancy.  Rotating the user priority of the client phases over three levels
        // (the serving phase keeps level 3) shares the issue slots.  Only when the whole grid is resident from
        // the start (rank >= 0): with more workgroups than slots oldest-first is the better order.
        if (c.rank >= 0) {
            switch ((c.rank + round) % 3) {
                case 0: CF_SETPRIO(0); break;
                case 1: CF_SETPRIO(1); break;
                default: CF_SETPRIO(2); break;
            }
        }
#ifdef COFHE_WG_TIMING
        const unsigned long long tq0 = wall_clock64();
#endif
        if (!done) {
            CF_UNROLL for (int j = 0; j < CH; j++) {
                stash[c.gl * CH + j] = s.x.v[0][j];
                stash[PLIMBS + c.gl * CH + j] = s.y.v[0][j];
            }
        }
        CF_WG_BARRIER(c);
        if (c.wave == 0) {
            // the other wavefronts of the workgroup wait for this one: let it issue first
            CF_SETPRIO(3);
            const int l = (int)(CF_WG_TID(c) & 63);       // lane = request index
#ifdef COFHE_WG_TIMING
            const unsigned long long ts0 = wall_clock64();
#endif
            (void)euclid_serve_round(c, l, stopw, tx, ty, sdone);
#ifdef COFHE_WG_TIMING
            c.t_serve += wall_clock64() - ts0;
#endif
            CF_SETPRIO(0);
        }
        CF_WG_BARRIER(c);
#ifdef COFHE_WG_TIMING
        const unsigned long long tq1 = wall_clock64();
        c.t_wait += tq1 - tq0;
        c.n_rounds++;
#endif
        // the four reply words in ONE LDS load (read one after the other behind a flag word and the branches that need
        // them, they were three serialized round trips on every wavefront's critical path)
        const uint32_t a0 = res[0], b0 = res[1], c0 = res[2], d0 = res[3];
        if ((b0 & 0x40000000u) == 0) {            // nobody is still running
            capped = false;
            break;
        }
        if (!done) {
            if (CF_UNLIKELY(b0 >> 31)) {
                done = true;
            } else if (CF_LIKELY(a0 >> 31)) {
                const uint32_t A = a0 & 0x7FFFFFFFu, B = b0 & 0x3FFFFFFFu, C = c0, D = d0;
                Mp<P> nx, ny;
                mp_lincomb_sub(c, nx, A, s.x, B, s.y);
                mp_lincomb_sub(c, ny, D, s.y, C, s.x);
                s.x = nx; s.y = ny;
                (void)mp_lincomb_add(c, nx, A, s.ux, B, s.uy);
                (void)mp_lincomb_add(c, ny, D, s.uy, C, s.ux);
                s.ux = nx; s.uy = ny;
            } else {
                // rare: quotient beyond a batch (or equal windows) -- order the pair, one long-division step
                CF_FLAG(4u);
                euclid_order(c, s);
                const int xb = mp_bitlen(c, s.x), yb = mp_bitlen(c, s.y);
                int sh;
                uint32_t qd = mp_quot_digit(c, s.x, xb, s.y, yb, sh);
                Mp<P> ys = sh ? mp_shl(c, s.y, sh) : s.y;
                mp_lincomb_sub(c, s.x, 1u, s.x, qd, ys);
                Mp<P> us = sh ? mp_shl(c, s.uy, sh) : s.uy;
                (void)mp_lincomb_add(c, s.ux, 1u, s.ux, qd, us);
            }
        }
#ifdef COFHE_WG_TIMING
        c.t_apply += wall_clock64() - tq1;
#endif
    }
    if (capped) CF_STATUS(c, CF_ST_EUCLID_CAP);
    // The client wavefronts deliberately keep the rotated priority of their last round through the phases that
    // follow (the serving wavefront is back at 0): resetting every wavefront to 0 here puts the co-resident
    // workgroups back into oldest-first order and measured 4 % slower on the 128x128 composition (0.543 vs
    // 0.522 ms, three interleaved rounds, gpurun_out/r2_variants.log).  Priorities end with the wavefront.
    // leave with x >= y like euclid_run
    euclid_order(c, s);
}


}  // namespace cofhe
