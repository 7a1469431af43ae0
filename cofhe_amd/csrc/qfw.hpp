// qfw.hpp -- quadratic forms in the wavefront-wide layout (wide.hpp): reduction and composition for ONE chain of dependent
// compositions (the ladder of a decryption).  The formulas are those of qf.hpp (Shanks / Atkin NUCOMP with one set of
// formulas, adaptive partial-Euclid bound, c' from (b'^2 + |Delta|) / 4a', full reduction: the unique reduced form, hence
// bit-identical results); only the COMMON route is restated here.  Whatever is rare -- a common factor of the first
// coefficients left over by the coprime-representative step, a second coefficient or an intermediate wider than expected, a
// loop that runs into its cap -- makes wf_compose return false with nothing written, and the kernel takes that one
// composition through qf_compose on 8 lanes of the same wavefront (cofhe_hip.hip: compose_wide_or_fallback), so every input
// the throughput kernels accept gives the same bytes here.
#pragma once
#include "layout.hpp"
#include "qf.hpp"
#include "wide.hpp"

namespace cofhe {
namespace wide {

// phase accounting of the diagnostic build tools/wide_timing.hip (shader-clock ticks summed in LDS by lane 0; nothing of
// this is compiled into the library)
#if defined(COFHE_WIDE_TIMING) && !defined(COFHE_HOSTSIM)
CF_W unsigned long long *wt_slots() {
    __shared__ unsigned long long s[32];
    return s;
}
#define WT_T() __builtin_amdgcn_s_memtime()
#define WT_ADD(id, t0) do { const unsigned long long d_ = WT_T() - (t0); if (lane_id() == 0) atomicAdd(&wt_slots()[id], d_); } while (0)
#define WT_LAP(id, t0) do { const unsigned long long n_ = WT_T(); if (lane_id() == 0) atomicAdd(&wt_slots()[id], n_ - (t0)); (t0) = n_; } while (0)
#else
#define WT_T() 0ull
#define WT_ADD(id, t0) do { } while (0)
#define WT_LAP(id, t0) do { (void)(t0); } while (0)
#endif

struct WForm {          // a, |b| within 40 limbs, c within 80; the sign of b is wave-uniform
    WN a, bm, c;
    int bneg;
};

#if defined(COFHE_HOSTSIM)
inline WForm wf_load(const uint32_t *rec) {
    WForm f;
    for (int L = 0; L < WL; L++) {
        f.a.a.v[L] = L < PLIMBS / 2 ? rec[REC_A + 2 * L] : 0u;
        f.a.b.v[L] = L < PLIMBS / 2 ? rec[REC_A + 2 * L + 1] : 0u;
        f.bm.a.v[L] = L < PLIMBS / 2 ? rec[REC_B + 2 * L] : 0u;
        f.bm.b.v[L] = L < PLIMBS / 2 ? rec[REC_B + 2 * L + 1] : 0u;
        f.c.a.v[L] = L < PLIMBS ? rec[REC_C + 2 * L] : 0u;
        f.c.b.v[L] = L < PLIMBS ? rec[REC_C + 2 * L + 1] : 0u;
    }
    f.bneg = (int)rec[REC_SIGN];
    return f;
}
inline void wf_store(const WForm &f, uint32_t *rec) {
    for (int L = 0; L < PLIMBS / 2; L++) {
        rec[REC_A + 2 * L] = f.a.a.v[L];
        rec[REC_A + 2 * L + 1] = f.a.b.v[L];
        rec[REC_B + 2 * L] = f.bm.a.v[L];
        rec[REC_B + 2 * L + 1] = f.bm.b.v[L];
    }
    for (int L = 0; L < PLIMBS; L++) {
        rec[REC_C + 2 * L] = f.c.a.v[L];
        rec[REC_C + 2 * L + 1] = f.c.b.v[L];
    }
    rec[REC_SIGN] = (uint32_t)f.bneg;
    for (int i = 1; i < 8; i++) rec[REC_SIGN + i] = 0u;
}
#else
CF_W WForm wf_load(const uint32_t *rec) {
    const uint32_t L = lane_id();
    WForm f;
    const bool p1 = L < PLIMBS / 2, p2 = L < PLIMBS;
    f.a.a = p1 ? rec[REC_A + 2 * L] : 0u;
    f.a.b = p1 ? rec[REC_A + 2 * L + 1] : 0u;
    f.bm.a = p1 ? rec[REC_B + 2 * L] : 0u;
    f.bm.b = p1 ? rec[REC_B + 2 * L + 1] : 0u;
    f.c.a = p2 ? rec[REC_C + 2 * L] : 0u;
    f.c.b = p2 ? rec[REC_C + 2 * L + 1] : 0u;
    f.bneg = (int)rec[REC_SIGN];
    return f;
}
CF_W void wf_store(const WForm &f, uint32_t *rec) {
    const uint32_t L = lane_id();
    if (L < PLIMBS / 2) {
        rec[REC_A + 2 * L] = f.a.a;
        rec[REC_A + 2 * L + 1] = f.a.b;
        rec[REC_B + 2 * L] = f.bm.a;
        rec[REC_B + 2 * L + 1] = f.bm.b;
    }
    if (L < PLIMBS) {
        rec[REC_C + 2 * L] = f.c.a;
        rec[REC_C + 2 * L + 1] = f.c.b;
    }
    if (L < 8) rec[REC_SIGN + L] = L == 0 ? (uint32_t)f.bneg : 0u;
}
#endif

// a conservative quotient digit for num / den (mp.hpp: mp_quot_digit): qd < 2^31 and sh >= 0 with (qd << sh) den <= num
CF_W uint32_t w_quot_digit(const WN &num, int nb, const WN &den, int db, int &sh, bool &ok) {
    const int npos = nb > 64 ? nb - 64 : 0, dpos = db > 32 ? db - 32 : 0;
    const uint64_t nt = w_bits64(num, npos);
    const uint64_t dt = (uint64_t)(uint32_t)w_bits64(den, dpos) + (dpos > 0 ? 1u : 0u);
    int e = npos - dpos;
    if (dt == 0) {
        ok = false;
        sh = 0;
        return 1;
    }
    uint64_t t = (dpos == 0 && npos == 0) ? nt / dt : div64_lower(nt, dt);
    if (e < 0) {
        t = (-e >= 64) ? 0 : (t >> (-e));
        e = 0;
    }
    const int extra = 33 - __builtin_clzll(t | 1);
    if (extra > 0) {
        t >>= extra;
        e += extra;
    }
    sh = e;
    if (t == 0) {
        sh = 0;
        return 1;
    }
    return (uint32_t)t;
}

// (a, b, c) any positive definite form within the capacity; reduced on return.  false: the step cap was hit (the caller
// falls back; never for the composition's outputs, which are a few steps from reduced)
CF_W bool wf_reduce(WN &a, SW &b, WN &cc) {
    for (int guard = 0; guard < 256; guard++) {
        const int cm = w_cmp(b.m, a);
        if (cm > 0 || (cm == 0 && b.neg)) {
            WN two_a;
            (void)w_add(two_a, a, a);
            if (w_cmp(b.m, two_a) < 0) {
                WN t, u;
                w_sub(t, two_a, b.m);
                (void)w_add(u, cc, a);
                w_sub(cc, u, b.m);
                b.m = t;
                b.neg ^= 1;
            } else {
                const int nb = w_bitlen(b.m), db = w_bitlen(two_a);
                int sh;
                bool ok = true;
                const uint32_t qd = w_quot_digit(b.m, nb, two_a, db, sh, ok);
                if (!ok) return false;
                const WN ds = sh ? w_shl(two_a, sh) : two_a;
                WN nbm, half;
                (void)w_lincomb_sub(nbm, 1u, b.m, qd, ds);             // |b'| = |b| - q 2a >= 0
                (void)w_add(half, b.m, nbm);
                half = w_shr(half, 1);                                  // (|b| + |b'|) / 2, exact
                const WN hs = sh ? w_shl(half, sh) : half;
                WN nc;
                (void)w_lincomb_sub(nc, 1u, cc, qd, hs);                // c' = c - q (|b| + |b'|) / 2
                cc = nc;
                b.m = nbm;
            }
            continue;
        }
        const int ac = w_cmp(a, cc);
        if (ac > 0) {
            const WN t = a;
            a = cc;
            cc = t;
            b.neg ^= 1;
            continue;
        }
        if (w_is_zero(b.m)) b.neg = 0;
        if (ac == 0 && b.neg) b.neg = 0;
        return true;
    }
    return false;
}

// x mod 223092870 (mp.hpp: mp_mod_primorial): two tabulated limb weights per lane, a butterfly sum over the wavefront
struct PrimorialWeightsW { uint32_t w[WLIMBS]; };
constexpr PrimorialWeightsW primorial_weights_w() {
    PrimorialWeightsW t{};
    uint64_t v = 1;
    for (int i = 0; i < WLIMBS; i++) {
        t.w[i] = (uint32_t)v;
        v = (v << 32) % PRIMORIAL23;
    }
    return t;
}
#if defined(COFHE_HOSTSIM)
static constexpr PrimorialWeightsW PRIMORIAL_WW = primorial_weights_w();
inline V32 primorial_weight(int odd) {
    V32 r;
    for (int L = 0; L < WL; L++) r.v[L] = PRIMORIAL_WW.w[2 * L + odd];
    return r;
}
inline V32 mod_const(const V64 &x) {
    V32 r;
    for (int L = 0; L < WL; L++) r.v[L] = (uint32_t)(x.v[L] % PRIMORIAL23);
    return r;
}
#else
__device__ static constexpr PrimorialWeightsW PRIMORIAL_WW = primorial_weights_w();
CF_W V32 primorial_weight(int odd) { return PRIMORIAL_WW.w[2 * lane_id() + odd]; }
CF_W V32 mod_const(V64 x) { return (uint32_t)(x % PRIMORIAL23); }
#endif
CF_W uint32_t w_mod_primorial(const WN &x) {
    V32 v = mod_const(mad(x.a, primorial_weight(0), mad(x.b, primorial_weight(1), V64(0ull))));
    const V32 L = lane_id();
    for (int k = 1; k < WL; k <<= 1) {                       // butterfly: every lane ends with the sum mod M
        const V32 o = bperm(v, L ^ V32((uint32_t)k));
        const V32 s = v + o;                                 // < 2^29
        v = sel(s >= V32(PRIMORIAL23), s - V32(PRIMORIAL23), s);
    }
    return rdlane(v, 0);
}

// the remainder sequence of euclid_run (mp.hpp) on the wavefront itself: windows from readlanes, the batch uniform on all
// lanes, four linear combinations.  false: cap hit / division by zero (fallback)
#ifndef WIDE_LEHMER_CAP
#define WIDE_LEHMER_CAP 12          // double-steps per batch at most (it ends by itself when the cofactors are full)
#endif
struct WEuclid {
    WN x, y, ux, uy;
    int sx, sy;
};
CF_W void we_order(WEuclid &s) {
    if (w_cmp(s.x, s.y) < 0) {
        WN t = s.x; s.x = s.y; s.y = t;
        t = s.ux; s.ux = s.uy; s.uy = t;
        const int q = s.sx; s.sx = s.sy; s.sy = q;
    }
}
CF_W bool w_euclid(WEuclid &s, int stop_bits) {
    bool fine = false;
    for (int guard = 0; guard < 600; guard++) {
        unsigned long long wt = WT_T();
        const WTop tp = w_top_pair(s.x, s.y);
        int xb0 = top_bitlen(tp.xh, tp.xl), yb0 = top_bitlen(tp.yh, tp.yl);
        xb0 = xb0 ? xb0 + tp.base : (tp.base ? w_bitlen(s.x) : 0);         // empty view: the number ends below it (or is zero)
        yb0 = yb0 ? yb0 + tp.base : (tp.base ? w_bitlen(s.y) : 0);
        const int lo_ = xb0 < yb0 ? xb0 : yb0, hi_ = xb0 < yb0 ? yb0 : xb0;
        if (lo_ == 0 || lo_ <= stop_bits) {
            fine = true;
            break;
        }
        bool done = false;
        if (hi_ - lo_ < LEHMER_WINDOW / 2) {
            const int sh = hi_ > LEHMER_WINDOW ? hi_ - LEHMER_WINDOW : 0;
            // both numbers reach into the view here (hi - lo < 27, the longer one fills lane L), and sh >= base
            const uint64_t xh = top_bits64(tp.xh, tp.xl, sh - tp.base), yh = top_bits64(tp.yh, tp.yl, sh - tp.base);
            uint32_t A = 1, B = 1, C = 0, D = 1;
            bool ok;
            WT_LAP(8, wt);
            if (sh == 0 && xh == yh) {
                ok = true;                                   // x == y: x' = x - y = 0, y' = y (see euclid_serve)
            } else {
                const double thr = stop_bits >= 0 ? lehmer_threshold(stop_bits - sh) : 0.0;
                ok = lehmer_batch_uniform_unordered<WIDE_LEHMER_CAP>(xh, yh, sh == 0, thr, A, B, C, D);
            }
            WT_LAP(9, wt);
            if (ok) {
                WN nx, ny;
                (void)w_lincomb_sub(nx, A, s.x, B, s.y);
                (void)w_lincomb_sub(ny, D, s.y, C, s.x);
                s.x = nx; s.y = ny;
                (void)w_lincomb_add(nx, A, s.ux, B, s.uy);
                (void)w_lincomb_add(ny, D, s.uy, C, s.ux);
                s.ux = nx; s.uy = ny;
                done = true;
                WT_LAP(10, wt);
            }
        }
        if (!done) {
            we_order(s);
            int sh;
            bool ok = true;
            const uint32_t qd = w_quot_digit(s.x, hi_, s.y, lo_, sh, ok);
            if (!ok) return false;
            const WN ys = sh ? w_shl(s.y, sh) : s.y;
            WN t;
            (void)w_lincomb_sub(t, 1u, s.x, qd, ys);
            s.x = t;
            const WN us = sh ? w_shl(s.uy, sh) : s.uy;
            (void)w_lincomb_add(t, 1u, s.ux, qd, us);
            s.ux = t;
            WT_LAP(11, wt);
        }
    }
    we_order(s);
    return fine;
}

// exact signed division n / v (v > 0 divides n)
CF_W SW sw_div_exact(const SW &n, const WN &v, bool &ok) {
    const int nb = w_bitlen(n.m), vb = w_bitlen(v);
    SW r;
    r.m = w_divexact(n.m, v, nb == 0 ? 0 : (nb - vb + 1 + 63) / 64, ok);
    r.neg = n.neg;
    return r;
}

// out = reduced(fa * fb): the common route of qf_compose (qf.hpp).  false: take the 8-lane route for this pair.
CF_W bool wf_compose(WForm &out, const WForm &fa, const WForm &fb, const QDisc &dd) {
    const int half_dbits = dd.half_dbits;
    const int plane_bits = PLIMBS * 32;
    // coprime representative of the second operand (qf.hpp has the derivation)
    // (the capacity is no constraint here -- 4096 bits -- so the representative may be anything a plane-and-a-bit long; forms
    // with an unusually small first coefficient, whose c is far longer, take the 8-lane route)
    if (w_bitlen(fb.c) > plane_bits + 64 || w_bitlen(fa.c) > 2 * plane_bits - 64) return false;
    unsigned long long wt = WT_T();
    WForm fbr = fb;
    const bool same = w_cmp(fa.a, fb.a) == 0;                     // a squaring, or a form with its inverse (qf.hpp)
    {
        const uint32_t M = PRIMORIAL23;
        const uint32_t ra1 = w_mod_primorial(fa.a), ra2 = w_mod_primorial(fb.a);
        uint32_t rb2 = w_mod_primorial(fb.bm);
        if (fb.bneg && rb2) rb2 = M - rb2;
        const uint32_t rc2 = w_mod_primorial(fb.c);
        const uint32_t rb2n = rb2 ? M - rb2 : 0u;
        const uint32_t cand[6] = {ra2, rc2, (uint32_t)(((uint64_t)ra2 + rb2 + rc2) % M), (uint32_t)(((uint64_t)ra2 + rb2n + rc2) % M),
                                  (uint32_t)(((uint64_t)ra2 + 2ull * rb2 + 4ull * rc2) % M),
                                  (uint32_t)(((uint64_t)ra2 + 2ull * rb2n + 4ull * rc2) % M)};
        int pick = -1;
        const uint32_t primes[9] = {2, 3, 5, 7, 11, 13, 17, 19, 23};
        for (int k = same ? 1 : 0; k < 6 && pick < 0; k++) {
            bool okc = true;
            for (int i = 0; i < 9; i++) okc = okc && !((cand[k] % primes[i]) == 0 && (ra1 % primes[i]) == 0);
            if (okc) pick = k;
        }
        if (pick == 1) {
            fbr.a = fb.c;
            fbr.c = fb.a;
            fbr.bneg = w_is_zero(fb.bm) ? 0 : (fb.bneg ^ 1);
        } else if (pick >= 2) {
            const uint32_t kk = pick >= 4 ? 2u : 1u;
            WN t, two_c, kb, na, nb;
            (void)w_lincomb_add(t, 1u, fb.a, kk * kk, fb.c);
            (void)w_lincomb_add(two_c, 2u * kk, fb.c, 0u, fb.c);
            (void)w_lincomb_add(kb, kk, fb.bm, 0u, fb.bm);
            const bool up = (pick & 1) == 0;
            const bool plus = up != (fb.bneg != 0);
            if (plus) {
                (void)w_add(na, t, kb);
                (void)w_add(nb, two_c, fb.bm);
            } else {
                w_sub(na, t, kb);
                w_sub(nb, two_c, fb.bm);
            }
            fbr.a = na;
            fbr.bm = nb;
            fbr.bneg = up ? 0 : 1;
        }
    }
    const bool sw = w_cmp(fa.a, fbr.a) < 0;
    const WForm &f1 = sw ? fbr : fa, &f2 = sw ? fa : fbr;          // a1 >= a2
    if (w_bitlen(f1.a) > plane_bits + 64 || w_bitlen(f2.c) > 2 * plane_bits - 64) return false;
    const SW b1{f1.bm, f1.bneg}, b2{f2.bm, f2.bneg};
    SW s = sw_add(b1, b2), m = sw_sub(b1, b2);
    s.m = w_shr(s.m, 1);
    m.m = w_shr(m.m, 1);

    // d = gcd(a1, a2), y1 a2 == d (mod a1)
    WEuclid e;
    e.x = f1.a; e.y = f2.a;
    e.ux = w_zero(); e.uy = w_word(1u);
    e.sx = -1; e.sy = 1;
    WT_LAP(0, wt);
    if (!w_euclid(e, -1)) return false;
    WT_LAP(1, wt);
    if (!w_is_word(e.x, 1u)) return false;                         // a common factor: the 8-lane route has all the formulas
    const WN &v1 = f1.a, &v2 = f2.a;
    const SW y1{e.ux, e.sx < 0};
    bool ok = true;
    WN r;
    {
        const SW t = sw_mul(y1, m);
        r = w_mod(t.m, v1, ok);
        if (!ok) return false;
        if (t.neg && !w_is_zero(r)) {
            WN u;
            w_sub(u, v1, r);
            r = u;
        }
    }
    WT_LAP(2, wt);
    // partial Euclid on (v1, r)
    const int lv1 = w_bitlen(v1), lv2 = w_bitlen(v2);
    const int stop = (lv1 - lv2 + half_dbits) / 2;
    WEuclid pe;
    pe.x = v1; pe.y = r;
    pe.ux = w_zero(); pe.uy = w_word(1u);
    pe.sx = -1; pe.sy = 1;
    if (!w_euclid(pe, stop)) return false;
    WT_LAP(3, wt);
    const SW C0{pe.ux, pe.sx < 0}, C1{pe.uy, pe.sy < 0};
    const int sg_neg = C1.neg;
    const SW R1{pe.y, 0}, R0{pe.x, 0};
    // M1 = (v2 R - m C) / v1, M2 = (s R + c2 C) / v1 for (R1, C1)
    const SW M1 = sw_div_exact(sw_sub(sw_mul(SW{v2, 0}, R1), sw_mul(m, C1)), v1, ok);
    const SW M2 = sw_div_exact(sw_add(sw_mul(s, R1), sw_mul(SW{f2.c, 0}, C1)), v1, ok);
    if (!ok) return false;
    WT_LAP(4, wt);
    const SW an = sw_add(sw_mul(R1, M1), sw_mul(C1, M2));
    const SW bs = sw_add(sw_mul(R0, M1), sw_mul(C0, M2));
    // b' = -sg 2 bs - b1
    SW two_bs;
    (void)w_add(two_bs.m, bs.m, bs.m);
    two_bs.neg = bs.neg ^ (sg_neg ? 0 : 1);
    SW bn = sw_sub(two_bs, b1);
    if (w_bitlen(an.m) >= plane_bits - 8 || w_bitlen(bn.m) >= plane_bits - 8 || w_is_zero(an.m)) return false;
    WT_LAP(5, wt);
    // c' = (b'^2 + |Delta|) / (4 a')
    WN num = w_mul(bn.m, bn.m), dl;
#if defined(COFHE_HOSTSIM)
    for (int L = 0; L < WL; L++) {
        dl.a.v[L] = L < PLIMBS ? dd.absdelta[2 * L] : 0u;
        dl.b.v[L] = L < PLIMBS ? dd.absdelta[2 * L + 1] : 0u;
    }
#else
    {
        const uint32_t L = lane_id();
        dl.a = L < PLIMBS ? dd.absdelta[2 * L] : 0u;
        dl.b = L < PLIMBS ? dd.absdelta[2 * L + 1] : 0u;
    }
#endif
    WN nd;
    (void)w_add(nd, num, dl);
    nd = w_shr(nd, 2);
    WN cn = w_divexact(nd, an.m, (w_bitlen(nd) - w_bitlen(an.m) + 1 + 63) / 64, ok);
    if (!ok) return false;
    if (w_bitlen(cn) >= plane_bits - 8) return false;              // the throughput kernels reduce such a form at double width
    WN a1 = an.m;
    WT_LAP(6, wt);
    if (!wf_reduce(a1, bn, cn)) return false;
    WT_LAP(7, wt);
    out.a = a1;
    out.bm = bn.m;
    out.bneg = bn.neg;
    out.c = cn;
    return true;
}

}  // namespace wide
}  // namespace cofhe
