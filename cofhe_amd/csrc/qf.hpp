// qf.hpp -- binary quadratic forms of negative discriminant on a limb group: reduction,
// composition (NUCOMP family) and exponentiation.  These are the device-side replacements of
// the BICYCL calls the reference makes on its hot path:
//   ClassGroup::nucomp / QFI::nucomp    include/x86_64/cpu_cryptosystem_tensor_ops.inl:260-261,
//                                       :409-414; include/x86_64/qfi.inl:26,111,127
//   QFI::nudupl                         include/x86_64/qfi.inl:23,57
//   ClassGroup::nupow / qfi_nupow       cpu_cryptosystem_tensor_ops.inl:334-335; qfi.inl:1-135
// Every routine returns the REDUCED form of the class (-a < b <= a <= c, b >= 0 when a == c),
// which is unique, so the results are bit-identical to any correct CPU implementation
// regardless of the partial-Euclid bound, window width or association order used there.
//
// Composition follows Shanks/Atkin NUCOMP (Cohen, CCANT Alg. 5.4.7 + 5.4.9; Jacobson-van der
// Poorten) with one set of formulas for every gcd structure (so squaring = NUDUPL is the same
// code with f1 == f2):
//   order a1 >= a2;  s = (b1+b2)/2, m = (b1-b2)/2
//   d = gcd(a1, a2) = y1*a2 (mod a1);   d1 = gcd(s, d) = x2*s + (-y2)*d
//   v1 = a1/d1, v2 = a2/d1;  r = (y1*y2*(-m) - x2*c2) mod v1       (d == 1:  r = y1*m mod a1)
//   partial Euclid on (v1, r) with cofactors C_i (R_i == C_i * r mod v1) down to
//       bits(R1) <= (bits v1 - bits v2 + bits(Delta)/2) / 2
//   for a pair (R, C):   M1 = (v2 R - m C)/v1,   M2 = (s R + c2 d1 C)/v1     (exact)
//   a' = R1 M1(R1,C1) + C1 M2(R1,C1)        c' = R0 M1(R0,C0) + C0 M2(R0,C0)
//   b' = -sign(C1) * 2 (R0 M1(R1,C1) + C0 M2(R1,C1)) - b1
//   (a', b', c') is equivalent to f1*f2 with every coefficient near sqrt|Delta|; reduce it.
// No coefficient of Delta is needed on the device, only its bit length.
#pragma once
#include "mp.hpp"

namespace cofhe {

struct QForm {          // registers of one lane: a, |b| single width, c double width
    Mp<1> a;
    Mp<1> bm;
    int bneg;
    Mp<2> c;
};

// ---------------------------------------------------------------------------- reduction
// (a, b, c) any positive definite form whose coefficients fit P planes with a bit to spare; on
// return it is reduced.
template <int P>
CF_DEV void qf_reduce(Ctx &c, Mp<P> &a, SMp<P> &b, Mp<P> &cc) {
    // every normalisation + swap pair shrinks a (a' = c' < a): bounded by the bit length; the cap only ever
    // triggers on garbage (coefficients that are not a positive definite form)
    for (int guard = 0;; guard++) {
        if (CF_UNLIKELY(guard > 4 * P * PLIMBS * 32)) {
            CF_STATUS(c, CF_ST_REDUCE_CAP);
            return;
        }
        int cm = mp_cmp(c, b.m, a);
        if (cm > 0 || (cm == 0 && b.neg)) {
            // normalise b into (-a, a]
            Mp<P> two_a;
            (void)mp_add(c, two_a, a, a);
            if (mp_cmp(c, b.m, two_a) < 0) {
                // a < |b| < 2a, or b == -a:  b' = b - 2a*sgn(b),  c' = c + a - |b|
                Mp<P> t;
                mp_sub(c, t, two_a, b.m);
                Mp<P> u;
                (void)mp_add(c, u, cc, a);
                mp_sub(c, cc, u, b.m);
                b.m = t;
                b.neg ^= 1;
            } else {
                int nb = mp_bitlen(c, b.m), db = mp_bitlen(c, two_a), sh;
                uint32_t qd = mp_quot_digit(c, b.m, nb, two_a, db, sh);
                Mp<P> ds = sh ? mp_shl(c, two_a, sh) : two_a;
                Mp<P> nbm;
                mp_lincomb_sub(c, nbm, 1u, b.m, qd, ds);          // |b'| = |b| - q*2a >= 0
                Mp<P> half;
                uint32_t top = mp_add(c, half, b.m, nbm);         // (|b| + |b'|)/2, exact
                half = mp_shr1(c, half);
                if (top) {   // the sum carried out of the top plane: restore the lost bit
                    Mp<P> tb;
                    mp_zero(tb);
                    tb.v[P - 1][CH - 1] = (c.gl == G - 1) ? 0x80000000u : 0u;
                    (void)mp_add(c, half, half, tb);
                }
                Mp<P> hs = sh ? mp_shl(c, half, sh) : half;
                mp_lincomb_sub(c, cc, 1u, cc, qd, hs);            // c' = c - q*(|b|+|b'|)/2
                b.m = nbm;
            }
            continue;
        }
        int ac = mp_cmp(c, a, cc);
        if (ac > 0) {
            mp_swap(a, cc);
            b.neg ^= 1;
            continue;
        }
        if (mp_is_zero(c, b.m)) b.neg = 0;
        if (ac == 0 && b.neg) b.neg = 0;
        return;
    }
}

// ---------------------------------------------------------------------------- helpers
// |t| mod v as a residue in [0, v) of the signed value t
template <int P>
CF_DEV Mp<1> smod(Ctx &c, const SMp<P> &t, const Mp<1> &v) {
    Mp<P> rem = t.m, q;
    mp_divrem(c, rem, v, q);
    Mp<1> r = mp_resize<1>(rem);
    if (t.neg && !mp_is_zero(c, r)) {
        Mp<1> u;
        mp_sub(c, u, v, r);
        r = u;
    }
    return r;
}

// exact signed division n / v (v > 0 divides n) -> quotient of QP planes (2-adic: mp_divexact)
template <int QP, int P>
CF_DEV SMp<QP> sdiv_exact(Ctx &c, const SMp<P> &n, const Mp<1> &v) {
    SMp<QP> r;
    const int nb = mp_bitlen(c, n.m), vb = mp_bitlen(c, v);
    mp_divexact(c, n.m, v, r.m, nb == 0 ? 0 : (nb - vb + 1 + 31) / 32);
    r.neg = n.neg;
    return r;
}

// M1 = (v2 R - m C)/v1, M2 = (s R + c2d C)/v1 for one remainder/cofactor pair
CF_DEV void nucomp_m12(Ctx &c, SMp<1> &M1, SMp<2> &M2, const Mp<1> &R, const SMp<1> &C, const Mp<1> &v1,
                       const Mp<1> &v2, const SMp<1> &m, const SMp<1> &s, const Mp<2> &c2d) {
    SMp<1> Rs{R, 0};
    SMp<1> v2s{v2, 0};
    SMp<2> t1 = smp_mul(c, v2s, Rs);
    SMp<2> t2 = smp_mul(c, m, C);
    SMp<2> n1;
    smp_sub(c, n1, t1, t2);
    M1 = sdiv_exact<1>(c, n1, v1);
    SMp<2> t3 = smp_mul(c, s, Rs);
    SMp<2> c2s{c2d, 0};
    SMp<3> t4w = smp_mul(c, c2s, C);
    SMp<2> t4{mp_resize<2>(t4w.m), t4w.neg};
    SMp<2> n2;
    smp_add(c, n2, t3, t4);
    M2 = sdiv_exact<2>(c, n2, v1);
}

// R*M1 + C*M2 (signed, double width)
CF_DEV SMp<2> nucomp_dot(Ctx &c, const Mp<1> &R, const SMp<1> &C, const SMp<1> &M1, const SMp<2> &M2) {
    SMp<1> Rs{R, 0};
    SMp<2> p1 = smp_mul(c, Rs, M1);
    SMp<3> p2w = smp_mul(c, C, M2);
    SMp<2> p2{mp_resize<2>(p2w.m), p2w.neg};
    SMp<2> r;
    smp_add(c, r, p1, p2);
    return r;
}

// ---------------------------------------------------------------------------- composition
// Discriminant as the kernels see it: 80 little-endian words of |Delta| and its half bit length.
struct QDisc {
    const uint32_t *absdelta;      // 2 planes, same limb order as Mp<2>
    int half_dbits;                // ceil(bits(|Delta|) / 2)
};

// The two big remainder sequences of a composition.  WG: 0 = inside the limb group (euclid_run), 1 = served by wavefront 0 of
// the workgroup (euclid_run_wg: every thread of the workgroup must then reach both calls).  (A dedicated fifth serving
// wavefront was WG = 2 in round 4: experiments/dedicated_server/.)
template <int WG>
CF_DEV void qf_euclid(Ctx &c, Euclid<1> &e, int stop_bits) {
#if defined(COFHE_HOSTSIM)
    if (WG != 0 && c.wg) {          // simulated workgroup (tests/hostsim: run_workgroup)
        euclid_run_wg(c, e, stop_bits);
        return;
    }
#else
    if (WG != 0) {
        euclid_run_wg(c, e, stop_bits);
        return;
    }
#endif
    euclid_run(c, e, stop_bits);
}

// g = gcd(m, a) and inv with inv * a == g (mod m), 0 <= inv < m, for words 0 < a < m: the scalar extended sequence (the same
// values in all lanes of the group)
CF_DEV void word_xgcd(uint32_t m, uint32_t a, uint32_t &g, uint32_t &inv) {
    uint32_t r0 = m, r1 = a;
    int64_t t0 = 0, t1 = 1;
    for (int it = 0; it < 64 && r1 != 0; it++) {          // <= 47 steps for 32-bit operands
        const uint32_t q = r0 / r1, r2 = r0 - q * r1;
        const int64_t t2 = t0 - (int64_t)q * t1;
        r0 = r1; r1 = r2;
        t0 = t1; t1 = t2;
    }
    g = r0;
    inv = (uint32_t)(t0 < 0 ? t0 + (int64_t)m : t0);
}
// word_xgcd for operands below 2^16: g = gcd(m, a), inv * a == g (mod m), 0 <= inv < m, for 0 < a < m < 2^16.  The
// quotients come from a float reciprocal (operands exact in f32; biased low, so never above the true quotient and at most
// one below it) instead of the ~40-instruction integer division.
CF_DEV void word_xgcd16(uint32_t m, uint32_t a, uint32_t &g, uint32_t &inv) {
    uint32_t r0 = m, r1 = a;
    int32_t t0 = 0, t1 = 1;                                      // |t| <= m < 2^16
    for (int it = 0; it < 32 && r1 != 0; it++) {              // <= 24 steps for 16-bit operands
        uint32_t q = f32_to_u32_sat((float)r0 * fast_rcp((float)r1) * 0.99999905f);
        uint32_t r2 = r0 - q * r1;
        if (r2 >= r1) { r2 -= r1; q++; }
        const int32_t t2 = t0 - (int32_t)q * t1;
        r0 = r1; r1 = r2;
        t0 = t1; t1 = t2;
    }
    g = r0;
    inv = (uint32_t)(t0 < 0 ? t0 + (int32_t)m : t0);
}
// The composition's route for a common factor d < 2^16 of the first coefficients (qf_compose below has the derivation).
// In: r = r0 = y1 m mod a1, v1 = a1, v2 = a2, c2d = c2.  A common prime p of a1 and a2 makes b1 and b2 square roots of Delta
// modulo p, so b2 == +-b1: either p | m or p | s, half of the cases each.
//   d and s coprime (and d, a1 / d coprime):  r = r0 / d + j (a1 / d), j from residues modulo d and d^2; v1, v2, c2d stay
//   d | s:  d1 = d, x2 = 0, y2 = -1 in the general formula:  v1 = a1 / d, v2 = a2 / d, c2d = c2 d, r = r0 mod v1 (quotient < d)
// false: neither (d >= 2^16, d and s share part of d, d^2 | a1 -- together ~1/d of these pairs); nothing is modified then.
CF_DEV_COLD bool qf_word_factor_residue(Ctx &c, Mp<1> &r, Mp<1> &v1, Mp<1> &v2, Mp<2> &c2d, const Mp<1> &d, const SMp<1> &s,
                                        const SMp<1> &m, const SMp<1> &y1) {
    const uint32_t dw = bcast_first(c, d.v[0][0]);
    if (CF_UNLIKELY(mp_bitlen(c, d) > 16)) { CF_FLAG(16u); return false; }
    const WordDiv dv = worddiv_make(dw);
    const ModW mw = modw_make(c, dw * dw);
    uint32_t sd, rem;
    const uint32_t sW = mp_mod_word_fast(c, s.m, mw);
    (void)worddiv_divmod(dv, sW, sd);                                 // |s| mod d
    if (sd == 0) {
        Mp<1> q1 = v1, q2 = v2;
        const uint32_t rem1 = mp_divrem_word(c, q1, dv), rem2 = mp_divrem_word(c, q2, dv);
        if (CF_UNLIKELY((rem1 | rem2) != 0)) { CF_FLAG(32u); return false; }     // d divides both by construction
        Mp<2> cd;
        (void)mp_lincomb_add(c, cd, dw, c2d, 0u, c2d);                // c2 d: within two planes like the general route's product
        Mp<1> rr = r, qq;
        mp_divrem(c, rr, q1, qq);                                      // r0 < a1 = d v1: one or two quotient digits
        r = rr; v1 = q1; v2 = q2; c2d = cd;
        return true;
    }
    uint32_t g = 0, xc = 0;
    word_xgcd16(dw, sd, g, xc);                                       // xc |s| == g (mod d), 0 < xc < d
    const uint32_t r0W = mp_mod_word_fast(c, r, mw), a1W = mp_mod_word_fast(c, v1, mw);
    uint32_t r0rem, a1rem;
    const uint32_t r0q = (uint32_t)worddiv_divmod(dv, r0W, r0rem);    // (r0 / d) mod d
    const uint32_t a1q = (uint32_t)worddiv_divmod(dv, a1W, a1rem);    // (a1 / d) mod d
    uint32_t ga = 0, ainv = 0;
    if (a1q != 0) word_xgcd16(dw, a1q, ga, ainv);
    if (CF_UNLIKELY(!(g == 1 && ga == 1 && r0rem == 0 && a1rem == 0))) {
        CF_FLAG((g != 1 ? 64u : 0u) | (ga != 1 ? 128u : 0u) | ((r0rem | a1rem) != 0 ? 256u : 0u));
        return false;
    }
    // y2 = (xc |s| - 1) / d >= 0 with x2 = sign(s) xc; modulo d from the residue of |s| modulo d^2
    (void)worddiv_divmod(mw.dv, (uint64_t)xc * sW, rem);
    const uint32_t y2d = (uint32_t)worddiv_divmod(dv, rem - 1u, rem);
    uint32_t y1d, md, c2w;
    (void)worddiv_divmod(dv, mp_mod_word_fast(c, y1.m, mw), y1d);
    if (y1.neg && y1d) y1d = dw - y1d;
    (void)worddiv_divmod(dv, mp_mod_word_fast(c, m.m, mw), md);
    if (!m.neg && md) md = dw - md;                               // residue of -m
    (void)worddiv_divmod(dv, mp_mod_word_fast(c, c2d, mw), c2w);
    const uint32_t x2d = s.neg ? dw - xc : xc;
    const uint32_t u = worddiv_mulmod(dv, worddiv_mulmod(dv, y1d, y2d), md);
    const uint32_t v = worddiv_mulmod(dv, x2d, c2w);
    const uint32_t rd = worddiv_addmod(dv, u, v ? dw - v : 0u);   // r mod d
    const uint32_t j = worddiv_mulmod(dv, worddiv_addmod(dv, rd, r0q ? dw - r0q : 0u), ainv);
    Mp<1> T;
    (void)mp_lincomb_add(c, T, 1u, r, j, v1);                     // r0 + j a1 < 2^16 a1: within the plane
    (void)mp_divrem_word(c, T, dv);                               // exact
    r = T;
    return true;
}
// out = reduced(f1 * f2).  WG: the remainder sequences are served by the workgroup's serving wavefront (every kernel).
// WORD_ROUTE: common word-sized factors of the first coefficients take qf_word_factor_residue instead of the general
// formula.  On in the tensor-addition kernels (k_compose_wg, k_add_ct: a 128x128 launch is ONE residency round, it ends with
// its slowest workgroup, and a workgroup with such a pair used to be the slowest) and in the matrix product (-1.8 %,
// interleaved runs); off in the other sequence kernels and the product-tree kernel, whose launches are many rounds deep:
// there a round that waits for the general formula costs ~2 % on average and the inlined route's registers cost as much
// or more (encrypt_tensor 128x128 8.0 -> 9.6 ms with the route in k_compose_pairs,
// profiles/r03_b/ops_word_route_everywhere.jsonl).  The host simulator runs both.
template <int WG = 0, bool WORD_ROUTE = true>
CF_DEV void qf_compose(Ctx &c, QForm &out, const QForm &fa, const QForm &fb, const QDisc &dd) {
    const int half_dbits = dd.half_dbits;
    CF_PHASE(0);
    // Coprime representative.  Random first coefficients share a small prime factor 38 % of the
    // time (and a squaring has a1 == a2), which would send the group -- and with it the whole
    // wavefront -- through the general-gcd route.  The class of f2 has other representatives:
    //   (a, b, c) ~ (c, -b, a) ~ (a+b+c, b+2c, c) ~ (a-b+c, b-2c, c)
    // so the first one whose leading coefficient shares no prime <= 23 with a1 (tested on
    // residues mod 2*3*...*23) is used instead; what is left (a common prime >= 29, chance
    // ~0.8 %) still takes the general route below.  Needs c2 within a plane (always, unless a2 is
    // unusually small).
    QForm fbr = fb;
    // equal first coefficients -- a squaring, or a form with its inverse (a tensor minus itself: every element) -- never keep
    // the second operand as it is: gcd(a1, a2) would be a1, the longest common factor there is
    const bool same = mp_cmp(c, fa.a, fb.a) == 0;
    if (CF_LIKELY(mp_bitlen(c, fb.c) <= PLIMBS * 32 - 110)) {
        const uint32_t M = PRIMORIAL23;                          // 2*3*5*7*11*13*17*19*23: tabulated limb weights (mp.hpp)
        const uint32_t ra1 = mp_mod_primorial(c, fa.a), ra2 = mp_mod_primorial(c, fb.a);
        uint32_t rb2 = mp_mod_primorial(c, fb.bm);
        if (fb.bneg && rb2) rb2 = M - rb2;
        const uint32_t rc2 = mp_mod_primorial(c, mp_resize<1>(fb.c));      // c2 within a plane (tested above)
        // candidates a x^2 + b x y + c y^2 for (x, y) = (1, 0), (0, 1), (1, 1), (1, -1), (1, 2), (1, -2): with the first four
        // 0.37 % of random pairs had NO admissible representative and went on with a common factor 2, 3, 5 ... -- often a
        // composite d, or one whose square divides a1, which the word route below has to decline (round 3: 16 of 1024
        // workgroups of a 128x128 launch, and they were the 16 slowest); with six, 0.013 %
        const uint32_t rb2n = rb2 ? M - rb2 : 0u;
        const uint32_t cand[6] = {ra2, rc2, (uint32_t)(((uint64_t)ra2 + rb2 + rc2) % M), (uint32_t)(((uint64_t)ra2 + rb2n + rc2) % M),
                                  (uint32_t)(((uint64_t)ra2 + 2ull * rb2 + 4ull * rc2) % M),
                                  (uint32_t)(((uint64_t)ra2 + 2ull * rb2n + 4ull * rc2) % M)};
        int pick = -1;
        for (int k = same ? 1 : 0; k < 6 && pick < 0; k++) {
            const uint32_t x = cand[k];
            bool ok = true;
            const uint32_t primes[9] = {2, 3, 5, 7, 11, 13, 17, 19, 23};
            CF_UNROLL for (int i = 0; i < 9; i++) ok = ok && !((x % primes[i]) == 0 && (ra1 % primes[i]) == 0);
            if (ok) pick = k;
        }
        if (pick == 1) {
            fbr.a = mp_resize<1>(fb.c);
            fbr.c = mp_resize<2>(fb.a);
            fbr.bneg = mp_is_zero(c, fb.bm) ? 0 : (fb.bneg ^ 1);
        } else if (pick >= 2) {
            // (x, y) = (1, +-k), k = 1 or 2, completed by (0, 1):  a' = a + k^2 c +- k b,  b' = b +- 2 k c,  c' = c
            // (a + k^2 c > k |b| and 2 k c > |b| for a reduced form)
            const uint32_t kk = pick >= 4 ? 2u : 1u;
            const Mp<1> cs = mp_resize<1>(fb.c);
            Mp<1> t, two_c, kb;
            (void)mp_lincomb_add(c, t, 1u, fb.a, kk * kk, cs);
            (void)mp_lincomb_add(c, two_c, 2u * kk, cs, 0u, cs);
            (void)mp_lincomb_add(c, kb, kk, fb.bm, 0u, fb.bm);
            const bool up = (pick & 1) == 0;                        // y = +k
            const bool plus = up != (fb.bneg != 0);                 // does k |b| add to a + k^2 c ?
            Mp<1> na, nb;
            if (plus) {
                (void)mp_add(c, na, t, kb);
                (void)mp_add(c, nb, two_c, fb.bm);
            } else {
                mp_sub(c, na, t, kb);
                mp_sub(c, nb, two_c, fb.bm);
            }
            fbr.a = na;
            fbr.bm = nb;
            fbr.bneg = up ? 0 : 1;                                  // b + 2 k c > 0, b - 2 k c < 0
        }
    }
    const QForm &fbx = fbr;
    const bool sw = mp_cmp(c, fa.a, fbx.a) < 0;
    QForm f1, f2;                         // a1 >= a2 (register selects, no addresses taken)
    mp_select(f1.a, sw, fa.a, fbx.a);   mp_select(f2.a, sw, fbx.a, fa.a);
    mp_select(f1.bm, sw, fa.bm, fbx.bm); mp_select(f2.bm, sw, fbx.bm, fa.bm);
    mp_select(f2.c, sw, fbx.c, fa.c);
    f1.bneg = sw ? fbx.bneg : fa.bneg;
    f2.bneg = sw ? fa.bneg : fbx.bneg;
    SMp<1> b1{f1.bm, f1.bneg}, b2{f2.bm, f2.bneg};
    SMp<1> s, m;
    smp_add(c, s, b1, b2);
    smp_sub(c, m, b1, b2);
    s.m = mp_shr1(c, s.m);
    m.m = mp_shr1(c, m.m);

    CF_PHASE(1);
    // d = gcd(a1, a2), y1*a2 == d (mod a1)
    Euclid<1> e;
    e.x = f1.a; e.y = f2.a;
    mp_zero(e.ux); mp_set_word(c, e.uy, 1);
    e.sx = -1; e.sy = 1;
    qf_euclid<WG>(c, e, -1);
    CF_PHASE(2);
#ifdef COFHE_WG_TIMING
    CF_PHASE_VAL(8, c.t_wait); CF_PHASE_VAL(9, c.t_apply); CF_PHASE_VAL(12, c.n_rounds); CF_PHASE_VAL(14, c.t_serve);
    c.t_wait = 0; c.t_apply = 0; c.n_rounds = 0; c.t_serve = 0;
#endif

    // r0 = y1 m mod a1: the residue the partial sequence starts from when the first coefficients are coprime (all but
    // ~0.8 % of the pairs), and the one long division of the word-factor route below
    Mp<1> v1 = f1.a, v2 = f2.a, r;
    Mp<2> c2d = f2.c;
    const SMp<1> y1{e.ux, e.sx < 0};
    {
        SMp<2> t = smp_mul(c, y1, m);
        r = smod(c, t, v1);
    }
    bool general = false;
    if (CF_UNLIKELY(!mp_is_word(c, e.x, 1))) {
        CF_FLAG(1u);
        // d = gcd(a1, a2) > 1.  What reaches this branch after the representative step is a common prime >= 29: 0.8 % of
        // random pairs, but a workgroup of 32 has one with probability 0.22, waits for it at the next barrier, and the
        // launch ends with its slowest workgroup (round 3, tools/wg_timing: all eight slowest workgroups of a 128x128
        // launch had one; with the general formula below -- three more long divisions -- it cost its workgroup 34 us
        // alone on a CU and 60-115 us among four).  For d below 2^16 that shares nothing with s or a1 / d:
        //   x2 s - y2 d = 1,  y1 a2 == d,  s m = a1 c1 - a2 c2   =>   d r == y1 m - x2 (y1 s m + d c2) == r0  (mod a1)
        // so r = r0 / d + j (a1 / d), and j in [0, d) follows from r mod d = -(y1 y2 m + x2 c2) mod d: word arithmetic on
        // residues modulo d and d^2 (mp_mod_word_fast), one word-multiple addition and one exact division by d.
        general = true;
        if constexpr (WORD_ROUTE) general = !qf_word_factor_residue(c, r, v1, v2, c2d, e.x, s, m, y1);
    }
    if (CF_UNLIKELY(general)) {
        CF_FLAG(2u);
        // general gcd structure (Cohen 5.4.7 steps 2-4): d of any size, d and s with a common factor, d^2 | a1.  For a
        // word-sized d everything about (d, s) is word arithmetic (no second remainder sequence on limb groups, no
        // multi-limb multiplication for y2 = (x2 s - d1) / d, none for c2 d1 when d1 == 1, a word multiple for x2 c2).
        const Mp<1> d = e.x;
        Mp<1> d1;
        SMp<1> x2, y2;
        uint32_t x2w = 0;                       // |x2| when it is known to fit a word (d word-sized), else 0
        const bool dword = mp_bitlen(c, d) <= 32;
        if (dword) {
            const uint32_t dw = bcast_first(c, d.v[0][0]);
            const WordDiv dv = worddiv_make(dw);
            const uint32_t sw = mp_mod_word(c, s.m, dv);          // |s| mod d
            if (sw == 0) {
                d1 = d;
                mp_zero(x2.m); x2.neg = 0;
                mp_set_word(c, y2.m, 1); y2.neg = 1;
            } else {
                uint32_t g, xc;
                word_xgcd(dw, sw, g, xc);                         // 0 < xc < d, xc |s| == g (mod d)
                mp_set_word(c, d1, g);
                mp_set_word(c, x2.m, xc);
                x2.neg = s.neg;                                   // x2 s == g (mod d) with s = +-|s|
                x2w = xc;
                // y2 = (xc |s| - g) / d >= 0, exact
                Mp<1> one, t;
                mp_set_word(c, one, 1);
                mp_lincomb_sub(c, t, xc, s.m, g, one);
                (void)mp_divrem_word(c, t, dv);
                y2.m = t;
                y2.neg = 0;
            }
        } else {
            Mp<1> sm = s.m, q;
            mp_divrem(c, sm, d, q);
            if (mp_is_zero(c, sm)) {
                d1 = d;
                mp_zero(x2.m); x2.neg = 0;
                mp_set_word(c, y2.m, 1); y2.neg = 1;
            } else {
                Euclid<1> e2;
                e2.x = d; e2.y = sm;
                mp_zero(e2.ux); mp_set_word(c, e2.uy, 1);
                e2.sx = -1; e2.sy = 1;
                euclid_run(c, e2, -1);
                d1 = e2.x;                          // d1 == (sx*ux) * sm (mod d)
                SMp<1> x2p{e2.ux, e2.sx < 0};
                x2.m = x2p.m; x2.neg = x2p.neg ^ s.neg;
                // y2 = (x2*s - d1)/d, exact
                SMp<1> sabs{s.m, 0};
                SMp<2> t = smp_mul(c, x2p, sabs);
                SMp<2> d1w{mp_resize<2>(d1), 0};
                SMp<2> t2;
                smp_sub(c, t2, t, d1w);
                y2 = sdiv_exact<1>(c, t2, d);
            }
        }
        if (mp_is_word(c, d1, 1)) {                 // the usual case: d and s share nothing
            v1 = f1.a; v2 = f2.a; c2d = f2.c;
        } else {
            Mp<1> qq;
            v1 = f1.a; mp_divrem(c, v1, d1, qq); v1 = qq;
            v2 = f2.a; mp_divrem(c, v2, d1, qq); v2 = qq;
            Mp<3> cw = mp_mul(c, f2.c, d1);
            c2d = mp_resize<2>(cw);
        }
        // r = (y1*y2*(-m) - x2*c2) mod v1, and y1*m == r0 (mod a1, hence mod v1 | a1):  r = (-y2 r0 - x2 c2) mod v1 -- one
        // product and ONE long division (x2 a word: x2 c2 stays within the two planes of c2; until round 3 this was two
        // products and five divisions, and its workgroup the slowest of the launch)
        const SMp<1> r0s{r, 1};                       // -r0
        SMp<2> num = smp_mul(c, y2, r0s);
        if (x2w != 0) {
            SMp<2> u, df;
            (void)mp_lincomb_add(c, u.m, x2w, f2.c, 0u, f2.c);
            u.neg = x2.neg;
            smp_sub(c, df, num, u);
            num = df;
        } else if (!mp_is_zero(c, x2.m)) {            // d beyond a word: x2 is multi-limb, reduce c2 first
            SMp<2> c2s{f2.c, 0};
            SMp<1> c2r{smod(c, c2s, v1), 0};
            SMp<2> w3 = smp_mul(c, x2, c2r), df;
            smp_sub(c, df, num, w3);
            num = df;
        }
        r = smod(c, num, v1);
    }

    CF_PHASE(3);
    // partial Euclid on (v1, r)
    const int lv1 = mp_bitlen(c, v1), lv2 = mp_bitlen(c, v2);
    int stop = (lv1 - lv2 + half_dbits) / 2;
    Euclid<1> pe;
    pe.x = v1; pe.y = r;
    mp_zero(pe.ux); mp_set_word(c, pe.uy, 1);
    pe.sx = -1; pe.sy = 1;
    qf_euclid<WG>(c, pe, stop);
    CF_PHASE(4);
#ifdef COFHE_WG_TIMING
    CF_PHASE_VAL(10, c.t_wait); CF_PHASE_VAL(11, c.t_apply); CF_PHASE_VAL(13, c.n_rounds); CF_PHASE_VAL(15, c.t_serve);
    c.t_wait = 0; c.t_apply = 0; c.n_rounds = 0; c.t_serve = 0;
#endif
    const SMp<1> C0{pe.ux, pe.sx < 0}, C1{pe.uy, pe.sy < 0};
    const int sg_neg = C1.neg;             // det(R0 C1 - R1 C0) has the sign of C1

    SMp<1> M1;
    SMp<2> M2;
    nucomp_m12(c, M1, M2, pe.y, C1, v1, v2, m, s, c2d);
    SMp<2> an = nucomp_dot(c, pe.y, C1, M1, M2);
    SMp<2> bs = nucomp_dot(c, pe.x, C0, M1, M2);
    // b' = -sg * 2 * bs - b1
    SMp<2> bn;
    {
        SMp<2> two_bs;
        (void)mp_add(c, two_bs.m, bs.m, bs.m);
        two_bs.neg = bs.neg ^ (sg_neg ? 0 : 1);
        SMp<2> b1w{mp_resize<2>(b1.m), b1.neg};
        smp_sub(c, bn, two_bs, b1w);
    }
    SMp<2> cn;
    CF_PHASE(5);
    if (CF_LIKELY(mp_high_planes_zero(c, an.m, 1) && mp_high_planes_zero(c, bn.m, 1))) {
        // usual case (a', b' near sqrt|Delta|): c' = (b'^2 + |Delta|) / (4 a')
        const Mp<1> bw = mp_resize<1>(bn.m);
        Mp<2> num = mp_mul(c, bw, bw), dl;
        CF_UNROLL for (int p = 0; p < 2; p++)
            CF_UNROLL for (int j = 0; j < CH; j++) dl.v[p][j] = dd.absdelta[p * PLIMBS + c.gl * CH + j];
        (void)mp_add(c, num, num, dl);
        num = mp_shr_small(c, num, 2);
        const Mp<1> aw = mp_resize<1>(an.m);
        mp_divexact(c, num, aw, cn.m, (mp_bitlen(c, num) - mp_bitlen(c, aw) + 1 + 31) / 32);      // exact: b'^2 - Delta == 4 a' c'
        cn.neg = 0;
    } else {
        // a' or b' wider than a plane (tiny v1*v2, e.g. inverse pairs): c' from the other pair
        SMp<1> M1p;
        SMp<2> M2p;
        nucomp_m12(c, M1p, M2p, pe.x, C0, v1, v2, m, s, c2d);
        cn = nucomp_dot(c, pe.x, C0, M1p, M2p);
    }

    CF_PHASE(6);
    if (CF_LIKELY(mp_bitlen(c, an.m) < PLIMBS * 32 - 8 && mp_bitlen(c, bn.m) < PLIMBS * 32 - 8 &&
                  mp_bitlen(c, cn.m) < PLIMBS * 32 - 8)) {
        // the usual case: everything near sqrt|Delta| -- reduce at single width
        Mp<1> a1 = mp_resize<1>(an.m), c1 = mp_resize<1>(cn.m);
        SMp<1> b1r{mp_resize<1>(bn.m), bn.neg};
        qf_reduce<1>(c, a1, b1r, c1);
        out.a = a1;
        out.bm = b1r.m;
        out.bneg = b1r.neg;
        out.c = mp_resize<2>(c1);
    } else {
        qf_reduce<2>(c, an.m, bn, cn.m);
        out.a = mp_resize<1>(an.m);
        out.bm = mp_resize<1>(bn.m);
        out.bneg = bn.neg;
        out.c = cn.m;
    }
    CF_PHASE(7);
}

// form inverse: (a, -b, c), re-normalised for the two boundary cases of the reduced domain
CF_DEV void qf_inverse(Ctx &c, QForm &f) {
    if (mp_is_zero(c, f.bm)) return;
    if (mp_cmp(c, f.bm, f.a) == 0) return;                       // b == a stays a
    if (mp_cmp(c, mp_resize<2>(f.a), f.c) == 0) return;          // a == c keeps b >= 0
    f.bneg ^= 1;
}

// ---------------------------------------------------------------------------- powering
// Exponent record: EXP_MAG_WORDS little-endian magnitude words followed by one sign word.
constexpr int EXP_MAG_WORDS = 31;     // 992-bit magnitudes: covers encryption randomness (~970 bits)
constexpr int EXP_REC_WORDS = 32;

CF_DEV int exp_bitlen(const uint32_t *e) {
    int n = 0;
    for (int i = 0; i < EXP_MAG_WORDS; i++)
        if (e[i]) n = i * 32 + 32 - clz32(e[i]);
    return n;
}
CF_DEV int exp_bit(const uint32_t *e, int t) { return (int)((e[t >> 5] >> (t & 31)) & 1u); }
// w-bit digit (w <= 8) at bit position pos of the magnitude
CF_DEV uint32_t exp_digit(const uint32_t *e, int pos, int w) {
    int i = pos >> 5, o = pos & 31;
    uint64_t v = e[i];
    if (i + 1 < EXP_MAG_WORDS) v |= (uint64_t)e[i + 1] << 32;
    return (uint32_t)(v >> o) & ((1u << w) - 1u);
}

// Non-adjacent form of the magnitude, read on the fly: digit_i = bit_(i+1)(3x) - bit_(i+1)(x).
// The words of 3x are x[w]*3 + carry_in(w); the 2-bit carries into all 32 word positions are
// packed into one 64-bit value by exp_naf_prepare (one pass over the exponent).  Inversion is
// free in a class group, so the ladder multiplies on a third of the digits instead of half of
// the bits; 2^k - 1 (the reference's plaintext -1, tensor_ops.inl:137) costs one multiplication.
CF_DEV uint64_t exp_naf_prepare(const uint32_t *e) {
    uint64_t pack = 0;
    uint32_t carry = 0;
    for (int w = 0; w < EXP_MAG_WORDS; w++) {
        const uint64_t t = (uint64_t)e[w] * 3u + carry;
        carry = (uint32_t)(t >> 32);
        pack |= (uint64_t)carry << (2 * (w + 1));
    }
    return pack;
}
CF_DEV int exp_naf_digit(const uint32_t *e, uint64_t pack, int i) {
    const int j = i + 1, w = j >> 5, o = j & 31;
    const uint32_t xw = w < EXP_MAG_WORDS ? e[w] : 0u;
    const uint32_t x3w = xw * 3u + (uint32_t)((pack >> (2 * w)) & 3u);
    return (int)((x3w >> o) & 1u) - (int)((xw >> o) & 1u);
}
// index of the leading (+1) digit of the non-adjacent form of a non-zero exponent of nb bits
CF_DEV int exp_naf_top(const uint32_t *e, uint64_t pack, int nb) { return exp_naf_digit(e, pack, nb) != 0 ? nb : nb - 1; }

// out = reduced(base^e), left-to-right signed-digit ladder (e == 0 gives the principal form
// `one`; negative exponents invert).  What ClassGroup::nupow returns
// (cpu_cryptosystem_tensor_ops.inl:334-335).  Squarings and multiplications share ONE
// qf_compose call site (the ladder is a two-phase state machine) to keep the code object small.
CF_DEV void qf_pow(Ctx &c, QForm &out, const QForm &base, const uint32_t *e, const QForm &one, const QDisc &dd) {
    const int nb = exp_bitlen(e);
    if (nb == 0) {
        out = one;
        return;
    }
    const uint64_t pack = exp_naf_prepare(e);
    QForm acc = base;
    bool inv_bneg;                      // sign of b in base^-1
    {
        QForm t = base;
        qf_inverse(c, t);
        inv_bneg = t.bneg;
    }
    int t = exp_naf_top(e, pack, nb) - 1;
    bool mul_phase = false;
    while (t >= 0) {
        const int dgt = exp_naf_digit(e, pack, t);
        QForm rhs, r;
        mp_select(rhs.a, mul_phase, acc.a, base.a);
        mp_select(rhs.bm, mul_phase, acc.bm, base.bm);
        mp_select(rhs.c, mul_phase, acc.c, base.c);
        rhs.bneg = mul_phase ? (dgt < 0 ? inv_bneg : base.bneg) : acc.bneg;
        qf_compose(c, r, acc, rhs, dd);
        acc = r;
        if (!mul_phase && dgt != 0) {
            mul_phase = true;
        } else {
            mul_phase = false;
            t--;
        }
    }
    if (e[EXP_MAG_WORDS]) qf_inverse(c, acc);
    out = acc;
}

}  // namespace cofhe
