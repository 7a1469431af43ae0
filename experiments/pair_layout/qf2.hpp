// qf2.hpp -- composition of binary quadratic forms on a lane PAIR (pair.hpp): the FAST PATH of
// add_ciphertext_tensors (reference: include/x86_64/cpu_cryptosystem_tensor_ops.inl:242-264, BICYCL nucomp).
//
// Same mathematics as qf.hpp (Shanks/Atkin NUCOMP with one formula set for every gcd structure, reduced
// representative at the end -- hence the same bytes), organised for latency instead of occupancy: the Lehmer
// batch runs in the lane, the window of the remainder sequence sits at a WAVE-uniform limb index (a scalar jump
// selects the registers; pairs whose numbers are shorter simply see a less precise window for a round), and
// nothing waits on another wavefront.
//
// Scope of the fast path: operands of generic size (every coefficient within a few limbs of sqrt|Delta|) and a
// word-sized gcd(a1, a2).  Anything else -- tiny or huge coefficients, a quotient beyond a batch, an
// intermediate that leaves its limbs -- clears `ok` for that pair; the launcher then hands the element to the
// general kernel (k_compose_wg, qf.hpp).  No loop here depends on the data for its exit (bounded trip counts).
#pragma once
#include "../../cofhe_amd/csrc/mp.hpp"        // scalar helpers shared with the 8-lane layout
#include "../lehmer_variants/lehmer_variants.hpp"   // the 64-bit integer batch this experiment was built on (the product's is f64 now)
#include "pair.hpp"
#include "../../cofhe_amd/csrc/layout.hpp"

namespace cofhe2 {

using cofhe::lehmer_batch_u64_unordered;

template <int N>
struct QForm2 {          // a, |b|, c of one form; c within 2N limbs on the fast path
    BN<N> a, bm, c;
    int bneg;
};

// ---------------------------------------------------------------------------------------- signed helpers
template <int N>
P2_DEV void sbn_add(PCtx &c, SBN<N> &r, const SBN<N> &x, const SBN<N> &y) {
    // magnitudes stay below 2^(64N - 1) on the fast path (callers check bit lengths)
    BN<N> s, d1, d2;
    (void)bn_add(c, s, x.m, y.m);
    const uint32_t bw = bn_sub(c, d1, x.m, y.m);          // bw == 1: |x| < |y|
    (void)bn_sub(c, d2, y.m, x.m);
    const bool same = x.neg == y.neg;
    P2_UNROLL for (int j = 0; j < N; j++) r.m.v[j] = same ? s.v[j] : (bw ? d2.v[j] : d1.v[j]);
    r.neg = same ? x.neg : (bw ? y.neg : x.neg);
}
template <int N>
P2_DEV void sbn_sub(PCtx &c, SBN<N> &r, const SBN<N> &x, const SBN<N> &y) {
    SBN<N> ny = y;
    ny.neg ^= 1;
    sbn_add(c, r, x, ny);
}

// ---------------------------------------------------------------------------------------- remainder sequences
template <int N>
struct Euclid2 {         // x, y >= 0;  x == sx * u * w, y == sy * v * w (mod modulus), sx, sy opposite
    BN<N> x, y, u, v;
    int sx, sy;
};

// limbs T, T-1, T-2, T-3 of x and y for a wave-uniform T in [3, 2N)
#define P2_CASE(K)                                                                                       \
    case K:                                                                                              \
        if (K < 2 * N) {                                                                                 \
            constexpr int k_ = K < 2 * N ? K : 3;                                                        \
            xw[3] = bn_limb<k_>(c, x); xw[2] = bn_limb<k_ - 1>(c, x); xw[1] = bn_limb<k_ - 2>(c, x); xw[0] = bn_limb<k_ - 3>(c, x); \
            yw[3] = bn_limb<k_>(c, y); yw[2] = bn_limb<k_ - 1>(c, y); yw[1] = bn_limb<k_ - 2>(c, y); yw[0] = bn_limb<k_ - 3>(c, y); \
        }                                                                                                \
        break;
template <int N>
P2_DEV void fetch_window(PCtx &c, const BN<N> &x, const BN<N> &y, int T, uint32_t (&xw)[4], uint32_t (&yw)[4]) {
    xw[0] = xw[1] = xw[2] = xw[3] = 0u;
    yw[0] = yw[1] = yw[2] = yw[3] = 0u;
    switch (T) {
        P2_CASE(3) P2_CASE(4) P2_CASE(5) P2_CASE(6) P2_CASE(7) P2_CASE(8) P2_CASE(9)
        P2_CASE(10) P2_CASE(11) P2_CASE(12) P2_CASE(13) P2_CASE(14) P2_CASE(15) P2_CASE(16) P2_CASE(17) P2_CASE(18) P2_CASE(19)
        P2_CASE(20) P2_CASE(21) P2_CASE(22) P2_CASE(23) P2_CASE(24) P2_CASE(25) P2_CASE(26) P2_CASE(27) P2_CASE(28) P2_CASE(29)
        P2_CASE(30) P2_CASE(31) P2_CASE(32) P2_CASE(33) P2_CASE(34) P2_CASE(35) P2_CASE(36) P2_CASE(37) P2_CASE(38) P2_CASE(39)
        default: break;
    }
}
#undef P2_CASE


// 64-bit windows of x and y aligned at the top of max(x, y), from the four limbs at the wave-uniform index T
// (fetch_window).  lag: 0 / 1 = the pair's top limb is T / T - 1; 2 = further behind (no window: `valid` false unless
// T == 3, where the numbers fit 64 bits and the windows are exact).  pos = bit position of the windows' low end.
struct Windows {
    uint64_t xh, yh;
    int pos;
    bool valid, exact, top_zero;
};
P2_DEV Windows make_windows(const uint32_t (&xw)[4], const uint32_t (&yw)[4], int T) {
    Windows w;
    const uint32_t m3 = xw[3] | yw[3], m2 = xw[2] | yw[2];
    const bool lag0 = m3 != 0, lag1 = !lag0 && m2 != 0;
    w.top_zero = !lag0;
    w.exact = !lag0 && !lag1 && T == 3;
    w.valid = lag0 || lag1 || w.exact;
    const uint32_t x3 = lag0 ? xw[3] : xw[2], x2 = lag0 ? xw[2] : xw[1], x1 = lag0 ? xw[1] : xw[0];
    const uint32_t y3 = lag0 ? yw[3] : yw[2], y2 = lag0 ? yw[2] : yw[1], y1 = lag0 ? yw[1] : yw[0];
    const uint32_t topw = x3 > y3 ? x3 : y3;
    const int sh = w.exact ? 0 : (topw ? __builtin_clz(topw) : 0);
    const uint64_t xl64 = ((uint64_t)x3 << 32) | x2, yl64 = ((uint64_t)y3 << 32) | y2;
    w.xh = sh ? ((xl64 << sh) | (uint64_t)(x1 >> (32 - sh))) : xl64;
    w.yh = sh ? ((yl64 << sh) | (uint64_t)(y1 >> (32 - sh))) : yl64;
    if (w.exact) {
        w.xh = ((uint64_t)xw[1] << 32) | xw[0];
        w.yh = ((uint64_t)yw[1] << 32) | yw[0];
    }
    const int tl = lag0 ? T : T - 1;
    w.pos = w.exact ? 0 : 32 * (tl - 1) - sh;
    return w;
}

// Runs the remainder sequence of (x, y) until the smaller one has at most stop_bits bits (stop_bits < 0: until
// it is 0).  On return x >= y.  `active`: this pair takes part; `ok` is cleared when the pair leaves the fast
// path (quotient beyond a batch: sizes far apart).  Every lane of the wavefront must call this.
template <int N>
P2_DEV void euclid_pair(PCtx &c, Euclid2<N> &s, int stop_bits, bool active, bool &ok) {
    int T = 2 * N - 1;                       // wave-uniform top limb of the window
    bool done = !active || !ok;
    for (int round = 0; round < 64 * N && wave_any(c, !done); round++) {
#if !defined(COFHE_HOSTSIM)
        T = __builtin_amdgcn_readfirstlane(T);
#endif
        uint32_t xw[4], yw[4];
        fetch_window(c, s.x, s.y, T, xw, yw);
        const Windows w = make_windows(xw, yw, T);
        if (T > 3 && !wave_any(c, !done && !w.top_zero)) {       // every running pair has shrunk below limb T
            T--;
            continue;
        }
        // a pair more than a limb behind the wave's window waits for the others (skip); at T == 3 its numbers fit
        // 64 bits and the windows are exact
        const bool small = w.exact, skip = !w.valid;
        const uint64_t xh = w.xh, yh = w.yh;
        uint64_t thr = 0;
        if (stop_bits >= 0) {
            const int tb = stop_bits - w.pos;
            thr = tb <= 0 ? 0 : (tb >= 64 ? ~0ull : (1ull << tb));
        }
        const uint64_t lo = xh < yh ? xh : yh;
        bool stop_now = false;
        if (!done && !skip) {
            if (lo < thr || (small && lo == 0)) stop_now = true;       // the smaller one is within the bound / zero
            else if (!small && lo < (1ull << 33)) P2_LEAVE(ok, "qf2.hpp:141");          // sizes >= 31 bits apart: long step, not here
        }
        uint32_t A = 1, B = 0, C = 0, D = 1;
        const bool run = !done && !skip && !stop_now && ok;
        bool got = lehmer_batch_u64_unordered(run ? xh : 0, run ? yh : 0, small, thr, A, B, C, D);
        if (run && !got) {
            if (small) {
                // exact 64-bit tail with a quotient beyond a batch (or equal values): one division step
                const bool xbig = xh >= yh;
                const uint64_t qq = xbig ? xh / yh : yh / xh;             // lo != 0 here
                const uint32_t q = (uint32_t)qq;
                A = 1; D = 1;
                B = xbig ? q : 0u;
                C = xbig ? 0u : q;
                got = true;
                if (qq > 0xFFFFFFFFull) { P2_LEAVE(ok, "qf2.hpp:156"); got = false; }      // beyond a word: not on the fast path
            } else {
                P2_LEAVE(ok, "qf2.hpp:158");
            }
        }
        if (!(run && got)) { A = 1; B = 0; C = 0; D = 1; }
        done = done || stop_now || !ok;
        // x' = A x - B y, y' = D y - C x, cofactor magnitudes by additions (signs opposite: see mp.hpp)
        BN<N> nx, ny;
        bn_lincomb_sub(c, nx, A, s.x, B, s.y);
        bn_lincomb_sub(c, ny, D, s.y, C, s.x);
        s.x = nx; s.y = ny;
        (void)bn_lincomb_add(c, nx, A, s.u, B, s.v);
        (void)bn_lincomb_add(c, ny, D, s.v, C, s.u);
        s.u = nx; s.v = ny;
    }
    if (wave_any(c, !done)) ok = ok && done;                 // round cap hit (cannot happen for bounded inputs)
    // leave with x >= y
    const bool sw = bn_cmp(c, s.x, s.y) < 0;
    BN<N> t;
    bn_select(t, sw, s.x, s.y); bn_select(s.y, sw, s.y, s.x); s.x = t;
    bn_select(t, sw, s.u, s.v); bn_select(s.v, sw, s.v, s.u); s.u = t;
    const int ts = sw ? s.sy : s.sx;
    s.sy = sw ? s.sx : s.sy;
    s.sx = ts;
}

// ---------------------------------------------------------------------------------------- reduction
// (a, b, c) positive definite with every coefficient well inside 2N limbs; on return reduced
// (-a < b <= a <= c, b >= 0 when a == c).  Quotients beyond 2^30 leave the fast path.
template <int N>
P2_DEV void qf2_reduce(PCtx &c, BN<N> &a, SBN<N> &b, BN<N> &cc, bool active, bool &ok) {
    for (int it = 0; it < 64 * N; it++) {
        const int cm = bn_cmp(c, b.m, a);
        const bool need_norm = active && ok && (cm > 0 || (cm == 0 && b.neg));
        const int ac = bn_cmp(c, a, cc);
        const bool need_swap = active && ok && !need_norm && ac > 0;
        if (!wave_any(c, need_norm || need_swap)) break;
        if (wave_any(c, need_norm)) {
            // b = 2 a q sgn(b) + b' with q = floor((|b| + a) / 2a) >= 1;  c' = c + q (a q - |b|).
            // q comes from the leading 64 bits (never above the true quotient; when it is below, the loop runs again)
            BN<N> two_a, t;
            (void)bn_add(c, two_a, a, a);
            (void)bn_add(c, t, b.m, a);                      // |b| + a >= 2a
            const int nb = bn_bitlen(c, t);
            const int T0 = wave_max_small(c, nb > 0 ? (nb - 1) >> 5 : 0, need_norm);
            const int T = T0 < 3 ? 3 : T0;
            uint32_t tw[4], dw[4];
            fetch_window(c, t, two_a, T, tw, dw);
            const Windows w = make_windows(tw, dw, T);
            uint32_t q = 1;
            if (need_norm) {
                if (!w.valid) {
                    P2_LEAVE(ok, "qf2.hpp:209");
                } else {
                    // floor(xh / (yh + 1)) scaled down by 2^-40: never above the true quotient
                    const double qf = (double)w.xh / ((double)w.yh + 1.0) * (1.0 - 1.0 / 1099511627776.0);
                    if (!(qf < 1073741824.0)) P2_LEAVE(ok, "qf2.hpp:213");    // quotient beyond 2^30 (or no divisor): general kernel
                    q = (qf < 1073741824.0 && qf >= 1.0) ? (uint32_t)qf : 1u;      // |b| >= a: one step of 2a is always valid
                }
            }
            BN<N> nbm;
            const uint32_t cw = bn_lincomb_sub_carry(c, nbm, 1u, b.m, q, two_a);     // |b| - q 2a == nbm + (cw - q) 2^(64N)
            const bool negres = cw != q;                                              // went negative: b' changes sign
            BN<N> z, neg;
            bn_zero(z);
            (void)bn_sub(c, neg, z, nbm);
            bn_select(nbm, negres, nbm, neg);
            // c' = c + q (a q - |b|)
            BN<N> aq, dq;
            (void)bn_mul_word(c, aq, a, q);
            SBN<N> u{aq, 0}, bs{b.m, 0}, d;
            sbn_sub(c, d, u, bs);                            // a q - |b|
            (void)bn_mul_word(c, dq, d.m, q);
            SBN<N> cs{cc, 0}, dqs{dq, d.neg}, cn;
            sbn_add(c, cn, cs, dqs);
            if (need_norm && ok) {
                cc = cn.m;
                b.m = nbm;
                b.neg = negres ? (b.neg ^ 1) : b.neg;
            }
            if (bn_is_zero(c, b.m)) b.neg = 0;
        }
        if (need_swap) {
            BN<N> t = a;
            a = cc;
            cc = t;
            b.neg ^= 1;
        }
    }
    if (bn_is_zero(c, b.m)) b.neg = 0;
    if (b.neg && bn_cmp(c, a, cc) == 0) b.neg = 0;
    if (b.neg && bn_cmp(c, b.m, a) == 0) b.neg = 0;           // b == -a is written b == a
}

// ---------------------------------------------------------------------------------------- composition
struct QDisc2 {
    const uint32_t *absdelta;      // 80 little-endian words of |Delta|
    int half_dbits;                // ceil(bits(|Delta|) / 2)
};

// residues of the coprime-representative screen: modulus 2*3*...*23 and its powers of 2^32
constexpr uint32_t SCREEN_M = 223092870u;
struct ScreenTable {
    uint32_t pw[24];
};
constexpr ScreenTable make_screen_table() {
    ScreenTable t{};
    uint64_t v = 1;
    for (int j = 0; j < 24; j++) {
        t.pw[j] = (uint32_t)v;
        v = (v << 32) % SCREEN_M;
    }
    return t;
}

// out = reduced(fa * fb); `ok` is cleared when the pair left the fast path (out is then meaningless)
template <int N>
P2_DEV void qf2_compose(PCtx &c, QForm2<N> &out, const QForm2<N> &fa, const QForm2<N> &fb, const QDisc2 &dd, bool active, bool &ok,
                        bool screen = true /* tests switch the coprime-representative screen off to reach the general gcd structure */) {
    constexpr ScreenTable ST = make_screen_table();
    static_assert(N < 24, "screen table");
    const int half_dbits = dd.half_dbits;
    P2_PHASE(0);
    const int LIM = 64 * N - 34;               // every coefficient and intermediate must stay below 2^LIM
    // ---- sizes: generic operands only
    {
        const int la = bn_bitlen(c, fa.a), lb = bn_bitlen(c, fb.a), lc = bn_bitlen(c, fb.c), lca = bn_bitlen(c, fa.c);
        if (la < 64 * N - 160 || lb < 64 * N - 160 || la > 64 * N - 3 || lb > 64 * N - 3 || lc > 64 * N - 3 || lca > 64 * N - 3) P2_LEAVE(ok, "qf2.hpp:282");
    }
    // ---- coprime representative of fb (qf.hpp): (a, b, c) ~ (c, -b, a) ~ (a+b+c, b+2c, c) ~ (a-b+c, b-2c, c)
    QForm2<N> fbr = fb;
    {
        const bool same = bn_cmp(c, fa.a, fb.a) == 0 && fa.bneg == fb.bneg && bn_cmp(c, fa.bm, fb.bm) == 0;
        const uint32_t M = SCREEN_M;
        const uint32_t ra1 = bn_mod_small(c, fa.a, M, ST.pw), ra2 = bn_mod_small(c, fb.a, M, ST.pw);
        uint32_t rb2 = bn_mod_small(c, fb.bm, M, ST.pw);
        if (fb.bneg && rb2) rb2 = M - rb2;
        const uint32_t rc2 = bn_mod_small(c, fb.c, M, ST.pw);
        const uint32_t cand[4] = {ra2, rc2, (uint32_t)(((uint64_t)ra2 + rb2 + rc2) % M), (uint32_t)(((uint64_t)ra2 + (M - rb2) + rc2) % M)};
        const uint32_t primes[9] = {2, 3, 5, 7, 11, 13, 17, 19, 23};
        uint32_t m1 = 0;                               // bit i: primes[i] divides a1
        P2_UNROLL for (int i = 0; i < 9; i++) m1 |= ((ra1 % primes[i]) == 0 ? 1u : 0u) << i;
        int pick = -1;
        P2_UNROLL for (int k = 3; k >= 0; k--) {
            uint32_t mk = 0;
            P2_UNROLL for (int i = 0; i < 9; i++) mk |= ((cand[k] % primes[i]) == 0 ? 1u : 0u) << i;
            if ((mk & m1) == 0 && !(same && k == 0)) pick = k;
            if (!screen && k == 0 && !same) pick = 0;
        }
        if (pick == 1) {
            fbr.a = fb.c;
            fbr.c = fb.a;
            fbr.bneg = bn_is_zero(c, fb.bm) ? 0 : (fb.bneg ^ 1);
        } else if (pick >= 2) {
            BN<N> t, two_c, na, nb2, na_m, nb_m;
            (void)bn_add(c, t, fb.a, fb.c);
            (void)bn_add(c, two_c, fb.c, fb.c);
            const bool plus = (pick == 2) != (fb.bneg != 0);     // does |b| add to a + c ?
            (void)bn_add(c, na, t, fb.bm);
            (void)bn_add(c, nb2, two_c, fb.bm);
            (void)bn_sub(c, na_m, t, fb.bm);
            (void)bn_sub(c, nb_m, two_c, fb.bm);
            bn_select(fbr.a, !plus, na, na_m);
            bn_select(fbr.bm, !plus, nb2, nb_m);
            fbr.bneg = (pick == 2) ? 0 : 1;
        } else if (pick < 0 && same) {
            P2_LEAVE(ok, "qf2.hpp:320");                                // squaring with no coprime representative among the four
        }
    }
    // ---- order a1 >= a2
    const bool sw = bn_cmp(c, fa.a, fbr.a) < 0;
    BN<N> a1, a2, b1m, b2m, c2;
    bn_select(a1, sw, fa.a, fbr.a);   bn_select(a2, sw, fbr.a, fa.a);
    bn_select(b1m, sw, fa.bm, fbr.bm); bn_select(b2m, sw, fbr.bm, fa.bm);
    bn_select(c2, sw, fbr.c, fa.c);
    const int b1neg = sw ? fbr.bneg : fa.bneg, b2neg = sw ? fa.bneg : fbr.bneg;
    SBN<N> b1{b1m, b1neg}, b2{b2m, b2neg}, s, m;
    sbn_add(c, s, b1, b2);
    sbn_sub(c, m, b1, b2);
    s.m = bn_shr_small(c, s.m, 1);
    m.m = bn_shr_small(c, m.m, 1);

    P2_PHASE(1);
    // ---- d = gcd(a1, a2), y1 * a2 == d (mod a1)
    Euclid2<N> e;
    e.x = a1; e.y = a2;
    bn_zero(e.u); bn_set_word(c, e.v, 1);
    e.sx = -1; e.sy = 1;
    euclid_pair(c, e, -1, active, ok);
    P2_PHASE(2);
    const SBN<N> y1{e.u, e.sx < 0};
    const int dbits_g = bn_bitlen(c, e.x);
    if (dbits_g > 32 || dbits_g == 0) P2_LEAVE(ok, "qf2.hpp:344");          // gcd beyond a word: general kernel
    const uint32_t d = bn_limb<0>(c, e.x);

    BN<N> v1 = a1, v2 = a2, c2d = c2, r;
    SBN<N> rr_general;
    bool general = ok && d != 1;
    {
        // r = y1 * m mod a1 (d == 1); the general structure below patches v1, v2, c2d and r for the pairs with d != 1
        BN<N> L, H;
        bn_mul(c, L, H, y1.m, m.m);
        BN<N> rem;
        bn_rem(c, rem, L, H, a1, active && !general, ok);
        const bool tneg = (y1.neg ^ m.neg) != 0;
        BN<N> comp;
        (void)bn_sub(c, comp, a1, rem);
        const bool z = bn_is_zero(c, rem);
        bn_select(r, tneg && !z, rem, comp);
    }
    P2_PHASE(3);
    if (wave_any(c, general)) {
        // general gcd structure (Cohen 5.4.7 steps 2-4) for a word-sized d; pairs with d == 1 idle through it
        const uint32_t dw = general ? d : 3u;
        const uint32_t sd = bn_mod_word(c, s.m, dw);          // |s| mod d
        // d1 = gcd(sd, d) = x2p * sd (mod d) by a word Euclid with one cofactor
        uint32_t gx = dw, gy = sd, ux = 0, uy = 1;
        int sgx = -1, sgy = 1;                                // gx == sgx*ux*sd, gy == sgy*uy*sd (mod d)
        for (int it = 0; it < 64 && wave_any(c, general && gy != 0); it++) {
            if (gy != 0) {
                const uint32_t q = gx / gy, t = gx - q * gy, tu = ux + q * uy;
                gx = gy; gy = t;
                ux = uy; uy = tu;
                const int ts = sgx; sgx = sgy; sgy = ts;
            }
        }
        const uint32_t d1 = gx;                               // gcd(sd, d) (sd == 0: d1 = d, ux = 0)
        BN<N> x2m;
        bn_set_word(c, x2m, ux);
        SBN<N> x2{x2m, (sgx < 0) ^ (s.neg != 0)};              // x2 * s == d1 (mod d) with the sign of s folded in
        // y2 = (x2p * |s| - d1) / d, exact (x2p = sgx * ux)
        SBN<N> y2;
        {
            BN<N> t;
            (void)bn_mul_word(c, t, s.m, ux);                 // |s| * ux
            SBN<N> ts{t, sgx < 0}, d1s;
            bn_set_word(c, d1s.m, d1);
            d1s.neg = 0;
            SBN<N> num;
            sbn_sub(c, num, ts, d1s);
            BN<N> q;
            (void)bn_divrem_word(c, q, num.m, dw);
            y2.m = q;
            y2.neg = num.neg;
            if (sd == 0) {                                    // s divisible by d: d1 = d, x2 = 0, y2 = -1
                bn_zero(x2.m); x2.neg = 0;
                bn_set_word(c, y2.m, 1); y2.neg = 1;
            }
        }
        BN<N> gv1, gv2, gc2d;
        (void)bn_divrem_word(c, gv1, a1, d1 ? d1 : 1u);
        (void)bn_divrem_word(c, gv2, a2, d1 ? d1 : 1u);
        const uint32_t ctop = bn_mul_word(c, gc2d, c2, d1);
        if (general && (ctop != 0 || bn_bitlen(c, gc2d) > LIM)) P2_LEAVE(ok, "qf2.hpp:404");
        // r = (y1*y2*(-m) - x2*c2) mod v1
        BN<N> L, H, wr, t1, c2r, t2;
        bool okg = true;
        bn_mul(c, L, H, y1.m, y2.m);
        bn_rem(c, wr, L, H, gv1, general, okg);
        const bool wneg = (y1.neg ^ y2.neg) != 0;              // sign of y1*y2; times (-m): flips with !m.neg
        bn_mul(c, L, H, wr, m.m);
        bn_rem(c, t1, L, H, gv1, general, okg);
        const bool t1neg = wneg ^ (m.neg == 0);                // residue t1 stands for (+-) t1
        BN<N> z;
        bn_zero(z);
        bn_rem(c, c2r, c2, z, gv1, general, okg);
        bn_mul(c, L, H, x2.m, c2r);
        bn_rem(c, t2, L, H, gv1, general, okg);
        SBN<N> T1{t1, t1neg}, T2{t2, x2.neg != 0}, df;
        sbn_sub(c, df, T1, T2);
        // df in (-2 v1, 2 v1): bring into [0, v1)
        BN<N> rg = df.m;
        {
            BN<N> t;
            const uint32_t bw = bn_sub(c, t, rg, gv1);
            if (!bw) rg = t;                                   // |df| >= v1: subtract once (|df| < 2 v1)
            const bool zz = bn_is_zero(c, rg);
            BN<N> comp;
            (void)bn_sub(c, comp, gv1, rg);
            if (df.neg && !zz) rg = comp;
        }
        if (general) {
            if (!okg) P2_LEAVE(ok, "qf2.hpp:433");
            v1 = gv1; v2 = gv2; c2d = gc2d; r = rg;
        }
    }

    P2_PHASE(4);
    // ---- partial Euclid on (v1, r)
    const int lv1 = bn_bitlen(c, v1), lv2 = bn_bitlen(c, v2);
    const int stop = (lv1 - lv2 + half_dbits) / 2;
    Euclid2<N> pe;
    pe.x = v1; pe.y = r;
    bn_zero(pe.u); bn_set_word(c, pe.v, 1);
    pe.sx = -1; pe.sy = 1;
    euclid_pair(c, pe, stop, active, ok);
    P2_PHASE(5);
    const SBN<N> C0{pe.u, pe.sx < 0}, C1{pe.v, pe.sy < 0};
    const int sg_neg = C1.neg;
    const BN<N> &R0 = pe.x, &R1 = pe.y;

    // ---- M1 = (v2 R1 - m C1)/v1, M2 = (s R1 + c2d C1)/v1, exact and SIGNED.  A 2-adic division only needs the
    // numerator modulo 2^(32 nq): the low limbs of the two products are combined modulo 2^(64N) (two's
    // complement) and the quotient modulo 2^(32 nq) is the two's complement of the signed quotient as soon as nq
    // limbs hold |M| plus a sign bit -- nq comes from the bit lengths of the factors.
    SBN<N> M1, M2;
    {
        const int lR1 = bn_bitlen(c, R1), lC1 = bn_bitlen(c, C1.m), lm = bn_bitlen(c, m.m), ls = bn_bitlen(c, s.m), lc2 = bn_bitlen(c, c2d);
        auto signed_quot = [&](SBN<N> &M, const BN<N> &Pl, const BN<N> &Sl, bool add, int bits_p, int bits_s) {
            // numerator low limbs: P + S or P - S (mod 2^(64N))
            BN<N> lo_add, lo_sub, lo;
            (void)bn_add(c, lo_add, Pl, Sl);
            (void)bn_sub(c, lo_sub, Pl, Sl);
            bn_select(lo, add, lo_sub, lo_add);
            const int nbits = (bits_p > bits_s ? bits_p : bits_s) + 1 - lv1 + 1 + 1;      // |M| + sign bit
            int nq = nbits <= 0 ? 1 : (nbits + 31) / 32;
            if (nq > 2 * N) { P2_LEAVE(ok, "qf2.hpp:465"); nq = 1; }
            const int nqu = wave_max_small(c, nq, active && ok);
            BN<N> Q;
            bn_divexact(c, Q, lo, 0u, 0, v1, nqu, ok);
            // sign: top bit of limb nqu - 1; magnitude of a negative quotient = 2^(32 nqu) - Q
            const uint32_t topl = nqu > 0 ? bn_limb_at(c, Q, nqu - 1) : 0u;
            const bool neg = (topl >> 31) != 0;
            BN<N> z, nQ;
            bn_zero(z);
            (void)bn_sub(c, nQ, z, Q);
            P2_UNROLL for (int j = 0; j < N; j++) nQ.v[j] = (c.hi * N + j < nqu) ? nQ.v[j] : 0u;
            bn_select(M.m, neg, Q, nQ);
            M.neg = neg;
        };
        BN<N> L1, H1, L2, H2;
        bn_mul(c, L1, H1, v2, R1);
        bn_mul(c, L2, H2, m.m, C1.m);
        // n1 = v2 R1 - (sign) |m C1|
        signed_quot(M1, L1, L2, /*add=*/((m.neg ^ C1.neg) != 0), lv2 + lR1, lm + lC1);
        bn_mul(c, L1, H1, s.m, R1);
        bn_mul(c, L2, H2, c2d, C1.m);
        // n2 = (+-)|s R1| + (+-)|c2d C1|: factor the sign of the first term out
        const bool p_neg = s.neg != 0, q_neg = C1.neg != 0;
        signed_quot(M2, L1, L2, /*add=*/(p_neg == q_neg), ls + lR1, lc2 + lC1);
        M2.neg ^= p_neg ? 1 : 0;
        if (bn_is_zero(c, M2.m)) M2.neg = 0;
        if (bn_is_zero(c, M1.m)) M1.neg = 0;
    }

    P2_PHASE(6);
    // ---- a' = R1 M1 + C1 M2,  b' = -sign(C1) 2 (R0 M1 + C0 M2) - b1   (all within 2N limbs on the fast path)
    SBN<N> an, bs;
    {
        BN<N> L, H, L2, H2;
        bn_mul(c, L, H, R1, M1.m);
        bn_mul(c, L2, H2, C1.m, M2.m);
        if (!bn_is_zero(c, H) || !bn_is_zero(c, H2)) P2_LEAVE(ok, "qf2.hpp:500");
        SBN<N> p1{L, M1.neg}, p2{L2, (C1.neg ^ M2.neg) != 0};
        sbn_add(c, an, p1, p2);
        bn_mul(c, L, H, R0, M1.m);
        bn_mul(c, L2, H2, C0.m, M2.m);
        if (!bn_is_zero(c, H) || !bn_is_zero(c, H2)) P2_LEAVE(ok, "qf2.hpp:505");
        SBN<N> q1{L, M1.neg}, q2{L2, (C0.neg ^ M2.neg) != 0};
        sbn_add(c, bs, q1, q2);
    }
    if (an.neg && !bn_is_zero(c, an.m)) P2_LEAVE(ok, "qf2.hpp:509");
    SBN<N> bn_;
    {
        SBN<N> two_bs;
        (void)bn_add(c, two_bs.m, bs.m, bs.m);
        two_bs.neg = bs.neg ^ (sg_neg ? 0 : 1);
        sbn_sub(c, bn_, two_bs, b1);
    }
    if (bn_bitlen(c, an.m) > LIM || bn_bitlen(c, bn_.m) > LIM || bn_is_zero(c, an.m)) P2_LEAVE(ok, "qf2.hpp:517");

    P2_PHASE(7);
    // ---- c' = (b'^2 + |Delta|) / (4 a'): exact, positive; the low 2N limbs of the numerator (after the shift
    // by two bits) are all the 2-adic division needs
    BN<N> cn;
    {
        BN<N> L, H;
        bn_mul(c, L, H, bn_.m, bn_.m);
        BN<N> dl;
        P2_UNROLL for (int j = 0; j < N; j++) dl.v[j] = dd.absdelta[c.hi * N + j];
        const uint32_t dh0 = dd.absdelta[2 * N];
        BN<N> nl;
        const uint32_t cy = bn_add(c, nl, L, dl);
        const uint32_t h0 = bn_limb<0>(c, H) + dh0 + cy;                 // limb 2N of the numerator (mod 2^32)
        const int lbn = bn_bitlen(c, bn_.m), lan = bn_bitlen(c, an.m);
        const int nbn = (2 * lbn > 2 * half_dbits ? 2 * lbn : 2 * half_dbits) + 1;      // bits of b'^2 + |Delta| (upper bound)
        const int qbits = nbn - 2 - lan + 2;                                            // c' < 2^qbits
        if (qbits > 64 * N - 3) P2_LEAVE(ok, "c' beyond 2N limbs");
        int nq = (qbits + 31) / 32;
        if (nq < 1) nq = 1;
        if (nq > 2 * N) nq = 2 * N;
        const int nqu = wave_max_small(c, nq, active && ok);
        bn_divexact(c, cn, nl, h0, 2, an.m, nqu, ok);
        cn = bn_mask_bits(c, cn, qbits);
    }
    P2_PHASE(8);
    // ---- reduce
    BN<N> ra = an.m, rc = cn;
    SBN<N> rb = bn_;
    qf2_reduce(c, ra, rb, rc, active, ok);
    out.a = ra;
    out.bm = rb.m;
    out.bneg = rb.neg;
    out.c = rc;
    P2_PHASE(9);
}

// ---------------------------------------------------------------------------------------- records
// record words (layout.hpp): a[40] | |b|[40] | c[80] | sign | pad; the pair reads 2N limbs of each plane and checks
// that the rest is zero (otherwise the form is not on the fast path)
template <int N>
P2_DEV void qf2_load(PCtx &c, QForm2<N> &f, const uint32_t *rec, bool &ok) {
    using namespace cofhe;
    uint32_t extra = 0;
    P2_UNROLL for (int j = 0; j < N; j++) {
        f.a.v[j] = rec[REC_A + c.hi * N + j];
        f.bm.v[j] = rec[REC_B + c.hi * N + j];
        f.c.v[j] = rec[REC_C + c.hi * N + j];
    }
    // words beyond 2N limbs: a, b planes have 40 - 2N, c has 80 - 2N; each lane checks half of them
    for (int j = 2 * N + c.hi; j < PLIMBS; j += 2) extra |= rec[REC_A + j] | rec[REC_B + j];
    for (int j = 2 * N + c.hi; j < 2 * PLIMBS; j += 2) extra |= rec[REC_C + j];
    if ((extra | xl(c, extra)) != 0) P2_LEAVE(ok, "qf2.hpp:571");
    f.bneg = (int)rec[REC_SIGN];
}
template <int N>
P2_DEV void qf2_store(PCtx &c, const QForm2<N> &f, uint32_t *rec) {
    using namespace cofhe;
    P2_UNROLL for (int j = 0; j < N; j++) {
        rec[REC_A + c.hi * N + j] = f.a.v[j];
        rec[REC_B + c.hi * N + j] = f.bm.v[j];
        rec[REC_C + c.hi * N + j] = f.c.v[j];
    }
    for (int j = 2 * N + c.hi; j < PLIMBS; j += 2) { rec[REC_A + j] = 0u; rec[REC_B + j] = 0u; }
    for (int j = 2 * N + c.hi; j < 2 * PLIMBS; j += 2) rec[REC_C + j] = 0u;
    if (c.hi == 0) rec[REC_SIGN] = (uint32_t)f.bneg;
    else for (int j = 1; j < 8; j++) rec[REC_SIGN + j] = 0u;
}

}  // namespace cofhe2
