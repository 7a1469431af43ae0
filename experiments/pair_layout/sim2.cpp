// Host lane-PAIR simulator: compiles the device arithmetic of the pair layout (cofhe_amd/csrc/pair.hpp,
// qf2.hpp) with COFHE_HOSTSIM, the two lanes of a pair as two host threads.  TEST INFRASTRUCTURE ONLY.
#define COFHE_HOSTSIM 1
#include <cstring>
#include <thread>
#include <vector>

#include "qf2.hpp"

using namespace cofhe2;

#ifndef SIM2_N
#define SIM2_N 17
#endif
constexpr int N = SIM2_N;
constexpr int W = 2 * N;            // limbs of a BN

template <typename F>
static void run_pair(F &&fn) {
    static PairShared ps;
    std::thread t0([&]() { PCtx c; c.hi = 0; c.ps = &ps; c.sense = ps.sense.load(); fn(c); });
    std::thread t1([&]() { PCtx c; c.hi = 1; c.ps = &ps; c.sense = ps.sense.load(); fn(c); });
    t0.join();
    t1.join();
}
static BN<N> ld(const PCtx &c, const uint32_t *w) {
    BN<N> x;
    for (int j = 0; j < N; j++) x.v[j] = w[c.hi * N + j];
    return x;
}
static void st(const PCtx &c, const BN<N> &x, uint32_t *w) {
    for (int j = 0; j < N; j++) w[c.hi * N + j] = x.v[j];
}

extern "C" {
int sim2_limbs(void) { return W; }
const char *sim2_last_reason(void) { return g_p2_last_reason; }

// basic: out[0..W) = x + y, out[W..2W) = x - y, misc[0] = carry, [1] = borrow, [2] = cmp + 1, [3] = bitlen(x), [4] = is_zero(y)
void sim2_addsub(const uint32_t *x, const uint32_t *y, uint32_t *out, int *misc, int count) {
    run_pair([&](PCtx &c) {
        for (int i = 0; i < count; i++) {
            BN<N> a = ld(c, x + W * i), b = ld(c, y + W * i), r;
            const uint32_t cy = bn_add(c, r, a, b);
            st(c, r, out + 2 * W * i);
            const uint32_t bw = bn_sub(c, r, a, b);
            st(c, r, out + 2 * W * i + W);
            const int cm = bn_cmp(c, a, b), bl = bn_bitlen(c, a), z = bn_is_zero(c, b) ? 1 : 0;
            if (c.hi == 0) { misc[5 * i] = (int)cy; misc[5 * i + 1] = (int)bw; misc[5 * i + 2] = cm + 1; misc[5 * i + 3] = bl; misc[5 * i + 4] = z; }
        }
    });
}
// shifts: out = [x >> n | x << n | x << 32 | x >> 32]
void sim2_shift(const uint32_t *x, int n, uint32_t *out, int count) {
    run_pair([&](PCtx &c) {
        for (int i = 0; i < count; i++) {
            BN<N> a = ld(c, x + W * i);
            st(c, bn_shr_small(c, a, n), out + 4 * W * i);
            st(c, bn_shl_small(c, a, n), out + 4 * W * i + W);
            st(c, bn_shl_limb(c, a), out + 4 * W * i + 2 * W);
            st(c, bn_shr_limb(c, a), out + 4 * W * i + 3 * W);
        }
    });
}
// r = A x - B y (mod), s = A x + B y (mod), tops[2i] = word of sub_carry, tops[2i+1] = word of add
void sim2_lincomb(const uint32_t *x, const uint32_t *y, uint32_t A, uint32_t B, uint32_t *r, uint32_t *s, uint32_t *tops, int count) {
    run_pair([&](PCtx &c) {
        for (int i = 0; i < count; i++) {
            BN<N> a = ld(c, x + W * i), b = ld(c, y + W * i), o;
            const uint32_t w1 = bn_lincomb_sub_carry(c, o, A, a, B, b);
            st(c, o, r + W * i);
            const uint32_t w2 = bn_lincomb_add(c, o, A, a, B, b);
            st(c, o, s + W * i);
            if (c.hi == 0) { tops[2 * i] = w1; tops[2 * i + 1] = w2; }
        }
    });
}
// R = (L + H B^W) mod D; ok flags out
void sim2_rem(const uint32_t *lo, const uint32_t *hi, const uint32_t *den, uint32_t *rem, int *okout, int count) {
    run_pair([&](PCtx &c) {
        for (int i = 0; i < count; i++) {
            BN<N> r;
            bool ok = true;
            bn_rem(c, r, ld(c, lo + W * i), ld(c, hi + W * i), ld(c, den + W * i), true, ok);
            st(c, r, rem + W * i);
            if (c.hi == 0) okout[i] = ok ? 1 : 0;
        }
    });
}
// Q = Wl / D mod 2^(32 nq) (exact 2-adic)
void sim2_divexact(const uint32_t *w, const uint32_t *den, uint32_t *quot, const int *nq, int *okout, int count) {
    run_pair([&](PCtx &c) {
        for (int i = 0; i < count; i++) {
            BN<N> q;
            bool ok = true;
            bn_divexact(c, q, ld(c, w + W * i), 0u, 0, ld(c, den + W * i), nq[i], ok);
            st(c, q, quot + W * i);
            if (c.hi == 0) okout[i] = ok ? 1 : 0;
        }
    });
}
// word helpers: out[i] = [x mod 223092870 | x mod w | (x / w) limbs... ]
void sim2_words(const uint32_t *x, const uint32_t *ws, uint32_t *small, uint32_t *modw, uint32_t *quot, uint32_t *rem, int count) {
    constexpr ScreenTable ST = make_screen_table();
    run_pair([&](PCtx &c) {
        for (int i = 0; i < count; i++) {
            BN<N> a = ld(c, x + W * i), q;
            const uint32_t s = bn_mod_small(c, a, SCREEN_M, ST.pw), m = bn_mod_word(c, a, ws[i]);
            const uint32_t r = bn_divrem_word(c, q, a, ws[i]);
            st(c, q, quot + W * i);
            if (c.hi == 0) { small[i] = s; modw[i] = m; rem[i] = r; }
        }
    });
}
// full xgcd / partial sequence on (x, y): out x, y, u, v; signs[2i], [2i+1]; okout
void sim2_euclid(const uint32_t *x, const uint32_t *y, int stop_bits, uint32_t *out, int *signs, int *okout, int count) {
    run_pair([&](PCtx &c) {
        for (int i = 0; i < count; i++) {
            Euclid2<N> e;
            e.x = ld(c, x + W * i); e.y = ld(c, y + W * i);
            bn_zero(e.u); bn_set_word(c, e.v, 1);
            e.sx = -1; e.sy = 1;
            bool ok = true;
            euclid_pair(c, e, stop_bits, true, ok);
            st(c, e.x, out + 4 * W * i); st(c, e.y, out + 4 * W * i + W); st(c, e.u, out + 4 * W * i + 2 * W); st(c, e.v, out + 4 * W * i + 3 * W);
            if (c.hi == 0) { signs[2 * i] = e.sx; signs[2 * i + 1] = e.sy; okout[i] = ok ? 1 : 0; }
        }
    });
}
// out[i] = f1[i] * f2[i] on form records (layout.hpp); okout[i] = 0 when the element left the fast path
void sim2_compose(const uint32_t *f1, const uint32_t *f2, uint32_t *out, int *okout, int count, int half_dbits, const uint32_t *absdelta, int screen) {
    const QDisc2 dd{absdelta, half_dbits};
    run_pair([&](PCtx &c) {
        for (int i = 0; i < count; i++) {
            QForm2<N> a, b, r;
            bool ok = true;
            if (c.hi == 0) g_p2_last_reason = "";
            qf2_load(c, a, f1 + (size_t)cofhe::REC_WORDS * i, ok);
            qf2_load(c, b, f2 + (size_t)cofhe::REC_WORDS * i, ok);
            qf2_compose(c, r, a, b, dd, true, ok, screen != 0);
            if (ok) qf2_store(c, r, out + (size_t)cofhe::REC_WORDS * i);
            if (c.hi == 0) okout[i] = ok ? 1 : 0;
            if (c.hi == 0 && !ok && getenv("SIM2_VERBOSE")) fprintf(stderr, "sim2_compose[%d]: left the fast path at %s\n", i, g_p2_last_reason);
        }
    });
}
// out[2W] = x * y
void sim2_mul(const uint32_t *x, const uint32_t *y, uint32_t *out, int count) {
    run_pair([&](PCtx &c) {
        for (int i = 0; i < count; i++) {
            BN<N> L, H;
            bn_mul(c, L, H, ld(c, x + W * i), ld(c, y + W * i));
            st(c, L, out + 2 * W * i);
            st(c, H, out + 2 * W * i + W);
        }
    });
}
}
