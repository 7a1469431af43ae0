// Host lane-group simulator: compiles the DEVICE arithmetic (cofhe_amd/csrc/{mp,qf}.hpp)
// with COFHE_HOSTSIM, running the 8 lanes of a limb group as 8 host threads.  TEST
// INFRASTRUCTURE ONLY: lets the CPU-only test tier exercise the very code the HIP kernels
// run; it is not linked into, nor loadable by, the product library.
#define COFHE_HOSTSIM 1
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../cofhe_amd/csrc/form_io.hpp"

using namespace cofhe;

template <typename F>
static void run_group(F &&fn) {
    static GroupShared gs;
    std::vector<std::thread> th;
    for (int l = 0; l < G; l++)
        th.emplace_back([&, l]() {
            Ctx c;
            c.gl = l;
            c.gs = &gs;
            c.sense = gs.bar.sense.load();
            fn(c);
        });
    for (auto &t : th) t.join();
}

// A whole workgroup as host threads: WG_GROUPS groups of G lanes, one LDS image (the groups' slices followed by the
// mailbox, as in the kernels), a barrier over all threads for __syncthreads and one over the serving wavefront's
// threads for its ballot.  Lets qf_compose<true> -- the workgroup-served remainder sequence of every kernel -- run here.
template <typename F>
static void run_workgroup(F &&fn) {
    constexpr int NT = WG_GROUPS * G;
    static_assert(NT <= 64 || NT % 64 == 0, "whole wavefronts");
    std::vector<GroupShared> groups(WG_GROUPS);
    std::vector<uint32_t> lds(WG_GROUPS * SCRATCH_WORDS + WG_MAIL_WORDS, 0u);
    WgShared wg;
    wg.bar.n = NT;
    wg.bar.yield = true;
    wg.wave_threads = NT < 64 ? NT : 64;
    wg.wave_bar.n = wg.wave_threads;
    wg.wave_bar.yield = true;
    for (int g = 0; g < WG_GROUPS; g++) {
        groups[g].scratch = lds.data() + g * SCRATCH_WORDS;
        groups[g].bar.yield = true;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < NT; t++)
        th.emplace_back([&, t]() {
            Ctx c;
            c.gl = t % G;
            c.gs = &groups[t / G];
            c.sense = c.gs->bar.sense.load();
            c.wg = &wg;
            c.wg_sense = wg.bar.sense.load();
            c.wave_sense = wg.wave_bar.sense.load();
            c.tid = t;
            c.gi = t / G;
            c.wave = t / 64;                       // wavefront 0 serves (the kernels rotate the index per workgroup)
            c.rank = -1;
            c.wg_mail = lds.data() + WG_GROUPS * SCRATCH_WORDS;
            c.wg_scr0 = lds.data();
            fn(c);
        });
    for (auto &t : th) t.join();
}

template <int P>
static Mp<P> ld(const Ctx &c, const uint32_t *w) {
    Mp<P> x;
    for (int p = 0; p < P; p++)
        for (int j = 0; j < CH; j++) x.v[p][j] = w[p * PLIMBS + c.gl * CH + j];
    return x;
}
template <int P>
static void st(const Ctx &c, const Mp<P> &x, uint32_t *w) {
    for (int p = 0; p < P; p++)
        for (int j = 0; j < CH; j++) w[p * PLIMBS + c.gl * CH + j] = x.v[p][j];
}

extern "C" {
unsigned sim_status(void) { return g_sim_status.exchange(0); }     // device status word of the simulator (lane.hpp: CF_ST_*)
void sim_stats(long *out) { memcpy(out, &g_stats, sizeof(g_stats)); memset(&g_stats, 0, sizeof(g_stats)); }

// out[80] = x[40] * y[40], count instances
void sim_mul11(const uint32_t *x, const uint32_t *y, uint32_t *out, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            Mp<2> r = mp_mul(c, ld<1>(c, x + 40 * i), ld<1>(c, y + 40 * i));
            st(c, r, out + 80 * i);
        }
    });
}
// out[120] = x[80] * y[40]
void sim_mul21(const uint32_t *x, const uint32_t *y, uint32_t *out, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            Mp<3> r = mp_mul(c, ld<2>(c, x + 80 * i), ld<1>(c, y + 40 * i));
            st(c, r, out + 120 * i);
        }
    });
}
// num[80] / den[40] -> quot[80], rem[80]
void sim_divrem21(const uint32_t *num, const uint32_t *den, uint32_t *quot, uint32_t *rem, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            Mp<2> n = ld<2>(c, num + 80 * i), q;
            mp_divrem(c, n, ld<1>(c, den + 40 * i), q);
            st(c, q, quot + 80 * i);
            st(c, n, rem + 80 * i);
        }
    });
}
// exact division num[80] / den[40] -> quot[80] (2-adic, mp_divexact); nq quotient limbs requested
void sim_divexact21(const uint32_t *num, const uint32_t *den, uint32_t *quot, const int *nq, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            Mp<2> q;
            mp_divexact(c, ld<2>(c, num + 80 * i), ld<1>(c, den + 40 * i), q, nq[i]);
            st(c, q, quot + 80 * i);
        }
    });
}
// x[80] mod W through mp_mod_word_fast (the word route's residues), one and two planes; out[2 i] = low plane, out[2 i + 1] = both
void sim_mod_word_fast(const uint32_t *x, const uint32_t *W, uint32_t *out, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            const ModW mw = modw_make(c, W[i]);
            const Mp<2> v = ld<2>(c, x + 80 * i);
            const uint32_t r1 = mp_mod_word_fast(c, mp_resize<1>(v), mw), r2 = mp_mod_word_fast(c, v, mw);
            if (c.gl == 0) {
                out[2 * i] = r1;
                out[2 * i + 1] = r2;
            }
        }
    });
}
// x[40] mod 2*3*...*23 through mp_mod_primorial (the representative test of the composition)
void sim_mod_primorial(const uint32_t *x, uint32_t *out, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            const uint32_t r = mp_mod_primorial(c, ld<1>(c, x + 40 * i));
            if (c.gl == 0) out[i] = r;
        }
    });
}
// word_xgcd16: g = gcd(m, a), inv a == g (mod m) for 0 < a < m < 2^16
void sim_word_xgcd16(const uint32_t *m, const uint32_t *a, uint32_t *g, uint32_t *inv, int count) {
    for (int i = 0; i < count; i++) word_xgcd16(m[i], a[i], g[i], inv[i]);
}
// r[80] = A*x - B*y  and  s[80] = A*x + B*y (mod 2^2560)
void sim_lincomb(const uint32_t *x, const uint32_t *y, uint32_t A, uint32_t B, uint32_t *r, uint32_t *s, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            Mp<2> a = ld<2>(c, x + 80 * i), b = ld<2>(c, y + 80 * i), o;
            mp_lincomb_sub(c, o, A, a, B, b);
            st(c, o, r + 80 * i);
            (void)mp_lincomb_add(c, o, A, a, B, b);
            st(c, o, s + 80 * i);
        }
    });
}
// shifts / bit length / compare
void sim_shift(const uint32_t *x, int n, uint32_t *l, uint32_t *r, uint32_t *h, int *bits, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            Mp<2> a = ld<2>(c, x + 80 * i);
            st(c, mp_shl(c, a, n), l + 80 * i);
            st(c, mp_shr(c, a, n), r + 80 * i);
            st(c, mp_shr1(c, a), h + 80 * i);
            int b = mp_bitlen(c, a);
            if (c.gl == 0) bits[i] = b;
        }
    });
}
// full xgcd: d[40], u[40] with sign*u*y == d (mod x)
void sim_xgcd(const uint32_t *x, const uint32_t *y, uint32_t *d, uint32_t *u, int *sign, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            Euclid<1> e;
            e.x = ld<1>(c, x + 40 * i);
            e.y = ld<1>(c, y + 40 * i);
            mp_zero(e.ux);
            mp_set_word(c, e.uy, 1);
            e.sx = -1;
            e.sy = 1;
            euclid_run(c, e, -1);
            st(c, e.x, d + 40 * i);
            st(c, e.ux, u + 40 * i);
            if (c.gl == 0) sign[i] = e.sx;
        }
    });
}
// the product's batch (mp.hpp: lehmer_batch, double precision, windows below 2^53, cofactors below 2^26)
int sim_lehmer_f64(uint64_t xh, uint64_t yh, int exact, uint64_t thr, uint32_t *out) {
    return lehmer_batch(xh, yh, exact != 0, thr, out[0], out[1], out[2], out[3]) ? 1 : 0;
}
// the single-chain form of the batch (mp.hpp: lehmer_batch_uniform): cap 8 = the serving lane's, 12 = the wide layout's
int sim_lehmer_uniform(uint64_t xh, uint64_t yh, int exact, uint64_t thr, int cap, uint32_t *out) {
    if (cap == 8) return lehmer_batch_uniform<8>(xh, yh, exact != 0, (double)thr, out[0], out[1], out[2], out[3]) ? 1 : 0;
    return lehmer_batch_uniform<12>(xh, yh, exact != 0, (double)thr, out[0], out[1], out[2], out[3]) ? 1 : 0;
}
// the scalar routine of the serving lane (mp.hpp: euclid_serve) on one request: x[40] | y[40], state in/out
void sim_euclid_serve(const uint32_t *xy, int stop_bits, int *tx, int *ty, int *sdone, uint32_t *w) {
    uint32_t ww[SERVE_WORDS] = {0};
    bool sd = *sdone != 0;
    euclid_serve(xy, stop_bits, *tx, *ty, sd, ww);
    *sdone = sd ? 1 : 0;
    memset(w, 0, 8 * sizeof(uint32_t));           // always 8 words out: w4..w7 = second matrix of the round (bit 31 of w4: present)
    memcpy(w, ww, sizeof(ww));
}
// reduce records in place
void sim_reduce(uint32_t *a, uint32_t *b, int *bneg, uint32_t *cc, int count) {
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            Mp<2> A = ld<2>(c, a + 80 * i), C = ld<2>(c, cc + 80 * i);
            SMp<2> B{ld<2>(c, b + 80 * i), bneg[i]};
            group_sync(c);
            qf_reduce<2>(c, A, B, C);
            st(c, A, a + 80 * i);
            st(c, B.m, b + 80 * i);
            st(c, C, cc + 80 * i);
            group_sync(c);
            if (c.gl == 0) bneg[i] = B.neg;
            group_sync(c);
        }
    });
}
// out[i] = base[i] ^ exps[i]  (exponent records of 32 words, qf.hpp) -- the binary ladder,
// the sign / zero handling and qf_inverse
void sim_pow(const uint32_t *base, const uint32_t *exps, const uint32_t *one, uint32_t *out, int count, int half_dbits,
             const uint32_t *absdelta) {
    const QDisc dd{absdelta, half_dbits};
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            QForm b, o, r;
            qf_load(c, b, base + (size_t)REC_WORDS * i);
            qf_load(c, o, one);
            qf_pow(c, r, b, exps + (size_t)EXP_REC_WORDS * i, o, dd);
            qf_store(c, r, out + (size_t)REC_WORDS * i);
        }
    });
}
// out[i] = f1[i] * f2[i] on form records (layout.hpp)
// the kernels' form: one workgroup, group gi composes item min(gi, count - 1) with the remainder sequences served by
// wavefront 0 (mp.hpp: euclid_run_wg), count <= WG_GROUPS
int sim_wg_groups(void) { return WG_GROUPS; }
// The workgroup-served remainder sequence by itself (mp.hpp: euclid_run_wg, packed remainder / cofactor pairs), `count`
// pairs (x[40], y[40]) in one simulated workgroup; stop < 0: down to the gcd.  out per pair: x[40] y[40] ux[40] uy[40], and
// sign[2 i], sign[2 i + 1] = sx, sy.  Same entry state as qf_compose uses (ux = 0, uy = 1, sx = -1, sy = +1).
void sim_euclid_wg(const uint32_t *x, const uint32_t *y, int count, const int *stop, uint32_t *out, int *sign) {
    run_workgroup([&](Ctx &c) {
        const int i = c.gi < count ? c.gi : count - 1;
        Euclid<1> e;
        e.x = ld<1>(c, x + 40 * i);
        e.y = ld<1>(c, y + 40 * i);
        mp_zero(e.ux);
        mp_set_word(c, e.uy, 1);
        e.sx = -1;
        e.sy = 1;
        euclid_run_wg(c, e, stop[i]);
        if (c.gi < count) {
            st(c, e.x, out + 160 * i);
            st(c, e.y, out + 160 * i + 40);
            st(c, e.ux, out + 160 * i + 80);
            st(c, e.uy, out + 160 * i + 120);
            if (c.gl == 0) {
                sign[2 * i] = e.sx;
                sign[2 * i + 1] = e.sy;
            }
        }
    });
}
// which route common word-sized factors of the first coefficients take (qf_compose's WORD_ROUTE): 1 = the word route of the
// one-composition kernels, 0 = the general formula of the sequence kernels
static int g_word_route = 1;
void sim_set_word_route(int on) { g_word_route = on; }
void sim_compose_wg(const uint32_t *f1, const uint32_t *f2, uint32_t *out, int count, int half_dbits, const uint32_t *absdelta) {
    const QDisc dd{absdelta, half_dbits};
    run_workgroup([&](Ctx &c) {
        const int i = c.gi < count ? c.gi : count - 1;
        QForm a, b, r;
        qf_load(c, a, f1 + (size_t)REC_WORDS * i);
        qf_load(c, b, f2 + (size_t)REC_WORDS * i);
        if (g_word_route) qf_compose<true, true>(c, r, a, b, dd); else qf_compose<true, false>(c, r, a, b, dd);
        if (c.gi < count) qf_store(c, r, out + (size_t)REC_WORDS * i);
    });
}
void sim_compose(const uint32_t *f1, const uint32_t *f2, uint32_t *out, int count, int half_dbits, const uint32_t *absdelta) {
    const QDisc dd{absdelta, half_dbits};
    run_group([&](Ctx &c) {
        for (int i = 0; i < count; i++) {
            QForm a, b, r;
            qf_load(c, a, f1 + (size_t)REC_WORDS * i);
            qf_load(c, b, f2 + (size_t)REC_WORDS * i);
            if (g_word_route) qf_compose<false, true>(c, r, a, b, dd); else qf_compose<false, false>(c, r, a, b, dd);
            qf_store(c, r, out + (size_t)REC_WORDS * i);
        }
    });
}
}
