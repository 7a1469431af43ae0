// lehmer_variants.hpp -- measured-and-rejected forms of the serving lane's Lehmer batch (NOT part of the product; the
// product's batch is cofhe_amd/csrc/mp.hpp: lehmer_batch).  Kept with their property tests
// (tests/test_hostsim_device_code.py through tests/hostsim/sim.cpp) and the numbers in DESIGN.md 5:
//   lehmer_batch2       two-level batch (31-bit inner phases folded into the 64-bit pair): 43 instead of 90 instructions
//                       per double-step but 5 phases / 23 inner iterations per wavefront against 13: 4.8 vs 2.5 us per batch
//                       in isolation, 0.599 vs 0.510 ms in the 128x128 composition kernel
//   lehmer_batch_wide   the batch on windows known only to (xh - 1, xh + 2): the second batch of a two-batch round
//   serve_second_batch  the serving lane's derivation of that second batch from a 128-bit window (two batches per round:
//                       0.4925 vs 0.4960 ms -- the saved round trips are paid back by the 128-bit arithmetic)
#pragma once
#include "../../cofhe_amd/csrc/mp.hpp"

namespace cofhe {

// ---- the product's batch until round 3: 64-bit integer windows, 31-bit cofactors (lehmer_batch_ref: the plain loop it
// was derived from; lehmer_batch_u64: flattened for latency).  Replaced by the double-precision batch of mp.hpp.
// double-digit Lehmer batch on the leading 64 bits with 31-bit cofactors.  Conservative
// quotients keep the true remainders non-negative for every value the truncated operands can
// stand for:   x' = A x - B y >= 0,  y' = D y - C x >= 0.
// A quotient estimate t = floor(num/den * (1 - 2^-20)) in float is never above the true
// quotient (the margin covers the conversions, the sum in den and the reciprocal) and at most
// one below it for quotients < 2^20; only the RELATIVE error of the estimate matters, so 64-bit
// operands need no wider float.  thr: stop once the smaller approximate remainder drops below
// thr (partial Euclid).
// Double-steps per batch.  The serving wavefront runs until its slowest lane has finished, and a lane that meets a run
// of small quotients needs up to ~13 double-steps for its 31 cofactor bits where the average lane needs 8-9.  Capped,
// such a lane hands back a slightly smaller matrix and catches up in a later round; the round gets shorter for the
// whole workgroup.  Measured on the 128x128 composition: 64 (no cap) 0.5105 ms, 11: 0.505, 9: 0.5075.
constexpr int LEHMER_U64_CAP = 11;
CF_DEV bool lehmer_batch_ref(uint64_t xh, uint64_t yh, bool exact, uint64_t thr, uint32_t &A, uint32_t &B,
                         uint32_t &C, uint32_t &D) {
    uint64_t p = xh, q = yh;
    uint32_t a = 1, b = 0, cc = 0, d = 1;
    const uint32_t eb = exact ? 0u : 0xFFFFFFFFu;
    const float MARGIN = 0.99999905f, TWO31 = 2147483648.0f;
    for (int it = 0; it < LEHMER_U64_CAP; it++) {
        {   // x -= t*y : t <= (p - b) / (q + d)
            const uint32_t ub = b & eb, ud = d & eb;
            const float tf = u64_to_float(p - ub) * (fast_rcp(u64_to_float(q) + (float)ud) * MARGIN);
            const uint32_t t = (uint32_t)tf;
            const uint64_t na = a + (uint64_t)t * cc, nb = b + (uint64_t)t * d;
            // (NaN/inf from q + d == 0 fail the comparisons below)
            if (!((p >= ub) & (tf >= 1.0f) & (tf < TWO31) & (((na | nb) >> 31) == 0))) break;
            p -= (uint64_t)t * (uint32_t)q + (((uint64_t)(t * (uint32_t)(q >> 32))) << 32);
            a = (uint32_t)na; b = (uint32_t)nb;
            if (p < thr) break;
        }
        {   // y -= t*x : t <= (q - c) / (p + a)
            const uint32_t uc = cc & eb, ua = a & eb;
            const float tf = u64_to_float(q - uc) * (fast_rcp(u64_to_float(p) + (float)ua) * MARGIN);
            const uint32_t t = (uint32_t)tf;
            const uint64_t nd = d + (uint64_t)t * b, nc = cc + (uint64_t)t * a;
            if (!((q >= uc) & (tf >= 1.0f) & (tf < TWO31) & (((nd | nc) >> 31) == 0))) break;
            q -= (uint64_t)t * (uint32_t)p + (((uint64_t)(t * (uint32_t)(p >> 32))) << 32);
            d = (uint32_t)nd; cc = (uint32_t)nc;
            if (q < thr) break;
        }
    }
    A = a; B = b; C = cc; D = d;
    return (b | cc) != 0;
}

// The batch above with its control flow flattened for LATENCY: in the serving wavefront every lane runs its own
// batch and the wavefront's time per round is the critical path of ONE lane (measured with tools/wg_timing.hip:
// 4.4 us of a 5.4 us Euclid round were spent waiting for the server).  Here a lane that has stopped keeps
// executing on dead values -- no exec-mask region and no compare -> scalar branch inside a half-step -- and the
// last valid matrix is kept in a snapshot; the only branch is the wave-uniform "everybody has stopped" test,
// taken on the flags of the PREVIOUS iteration so that the chain never waits for it.  Same contract as
// lehmer_batch_ref; tests/test_hostsim_device_code.py checks every matrix against the window intervals and the
// progress against the reference loop.
CF_DEV bool lehmer_batch_u64(uint64_t xh, uint64_t yh, bool exact, uint64_t thr, uint32_t &A, uint32_t &B,
                         uint32_t &C, uint32_t &D) {
    // Quotient first, validity second: t = floor(p / q) biased low by 2^-20 (never above the true quotient of the
    // windows), then the step is kept iff it is non-negative for every value the truncated operands can stand for,
    //   P - t Q >= 0  for  P > p - b, Q < q + d   <=>   p - t q >= b + t d   (the new remainder >= the new cofactor),
    // which is one 64-bit compare on values the step computes anyway -- the reference loop biases the quotient itself,
    // t <= (p - b) / (q + d), at a two-word subtraction, a conversion and an addition more per half-step.  A step
    // whose full quotient is not provably safe ends the batch (the reference would take a smaller one and go on):
    // 0.6 % fewer cofactor bits per batch (tests/test_hostsim_device_code.py), 25 % fewer instructions.
    // The cofactor columns are continuants: a <= b and c <= d after the first step, so the 31-bit bound is tested
    // on the larger one only.
    uint64_t p = xh, q = yh;
    uint32_t a = 1, b = 0, cc = 0, d = 1;           // working state: runs on, meaningless once the lane has stopped
    uint32_t ra = 1, rb = 0, rc = 0, rd = 1;       // state after the last valid half-step
    const uint64_t eb = exact ? 0ull : ~0ull;
    const float MARGIN = 0.99999905f;
    bool alive = true, any_prev = true;
    for (int it = 0; it < LEHMER_U64_CAP; it++) {
        if (!any_prev) break;
        {   // x -= t*y
            const float tf = u64_to_float(p) * (fast_rcp(u64_to_float(q)) * MARGIN);
            const uint32_t t = f32_to_u32_sat(tf);          // q == 0: saturates (or NaN -> 0); both fail below
            const uint64_t nb = b + (uint64_t)t * d;
            a += t * cc;
            p -= (uint64_t)t * (uint32_t)q + (((uint64_t)(t * (uint32_t)(q >> 32))) << 32);
            b = (uint32_t)nb;
            alive = alive & (t != 0u) & (nb < 0x80000000ull) & (p >= (nb & eb));
            ra = alive ? a : ra; rb = alive ? b : rb;
            alive = alive & !(p < thr);
        }
        {   // y -= t*x
            const float tf = u64_to_float(q) * (fast_rcp(u64_to_float(p)) * MARGIN);
            const uint32_t t = f32_to_u32_sat(tf);
            const uint64_t nd = d + (uint64_t)t * b;
            cc += t * a;
            q -= (uint64_t)t * (uint32_t)p + (((uint64_t)(t * (uint32_t)(p >> 32))) << 32);
            d = (uint32_t)nd;
            alive = alive & (t != 0u) & (nd < 0x80000000ull) & (q >= ((uint64_t)cc & eb));
            rd = alive ? d : rd; rc = alive ? cc : rc;
            alive = alive & !(q < thr);
        }
        any_prev = CF_WAVE_ANY(alive);
    }
    A = ra; B = rb; C = rc; D = rd;
    return (rb | rc) != 0;
}


// Two-level batch: the same contract as lehmer_batch (xh >= yh; x' = A x - B y >= 0 and y' = D y - C x >= 0 for every
// value the windows can stand for; 31-bit cofactors), at under half the instructions per quotient.  The serving
// wavefront's batch IS the critical path of a Euclid round (tools/wg_timing.hip), and a half-step on 64-bit
// remainders costs ~50 instructions (two-word subtractions, three-instruction u64 -> f32 images, 64-bit
// multiply-subtract, 64-bit compares).  Here the 64-bit pair (p, q) is only touched between PHASES; inside a phase
// the sequence runs on the leading 31 bits of (p, q) with cofactors of at most 15 bits, every quantity one
// register and one instruction:
//   ph = p >> k, qh = q >> k;  the true P / 2^k lies in (ph - E, ph + E), same for Q, where E covers the truncation by
//   2^k and the outer cofactors (|P - p| < max(A, B, C, D) when the window was cut from longer numbers);
//   inner pair pi = ai ph - bi qh, qi = di qh - ci ph; true values within E (ai + bi) =: u resp. E (ci + di) =: v;
//   x-step with t <= (pi - u) / (qi + v): non-negative for every value in range; then u += t v (y-step mirrored).
// A phase ends when no lane can take a step; its matrix is folded into (A, B, C, D) and applied to (p, q) exactly;
// (ai + bi) <= room keeps the folded cofactors below 2^31.  Lanes that have stopped run on with t forced to 0, so
// their state does not move and no snapshots are needed.  Quotients are conservative like the one-level batch's:
// never above the true one, so the sequence may differ from lehmer_batch's by delayed steps -- the reduced form at
// the end of a composition is unique, and that is what parity is about.
CF_DEV bool lehmer_batch2(uint64_t xh, uint64_t yh, bool exact, uint64_t thr, uint32_t &A_, uint32_t &B_, uint32_t &C_,
                          uint32_t &D_) {
    uint64_t p = xh, q = yh;
    uint32_t A = 1, B = 0, C = 0, D = 1;
    const float MARGIN = 0.99999905f;
    bool oalive = true;
    for (int phase = 0; phase < 6; phase++) {
        if (!CF_WAVE_ANY(oalive)) break;
        // ---- set-up of the phase
        const uint32_t mab = A > B ? A : B, mcd = C > D ? C : D, M = mab > mcd ? mab : mcd;
        const uint64_t pq = p | q;
        const int bl = pq ? 64 - __builtin_clzll(pq) : 0;
        const int k = bl > 31 ? bl - 31 : 0;
        const uint32_t ph = (uint32_t)(p >> k), qh = (uint32_t)(q >> k);
        const uint32_t E = exact ? (k ? 1u : 0u) : (uint32_t)((uint64_t)M >> k) + 2u;
        // room for (ai + bi), (ci + di): folded cofactors < 2^31, inner cofactors <= 15 bits, E * room < 2^29
        uint32_t room = f32_to_u32_sat(2147483648.0f * (fast_rcp((float)M) * MARGIN));
        const uint32_t room_e = f32_to_u32_sat(536870912.0f * (fast_rcp((float)E) * MARGIN));      // E == 0: saturates
        room = room < 32767u ? room : 32767u;
        room = room < room_e ? room : room_e;
        const uint64_t thr_k = thr >> k;
        const uint32_t thr_i = thr_k > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)thr_k;
        // ---- the phase: single-register state
        uint32_t pi = ph, qi = qh, ai = 1, bi = 0, ci = 0, di = 1, si = 1, ti = 1, u = E, v = E;
        bool alive = oalive & (room >= 2u), any_prev = true;
        for (int it = 0; it < 16; it++) {
            if (!any_prev) break;
            uint32_t tmx, tmy;
            {   // x -= t y
                const float tf = (float)(int32_t)(pi - u) * (fast_rcp((float)(qi + v)) * MARGIN);     // pi < u: negative -> t = 0
                const uint32_t t = f32_to_u32_sat(tf);
                const uint64_t sn = si + (uint64_t)t * ti;
                alive = alive & (sn <= room);
                tmx = alive ? t : 0u;
                pi -= tmx * qi;
                ai += tmx * ci; bi += tmx * di; si += tmx * ti;
                u = E * si;
                alive = alive & !(pi < thr_i);
            }
            {   // y -= t x
                const float tf = (float)(int32_t)(qi - v) * (fast_rcp((float)(pi + u)) * MARGIN);
                const uint32_t t = f32_to_u32_sat(tf);
                const uint64_t sn = ti + (uint64_t)t * si;
                alive = alive & (sn <= room);
                tmy = alive ? t : 0u;
                qi -= tmy * pi;
                ci += tmy * ai; di += tmy * bi; ti += tmy * si;
                v = E * ti;
                alive = alive & !(qi < thr_i);
            }
            alive = alive & ((tmx | tmy) != 0u);
            any_prev = CF_WAVE_ANY(alive);
        }
        // ---- fold the phase into the batch (a lane that did not move folds the identity)
        const bool moved = (bi | ci) != 0u;
        const uint32_t nA = ai * A + bi * C, nB = ai * B + bi * D, nC = di * C + ci * A, nD = di * D + ci * B;
        const uint64_t np = (uint64_t)ai * p - (uint64_t)bi * q, nq = (uint64_t)di * q - (uint64_t)ci * p;
        A = nA; B = nB; C = nC; D = nD;
        p = np; q = nq;
        oalive = oalive & moved & !(p < thr) & !(q < thr);
    }
    A_ = A; B_ = B; C_ = C; D_ = D;
    return (B | C) != 0;
}


// WIDE: the windows are not cut from the numbers but derived from a 128-bit window and the matrix of a previous
// batch (euclid_serve, second batch of a round): the true values lie in (xh - 1, xh + 2) and (yh - 1, yh + 2) instead
// of [xh, xh + 1), and the validity test becomes  p - t q >= (a + t c) + 2 (b + t d)  (y-step: q' >= nd + 2 nc).
CF_DEV bool lehmer_batch_wide(uint64_t xh, uint64_t yh, bool exact, uint64_t thr, uint32_t &A, uint32_t &B,
                         uint32_t &C, uint32_t &D) {
    // Quotient first, validity second: t = floor(p / q) biased low by 2^-20 (never above the true quotient of the
    // windows), then the step is kept iff it is non-negative for every value the truncated operands can stand for,
    //   P - t Q >= 0  for  P > p - b, Q < q + d   <=>   p - t q >= b + t d   (the new remainder >= the new cofactor),
    // which is one 64-bit compare on values the step computes anyway -- the reference loop biases the quotient itself,
    // t <= (p - b) / (q + d), at a two-word subtraction, a conversion and an addition more per half-step.  A step
    // whose full quotient is not provably safe ends the batch (the reference would take a smaller one and go on):
    // 0.6 % fewer cofactor bits per batch (tests/test_hostsim_device_code.py), 25 % fewer instructions.
    // The cofactor columns are continuants: a <= b and c <= d after the first step, so the 31-bit bound is tested
    // on the larger one only.
    uint64_t p = xh, q = yh;
    uint32_t a = 1, b = 0, cc = 0, d = 1;           // working state: runs on, meaningless once the lane has stopped
    uint32_t ra = 1, rb = 0, rc = 0, rd = 1;       // state after the last valid half-step
    (void)exact;
    const float MARGIN = 0.99999905f;
    bool alive = true, any_prev = true;
    for (int it = 0; it < COFHE_LEHMER_CAP; it++) {
        if (!any_prev) break;
        {   // x -= t*y
            const float tf = u64_to_float(p) * (fast_rcp(u64_to_float(q)) * MARGIN);
            const uint32_t t = f32_to_u32_sat(tf);          // q == 0: saturates (or NaN -> 0); both fail below
            const uint64_t nb = b + (uint64_t)t * d;
            a += t * cc;
            p -= (uint64_t)t * (uint32_t)q + (((uint64_t)(t * (uint32_t)(q >> 32))) << 32);
            b = (uint32_t)nb;
            alive = alive & (t != 0u) & (nb < 0x80000000ull) & (p >= 2 * nb + a);
            ra = alive ? a : ra; rb = alive ? b : rb;
            alive = alive & !(p < thr);
        }
        {   // y -= t*x
            const float tf = u64_to_float(q) * (fast_rcp(u64_to_float(p)) * MARGIN);
            const uint32_t t = f32_to_u32_sat(tf);
            const uint64_t nd = d + (uint64_t)t * b;
            cc += t * a;
            q -= (uint64_t)t * (uint32_t)p + (((uint64_t)(t * (uint32_t)(p >> 32))) << 32);
            d = (uint32_t)nd;
            alive = alive & (t != 0u) & (nd < 0x80000000ull) & (q >= 2 * (uint64_t)cc + nd);
            rd = alive ? d : rd; rc = alive ? cc : rc;
            alive = alive & !(q < thr);
        }
        any_prev = CF_WAVE_ANY(alive);
    }
    A = ra; B = rb; C = rc; D = rd;
    return (rb | rc) != 0;
}


// The second batch of a round (what euclid_serve did under COFHE_BATCHES_PER_ROUND == 2, after its first batch (A, B, C, D) on
// the windows (xh, yh) cut at bit sh of the pair stashed at xs / ys): w[0] bit 31 set = present, w = (A2, B2, C2, D2).
CF_DEV void serve_second_batch(const uint32_t *xs, const uint32_t *ys, uint64_t xh, uint64_t yh, int sh, int stop_bits, uint32_t A, uint32_t B,
                               uint32_t C, uint32_t D, uint32_t (&w)[4]) {
    w[0] = 1u; w[1] = 0u; w[2] = 0u; w[3] = 1u;
    if (sh < 64) return;
    const int s2 = sh - 64, j0 = s2 >> 5, o2 = s2 & 31;
    const uint32_t u0 = xs[j0], u1 = xs[j0 + 1], u2 = xs[j0 + 2], v0 = ys[j0], v1 = ys[j0 + 1], v2 = ys[j0 + 2];
    const uint64_t ul = ((uint64_t)u1 << 32) | u0, vl = ((uint64_t)v1 << 32) | v0;
    const uint64_t xlo = o2 ? ((ul >> o2) | ((uint64_t)u2 << (64 - o2))) : ul;
    const uint64_t ylo = o2 ? ((vl >> o2) | ((uint64_t)v2 << (64 - o2))) : vl;
    typedef unsigned __int128 u128;
    const u128 xw = ((u128)xh << 64) | xlo, yw = ((u128)yh << 64) | ylo;
    const u128 pw = (u128)A * xw - (u128)B * yw, qw = (u128)D * yw - (u128)C * xw;     // mod 2^128: the true values are in range
    const uint64_t ph_ = (uint64_t)(pw >> 64), qh_ = (uint64_t)(qw >> 64);
    const uint64_t top = ph_ | qh_;
    if ((top >> 63) != 0 || top < (1ull << 31)) return;          // both images non-negative and long enough that t >= 31
    const int t = 64 - __builtin_clzll(top);
    const uint64_t p2 = (uint64_t)(pw >> t), q2 = (uint64_t)(qw >> t);
    const int base = s2 + t;
    const int pb = p2 ? 64 - __builtin_clzll(p2) : 0, qb = q2 ? 64 - __builtin_clzll(q2) : 0;
    const int hi2 = pb > qb ? pb : qb, lo2 = pb > qb ? qb : pb;
    if (!(lo2 > 34 && hi2 - lo2 < 31 && (stop_bits < 0 || base + lo2 > stop_bits + 2))) return;
    uint64_t thr2 = 0;
    if (stop_bits >= 0) {
        const int tb = stop_bits - base;
        thr2 = tb <= 0 ? 0 : (tb >= 64 ? ~0ull : (1ull << tb));
    }
    const bool sw = p2 < q2;
    uint32_t a, b, cc, d;
    if (lehmer_batch_wide(sw ? q2 : p2, sw ? p2 : q2, false, thr2, a, b, cc, d)) {
        w[0] = (sw ? d : a) | 0x80000000u; w[1] = sw ? cc : b; w[2] = sw ? b : cc; w[3] = sw ? a : d;
    }
}

// the integer batch for a pair whose order is unknown (what mp.hpp: lehmer_batch_unordered does for the product's batch)
CF_DEV bool lehmer_batch_u64_unordered(uint64_t xh, uint64_t yh, bool exact, uint64_t thr, uint32_t &A, uint32_t &B, uint32_t &C, uint32_t &D) {
    const bool sw = xh < yh;
    uint32_t a, b, cc, d;
    const bool ok = lehmer_batch_u64(sw ? yh : xh, sw ? xh : yh, exact, thr, a, b, cc, d);
    A = sw ? d : a;
    B = sw ? cc : b;
    C = sw ? b : cc;
    D = sw ? a : d;
    return ok;
}

}  // namespace cofhe
