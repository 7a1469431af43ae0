// Host emulation of the wavefront-wide layout (cofhe_amd/csrc/wide.hpp, qfw.hpp): with COFHE_HOSTSIM the lane values are
// 64-element vectors and the cross-lane primitives loops, so the device source runs single-threaded here.  TEST
// INFRASTRUCTURE ONLY: not linked into, nor loadable by, the product library.
#define COFHE_HOSTSIM 1
#include <cstring>

#include "../../cofhe_amd/csrc/qfw.hpp"

using namespace cofhe;
using namespace cofhe::wide;

static WN ldw(const uint32_t *w) {          // 128 limbs
    WN x;
    for (int i = 0; i < WL; i++) {
        x.a.v[i] = w[2 * i];
        x.b.v[i] = w[2 * i + 1];
    }
    return x;
}
static void stw(const WN &x, uint32_t *w) {
    for (int i = 0; i < WL; i++) {
        w[2 * i] = x.a.v[i];
        w[2 * i + 1] = x.b.v[i];
    }
}

extern "C" {
unsigned simw_status(void) { return g_sim_status.exchange(0); }
void simw_mul(const uint32_t *x, const uint32_t *y, uint32_t *out, int count) {
    for (int i = 0; i < count; i++) stw(w_mul(ldw(x + 128 * i), ldw(y + 128 * i)), out + 128 * i);
}
// r = A x - B y and s = A x + B y (mod 2^4096)
void simw_lincomb(const uint32_t *x, const uint32_t *y, uint32_t A, uint32_t B, uint32_t *r, uint32_t *s, int count) {
    for (int i = 0; i < count; i++) {
        WN o;
        (void)w_lincomb_sub(o, A, ldw(x + 128 * i), B, ldw(y + 128 * i));
        stw(o, r + 128 * i);
        (void)w_lincomb_add(o, A, ldw(x + 128 * i), B, ldw(y + 128 * i));
        stw(o, s + 128 * i);
    }
}
void simw_shift(const uint32_t *x, int n, uint32_t *l, uint32_t *r, int *bits, int count) {
    for (int i = 0; i < count; i++) {
        const WN a = ldw(x + 128 * i);
        stw(w_shl(a, n), l + 128 * i);
        stw(w_shr(a, n), r + 128 * i);
        bits[i] = w_bitlen(a);
    }
}
int simw_cmp(const uint32_t *x, const uint32_t *y) { return w_cmp(ldw(x), ldw(y)); }
// rem = num mod den; returns 0 when the routine asked for the fallback
int simw_mod(const uint32_t *num, const uint32_t *den, uint32_t *rem, int count) {
    int all = 1;
    for (int i = 0; i < count; i++) {
        bool ok = true;
        stw(w_mod(ldw(num + 128 * i), ldw(den + 128 * i), ok), rem + 128 * i);
        if (!ok) all = 0;
    }
    return all;
}
int simw_divexact(const uint32_t *num, const uint32_t *den, const int *nq, uint32_t *quot, int count) {
    int all = 1;
    for (int i = 0; i < count; i++) {
        bool ok = true;
        stw(w_divexact(ldw(num + 128 * i), ldw(den + 128 * i), nq[i], ok), quot + 128 * i);
        if (!ok) all = 0;
    }
    return all;
}
// the remainder sequence (qfw.hpp: w_euclid) from (x, y) with cofactor column (0, 1): out = x | y | ux | uy (128 limbs each),
// sg = {sx, sy}; returns w_euclid's flag
int simw_euclid(const uint32_t *x, const uint32_t *y, int stop_bits, uint32_t *out, int *sg) {
    WEuclid e;
    e.x = ldw(x); e.y = ldw(y);
    e.ux = w_zero(); e.uy = w_word(1u);
    e.sx = -1; e.sy = 1;
    const bool ok = w_euclid(e, stop_bits);
    stw(e.x, out); stw(e.y, out + 128); stw(e.ux, out + 256); stw(e.uy, out + 384);
    sg[0] = e.sx; sg[1] = e.sy;
    return ok ? 1 : 0;
}
// out = f1 o f2 on form records (layout.hpp); flags[i] = 1 when the wide composition asked for the fallback (the record is
// then left untouched)
void simw_compose(const uint32_t *f1, const uint32_t *f2, uint32_t *out, int *flags, int count, int half_dbits, const uint32_t *absdelta) {
    const QDisc dd{absdelta, half_dbits};
    for (int i = 0; i < count; i++) {
        WForm a = wf_load(f1 + (size_t)REC_WORDS * i), b = wf_load(f2 + (size_t)REC_WORDS * i), r;
        const bool ok = wf_compose(r, a, b, dd);
        flags[i] = ok ? 0 : 1;
        if (ok) wf_store(r, out + (size_t)REC_WORDS * i);
    }
}
}
