// prim_bench.hip -- per-primitive timing of the device arithmetic on a real GPU (tuning tool;
// not part of the product).  Each kernel repeats one primitive `iters` times per limb group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "../cofhe_amd/csrc/form_io.hpp"
using namespace cofhe;
constexpr int BLOCK = 256;
constexpr int GPB = BLOCK / G;
__device__ __forceinline__ Ctx make_ctx(uint32_t *lds) {
    Ctx c; const int lane = (int)(threadIdx.x & 63);
    c.gl = lane & (G - 1); c.base4 = (lane & ~(G - 1)) << 2; c.scr = lds + (threadIdx.x / G) * SCRATCH_WORDS; return c;
}
template <int P> __device__ Mp<P> ldm(const Ctx &c, const uint32_t *w) {
    Mp<P> x; for (int p = 0; p < P; p++) for (int j = 0; j < CH; j++) x.v[p][j] = w[p * PLIMBS + c.gl * CH + j]; return x; }
template <int P> __device__ void stm(const Ctx &c, const Mp<P> &x, uint32_t *w) {
    for (int p = 0; p < P; p++) for (int j = 0; j < CH; j++) w[p * PLIMBS + c.gl * CH + j] = x.v[p][j]; }

#define KERNEL(name, ...) \
__global__ void __launch_bounds__(BLOCK, 4) name(const uint32_t *in, uint32_t *out, int iters) { \
    __shared__ uint32_t lds[GPB * SCRATCH_WORDS]; Ctx c = make_ctx(lds); \
    const size_t g = (size_t)blockIdx.x * GPB + threadIdx.x / G; \
    const uint32_t *my = in + g * 400; \
    Mp<1> a = ldm<1>(c, my), b = ldm<1>(c, my + 40); Mp<2> d = ldm<2>(c, my + 80), e = ldm<2>(c, my + 160); \
    uint32_t acc = 0; \
    for (int it = 0; it < iters; it++) { __VA_ARGS__ } \
    stm(c, a, out + g * 400); stm(c, b, out + g * 400 + 40); stm(c, d, out + g * 400 + 80); stm(c, e, out + g*400 + 160); \
    if (c.gl == 0) out[g * 400 + 399] = acc; }

KERNEL(k_empty, { acc += a.v[0][0]; a.v[0][0] ^= (uint32_t)it; })
KERNEL(k_cmp1, { acc += (uint32_t)mp_cmp(c, a, b); a.v[0][0] ^= acc; })
KERNEL(k_bitlen1, { acc += (uint32_t)mp_bitlen(c, a); a.v[0][1] ^= acc; })
KERNEL(k_bits64, { acc += (uint32_t)mp_bits64(c, a, 900 + (acc & 63)); })
KERNEL(k_lincomb1, { Mp<1> r; mp_lincomb_sub(c, r, 40000u, a, 3u, b); a = r; a.v[0][4] |= 0x80000000u; })
KERNEL(k_lincomb2, { Mp<2> r; mp_lincomb_sub(c, r, 40000u, d, 3u, e); d = r; d.v[1][4] |= 0x80000000u; })
KERNEL(k_add2, { Mp<2> r; acc += mp_add(c, r, d, e); d = r; })
KERNEL(k_add1, { Mp<1> r; acc += mp_add(c, r, a, b); a = r; a.v[0][4] &= 0x7FFFFFFFu; })
KERNEL(k_modword, { const WordDiv dm = worddiv_make(223092870u); acc += mp_mod_word(c, a, dm); a.v[0][0] ^= acc; })
KERNEL(k_shl1, { a = mp_shl(c, a, 7 + (it & 31)); a.v[0][0] |= 1; })
KERNEL(k_bits64pair, { uint64_t xh, yh; mp_bits64_pair(c, a, b, 900 + (acc & 63), xh, yh); acc += (uint32_t)xh + (uint32_t)(yh >> 32); })
KERNEL(k_bcast, { acc += bcast(c, a.v[0][4] + acc, G - 1); })
KERNEL(k_reduce, { Mp<1> ra = a, rc = b; SMp<1> rb; rb.m = mp_shr(c, a, 3); rb.neg = it & 1; ra.v[0][4] = 0; rc.v[0][4] = 0; if (c.gl >= 7) { mp_zero(ra); mp_zero(rc); mp_zero(rb.m);} rc.v[0][0] |= 1; qf_reduce<1>(c, ra, rb, rc); acc += ra.v[0][0] + rb.m.v[0][0]; a.v[0][0] += acc; })
KERNEL(k_mul11, { Mp<2> r = mp_mul(c, a, b); a = mp_resize<1>(r); a.v[0][0] |= 1; })
KERNEL(k_mul21, { Mp<3> r = mp_mul(c, d, a); d = mp_resize<2>(r); d.v[0][0] |= 1; })
KERNEL(k_shl2, { d = mp_shl(c, d, 37 + (it & 31)); d.v[0][0] |= 1; })
KERNEL(k_divrem21, { Mp<2> n = d, q; n.v[1][4] &= 0x0000FFFFu; mp_divrem(c, n, a, q); acc += q.v[0][0] + n.v[0][0]; d.v[0][0] += acc; })
KERNEL(k_lehmer, { uint32_t A, B, C, D; uint64_t xh = ((uint64_t)(a.v[0][1] | 0x80000000u) << 32) | a.v[0][0], yh = ((uint64_t)(b.v[0][1] & 0x7FFFFFFFu) << 32) | b.v[0][0];
                   lehmer_batch(xh, yh, false, (uint64_t)0, A, B, C, D); acc += A + B + C + D; a.v[0][0] += acc; b.v[0][0] ^= acc; })
KERNEL(k_xgcd, { Euclid<1> s; s.x = a; s.y = b; s.x.v[0][4] &= 0x3FFFFu; s.y.v[0][4] &= 0x1FFFFu; if (c.gl > 6) { } mp_zero(s.ux); mp_set_word(c, s.uy, 1); s.sx = -1; s.sy = 1;
                 CF_UNROLL for (int j = 0; j < CH; j++) { if (c.gl * CH + j > 32) { s.x.v[0][j] = 0; s.y.v[0][j] = 0; } }
                 euclid_run(c, s, -1); acc += s.ux.v[0][0]; a.v[0][0] += acc | 1; })
KERNEL(k_partial, { Euclid<1> s; s.x = a; s.y = b; mp_zero(s.ux); mp_set_word(c, s.uy, 1); s.sx = -1; s.sy = 1;
                 CF_UNROLL for (int j = 0; j < CH; j++) { if (c.gl * CH + j > 32) { s.x.v[0][j] = 0; s.y.v[0][j] = 0; } if (c.gl * CH + j == 32) { s.x.v[0][j] &= 0x3FFFF; s.y.v[0][j] &= 0x1FFFF; } }
                 euclid_run(c, s, 522); acc += s.ux.v[0][0]; a.v[0][0] += acc | 1; })

int main(int argc, char **argv) {
    const int blocks = 1024, groups = blocks * GPB;
    std::vector<uint32_t> h((size_t)groups * 400);
    std::mt19937 rng(1);
    for (auto &x : h) x = rng();
    for (int g = 0; g < groups; g++) {   // a, b: 1044-bit; d, e: 2088-bit
        uint32_t *m = &h[(size_t)g * 400];
        for (int i = 33; i < 40; i++) { m[i] = 0; m[40 + i] = 0; }
        m[32] &= 0xFFFFF; m[32] |= 0x80000; m[72] &= 0x7FFFF;
        for (int i = 66; i < 80; i++) { m[80 + i] = 0; m[160 + i] = 0; }
        m[80 + 65] &= 0xFF; m[80 + 65] |= 0x80; m[160 + 65] &= 0x7F;
    }
    uint32_t *din, *dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, h.size() * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct K { const char *name; void (*fn)(const uint32_t *, uint32_t *, int); int iters; };
    K ks[] = {{"empty", k_empty, 200}, {"add1", k_add1, 200}, {"qf_reduce (arbitrary b<a/8)", k_reduce, 20}, {"mod_word", k_modword, 100}, {"shl1", k_shl1, 100}, {"bits64_pair", k_bits64pair, 200}, {"bcast", k_bcast, 400}, {"cmp1", k_cmp1, 200}, {"bitlen1", k_bitlen1, 200}, {"bits64", k_bits64, 200}, {"lincomb1", k_lincomb1, 200},
              {"lincomb2", k_lincomb2, 200}, {"add2", k_add2, 200}, {"mul11", k_mul11, 50}, {"mul21", k_mul21, 50}, {"shl2", k_shl2, 100},
              {"divrem21(33 digits)", k_divrem21, 4}, {"lehmer_batch", k_lehmer, 100}, {"xgcd1044", k_xgcd, 2}, {"partial1044->522", k_partial, 2}};
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(BLOCK), 0, 0, din, dout, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(BLOCK), 0, 0, din, dout, k.iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // 4096 waves on 1024 SIMDs = 4 waves per SIMD; VALU-slot estimate = time*clk/(4 cycles)/4 waves
        double per_iter_us = ms * 1e3 / k.iters;
        printf("%-22s %8.3f ms  %9.3f us/iter  ~%8.0f issue-slots/iter/wave (at 1.7 GHz, 4 cyc/instr, 4 waves/SIMD)\n", k.name, ms, per_iter_us,
               per_iter_us * 1e-6 * 1.7e9 / 4 / 4);
    }
    return 0;
}
