// pair_proto.hip -- feasibility prototype (tuning tool, not part of the product): the Euclid round of a
// lane-PAIR layout (one big integer spread over 2 lanes, 17 limbs per lane, one wave per SIMD):
// in-lane Lehmer batch on the leading 64 bits + the four linear combinations, timed per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#include "../cofhe_amd/csrc/mp.hpp"     // lehmer_batch_unordered (scalar code shared with the product)
using namespace cofhe;
constexpr int N = 17;

__device__ __forceinline__ uint32_t xl(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t flo(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xA0, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t fhi(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xF5, 0xF, 0xF, true); }

// r = A*x + B*(NOTY ? ~y : y) + cin over this lane's N limbs, flag-free (64-bit multiply-add chain); returns the word
// that leaves the lane
template <bool NOTY>
__device__ __forceinline__ uint32_t lincomb_lane(uint32_t (&r)[N], uint32_t A, const uint32_t (&x)[N], uint32_t B, const uint32_t (&y)[N],
                                                 uint32_t cin) {
    uint64_t c = cin;
#pragma unroll
    for (int j = 0; j < N; j++) {
        const uint64_t t = (uint64_t)A * x[j] + ((uint64_t)B * (NOTY ? ~y[j] : y[j]) + c);
        r[j] = (uint32_t)t;
        c = t >> 32;
    }
    return (uint32_t)c;
}
// adds the word handed over by the low lane into the high lane's limbs (ripple is rare beyond limb 1)
__device__ __forceinline__ void add_word(uint32_t (&r)[N], uint32_t w) {
    uint64_t t = (uint64_t)r[0] + w;
    r[0] = (uint32_t)t;
    uint32_t cy = (uint32_t)(t >> 32);
    t = (uint64_t)r[1] + cy;
    r[1] = (uint32_t)t;
    cy = (uint32_t)(t >> 32);
    if (__builtin_amdgcn_ballot_w64(cy != 0) != 0) {
#pragma unroll
        for (int j = 2; j < N; j++) {
            t = (uint64_t)r[j] + cy;
            r[j] = (uint32_t)t;
            cy = (uint32_t)(t >> 32);
        }
    }
}


// Lehmer batch in f64: 53-bit windows, cofactors < 2^26, every value an exactly represented integer.
// Conservative quotients t <= (p - b)/(q + d) keep the true remainders non-negative (same rule as lehmer_batch).
template <bool FASTRCP>
__device__ __forceinline__ double rcp64(double x) {
    if (FASTRCP) {
        const double r0 = (double)__builtin_amdgcn_rcpf((float)x);
        return r0 * (2.0 - x * r0);
    }
    return __builtin_amdgcn_rcp(x);
}
template <bool FASTRCP>
__device__ __forceinline__ bool lehmer53(double p, double q, bool exact, double thr, uint32_t &A, uint32_t &B, uint32_t &C, uint32_t &D) {
    double a = 1.0, b = 0.0, c = 0.0, d = 1.0;
    const double LIM = 67108864.0, MARGIN = 1.0 - 1.0 / 1099511627776.0;
    const double e = exact ? 0.0 : 1.0;
    for (int it = 0; it < 40; it++) {
        {
            const double num = p - b * e, den = q + d * e;
            const double t = __builtin_floor(num * rcp64<FASTRCP>(den) * MARGIN);
            const double na = __builtin_fma(t, c, a), nb = __builtin_fma(t, d, b);
            if (!((t >= 1.0) & (__builtin_fmax(na, nb) < LIM))) break;
            p = __builtin_fma(-t, q, p);
            a = na; b = nb;
            if (p < thr) break;
        }
        {
            const double num = q - c * e, den = p + a * e;
            const double t = __builtin_floor(num * rcp64<FASTRCP>(den) * MARGIN);
            const double nd = __builtin_fma(t, b, d), nc = __builtin_fma(t, a, c);
            if (!((t >= 1.0) & (__builtin_fmax(nd, nc) < LIM))) break;
            q = __builtin_fma(-t, p, q);
            d = nd; c = nc;
            if (q < thr) break;
        }
    }
    A = (uint32_t)a; B = (uint32_t)b; C = (uint32_t)c; D = (uint32_t)d;
    return (B | C) != 0;
}

template <int MODE>
__global__ void __launch_bounds__(256, 1) k_round(const uint32_t *in, uint32_t *out, unsigned long long *tm, int rounds) {
    const int lane = threadIdx.x & 63, hi = lane & 1;
    const size_t pair = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 1;
    const uint32_t *my = in + pair * 160;
    uint32_t x[N], y[N], u[N], v[N];
    for (int j = 0; j < N; j++) {
        x[j] = my[hi * N + j]; y[j] = my[40 + hi * N + j];
        u[j] = 0; v[j] = 0;
    }
    if (!hi) v[0] = 1;
    int T = 2 * N - 2;       // top limb index of the window (uniform)
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int rd = 0; rd < rounds; rd++) {
        // window: limbs T, T-1, T-2 of x and y (here: static top of the hi lane; the product version selects by T)
        const uint32_t x2 = fhi(x[N - 2]), x1 = fhi(x[N - 3]), x0 = fhi(x[N - 4]);
        const uint32_t y2 = fhi(y[N - 2]), y1 = fhi(y[N - 3]), y0 = fhi(y[N - 4]);
        const uint32_t top = x2 > y2 ? x2 : y2;
        const int sh = top ? __builtin_clz(top) : 0;
        const uint64_t xw = (((uint64_t)x2 << 32 | x1) << sh) | (sh ? (uint64_t)(x0 >> (32 - sh)) : 0);
        const uint64_t yw = (((uint64_t)y2 << 32 | y1) << sh) | (sh ? (uint64_t)(y0 >> (32 - sh)) : 0);
        uint32_t A = 1, B = 0, C = 0, D = 1;
        bool ok = true;
        if (MODE == 3 || MODE == 4) {
            const bool sw = xw < yw;
            const double pw = (double)((sw ? yw : xw) >> 11), qw = (double)((sw ? xw : yw) >> 11);
            uint32_t a_, b_, c_, d_;
            ok = MODE == 3 ? lehmer53<false>(pw, qw, false, 0.0, a_, b_, c_, d_) : lehmer53<true>(pw, qw, false, 0.0, a_, b_, c_, d_);
            A = sw ? d_ : a_; B = sw ? c_ : b_; C = sw ? b_ : c_; D = sw ? a_ : d_;
        } else if (MODE != 1) ok = lehmer_batch_unordered(xw, yw, false, 0, A, B, C, D);
        else { A = (uint32_t)(xw >> 34) | 1; B = (uint32_t)(yw >> 35); C = (uint32_t)(xw >> 36); D = (uint32_t)(yw >> 34) | 1; }
        if (!ok) { A = 1; B = 0; C = 0; D = 1; }
        acc += A ^ B ^ C ^ D;
        if (MODE < 2) {
            uint32_t nx[N], ny[N], nu[N], nv[N];
            uint32_t wx = lincomb_lane<true>(nx, A, x, B, y, hi ? 0u : B);
            uint32_t wy = lincomb_lane<true>(ny, D, y, C, x, hi ? 0u : C);
            uint32_t wu = lincomb_lane<false>(nu, A, u, B, v, 0u);
            uint32_t wv = lincomb_lane<false>(nv, D, v, C, u, 0u);
            wx = xl(wx); wy = xl(wy); wu = xl(wu); wv = xl(wv);
            add_word(nx, hi ? wx : 0u); add_word(ny, hi ? wy : 0u); add_word(nu, hi ? wu : 0u); add_word(nv, hi ? wv : 0u);
#pragma unroll
            for (int j = 0; j < N; j++) { x[j] = nx[j]; y[j] = ny[j]; u[j] = nu[j]; v[j] = nv[j]; }
            // keep the numbers from collapsing so that every round does the same work: re-seed the top limbs
            if (hi) { x[N - 2] |= 0x80000000u; y[N - 2] = (y[N - 2] | 0x40000000u) & 0x7FFFFFFFu; x[N - 1] = 0; y[N - 1] = 0; }
        } else {
            if (hi) { x[N - 3] += acc; y[N - 3] ^= acc; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) tm[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    uint32_t *o = out + pair * 160;
    for (int j = 0; j < N; j++) { o[hi * N + j] = x[j] ^ u[j]; o[40 + hi * N + j] = y[j] ^ v[j]; }
    if (lane == 0) o[159] = acc + T;
}

int main() {
    const int blocks_full = 256;
    std::vector<uint32_t> h((size_t)4096 * 128 * 160);
    std::mt19937 rng(1);
    for (auto &w : h) w = rng();
    uint32_t *din, *dout; unsigned long long *dt;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, h.size() * 4); hipMalloc(&dt, 4096 * 4 * 8);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    struct V { const char *name; void (*fn)(const uint32_t *, uint32_t *, unsigned long long *, int); };
    V vs[] = {{"full round (lehmer + 4 lincombs)", k_round<0>}, {"lincombs only", k_round<1>}, {"lehmer only", k_round<2>}, {"lehmer53 only (v_rcp_f64)", k_round<3>}, {"lehmer53 only (f32 rcp + newton)", k_round<4>}};
    for (int W : {1, 2}) {
        for (auto &v : vs) {
            const int blocks = blocks_full * W, rounds = 54;
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, din, dout, dt, 2);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, din, dout, dt, rounds);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> t(blocks * 4);
            hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost);
            std::sort(t.begin(), t.end());
            printf("W=%d %-36s %8.1f us for %d rounds (%d pairs): %7.0f cycles/round (median wave), %.2f us/round\n", W, v.name, ms * 1e3, rounds,
                   blocks * 128, (double)t[t.size() / 2] / rounds, ms * 1e3 / rounds);
        }
    }
    return 0;
}
