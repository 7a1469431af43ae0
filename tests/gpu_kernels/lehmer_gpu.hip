// lehmer_gpu.hip -- TEST INFRASTRUCTURE: the serving lane's batch (cofhe_amd/csrc/mp.hpp: lehmer_batch)
// run ON THE GPU, one window pair per lane as the serving wavefront runs it (divergent data, run-on lanes), and checked
// on the host in exact integer arithmetic.  The host simulator executes the same source with 1.0f / x and C++ conversions;
// the margin argument of the quotient (v_rcp_f32, v_cvt_u32_f32, the f32 image of a 53-bit window) holds for the
// instructions themselves only where they execute -- here.  Built by __graft_entry__.build() into
// tests/gpu_kernels/liblehmer_gpu.so; used by tests/test_gpu_parity.py::test_lehmer_batch_on_the_gpu.  Not part of the product.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../cofhe_amd/csrc/mp.hpp"

using namespace cofhe;

__global__ void k_lehmer(const uint64_t *__restrict__ xh, const uint64_t *__restrict__ yh, const uint32_t *__restrict__ flags,
                         const uint64_t *__restrict__ thr, uint64_t n, int full, uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t A, B, C, D;
    const bool exact = flags[i] & 1u;
    // full == 2: the single-chain form of the wide layout (mp.hpp: lehmer_batch_uniform<12>), no thresholds; here every lane has
    // windows of its own, so its scalar branches diverge -- the arithmetic and the tests are the same
    const bool ok = full == 2 ? lehmer_batch_uniform<12>(xh[i], yh[i], exact, 0.0, A, B, C, D)
                              : lehmer_batch(xh[i], yh[i], exact, full ? (uint64_t)0 : thr[i], A, B, C, D);
    out[5 * i + 0] = A;
    out[5 * i + 1] = B;
    out[5 * i + 2] = C;
    out[5 * i + 3] = D;
    out[5 * i + 4] = ok ? 1u : 0u;
}

namespace {
struct Sm64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint64_t bits(int b) { return b <= 0 ? 0 : next() >> (64 - b); }
};
typedef __int128 i128;
// A X - B Y >= 0 for X = x 2^64 + dx, Y = y 2^64 + dy  <=>  (A x - B y) + floor((A dx - B dy) / 2^64) >= 0
bool nonneg(uint32_t A, uint64_t x, uint64_t dx, uint32_t B, uint64_t y, uint64_t dy) {
    const i128 t1 = (i128)A * x - (i128)B * y;
    const i128 t2 = (i128)A * dx - (i128)B * dy;
    return t1 + (t2 >> 64) >= 0;
}
}  // namespace

// n window pairs of the families of tests/test_hostsim_device_code.py::_batch_cases at the product's window width (53 bits);
// full != 0: thresholds dropped (full sequences only).  stats: [0] batches that made a step, [1] sum of the largest cofactor's
// bit length over full-window inexact batches, [2] their count, [3] index of the first violation.  Returns the number of
// violations (unimodular, cofactors < 2^26, ok flag, both remainders >= 0 at the corners of the window intervals, and for a
// partial sequence: never more than one step past the threshold), or -1 when no GPU run was possible.
extern "C" long lehmer_gpu_check(uint64_t n, uint64_t seed, int full, uint64_t *stats) {
    std::vector<uint64_t> x(n), y(n), thr(n);
    std::vector<uint32_t> fl(n), kind(n);
    Sm64 r{seed};
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t k = (uint32_t)(i % 8);
        uint64_t a, b, t = 0;
        uint32_t ex = 0;
        const int small_x[6] = {53, 40, 33, 31, 8, 1}, small_y[6] = {52, 33, 30, 20, 3, 0};
        switch (k) {
            case 0: a = r.bits(small_x[r.next() % 6]); b = r.bits(small_y[r.next() % 6]); ex = 1; break;
            case 1: a = r.bits(53) | (1ull << 52); b = r.bits(51); t = full ? 0 : 1ull << (1 + r.next() % 51); break;
            case 2: a = r.bits(53) | (1ull << 52); b = a - r.next() % 3; break;
            case 3: { const int yb[3] = {23, 29, 39}; a = r.bits(53) | (1ull << 52); b = r.bits(yb[r.next() % 3]); break; }
            case 4: {
                const uint64_t mult[6] = {1, 2, 3, 1000, 65535, 1ull << 20};
                b = r.bits(29) | (1ull << 28);
                a = b * mult[r.next() % 6] + r.bits(19);
                if (a >> 53) a = (1ull << 53) - 1;
                break;
            }
            default: { const int yb[4] = {52, 51, 49, 44}; a = r.bits(53) | (1ull << 52); b = r.bits(53) | (1ull << yb[r.next() % 4]); ex = (r.next() % 4) == 3; b &= (1ull << 53) - 1; break; }
        }
        if (a < b) { const uint64_t s = a; a = b; b = s; }
        x[i] = a; y[i] = b; thr[i] = t; fl[i] = ex; kind[i] = k;
    }
    uint64_t *dx = nullptr, *dy = nullptr, *dt = nullptr;
    uint32_t *df = nullptr, *dout = nullptr;
    std::vector<uint32_t> out(5 * n);
    bool ok = hipMalloc((void **)&dx, n * 8) == hipSuccess && hipMalloc((void **)&dy, n * 8) == hipSuccess && hipMalloc((void **)&dt, n * 8) == hipSuccess &&
              hipMalloc((void **)&df, n * 4) == hipSuccess && hipMalloc((void **)&dout, n * 20) == hipSuccess;
    ok = ok && hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(dt, thr.data(), n * 8, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(df, fl.data(), n * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_lehmer, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dy, df, dt, n, full, dout);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(out.data(), dout, n * 20, hipMemcpyDeviceToHost) == hipSuccess;
    }
    hipFree(dx); hipFree(dy); hipFree(dt); hipFree(df); hipFree(dout);
    if (!ok) return -1;
    long bad = 0;
    stats[0] = stats[1] = stats[2] = 0;
    stats[3] = ~0ull;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t A = out[5 * i], B = out[5 * i + 1], C = out[5 * i + 2], D = out[5 * i + 3], okf = out[5 * i + 4];
        bool good = A < (1u << 26) && B < (1u << 26) && C < (1u << 26) && D < (1u << 26);
        good = good && (int64_t)A * D - (int64_t)B * C == 1;
        good = good && okf == ((B | C) ? 1u : 0u);
        const uint64_t top = ~0ull;
        if (fl[i] & 1u) {
            good = good && nonneg(A, x[i], 0, B, y[i], 0) && nonneg(D, y[i], 0, C, x[i], 0);
        } else {
            for (int cx = 0; cx < 2 && good; cx++)
                for (int cy = 0; cy < 2 && good; cy++)
                    good = nonneg(A, x[i], cx ? top : 0, B, y[i], cy ? top : 0) && nonneg(D, y[i], cy ? top : 0, C, x[i], cx ? top : 0);
        }
        if (!good) {
            if (stats[3] == ~0ull) stats[3] = i;
            bad++;
            continue;
        }
        stats[0] += okf;
        if (okf && thr[i] == 0 && !(fl[i] & 1u) && kind[i] >= 5) {
            uint32_t m = A > B ? A : B;
            m = C > m ? C : m;
            m = D > m ? D : m;
            stats[1] += 32 - __builtin_clz(m);
            stats[2]++;
        }
    }
    return bad;
}
