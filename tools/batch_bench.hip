// batch_bench.hip -- cycles of ONE Lehmer batch on a lone wavefront (the wide layout's critical path): the single-chain
// form (mp.hpp: lehmer_batch_uniform<12>) against the serving lane's (lehmer_batch, cap 8), random 53-bit windows.
#include <hip/hip_runtime.h>
#include "../cofhe_amd/csrc/mp.hpp"
#include <cstdio>
#include <vector>
using namespace cofhe;

template <int KIND>
__global__ void __launch_bounds__(64) k_bench(const uint64_t *xs, const uint64_t *ys, uint32_t n, unsigned long long *out) {
    uint32_t acc = 0, steps = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t x = __builtin_amdgcn_readfirstlane((uint32_t)xs[i]) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(xs[i] >> 32)) << 32);
        const uint64_t y = __builtin_amdgcn_readfirstlane((uint32_t)ys[i]) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(ys[i] >> 32)) << 32);
        uint32_t A, B, C, D;
        bool ok;
        if (KIND == 0) ok = lehmer_batch_uniform_unordered<12>(x, y, false, 0.0, A, B, C, D);
        else if (KIND == 1) ok = lehmer_batch_unordered(x, y, false, 0.0, A, B, C, D);
        else if (KIND == 3) ok = lehmer_batch_uniform_unordered<2>(x, y, false, 0.0, A, B, C, D);
        else if (KIND == 4) ok = lehmer_batch_uniform_unordered<4>(x, y, false, 0.0, A, B, C, D);
        else if (KIND == 5) ok = lehmer_batch_uniform_unordered<6>(x, y, false, 0.0, A, B, C, D);
        else { A = (uint32_t)x; B = (uint32_t)y; C = 1; D = 2; ok = true; }
        acc += A + B + C + D + (ok ? 1u : 0u);
        if (i < 8 && threadIdx.x == 0) { out[8 + 4 * i] = A; out[9 + 4 * i] = B; out[10 + 4 * i] = C; out[11 + 4 * i] = D; }
        steps += 32 - __builtin_clz(D | 1u);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; out[2] = steps; }
}

int main() {
    const uint32_t n = 4096;
    std::vector<uint64_t> xs(n), ys(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (uint32_t i = 0; i < n; i++) { xs[i] = (rnd() >> 11) | (1ull << 52); ys[i] = (rnd() >> 11) | (1ull << 51); }
    uint64_t *dx, *dy; unsigned long long *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dout, 64 * 8);
    hipMemcpy(dx, xs.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dy, ys.data(), n * 8, hipMemcpyHostToDevice);
    unsigned long long r[3], base = 0;
    for (int kind : {2, 0, 1, 3, 4, 5}) {
        for (int rep = 0; rep < 2; rep++) {
            if (kind == 0) hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(64), 0, 0, dx, dy, n, dout);
            if (kind == 1) hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(64), 0, 0, dx, dy, n, dout);
            if (kind == 3) hipLaunchKernelGGL(k_bench<3>, dim3(1), dim3(64), 0, 0, dx, dy, n, dout);
            if (kind == 4) hipLaunchKernelGGL(k_bench<4>, dim3(1), dim3(64), 0, 0, dx, dy, n, dout);
            if (kind == 5) hipLaunchKernelGGL(k_bench<5>, dim3(1), dim3(64), 0, 0, dx, dy, n, dout);
            if (kind == 2) hipLaunchKernelGGL(k_bench<2>, dim3(1), dim3(64), 0, 0, dx, dy, n, dout);
            hipDeviceSynchronize();
        }
        hipMemcpy(r, dout, 24, hipMemcpyDeviceToHost);
        if (kind == 2) base = r[0];
        if (kind != 2) {
            unsigned long long m[40];
            hipMemcpy(m, dout, 40 * 8, hipMemcpyDeviceToHost);
            for (int i = 0; i < 4; i++) printf("   x %llu y %llu -> %llu %llu %llu %llu\n", (unsigned long long)xs[i], (unsigned long long)ys[i], m[8 + 4 * i], m[9 + 4 * i], m[10 + 4 * i], m[11 + 4 * i]);
        }
        printf("%s: %.1f cycles per call (loop overhead %.1f), mean cofactor bits %.2f\n", kind == 0 ? "uniform<12>" : kind == 1 ? "lehmer_batch (cap 8)" : kind == 2 ? "empty loop" : kind == 3 ? "uniform<2>" : kind == 4 ? "uniform<4>" : "uniform<6>",
               (double)r[0] / n, (double)base / n, (double)r[2] / n);
    }
    return 0;
}
