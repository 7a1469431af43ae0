// lehmer_bench.hip -- latency of one Lehmer batch on the serving wavefront's terms: ONE wavefront, 32 lanes with
// different windows, back-to-back batches; one-level (lehmer_batch) vs two-level (lehmer_batch2).  Tuning tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../cofhe_amd/csrc/form_io.hpp"
#include "../experiments/lehmer_variants/lehmer_variants.hpp"
using namespace cofhe;

template <int WHICH>
__global__ void __launch_bounds__(64, 1) k_bench(const uint64_t *in, uint32_t *out, int iters, unsigned long long *cyc) {
    const int l = threadIdx.x;
    uint64_t xh = in[2 * l] | (1ull << 63), yh = in[2 * l + 1] >> 1;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        uint32_t A, B, C, D;
        if (l < 32) {
            if (WHICH == 0) lehmer_batch(xh, yh, false, (uint64_t)0, A, B, C, D);
            else lehmer_batch2(xh, yh, false, 0, A, B, C, D);
            acc += A ^ B ^ C ^ D;
            // next windows depend on the result (xorshift keeps them random)
            xh ^= (xh << 13) ^ acc; xh ^= xh >> 7; xh ^= xh << 17; xh |= 1ull << 63;
            yh ^= (yh << 13) ^ (acc * 2654435761u); yh ^= yh >> 7; yh ^= yh << 17; yh = (yh >> 1) | (1ull << 61);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[l] = acc;
    if (l == 0) cyc[0] = t1 - t0;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    std::mt19937_64 rng(5);
    std::vector<uint64_t> h(128);
    for (auto &v : h) v = rng();
    uint64_t *din; uint32_t *dout; unsigned long long *dc;
    hipMalloc(&din, 1024); hipMalloc(&dout, 256); hipMalloc(&dc, 8);
    hipMemcpy(din, h.data(), 1024, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 3; rep++) {
            if (which == 0) hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(64), 0, 0, din, dout, iters, dc);
            else hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(64), 0, 0, din, dout, iters, dc);
            hipDeviceSynchronize();
            unsigned long long c; uint32_t o[64];
            hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost);
            printf("%s: %.1f ns per batch (s_memrealtime at 100 MHz: %llu ticks / %d), check %08x\n", which ? "two-level" : "one-level",
                   c * 10.0 / iters, c, iters, o[0] ^ o[31]);
        }
    }
    return 0;
}
