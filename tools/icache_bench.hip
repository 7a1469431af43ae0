// icache_bench.hip -- does a single wavefront stream straight-line code at full rate?  (tuning tool)
// Kernels with 2 K / 16 K / 48 K instructions of independent v_add_u32 / v_xor per loop body
// (8 KB / 64 KB / 192 KB of code), one wave per SIMD and four: cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define I8 "v_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\t"
#define I64 I8 I8 I8 I8 I8 I8 I8 I8
#define I512 I64 I64 I64 I64 I64 I64 I64 I64
#define I2K I512 I512 I512 I512
#define BLK(S) asm volatile(S : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
#define KERN(name, BODY, NI)                                                                   \
    __global__ void __launch_bounds__(256) name(unsigned long long *t, unsigned *sink, int iters, unsigned seed) { \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, x = seed * 77 + 1; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
        for (int it = 0; it < iters; it++) { BODY }                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
        if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;         \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u) sink[0] = 1;              \
    }
KERN(k2k, BLK(I2K), 2048)
KERN(k16k, BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K), 16384)
KERN(k48k, BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K) BLK(I2K), 49152)
int main() {
    unsigned long long *dt; unsigned *ds;
    hipMalloc(&dt, 4096 * 4 * 8); hipMalloc(&ds, 64);
    struct K { const char *n; void (*f)(unsigned long long *, unsigned *, int, unsigned); int ni; } ks[] = {{"8 KB body", k2k, 2048}, {"64 KB body", k16k, 16384}, {"192 KB body", k48k, 49152}};
    for (int W : {1, 4})
        for (auto &k : ks) {
            const int blocks = 256 * W, iters = 196608 / k.ni;          // same instruction count for every body
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, dt, ds, 1, 1u);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, dt, ds, iters, 1u);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks * 4);
            hipMemcpy(h.data(), dt, h.size() * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("W=%d %-12s %6.2f cycles per instruction per wave (median), max wave %6.2f\n", W, k.n, (double)h[h.size() / 2] / (iters * (double)k.ni),
                   (double)h.back() / (iters * (double)k.ni));
        }
    return 0;
}
