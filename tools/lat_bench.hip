// lat_bench.hip -- latency of DEPENDENT instruction chains on a lone wavefront (what the wide layout's batch is made of)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int K>
__global__ void __launch_bounds__(64) k_lat(unsigned long long *out, double dseed, float fseed) {
    double d = dseed, e = dseed * 0.5;
    float f = fseed;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; i++) {
        if (K == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(e));) }
        if (K == 1) { REP16(asm volatile("v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1" : "+v"(d), "+v"(f));) }
        if (K == 2) { REP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(f));) }
        if (K == 3) { REP16(asm volatile("v_trunc_f32 %0, %0" : "+v"(f));) }
        if (K == 4) { REP16(asm volatile("v_fmaak_f32 %0, %0, %0, 0xb8800000" : "+v"(f));) }
        if (K == 5) { REP16(asm volatile("v_max_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
        if (K == 6) { REP16(asm volatile("v_cmp_ge_f64 vcc, %0, %1\n s_and_b64 vcc, exec, vcc\n s_cbranch_vccz 0" : : "v"(d), "v"(e) : "vcc");) }
        if (K == 7) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_fmaak_f32 %0, %0, %0, 0xb8800000" : "+v"(f));) }
        if (K == 8) { REP16(asm volatile("v_add_f32 %0, %0, %0" : "+v"(f));) }
        if (K == 9) { REP16(asm volatile("v_cvt_f32_f64 %1, %0\n v_rcp_f32 %1, %1\n v_fmaak_f32 %1, %1, %1, 0xb8800000\n v_trunc_f32 %1, %1\n v_cvt_f64_f32 %0, %1\n v_fma_f64 %0, %0, %2, %2" : "+v"(d), "+v"(f) : "v"(e));) }
        if (K == 10) { REP16(asm volatile("v_cmp_ge_f64 s[20:21], %0, %1\n s_and_b64 s[20:21], s[20:21], s[20:21]" : : "v"(d), "v"(e) : "s20", "s21");) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(d + f); }
}
int main() {
    unsigned long long *dout, r[2];
    hipMalloc(&dout, 16);
    const char *names[11] = {"v_fma_f64", "v_cvt_f32_f64 + v_cvt_f64_f32 (pair)", "v_rcp_f32", "v_trunc_f32", "v_fmaak_f32", "v_max_f64",
                             "v_cmp_f64 -> s_and vcc -> s_cbranch (triple)", "v_rcp_f32 + v_fmaak_f32 (pair)", "v_add_f32",
                             "half-step chain: cvt, rcp, fmaak, trunc, cvt, fma (six)", "v_cmp_f64 sgpr -> s_and (pair)"};
#define RUN(K) for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_lat<K>, dim3(1), dim3(64), 0, 0, dout, 1.000001, 1.5f); hipDeviceSynchronize(); } \
    hipMemcpy(r, dout, 16, hipMemcpyDeviceToHost); printf("%-62s %.2f cycles per group\n", names[K], (double)r[0] / 1024.0);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
    return 0;
}
