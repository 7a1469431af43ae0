// op_bench.hip -- issue cost of the integer instructions the arithmetic is built from (tuning tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int OP> __global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x9e3779b9u;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP16(a0 = a0 + a1; a1 = a1 + a2; a2 = a2 + a3; a3 = a3 + a0;) }
        if (OP == 1) { REP16(a0 = a0 * a1; a1 = a1 * a2; a2 = a2 * a3; a3 = a3 * a0;) }
        if (OP == 2) { REP16(a0 = __umulhi(a0, a1); a1 = __umulhi(a1, a2); a2 = __umulhi(a2, a3); a3 = __umulhi(a3, a0);) }
        if (OP == 3) { REP16(q0 = (uint64_t)a0 * a1 + q0; q1 = (uint64_t)a1 * a2 + q1; q2 = (uint64_t)a2 * a3 + q2; q3 = (uint64_t)a3 * a0 + q3;) a0 += (uint32_t)q0; a1 += (uint32_t)(q1 >> 32); }
        if (OP == 4) { REP16(q0 = q0 + q1; q1 = q1 + q2; q2 = q2 + q3; q3 = q3 + q0;) }
        if (OP == 5) { REP16(f0 = f0 * f1 + f2; f1 = f1 * f2 + f3; f2 = f2 * f3 + f0; f3 = f3 * f0 + f1;) }
        if (OP == 6) { REP16(a0 = (a0 > a1) ? a2 : a3; a1 = (a1 > a2) ? a3 : a0; a2 = (a2 > a3) ? a0 : a1; a3 = (a3 > a0) ? a1 : a2;) }
        if (OP == 7) { REP16(a0 = __builtin_amdgcn_ds_bpermute((a1 & 63) << 2, a0); a1 = __builtin_amdgcn_ds_bpermute((a2 & 63) << 2, a1); a2 = __builtin_amdgcn_ds_bpermute((a3 & 63) << 2, a2); a3 = __builtin_amdgcn_ds_bpermute((a0 & 63) << 2, a3);) }
        if (OP == 8) { REP16(a0 = __builtin_amdgcn_update_dpp(a1, a0, 0x111, 0xF, 0xF, false); a1 = __builtin_amdgcn_update_dpp(a2, a1, 0x111, 0xF, 0xF, false); a2 = __builtin_amdgcn_update_dpp(a3, a2, 0x111, 0xF, 0xF, false); a3 = __builtin_amdgcn_update_dpp(a0, a3, 0x111, 0xF, 0xF, false);) }
        if (OP == 9) { REP16(a0 += (uint32_t)__builtin_amdgcn_ballot_w64(a1 > a2); a1 += (uint32_t)__builtin_amdgcn_ballot_w64(a2 > a3); a2 += (uint32_t)__builtin_amdgcn_ballot_w64(a3 > a0); a3 += (uint32_t)__builtin_amdgcn_ballot_w64(a0 > a1);) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(q0 + q1 + q2 + q3) + (uint32_t)(f0 + f1 + f2 + f3);
}
int main() {
    uint32_t *d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "add_u64 (2 instr)", "v_fma_f32", "cmp+cndmask (2 instr)", "ds_bpermute (+and,lshl)", "v_mov_dpp", "ballot (cmp+s_mov..)"};
    void (*fns[])(uint32_t *, int, uint32_t) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>};
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD)
        printf("== %d wave(s) per SIMD\n", waves_per_simd);
        for (int i = 0; i < 10; i++) {
            int iters = 2000;
            hipLaunchKernelGGL(fns[i], dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(fns[i], dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double ops = (double)iters * 64;    // source-level ops per wave
            printf("%-26s %7.3f ms  %6.2f ns per op per wave = %5.1f cycles at 2.1 GHz (per SIMD: /%d waves)\n", names[i], ms, ms * 1e6 / ops,
                   ms * 1e6 / ops * 2.1 / waves_per_simd, waves_per_simd);
        }
    }
    return 0;
}
