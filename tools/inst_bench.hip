// inst_bench.hip -- issue cost of the VALU / DS instructions the multi-precision layer is built from,
// measured on the GPU itself (tuning tool; not part of the product).  Each kernel runs a block of
// REP copies of one instruction over 8 independent destination registers inside a loop and stamps
// s_memtime around it; reported: shader cycles per wave-instruction per SIMD with W waves resident
// on every SIMD (W = 1, 2, 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP 64      // instructions per loop body = 8 x 8

// body macros: X(i) emits ONE instruction writing register set i
#define KERN(name, DECL, BODY8, SINK)                                                       \
    __global__ void __launch_bounds__(256) name(unsigned long long *t, unsigned *sink, int iters, unsigned seed) {  \
        DECL                                                                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                     \
        for (int it = 0; it < iters; it++) {                                                    \
            BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8                                     \
        }                                                                                       \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                     \
        if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;          \
        SINK                                                                                    \
    }

#define DECL_U32                                                                             \
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5,  \
             a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;                                                \
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u + 12345u, y = x ^ 0x9E3779B9u;
#define SINK_U32 if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u) sink[0] = 1;

#define DECL_U64                                                                             \
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4,         \
                       a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;                     \
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u + 12345u, y = x ^ 0x9E3779B9u;
#define SINK_U64 if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678ull) sink[0] = 1;

#define DECL_F64                                                                             \
    double a0 = 1.0 + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17,         \
           a7 = a0 * 19;                                                                        \
    double x = 1.0000001 + seed * 1e-9, y = 1e-7;
#define SINK_F64 if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.125) sink[0] = 1;

#define DECL_F32                                                                             \
    float a0 = 1.0f + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17,         \
          a7 = a0 * 19;                                                                         \
    float x = 1.0000001f + seed * 1e-9f, y = 1e-7f;
#define SINK_F32 if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.125f) sink[0] = 1;

// ---- 32-bit integer
#define I_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(x));
#define I_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a##i) : "v"(x), "v"(y));
#define I_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a##i) : "v"(x) : "vcc");
#define I_ADDC(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##i) : "v"(x) : "vcc");
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(x));
#define I_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(x));
#define I_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##i) : "v"(x), "v"(y));
#define I_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(x));
#define I_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(x));
#define I_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(x) : "vcc");
#define I_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a##i) : "v"(x));
#define I_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(x));
#define I_DPPQ(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(x));
#define I_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(x));
#define I_LSHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a##i));
#define I_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a##i) : "v"(x));
#define I_AND_OR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(x), "v"(y));
#define I_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a##i), "v"(x) : "vcc");
#define I_FFBH(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a##i));
#define I_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a##i) : "v"(x), "v"(y));
#define I_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a##i) : "v"(x));
#define I_BPERM_NW(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a##i) : "v"(x));
#define I_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3\n\tv_add_u32 %0, s20, %0" : "+v"(a##i) : : "s20");
#define I_SNOP(i) asm volatile("s_nop 0");
// ---- 64-bit integer
#define I_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a##i) : "v"(x), "v"(y) : "vcc");
#define I_MAD64_SGPRCO(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a##i) : "v"(x), "v"(y) : "s20", "s21");
#define I_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a##i) : "v"(a7));
#define I_LSHR64(i) asm volatile("v_lshrrev_b64 %0, 5, %0" : "+v"(a##i));
// ---- f64 / f32
#define I_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(x), "v"(y));
#define I_MUL64F(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(x));
#define I_ADD64F(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(y));
#define I_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(x), "v"(y));
#define I_RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
#define I_CVTF32U(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##i));
#define I_CVTU32F(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a##i));

KERN(k_add, DECL_U32, R8(I_ADD), SINK_U32)
KERN(k_add3, DECL_U32, R8(I_ADD3), SINK_U32)
KERN(k_addco, DECL_U32, R8(I_ADDCO), SINK_U32)
KERN(k_addc, DECL_U32, R8(I_ADDC), SINK_U32)
KERN(k_mullo, DECL_U32, R8(I_MULLO), SINK_U32)
KERN(k_mulhi, DECL_U32, R8(I_MULHI), SINK_U32)
KERN(k_mad24, DECL_U32, R8(I_MAD24), SINK_U32)
KERN(k_mul24, DECL_U32, R8(I_MUL24), SINK_U32)
KERN(k_mulhi24, DECL_U32, R8(I_MULHI24), SINK_U32)
KERN(k_cndmask, DECL_U32, R8(I_CNDMASK), SINK_U32)
KERN(k_mov, DECL_U32, R8(I_MOV), SINK_U32)
KERN(k_dpp, DECL_U32, R8(I_DPP), SINK_U32)
KERN(k_dppq, DECL_U32, R8(I_DPPQ), SINK_U32)
KERN(k_adddpp, DECL_U32, R8(I_ADDDPP), SINK_U32)
KERN(k_lshl, DECL_U32, R8(I_LSHL), SINK_U32)
KERN(k_alignbit, DECL_U32, R8(I_ALIGNBIT), SINK_U32)
KERN(k_and_or, DECL_U32, R8(I_AND_OR), SINK_U32)
KERN(k_cmp, DECL_U32, R8(I_CMP), SINK_U32)
KERN(k_ffbh, DECL_U32, R8(I_FFBH), SINK_U32)
KERN(k_dot4, DECL_U32, R8(I_DOT4), SINK_U32)
KERN(k_bperm_wait, DECL_U32, R8(I_BPERM), SINK_U32)
KERN(k_bperm, DECL_U32, R8(I_BPERM_NW), SINK_U32)
KERN(k_readlane_add, DECL_U32, R8(I_READLANE), SINK_U32)
KERN(k_snop, DECL_U32, R8(I_SNOP), SINK_U32)
KERN(k_mad64, DECL_U64, R8(I_MAD64), SINK_U64)
KERN(k_mad64_sco, DECL_U64, R8(I_MAD64_SGPRCO), SINK_U64)
KERN(k_lshladd64, DECL_U64, R8(I_LSHLADD64), SINK_U64)
KERN(k_lshr64, DECL_U64, R8(I_LSHR64), SINK_U64)
KERN(k_fma64, DECL_F64, R8(I_FMA64), SINK_F64)
KERN(k_mul64f, DECL_F64, R8(I_MUL64F), SINK_F64)
KERN(k_add64f, DECL_F64, R8(I_ADD64F), SINK_F64)
KERN(k_fma32, DECL_F32, R8(I_FMA32), SINK_F32)
KERN(k_rcp32, DECL_F32, R8(I_RCP32), SINK_F32)
KERN(k_cvtf32u, DECL_U32, R8(I_CVTF32U), SINK_U32)
KERN(k_cvtu32f, DECL_F32, R8(I_CVTU32F), SINK_F32)


// ---- single asm block variants (no compiler-inserted wait states between the instructions)
#define BLK8(name, TY, INIT, SINKC, ASM8, ...)                                                  \
    __global__ void __launch_bounds__(256) name(unsigned long long *t, unsigned *sink, int iters, unsigned seed) {  \
        TY a0 = INIT(0), a1 = INIT(1), a2 = INIT(2), a3 = INIT(3), a4 = INIT(4), a5 = INIT(5), a6 = INIT(6), a7 = INIT(7); \
        unsigned x = seed * 2654435761u + threadIdx.x * 40503u + 12345u, y = x ^ 0x9E3779B9u;  \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                     \
        for (int it = 0; it < iters; it++) {                                                    \
            asm volatile(ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8                                \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(x), "v"(y) : __VA_ARGS__);                                       \
        }                                                                                       \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                     \
        if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;          \
        if (SINKC) sink[0] = 1;                                                                 \
    }
#define INIT_U(i) (seed * (2 * i + 3) + threadIdx.x + i)
#define SINK_X ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u)
#define A8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define S_(x) #x
// cndmask with VCC set once in front (scalar write), e32 and e64 forms, and with an SGPR-pair mask
#define C_VCC(i) "v_cndmask_b32 %" S_(i) ", %" S_(i) ", %8, vcc\n\t"
#define C_SG(i) "v_cndmask_b32 %" S_(i) ", %" S_(i) ", %8, s[20:21]\n\t"
BLK8(b_cnd_vcc, unsigned, INIT_U, SINK_X, A8(C_VCC), "vcc")
BLK8(b_cnd_sgpr, unsigned, INIT_U, SINK_X, A8(C_SG), "s20", "s21")
// compare + select pairs (what the compiler emits for a ?: on lane data)
#define C_PAIR(i) "v_cmp_lt_u32 vcc, %" S_(i) ", %9\n\tv_cndmask_b32 %" S_(i) ", %" S_(i) ", %8, vcc\n\t"
BLK8(b_cmp_cnd, unsigned, INIT_U, SINK_X, A8(C_PAIR), "vcc")
#define C_PAIR_S(i) "v_cmp_lt_u32 s[20:21], %" S_(i) ", %9\n\tv_cndmask_b32 %" S_(i) ", %" S_(i) ", %8, s[20:21]\n\t"
BLK8(b_cmp_cnd_sgpr, unsigned, INIT_U, SINK_X, A8(C_PAIR_S), "s20", "s21")
// carry chains
#define C_ADDC(i) "v_addc_co_u32 %" S_(i) ", vcc, %" S_(i) ", %8, vcc\n\t"
BLK8(b_addc_chain, unsigned, INIT_U, SINK_X, A8(C_ADDC), "vcc")
#define C_ADDC_NOP(i) "v_addc_co_u32 %" S_(i) ", vcc, %" S_(i) ", %8, vcc\n\ts_nop 1\n\t"
BLK8(b_addc_chain_nop1, unsigned, INIT_U, SINK_X, A8(C_ADDC_NOP), "vcc")
#define C_ADD(i) "v_add_u32 %" S_(i) ", %" S_(i) ", %8\n\t"
BLK8(b_add, unsigned, INIT_U, SINK_X, A8(C_ADD), "vcc")
#define C_MULLO(i) "v_mul_lo_u32 %" S_(i) ", %" S_(i) ", %8\n\t"
BLK8(b_mullo, unsigned, INIT_U, SINK_X, A8(C_MULLO), "vcc")
#define C_XOR(i) "v_xor_b32 %" S_(i) ", %" S_(i) ", %8\n\t"
BLK8(b_xor, unsigned, INIT_U, SINK_X, A8(C_XOR), "vcc")
#define C_BFE(i) "v_bfe_u32 %" S_(i) ", %" S_(i) ", 3, 20\n\t"
BLK8(b_bfe, unsigned, INIT_U, SINK_X, A8(C_BFE), "vcc")
#define C_PERM(i) "v_perm_b32 %" S_(i) ", %" S_(i) ", %8, %9\n\t"
BLK8(b_perm, unsigned, INIT_U, SINK_X, A8(C_PERM), "vcc")
#define C_RFL(i) "v_readfirstlane_b32 s20, %" S_(i) "\n\t"
BLK8(b_readfirstlane, unsigned, INIT_U, SINK_X, A8(C_RFL), "s20")
#define C_SALU(i) "s_add_u32 s20, s20, 7\n\t"
BLK8(b_salu, unsigned, INIT_U, SINK_X, A8(C_SALU), "s20", "scc")
#define C_DPPROW(i) "v_mov_b32_dpp %" S_(i) ", %" S_(i) " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
BLK8(b_dpp_dep, unsigned, INIT_U, SINK_X, A8(C_DPPROW), "vcc")
// dependent chain on ONE register: latency
#define C_ADD_DEP(i) "v_add_u32 %0, %0, %8\n\t"
BLK8(b_add_dep1, unsigned, INIT_U, SINK_X, A8(C_ADD_DEP), "vcc")


// ---- the VCC-read anomaly: v_cndmask with the mask in VCC is slow unless VCC was just written by a VALU compare?
#define C1_CMP_V "v_cmp_lt_u32 vcc, %0, %9\n\t"
#define C7_VCC "v_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc\n\t"
BLK8(b_cmp_7cnd_vcc, unsigned, INIT_U, SINK_X, C1_CMP_V C7_VCC, "vcc")
#define C1_CMP_S "v_cmp_lt_u32 s[20:21], %0, %9\n\t"
#define C7_S "v_cndmask_b32 %1, %1, %8, s[20:21]\n\tv_cndmask_b32 %2, %2, %8, s[20:21]\n\tv_cndmask_b32 %3, %3, %8, s[20:21]\n\tv_cndmask_b32 %4, %4, %8, s[20:21]\n\tv_cndmask_b32 %5, %5, %8, s[20:21]\n\tv_cndmask_b32 %6, %6, %8, s[20:21]\n\tv_cndmask_b32 %7, %7, %8, s[20:21]\n\t"
BLK8(b_cmp_7cnd_sgpr, unsigned, INIT_U, SINK_X, C1_CMP_S C7_S, "s20", "s21")
#define C1_SMOV "s_mov_b64 vcc, s[20:21]\n\t"
BLK8(b_smov_7cnd_vcc, unsigned, INIT_U, SINK_X, C1_SMOV C7_VCC, "vcc")
#define C7_VCC64 "v_cndmask_b32_e64 %1, %1, %8, vcc\n\tv_cndmask_b32_e64 %2, %2, %8, vcc\n\tv_cndmask_b32_e64 %3, %3, %8, vcc\n\tv_cndmask_b32_e64 %4, %4, %8, vcc\n\tv_cndmask_b32_e64 %5, %5, %8, vcc\n\tv_cndmask_b32_e64 %6, %6, %8, vcc\n\tv_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
BLK8(b_cmp_7cnd_vcc_e64, unsigned, INIT_U, SINK_X, C1_CMP_V C7_VCC64, "vcc")
// the same mask read by v_addc (carry-in from vcc, not rewritten): v_addc with sgpr carry-out
#define C7_ADDCI "v_addc_co_u32 %1, s[22:23], %1, %8, vcc\n\tv_addc_co_u32 %2, s[22:23], %2, %8, vcc\n\tv_addc_co_u32 %3, s[22:23], %3, %8, vcc\n\tv_addc_co_u32 %4, s[22:23], %4, %8, vcc\n\tv_addc_co_u32 %5, s[22:23], %5, %8, vcc\n\tv_addc_co_u32 %6, s[22:23], %6, %8, vcc\n\tv_addc_co_u32 %7, s[22:23], %7, %8, vcc\n\t"
BLK8(b_cmp_7addc_vccin, unsigned, INIT_U, SINK_X, C1_CMP_V C7_ADDCI, "vcc", "s22", "s23")

// mixed: one v_mad_u64_u32 followed by N cheap VALU (does the multiplier overlap with other VALU work?)
#define I_MIX1(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a##i) : "v"(x), "v"(y) : "vcc"); \
                  asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
#define I_MIX3(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a##i) : "v"(x), "v"(y) : "vcc"); \
                  asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(x), "+v"(y));
KERN(k_mix_mad64_1add, DECL_U64, R8(I_MIX1), SINK_U64)
KERN(k_mix_mad64_3alu, DECL_U64, R8(I_MIX3), SINK_U64)

// LDS: write then read one dword per lane (private slot), waits included
__global__ void __launch_bounds__(256) k_lds_rw(unsigned long long *t, unsigned *sink, int iters, unsigned seed) {
    __shared__ unsigned lds[256 * 9];
    unsigned a0 = seed + threadIdx.x;
    unsigned *p = lds + threadIdx.x * 9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            p[r & 7] = a0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            a0 += p[(r + 1) & 7];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 == 0x12345678u) sink[0] = 1;
}

struct K { const char *name; void (*fn)(unsigned long long *, unsigned *, int, unsigned); int per_body; };

int main() {
    const int CUS = 256;
    unsigned long long *dt; unsigned *dsink;
    HIPCHK(hipMalloc(&dt, 4096 * 4 * 8)); HIPCHK(hipMalloc(&dsink, 64));
    K ks[] = {
        {"v_add_u32", k_add, REP}, {"v_add3_u32", k_add3, REP}, {"v_add_co_u32", k_addco, REP}, {"v_addc_co_u32 (chain)", k_addc, REP},
        {"v_mul_lo_u32", k_mullo, REP}, {"v_mul_hi_u32", k_mulhi, REP}, {"v_mad_u32_u24", k_mad24, REP}, {"v_mul_u32_u24", k_mul24, REP},
        {"v_mul_hi_u32_u24", k_mulhi24, REP}, {"v_cndmask_b32", k_cndmask, REP}, {"v_mov_b32", k_mov, REP}, {"v_mov_b32_dpp row_shr", k_dpp, REP},
        {"v_mov_b32_dpp quad_perm", k_dppq, REP}, {"v_add_u32_dpp", k_adddpp, REP}, {"v_lshlrev_b32", k_lshl, REP}, {"v_alignbit_b32", k_alignbit, REP},
        {"v_and_or_b32", k_and_or, REP}, {"v_cmp_lt_u32", k_cmp, REP}, {"v_ffbh_u32", k_ffbh, REP}, {"v_dot4_u32_u8", k_dot4, REP},
        {"ds_bpermute + wait", k_bperm_wait, REP}, {"ds_bpermute (dep chain)", k_bperm, REP}, {"v_readlane + v_add (2 instr)", k_readlane_add, REP},
        {"s_nop 0", k_snop, REP},
        {"v_mad_u64_u32", k_mad64, REP}, {"v_mad_u64_u32 (sgpr carry-out)", k_mad64_sco, REP}, {"v_lshl_add_u64", k_lshladd64, REP},
        {"v_lshrrev_b64", k_lshr64, REP},
        {"v_fma_f64", k_fma64, REP}, {"v_mul_f64", k_mul64f, REP}, {"v_add_f64", k_add64f, REP}, {"v_fma_f32", k_fma32, REP},
        {"v_rcp_f32", k_rcp32, REP}, {"v_cvt_f32_u32", k_cvtf32u, REP}, {"v_cvt_u32_f32", k_cvtu32f, REP},
        {"mad64 + 1 add (per pair)", k_mix_mad64_1add, REP}, {"mad64 + 3 alu (per quad)", k_mix_mad64_3alu, REP},
        {"[blk] v_cndmask vcc", b_cnd_vcc, REP}, {"[blk] v_cndmask s[20:21]", b_cnd_sgpr, REP}, {"[blk] v_cmp+v_cndmask vcc (pair)", b_cmp_cnd, REP},
        {"[blk] v_cmp+v_cndmask sgpr (pair)", b_cmp_cnd_sgpr, REP}, {"[blk] v_addc chain", b_addc_chain, REP}, {"[blk] v_addc + s_nop 1", b_addc_chain_nop1, REP},
        {"[blk] v_add_u32", b_add, REP}, {"[blk] v_mul_lo_u32", b_mullo, REP}, {"[blk] v_xor_b32", b_xor, REP}, {"[blk] v_bfe_u32", b_bfe, REP},
        {"[blk] v_perm_b32", b_perm, REP}, {"[blk] v_readfirstlane", b_readfirstlane, REP}, {"[blk] s_add_u32", b_salu, REP},
        {"[blk] dpp row_shr dep", b_dpp_dep, REP}, {"[blk] v_add_u32 one-reg dep chain", b_add_dep1, REP},
        {"[blk] v_cmp vcc + 7 cndmask vcc", b_cmp_7cnd_vcc, REP}, {"[blk] v_cmp sgpr + 7 cndmask sgpr", b_cmp_7cnd_sgpr, REP},
        {"[blk] s_mov vcc + 7 cndmask vcc", b_smov_7cnd_vcc, REP}, {"[blk] v_cmp vcc + 7 cndmask_e64 vcc", b_cmp_7cnd_vcc_e64, REP},
        {"[blk] v_cmp vcc + 7 addc(vcc in)", b_cmp_7addc_vccin, REP},
        {"lds write+read dword (pair)", k_lds_rw, 32},
    };
    printf("%-34s %10s %10s %10s   (shader cycles per wave-instruction per SIMD; W waves resident per SIMD)\n", "instruction", "W=1", "W=2", "W=4");
    const int iters = 200;
    for (auto &k : ks) {
        double res[3];
        int wi = 0;
        for (int W : {1, 2, 4}) {
            const int blocks = CUS * W;           // 256-thread blocks: one wave per SIMD each
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, dt, dsink, 4, 1u);
            HIPCHK(hipDeviceSynchronize());
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, dt, dsink, iters, 1u);
            HIPCHK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(blocks * 4);
            HIPCHK(hipMemcpy(h.data(), dt, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[h.size() / 2];
            // a wave saw `med` cycles for iters * per_body instructions while W waves shared the SIMD
            res[wi++] = med / ((double)iters * k.per_body) / W;
        }
        printf("%-34s %10.2f %10.2f %10.2f\n", k.name, res[0], res[1], res[2]);
    }
    // clock: s_memtime ticks vs wall time of a long launch
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_add, dim3(1024), dim3(256), 0, 0, dt, dsink, 20000, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(1024 * 4);
        HIPCHK(hipMemcpy(h.data(), dt, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("clock: median wave %llu memtime ticks in a %.3f ms launch -> %.1f MHz if ticks are shader cycles\n", h[h.size() / 2], ms,
               (double)h[h.size() / 2] / (ms * 1e3));
    }
    return 0;
}
