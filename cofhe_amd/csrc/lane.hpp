// lane.hpp -- cross-lane primitives of an 8-lane "limb group" inside a 64-wide wavefront.
//
// One big integer is spread over G = 8 consecutive lanes of a wave64 (8 independent
// instances per wavefront).  All control flow in the arithmetic layers above is
// GROUP-UNIFORM: the 8 lanes of a group always take the same branches, different groups of
// one wave may diverge.  Every cross-lane operation below only ever reads lanes of the
// caller's own group, so it is well defined under that partial EXEC mask.
//
// gfx950 mapping:
//   shfl / bcast      -> ds_bpermute_b32 (LDS crossbar, no LDS memory)
//   shfl_up1/down1    -> DPP row_shr:1 / row_shl:1 (v_mov_b32_dpp), group edge patched
//   ballot8           -> v_cmp + s_mov of the 64-bit wave mask, shifted to the group
//   group scratch     -> a slice of LDS private to the group (multiplication operand
//                        staging, limb-granular shifts); ordering inside a wave is program
//                        order of the DS queue, made explicit with a wavefront fence
//
// COFHE_HOSTSIM (tests/hostsim only): the same primitives over 8 host threads and a spin
// barrier, so the arithmetic above can be unit-tested on a CPU-only machine.  That build is
// test infrastructure and is never linked into the product library.
#pragma once
#include <stdint.h>

#if defined(COFHE_HOSTSIM)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CF_DEV inline
#define CF_DEV_COLD inline
#define CF_UNROLL _Pragma("GCC unroll 16")
#define CF_NOUNROLL
#else
#include <hip/hip_runtime.h>
#define CF_DEV __device__ __forceinline__
// Rare routes of the arithmetic.  Inlined like everything else: as real (noinline) functions their operands have to live in
// memory -- passed by reference the composition's own values moved to scratch (30.2 M instead of 35.3 M ciphertext-ops/s at
// 128x128), passed as copies the hot path still lost 3 % (profiles/r03_b/variants_word_factor.txt).
#ifdef COFHE_COLD_CALLS
#define CF_DEV_COLD __device__ __attribute__((noinline, cold))
#else
#define CF_DEV_COLD __device__ __forceinline__
#endif
#define CF_UNROLL _Pragma("unroll")
#define CF_NOUNROLL _Pragma("nounroll")
#endif

// Branch weights for the register allocator and the block layout: the kernels are one ~60 k-instruction function each, and
// without weights every rarely taken route (a common factor, a long-division step, a carry that ripples, an add-back)
// counts as much as the route every composition takes when spill slots and live-range splits are placed.
#define CF_LIKELY(x) __builtin_expect(!!(x), 1)
#define CF_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace cofhe {

constexpr int G = 8;              // lanes per limb group
constexpr int CH = 40 / G;        // 32-bit limbs per lane per plane
constexpr int PLIMBS = G * CH;    // limbs per plane (40 limbs = 1280 bits)
#ifndef COFHE_WG_GROUPS
#define COFHE_WG_GROUPS 32
#endif
constexpr int WG_GROUPS = COFHE_WG_GROUPS;   // one request per lane of the serving wavefront (<= 64)
constexpr int WG_MAIL_WORDS = WG_GROUPS * 8 + WG_GROUPS * 4 + 4;     // replies (up to 8 words per group) | any-flag | stop bits per group
constexpr int SCRATCH_WORDS = 209;  // group scratch (LDS slice): 4 operand planes / 4x8 chunk tails; odd stride: the
                                    // serving lanes read one word of every slice at once (bank = 17 l + i mod 32)

#if defined(COFHE_HOSTSIM)
#define CF_PHASE(id) do { } while (0)
#define CF_PHASE_VAL(id, v) do { } while (0)
#define CF_FLAG(bits) do { } while (0)
#define CF_ST_EUCLID_CAP 1u
#define CF_ST_REDUCE_CAP 2u
#define CF_ST_DIV_CAP 4u
#define CF_ST_BAD_FORM 8u
inline std::atomic<unsigned> g_sim_status{0};      // host simulator: the status word
#define CF_STATUS(c, bits) do { if ((c).gl == 0) g_sim_status.fetch_or(bits); } while (0)

struct SpinBarrier {
    std::atomic<int> count{0};
    std::atomic<int> sense{0};
    int n = G;                      // participants: the lanes of a group, or the threads of a simulated workgroup
    bool yield = false;             // more threads than cores (workgroup simulation): give the core away while waiting
    void wait(int &local_sense) {
        local_sense ^= 1;
        if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            count.store(0, std::memory_order_relaxed);
            sense.store(local_sense, std::memory_order_release);
        } else {
            long spins = 0;
            while (sense.load(std::memory_order_acquire) != local_sense) {
                if (yield) std::this_thread::yield();
                if (++spins > 4000000000L) {
                    fprintf(stderr, "hostsim: barrier timeout (group-divergent control flow?)\n");
                    abort();
                }
            }
        }
    }
};

struct GroupShared {
    alignas(64) uint32_t xchg[G];
    alignas(64) uint32_t own_scratch[SCRATCH_WORDS];
    uint32_t *scratch = own_scratch;        // workgroup simulation: the group's slice of the workgroup's LDS image
    SpinBarrier bar;
};

// Simulated workgroup (tests/hostsim: run_workgroup): WG_GROUPS groups x G lanes as host threads with one LDS image,
// so that the workgroup-cooperative remainder sequence (mp.hpp: euclid_run_wg -- the form every kernel uses) runs on
// the CPU tier as well: same code, __syncthreads / the serving wavefront's ballot mapped to thread barriers.
struct WgShared {
    SpinBarrier bar;                // all threads of the workgroup (__syncthreads)
    SpinBarrier wave_bar;           // the threads of the serving wavefront (its ballot)
    uint32_t vote[64];
    int wave_threads = 0;
};

struct Ctx {
    int gl;              // lane index inside the group, 0..7
    GroupShared *gs;
    int sense = 0;
    // workgroup simulation (null / unused for the plain 8-thread group runs)
    WgShared *wg = nullptr;
    int wg_sense = 0, wave_sense = 0;
    int tid = 0;                    // threadIdx.x
    uint32_t *wg_mail = nullptr, *wg_scr0 = nullptr;
    int gi = 0, wave = 0, rank = -1;
    uint32_t *scratch() const { return gs->scratch; }
};

CF_DEV void group_sync(Ctx &c) { c.gs->bar.wait(c.sense); }

CF_DEV uint32_t shfl(Ctx &c, uint32_t v, int src) {
    c.gs->xchg[c.gl] = v;
    c.gs->bar.wait(c.sense);
    uint32_t r = c.gs->xchg[src & (G - 1)];
    c.gs->bar.wait(c.sense);
    return r;
}
CF_DEV uint32_t shfl_up1(Ctx &c, uint32_t v, uint32_t fill) {      // value of lane gl-1
    uint32_t r = shfl(c, v, (c.gl + G - 1) & (G - 1));
    return c.gl == 0 ? fill : r;
}
CF_DEV uint32_t shfl_down1(Ctx &c, uint32_t v, uint32_t fill) {    // value of lane gl+1
    uint32_t r = shfl(c, v, (c.gl + 1) & (G - 1));
    return c.gl == G - 1 ? fill : r;
}
CF_DEV uint32_t shfl_xor1(Ctx &c, uint32_t v) { return shfl(c, v, c.gl ^ 1); }
CF_DEV uint32_t shfl_xor2(Ctx &c, uint32_t v) { return shfl(c, v, c.gl ^ 2); }
CF_DEV uint32_t shfl_mirror(Ctx &c, uint32_t v) { return shfl(c, v, (G - 1) - c.gl); }   // lane i <- lane 7-i
CF_DEV uint32_t bcast_first(Ctx &c, uint32_t v) { return shfl(c, v, 0); }
CF_DEV uint32_t bcast_last(Ctx &c, uint32_t v) { return shfl(c, v, G - 1); }
CF_DEV uint32_t ballot8(Ctx &c, bool p) {
    uint32_t m = 0;
    c.gs->xchg[c.gl] = p ? 1u : 0u;
    c.gs->bar.wait(c.sense);
    for (int i = 0; i < G; i++) m |= c.gs->xchg[i] << i;
    c.gs->bar.wait(c.sense);
    return m;
}

#else  // ---------------------------------------------------------------- gfx950 device

struct Ctx {
    int gl;              // lane index inside the group, 0..7
    int base4;           // (first lane of the group) * 4: byte index for ds_bpermute
    uint32_t *scr;       // this group's LDS slice
    // workgroup-cooperative Euclid (k_compose only): mailbox in LDS, group / wave index in the WG
    uint32_t *wg_mail = nullptr;
    uint32_t *wg_scr0 = nullptr;   // LDS slice of group 0 of the workgroup (the serving wavefront reads every slice)
    int gi = 0, wave = 0;
    int rank = 0;          // arrival order of the workgroup on its CU (first grid wave), for issue-priority rotation
    uint32_t *status = nullptr;   // device status word of the context (CF_STATUS): set when a safety cap is hit
#ifdef COFHE_WG_TIMING
    unsigned long long t_wait = 0, t_apply = 0, n_rounds = 0, t_serve = 0;     // tools/wg_timing.hip: Euclid phase accounting
#endif
    __device__ uint32_t *scratch() const { return scr; }
};
// Device status bits (cofhe_hip_device_status): every data-dependent loop of the arithmetic has a trip-count cap;
// hitting one means the input was not what the path assumes (not a reduced form of the context's discriminant)
// and the result of that element is meaningless -- never a hang.
#define CF_ST_EUCLID_CAP 1u     /* remainder sequence did not end within its round cap */
#define CF_ST_REDUCE_CAP 2u     /* reduction did not end within its cap */
#define CF_ST_DIV_CAP 4u        /* division by zero / long division did not end */
#define CF_ST_BAD_FORM 8u       /* validation: not a reduced form of this discriminant */
#define CF_ST_SCHEDULE_CAP 16u  /* matrix product: a column's op list did not fit its slot (the product is truncated) */
#if !defined(COFHE_HOSTSIM)
#define CF_STATUS(c, bits) do { if ((c).status && (c).gl == 0) atomicOr((c).status, (bits)); } while (0)
#endif
// phase stamps of the diagnostic build tools/wg_timing.hip (thread 0 of every workgroup, 100 MHz wall clock)
#ifdef COFHE_WG_TIMING
#define CF_PHASE(id) do { if (threadIdx.x == 0) g_wg_phase[blockIdx.x * 16 + (id)] = wall_clock64(); } while (0)
#define CF_PHASE_VAL(id, v) do { if (threadIdx.x == 0) g_wg_phase[blockIdx.x * 16 + (id)] = (v); } while (0)
#define CF_FLAG(bits) do { if (c.gl == 0) atomicOr(&g_wg_flags[blockIdx.x], (bits)); } while (0)    /* which rare routes a workgroup took */
#else
#define CF_PHASE(id) do { } while (0)
#define CF_PHASE_VAL(id, v) do { } while (0)
#define CF_FLAG(bits) do { } while (0)
#endif

// LDS traffic between lanes of ONE wave: the DS queue is in order, the fence only stops the
// compiler from moving the reads above the writes.
CF_DEV void group_sync(Ctx &) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

CF_DEV uint32_t shfl(Ctx &c, uint32_t v, int src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(c.base4 + ((src & (G - 1)) << 2), (int)v);
}
CF_DEV uint32_t shfl_up1(Ctx &c, uint32_t v, uint32_t fill) {
    // row_shr:1 -- lane i reads lane i-1 of its 16-lane row; lane 0 of the group is patched
    uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x111, 0xF, 0xF, false);
    return c.gl == 0 ? fill : r;
}
CF_DEV uint32_t shfl_down1(Ctx &c, uint32_t v, uint32_t fill) {
    // row_shl:1 -- lane i reads lane i+1 of its row; last lane of the group is patched
    uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x101, 0xF, 0xF, false);
    return c.gl == G - 1 ? fill : r;
}
// DPP lane swaps inside a quad / a half row: no LDS crossbar traffic (ds_bpermute costs ~4x)
CF_DEV uint32_t shfl_xor1(Ctx &, uint32_t v) {      // quad_perm:[1,0,3,2]
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
CF_DEV uint32_t shfl_xor2(Ctx &, uint32_t v) {      // quad_perm:[2,3,0,1]
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
}
CF_DEV uint32_t shfl_mirror(Ctx &, uint32_t v) {    // lane i <- lane G-1-i of its group
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
}
// broadcast from the first / last lane of the group: two DPP moves (~6 issue cycles) instead of a
// ds_bpermute (21-28 cycles measured, tools/inst_bench.hip): a quad broadcast, then the half-row mirror written only
// into the other quad of the group (bank mask).
CF_DEV uint32_t bcast_first(Ctx &, uint32_t v) {
    const int q = __builtin_amdgcn_mov_dpp((int)v, 0x00, 0xF, 0xF, true);            // quad_perm:[0,0,0,0]
    return (uint32_t)__builtin_amdgcn_update_dpp(q, q, 0x141, 0xF, 0xA, false);      // lanes 4..7 <- lanes 3..0
}
CF_DEV uint32_t bcast_last(Ctx &, uint32_t v) {
    const int q = __builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xF, 0xF, true);            // quad_perm:[3,3,3,3]
    return (uint32_t)__builtin_amdgcn_update_dpp(q, q, 0x141, 0xF, 0x5, false);      // lanes 0..3 <- lanes 7..4
}
CF_DEV uint32_t ballot8(Ctx &c, bool p) {
    uint64_t m = __builtin_amdgcn_ballot_w64(p);
    return (uint32_t)(m >> (c.base4 >> 2)) & ((1u << G) - 1u);
}

#endif

CF_DEV uint32_t bcast(Ctx &c, uint32_t v, int src) { return shfl(c, v, src); }

// true when p holds in some lane of the caller's group -- on the GPU: in some active lane of the wavefront
// (a scalar branch instead of an exec-mask region; code guarded by it must be a no-op for lanes where p is false)
CF_DEV bool any_lane(Ctx &c, bool p) {
#if defined(COFHE_HOSTSIM)
    return ballot8(c, p) != 0;
#else
    (void)c;
    return __builtin_amdgcn_ballot_w64(p) != 0;
#endif
}

CF_DEV uint32_t group_max(Ctx &c, uint32_t v) {
    uint32_t o;
    o = shfl_xor1(c, v); v = o > v ? o : v;
    o = shfl_xor2(c, v); v = o > v ? o : v;      // every lane of a quad now holds the quad's max
    o = shfl_mirror(c, v); v = o > v ? o : v;        // the mirror pairs quad 0 with quad 1
    return v;
}

CF_DEV int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

}  // namespace cofhe
