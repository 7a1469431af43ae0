// ctx.hpp -- what the translation units of libcofhe_hip.so share: the context object behind the
// opaque cofhe_hip_ctx handle, the per-thread error message and the HIP error check.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/cofhe_hip.h"
#include "layout.hpp"

struct cofhe_hip_ctx {
    int device;
    int dbits;
    int half_dbits;
    uint32_t *d_one;     // principal form record
    uint32_t *d_absdelta; // |Delta|, 80 words
    uint32_t *d_status = nullptr;   // [0] status word of the kernels (lane.hpp: CF_ST_*), [1] verdict of the last validation
    // per-call flag words of cofhe_hip_add_ciphertext_records ("the c1 of these tensors differ"), handed out round robin so
    // that calls in flight on different streams do not share one
    static constexpr uint32_t N_FLAGS = 256;
    uint32_t *d_flags = nullptr;
    uint32_t flag_next = 0;
    uint32_t *d_ftab = nullptr;     // f^(-2^j), j < ftab_k (2 records each), for decryption
    uint32_t ftab_k = 0;
    uint32_t ftab_f[cofhe::REC_WORDS];
    void *workspace = nullptr;      // grow-only scratch for the power tables of the matrix product
    size_t workspace_bytes = 0;
    // the workspace belongs to one call at a time on the HOST (mu) -- but the kernels of that call are still running when
    // it returns: the last user records ws_event on its stream and the next user's stream waits for it (WsUse)
    hipEvent_t ws_event = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_event_set = false;
    // serialises the entry points that use the workspace, the cached tables or the status area: a context may be
    // shared by the threads of a server (the reference's compute node calls one instance from 8 threads)
    uint32_t opt_wnaf_width = 0, opt_matmul_segments = 0;      // cofhe_hip_ctx_set_option; 0 = the launcher decides
    int opt_ladder_form = 0;                                    // 0: by the number of ladders, 1: wide pair, 2: solo, 3: throughput kernel, 4: wide, one wavefront
    int opt_matmul_tree = -1;                                   // -1: the launcher decides, 0: lockstep chains, 1: product tree
    // "profile_kernels": the matrix product brackets each of its kernels with HIP events on the launch stream
    // (cofhe_hip_profile_read sums them per kernel name): bench.py's roofline leg for the C3 workload
    bool opt_profile = false;
    struct ProfSpan {
        const char *name;
        hipEvent_t a, b;
    };
    std::vector<ProfSpan> prof;
    std::recursive_mutex mu;
    // fixed-base tables base^(2^j) (h of the cryptosystem, public keys): built on first use, a few kept per context
    struct FixedBase {
        uint32_t base[cofhe::REC_WORDS];
        uint32_t *d_table = nullptr;
        uint32_t len = 0;
        uint64_t stamp = 0;
    } fb[4];
    uint64_t fb_clock = 0;
    // block cache behind cofhe_hip_malloc / cofhe_hip_free (hipFree synchronises the device: a chain of tensor
    // operations that allocates its result and drops its operand paid ~0.6 ms per operation for that)
    struct Pooled {
        void *p;
        hipEvent_t ev;
    };
    std::unordered_map<void *, size_t> live;
    std::multimap<size_t, Pooled> pool;
    size_t pooled_bytes = 0;
    size_t pool_cap = (size_t)64 << 30;
};

namespace cofhe {
inline thread_local std::string g_err;
inline int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
}  // namespace cofhe

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return cofhe::fail(COFHE_HIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
