// ctx.hpp -- what the translation units of libcofhe_hip.so share: the context object behind the
// opaque cofhe_hip_ctx handle, the per-thread error message and the HIP error check.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "../../include/cofhe_hip.h"
#include "layout.hpp"

struct cofhe_hip_ctx {
    int device;
    int dbits;
    int half_dbits;
    uint32_t *d_one;     // principal form record
    uint32_t *d_absdelta; // |Delta|, 80 words
    uint32_t *d_status = nullptr;   // [0] status word of the kernels (lane.hpp: CF_ST_*), [1] verdict of the last validation
    uint32_t *d_ftab = nullptr;     // f^(-2^j), j < ftab_k (2 records each), for decryption
    uint32_t ftab_k = 0;
    uint32_t ftab_f[cofhe::REC_WORDS];
    void *workspace = nullptr;      // grow-only scratch for the power tables of the matrix product
    size_t workspace_bytes = 0;
    // serialises the entry points that use the workspace, the cached tables or the status area: a context may be
    // shared by the threads of a server (the reference's compute node calls one instance from 8 threads)
    std::recursive_mutex mu;
    // fixed-base tables base^(2^j) (h of the cryptosystem, public keys): built on first use, a few kept per context
    struct FixedBase {
        uint32_t base[cofhe::REC_WORDS];
        uint32_t *d_table = nullptr;
        uint32_t len = 0;
        uint64_t stamp = 0;
    } fb[4];
    uint64_t fb_clock = 0;
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
