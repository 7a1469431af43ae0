// compose2.hip -- k_compose_pair: the latency-oriented composition kernel of add_ciphertext_tensors
// (pair.hpp / qf2.hpp: two lanes per form, 32 compositions per wavefront, one wavefront per SIMD, no
// workgroup cooperation).  Elements that leave its fast path (non-generic sizes, a gcd beyond a word, ...)
// are appended to a list that k_compose_wg_list (cofhe_hip.hip, the general 8-lane code) works off.
#include <hip/hip_runtime.h>

#include "qf2.hpp"

namespace cofhe_k {
using namespace cofhe2;

template <int N>
__global__ void __launch_bounds__(256, 1) k_compose_pair(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                          uint32_t *__restrict__ out, uint64_t n,
                                                          const uint32_t *__restrict__ absdelta, int half_dbits,
                                                          uint32_t *__restrict__ fb_count, uint32_t *__restrict__ fb_list) {
    PCtx c;
    c.hi = (int)(threadIdx.x & 1u);
    const uint64_t g0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 1;
    const bool active = g0 < n;
    const uint64_t g = active ? g0 : n - 1;
    const QDisc2 dd{absdelta, half_dbits};
    QForm2<N> x, y, r;
    bool ok = true;
    qf2_load(c, x, a + g * cofhe::REC_WORDS, ok);
    qf2_load(c, y, b + g * cofhe::REC_WORDS, ok);
    qf2_compose(c, r, x, y, dd, active, ok);
    if (active) {
        if (ok) {
            qf2_store(c, r, out + g * cofhe::REC_WORDS);
        } else if (c.hi == 0) {
            const uint32_t idx = atomicAdd(fb_count, 1u);
            fb_list[idx] = (uint32_t)g;
        }
    }
}

template __global__ void k_compose_pair<17>(const uint32_t *, const uint32_t *, uint32_t *, uint64_t, const uint32_t *, int, uint32_t *, uint32_t *);
template __global__ void k_compose_pair<19>(const uint32_t *, const uint32_t *, uint32_t *, uint64_t, const uint32_t *, int, uint32_t *, uint32_t *);

// launch helper (called from cofhe_hip.hip): n_limbs 17 or 19
void launch_compose_pair(int n_limbs, const uint32_t *a, const uint32_t *b, uint32_t *out, uint64_t n, const uint32_t *absdelta, int half_dbits,
                         uint32_t *fb_count, uint32_t *fb_list, hipStream_t st) {
    const unsigned blocks = (unsigned)((n + 127) / 128);
    if (n_limbs == 17)
        hipLaunchKernelGGL(k_compose_pair<17>, dim3(blocks), dim3(256), 0, st, a, b, out, n, absdelta, half_dbits, fb_count, fb_list);
    else
        hipLaunchKernelGGL(k_compose_pair<19>, dim3(blocks), dim3(256), 0, st, a, b, out, n, absdelta, half_dbits, fb_count, fb_list);
}
}  // namespace cofhe_k
