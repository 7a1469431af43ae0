// traffic_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the record access
// pattern of the composition kernel (MI355X_MICROARCH.md, "HBM": widths other than 16 B per lane
// are uncalibrated -- "calibrate on a known byte count in your own access pattern").
// k_copy2 reads two record arrays and writes one with the very qf_load / qf_store of the product
// (dword accesses, 5 consecutive words per lane in each 40-word plane, same grid geometry), and
// does nothing else: its algorithmic traffic is exactly 2 x N x 672 B read and N x 672 B written.
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); the ratio
// known / counted is the correction applied to the counters of k_compose_wg.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../cofhe_amd/csrc/form_io.hpp"

using namespace cofhe;

__global__ void __launch_bounds__(256) k_copy2(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                               uint32_t *__restrict__ out, uint64_t n) {
    __shared__ uint32_t lds[WG_GROUPS * SCRATCH_WORDS];
    Ctx c;
    const int lane = (int)(threadIdx.x & 63);
    c.gl = lane & (G - 1);
    c.base4 = (lane & ~(G - 1)) << 2;
    c.scr = lds + (threadIdx.x / G) * SCRATCH_WORDS;
    const uint64_t g = (uint64_t)blockIdx.x * WG_GROUPS + threadIdx.x / G;
    if (g >= n) return;
    QForm x, y;
    qf_load(c, x, a + g * REC_WORDS);
    qf_load(c, y, b + g * REC_WORDS);
    // keep both loads alive: one output word depends on y
    x.bneg ^= (int)(y.a.v[0][0] & y.bm.v[0][1] & y.c.v[0][2] & y.c.v[1][3] & 1u);
    CF_UNROLL for (int j = 0; j < CH; j++) {
        x.a.v[0][j] ^= y.a.v[0][j];
        x.bm.v[0][j] ^= y.bm.v[0][j];
        x.c.v[0][j] ^= y.c.v[0][j];
        x.c.v[1][j] ^= y.c.v[1][j];
    }
    qf_store(c, x, out + g * REC_WORDS);
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 32768;     // records (128x128 matadd = 32768)
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const size_t bytes = n * REC_WORDS * 4;
    uint32_t *a, *b, *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, bytes);
    std::vector<uint32_t> h(n * REC_WORDS);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
    hipMemcpy(a, h.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(b, h.data(), bytes, hipMemcpyHostToDevice);
    const unsigned blocks = (unsigned)((n + WG_GROUPS - 1) / WG_GROUPS);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_copy2, dim3(blocks), dim3(256), 0, 0, a, b, o, n);
    hipDeviceSynchronize();
    printf("{\"kernel\": \"k_copy2\", \"records\": %llu, \"read_bytes\": %llu, \"write_bytes\": %llu, \"launches\": %d}\n",
           (unsigned long long)n, (unsigned long long)(2 * bytes), (unsigned long long)bytes, reps);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
