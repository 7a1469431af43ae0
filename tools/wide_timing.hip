// wide_timing.hip -- where the time of ONE composition in the wavefront-wide layout goes (qfw.hpp: wf_compose): one
// wavefront composes the pairs of the two record files one after the other; shader-clock ticks per phase, summed in LDS.
// Inputs: the record files of tools/wg_timing.py gen.  Diagnostic build: the stamps are not compiled into the library.
#include <hip/hip_runtime.h>
#define COFHE_WIDE_TIMING
#include "../cofhe_amd/csrc/qfw.hpp"

#include <fstream>
#include <iostream>
#include <vector>

using namespace cofhe;
using namespace cofhe::wide;

__global__ void __launch_bounds__(64) k_wide_timing(const uint32_t *a, const uint32_t *b, uint32_t *out, uint32_t n, const uint32_t *absdelta,
                                                    int half_dbits, unsigned long long *report) {
    const QDisc dd{absdelta, half_dbits};
    unsigned long long *s = wt_slots();
    if (threadIdx.x < 32) s[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long fails = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = 0; i < n; i++) {
        const WForm x = wf_load(a + (uint64_t)i * REC_WORDS), y = wf_load(b + (uint64_t)i * REC_WORDS);
        WForm r;
        if (wf_compose(r, x, y, dd)) wf_store(r, out + (uint64_t)i * REC_WORDS); else fails++;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (threadIdx.x < 32) report[threadIdx.x] = s[threadIdx.x];
    if (threadIdx.x == 0) { report[32] = t1 - t0; report[33] = r1 - r0; report[34] = fails; }
}

static std::vector<char> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc < 4) return 1;
    std::vector<char> d = slurp(argv[1]), a = slurp(argv[2]), b = slurp(argv[3]);
    uint32_t n = (uint32_t)(a.size() / (REC_WORDS * 4));
    if (argc > 4 && (uint32_t)atoi(argv[4]) < n) n = (uint32_t)atoi(argv[4]);
    // |Delta| as 80 little-endian limbs, half_dbits = bitlen / 2 as the library's context does it
    std::vector<uint32_t> dl(2 * PLIMBS, 0);
    int bits = 0;
    for (size_t i = 0; i < d.size(); i++) {              // little-endian magnitude bytes (tools/wg_timing.py gen)
        const uint8_t v = (uint8_t)d[i];
        if (i / 4 < dl.size()) dl[i / 4] |= (uint32_t)v << (8 * (i % 4));
        if (v) { int hb = 0; for (int t = 0; t < 8; t++) if (v >> t & 1) hb = t + 1; bits = (int)i * 8 + hb; }
    }
    void *da, *db, *dout, *ddl, *drep;
    hipMalloc(&da, a.size()); hipMalloc(&db, b.size()); hipMalloc(&dout, a.size()); hipMalloc(&ddl, dl.size() * 4); hipMalloc(&drep, 40 * 8);
    hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice);
    hipMemcpy(ddl, dl.data(), dl.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_wide_timing, dim3(1), dim3(64), 0, 0, (const uint32_t *)da, (const uint32_t *)db, (uint32_t *)dout, n,
                           (const uint32_t *)ddl, (bits + 1) / 2, (unsigned long long *)drep);
        if (hipDeviceSynchronize() != hipSuccess) { std::cerr << "kernel failed\n"; return 3; }
    }
    unsigned long long rep[40];
    hipMemcpy(rep, drep, sizeof(rep), hipMemcpyDeviceToHost);
    const double ghz = (double)rep[32] / ((double)rep[33] * 10.0);          // s_memrealtime: 100 MHz
    const double us = 1e-3 / ghz / n;
    const char *names[12] = {"representative, s, m", "Euclid 1 (full)", "r = y1 m mod a1", "Euclid 2 (partial)", "M1, M2 (products, exact divisions)",
                             "a', b'", "c' (square, exact division)", "reduce", "  Euclid: bit lengths + windows", "  Euclid: batch",
                             "  Euclid: four linear combinations", "  Euclid: long-division steps"};
    std::cout << "wide composition, one wavefront, " << n << " pairs, " << rep[34] << " left to the 8-lane route; clock " << ghz << " GHz\n";
    std::cout << "total " << rep[32] * us << " us per composition\n";
    for (int k = 0; k < 12; k++) std::cout << "  " << names[k] << ": " << rep[k] * us << " us\n";
    return 0;
}
