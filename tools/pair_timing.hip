// pair_timing.hip -- where the time of a k_compose_pair launch goes (diagnostic build; not part of the product):
// P2_PHASE stamps of every wavefront (shader clock) -> mean cycles per phase of qf2_compose.
#include <hip/hip_runtime.h>
#define COFHE_PAIR_TIMING
__device__ unsigned long long g_pair_phase[4096 * 16];
#include "../experiments/pair_layout/compose2.hip"

#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

static std::vector<char> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
int main(int argc, char **argv) {
    if (argc < 4) return 1;
    std::vector<char> d = slurp(argv[1]), a = slurp(argv[2]), b = slurp(argv[3]);
    std::vector<uint32_t> ad(80, 0);
    memcpy(ad.data(), d.data(), d.size());
    int dbits = 0;
    for (int i = 79; i >= 0; i--) if (ad[i]) { dbits = i * 32 + 32 - __builtin_clz(ad[i]); break; }
    const uint64_t n = a.size() / (cofhe::REC_WORDS * 4);
    void *da, *db, *dout; uint32_t *dd, *fb;
    hipMalloc(&da, a.size()); hipMalloc(&db, b.size()); hipMalloc(&dout, a.size()); hipMalloc(&dd, 320); hipMalloc(&fb, (n + 1) * 4);
    hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice);
    hipMemcpy(dd, ad.data(), 320, hipMemcpyHostToDevice);
    for (int i = 0; i < 4; i++) {
        hipMemset(fb, 0, 4);
        cofhe_k::launch_compose_pair(17, (const uint32_t *)da, (const uint32_t *)db, (uint32_t *)dout, n, dd, (dbits + 1) / 2, fb, fb + 1, 0);
    }
    hipDeviceSynchronize();
    uint32_t cnt = 0;
    hipMemcpy(&cnt, fb, 4, hipMemcpyDeviceToHost);
    const size_t waves = (n + 31) / 32;
    std::vector<unsigned long long> ph(waves * 16);
    hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_pair_phase), waves * 16 * sizeof(unsigned long long));
    const char *names[9] = {"sizes + representative + s, m", "Euclid 1 (full)", "r = y1 m mod a1 (mul + Knuth)", "general gcd structure", "(bit lengths)",
                            "Euclid 2 (partial)", "M1, M2 (4 mul + 2 exact div)", "a', b' (4 mul)", "c' (square + exact div)"};
    double sum[10] = {0};
    for (size_t w = 0; w < waves; w++)
        for (int k = 0; k < 9; k++) sum[k] += (double)(ph[16 * w + k + 1] - ph[16 * w + k]);
    double tot = 0;
    for (int k = 0; k < 9; k++) tot += sum[k] / waves;
    std::cout << "elements " << n << ", left the fast path: " << cnt << "; mean cycles per wavefront and phase (total " << tot << "):\n";
    for (int k = 0; k < 9; k++) std::cout << "  " << names[k] << (k == 8 ? " + reduce" : "") << ": " << sum[k] / waves << "\n";
    return 0;
}
