// wg_timing.hip -- where the time of a one-wave k_compose_wg launch goes: start / end timestamp
// (100 MHz wall clock) and placement (XCC, SE, CU) of each of the 1024 workgroups of the 128x128
// matadd.  Inputs: two record files written by tools/wg_timing.py; output: CSV on stdout.
#include <hip/hip_runtime.h>
#define COFHE_WG_TIMING
__device__ unsigned long long g_wg_t[16384 * 4];
#include "../cofhe_amd/csrc/cofhe_hip.hip"

#include <fstream>
#include <iostream>

static std::vector<char> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc < 4) return 1;
    std::vector<char> d = slurp(argv[1]), a = slurp(argv[2]), b = slurp(argv[3]);
    cofhe_hip_ctx *ctx = nullptr;
    if (cofhe_hip_ctx_create(0, (const uint8_t *)d.data(), d.size(), &ctx)) { std::cerr << cofhe_hip_last_error() << "\n"; return 2; }
    const uint64_t n = a.size() / (REC_WORDS * 4);
    void *da, *db, *dout;
    hipMalloc(&da, a.size()); hipMalloc(&db, b.size()); hipMalloc(&dout, a.size());
    hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice);
    for (int i = 0; i < 5; i++) cofhe_hip_compose_records(ctx, da, db, dout, n, nullptr);     // the last launch is the one reported
    hipDeviceSynchronize();
    const size_t wgs = (n + 31) / 32;
    std::vector<unsigned long long> t(wgs * 4);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_wg_t), wgs * 4 * sizeof(unsigned long long));
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < wgs; i++) t0 = t[4 * i] < t0 ? t[4 * i] : t0;
    std::cout << "wg,start_us,end_us,hw_id,xcc_id\n";
    for (size_t i = 0; i < wgs; i++)
        std::cout << i << "," << (t[4 * i] - t0) / 100.0 << "," << (t[4 * i + 1] - t0) / 100.0 << "," << t[4 * i + 2] << "," << (t[4 * i + 3] & 15) << "\n";
    return 0;
}
