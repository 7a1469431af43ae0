// wg_timing.hip -- where the time of a one-wave k_compose_wg launch goes: start / end timestamp
// (100 MHz wall clock) and placement (XCC, SE, CU) of each of the 1024 workgroups of the 128x128
// matadd.  Inputs: two record files written by tools/wg_timing.py; output: CSV on stdout.
#include <hip/hip_runtime.h>
#define COFHE_WG_TIMING
__device__ unsigned long long g_wg_t[16384 * 4];
__device__ unsigned long long g_wg_phase[16384 * 16];     // CF_PHASE stamps (lane.hpp)
__device__ unsigned int g_wg_wave[16384 * 4];              // per hardware wavefront: HW_ID | logical wave index << 28 (0 = the serving one)
__device__ unsigned long long g_wg_clk[16384 * 2];
__device__ unsigned int g_wg_flags[16384];                  // CF_FLAG bits: 1 common factor, 2 general route, 4 long-division step, 8 add-back         // s_memtime (shader clock) at the start / end of every workgroup
#include "../cofhe_amd/csrc/cofhe_hip.hip"

#include <algorithm>
#include <chrono>
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
    cofhe_hip_ctx *ctx = nullptr;
    if (cofhe_hip_ctx_create(0, (const uint8_t *)d.data(), d.size(), &ctx)) { std::cerr << cofhe_hip_last_error() << "\n"; return 2; }
    const uint64_t n = a.size() / (REC_WORDS * 4);
    void *da, *db, *dout;
    hipMalloc(&da, a.size()); hipMalloc(&db, b.size()); hipMalloc(&dout, a.size());
    hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice);
    // The in-kernel clock (MI355X_MICROARCH.md, DVFS item 6): delta s_memtime / delta s_memrealtime x 100 MHz per
    // workgroup, median over the workgroups of the LAST launch after >= 2 s of back-to-back launches of this kernel on
    // this (random) data -- the clock the chip holds under exactly this load.  argv[4] = seconds of load (default 2).
    const double load_s = argc > 4 ? atof(argv[4]) : 2.0;
    {
        auto t0 = std::chrono::steady_clock::now();
        int launches = 0;
        do {
            for (int i = 0; i < 50; i++) cofhe_hip_compose_records(ctx, da, db, dout, n, nullptr);
            hipDeviceSynchronize();
            launches += 50;
        } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < load_s);
        std::cerr << "load: " << launches << " back-to-back launches\n";
    }
    for (int i = 0; i < 5; i++) cofhe_hip_compose_records(ctx, da, db, dout, n, nullptr);     // the last launch is the one reported
    hipDeviceSynchronize();
    {
        const size_t wgs0 = (n + 31) / 32;
        std::vector<unsigned long long> ck(wgs0 * 2), tt(wgs0 * 4);
        hipMemcpyFromSymbol(ck.data(), HIP_SYMBOL(g_wg_clk), wgs0 * 2 * sizeof(unsigned long long));
        hipMemcpyFromSymbol(tt.data(), HIP_SYMBOL(g_wg_t), wgs0 * 4 * sizeof(unsigned long long));
        std::vector<double> ghz;
        for (size_t i = 0; i < wgs0; i++) {
            const double dt = (double)(tt[4 * i + 1] - tt[4 * i]), dc = (double)(ck[2 * i + 1] - ck[2 * i]);
            if (dt > 0) ghz.push_back(dc / dt * 0.1);                 // ticks per 10 ns -> GHz
        }
        std::sort(ghz.begin(), ghz.end());
        if (!ghz.empty())
            std::cerr << "in-kernel clock (s_memtime / s_memrealtime): median " << ghz[ghz.size() / 2] << " GHz, min " << ghz.front()
                      << ", max " << ghz.back() << " over " << ghz.size() << " workgroups\n"
                      << "CLOCK_JSON {\"clock_ghz_in_kernel\": " << ghz[ghz.size() / 2] << ", \"min\": " << ghz.front() << ", \"max\": "
                      << ghz.back() << ", \"workgroups\": " << ghz.size() << ", \"load_seconds\": " << load_s << "}\n";
    }
    const size_t wgs = (n + 31) / 32;
    std::vector<unsigned long long> t(wgs * 4);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_wg_t), wgs * 4 * sizeof(unsigned long long));
    std::vector<unsigned long long> ph(wgs * 16);
    hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_wg_phase), wgs * 16 * sizeof(unsigned long long));
    {   // mean duration of the phases of qf_compose over the workgroups (us), and the Euclid accounting
        const char *names[7] = {"representative + s, m", "Euclid 1 (full)", "r = y1 m mod a1", "Euclid 2 (partial)", "M1, M2, a', b'",
                                "c' (square, exact division)", "reduce"};
        double sum[7] = {0}, ew[2] = {0}, ea[2] = {0}, er[2] = {0}, es[2] = {0}, pre = 0, post = 0;
        size_t nserve = 0;
        for (size_t i = 0; i < wgs; i++) {
            for (int k = 0; k < 7; k++) sum[k] += (double)(ph[16 * i + k + 1] - ph[16 * i + k]) / 100.0;
            pre += (double)(ph[16 * i] - t[4 * i]) / 100.0;
            post += (double)(t[4 * i + 1] - ph[16 * i + 7]) / 100.0;
            for (int e = 0; e < 2; e++) {
                ew[e] += (double)ph[16 * i + 8 + 2 * e] / 100.0;
                ea[e] += (double)ph[16 * i + 9 + 2 * e] / 100.0;
                er[e] += (double)ph[16 * i + 12 + e];
            }
            if (i % 4 == 0) {          // thread 0 reports: it sits in the serving wavefront when blockIdx % 4 == 0 (make_wg_ctx)
                nserve++;
                for (int e = 0; e < 2; e++) es[e] += (double)ph[16 * i + 14 + e] / 100.0;
            }
        }
        std::cerr << "phase means over " << wgs << " workgroups (us):\n  load " << pre / wgs << "\n";
        for (int k = 0; k < 7; k++) std::cerr << "  " << names[k] << ": " << sum[k] / wgs << "\n";
        std::cerr << "  store " << post / wgs << "\n";
        for (int e = 0; e < 2; e++)
            std::cerr << "  Euclid " << e + 1 << ": rounds " << er[e] / wgs << ", stash+barrier+serve+barrier " << ew[e] / wgs
                      << " us, apply " << ea[e] / wgs << " us; of the first, the serving lane's work (windows + batch + reply) "
                      << es[e] / (nserve ? nserve : 1) << " us (workgroups whose reporting thread serves)\n";
    }
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < wgs; i++) t0 = t[4 * i] < t0 ? t[4 * i] : t0;
    std::vector<unsigned int> wv(wgs * 4);
    hipMemcpyFromSymbol(wv.data(), HIP_SYMBOL(g_wg_wave), wgs * 4 * sizeof(unsigned int));
    std::cout << "wg,start_us,end_us,hw_id,xcc_id,simd0,simd1,simd2,simd3,server_simd,ph_rep,ph_e1,ph_r,ph_e2,ph_m,ph_c,ph_red,rounds1,rounds2,flags\n";
    std::vector<unsigned int> fl(wgs);
    hipMemcpyFromSymbol(fl.data(), HIP_SYMBOL(g_wg_flags), wgs * sizeof(unsigned int));
    for (size_t i = 0; i < wgs; i++) {
        std::cout << i << "," << (t[4 * i] - t0) / 100.0 << "," << (t[4 * i + 1] - t0) / 100.0 << "," << t[4 * i + 2] << "," << (t[4 * i + 3] & 15);
        int server = -1;
        for (int w = 0; w < 4; w++) {
            const unsigned v = wv[4 * i + w];
            std::cout << "," << ((v >> 4) & 3);
            if ((v >> 28) == 0) server = (int)((v >> 4) & 3);
        }
        std::cout << "," << server;
        for (int k = 0; k < 7; k++) std::cout << "," << (double)(ph[16 * i + k + 1] - ph[16 * i + k]) / 100.0;
        std::cout << "," << ph[16 * i + 12] << "," << ph[16 * i + 13] << "," << fl[i] << "\n";
    }
    return 0;
}
