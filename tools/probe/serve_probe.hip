// serve_probe.hip -- one call of the serving lane's routine (mp.hpp: euclid_serve) on the GPU, driven round by round
// from Python (tools/probe/serve_probe.py): separates the serving side from the client side when a remainder sequence
// misbehaves on the device only.  Built against csrc as it is, or against a patched copy (CSRC_DIR).  Diagnostic tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "form_io.hpp"
using namespace cofhe;

__global__ void k_serve(const uint32_t *xy, int stop_bits, int *state, uint32_t *w) {
    if (threadIdx.x != 0) return;
    int tx = state[0], ty = state[1];
    bool sd = state[2] != 0;
    uint32_t ww[SERVE_WORDS] = {0};
    euclid_serve(xy, stop_bits, tx, ty, sd, ww);
    state[0] = tx; state[1] = ty; state[2] = sd ? 1 : 0;
    for (int i = 0; i < 8; i++) w[i] = i < SERVE_WORDS ? ww[i] : 0u;
}

// stdin protocol, one request per line:  "<stop_bits> <tx> <ty> <sdone> <80 hex words x|y>"  ->  "<tx> <ty> <sdone> <8 hex words>"
int main() {
    uint32_t *dxy, *dw; int *dst;
    hipMalloc(&dxy, 80 * 4); hipMalloc(&dw, 8 * 4); hipMalloc(&dst, 3 * 4);
    char *line = nullptr; size_t cap = 0;
    while (getline(&line, &cap, stdin) > 0) {
        uint32_t xy[80]; int st[3], stop;
        char *p = line;
        stop = (int)strtol(p, &p, 10);
        for (int i = 0; i < 3; i++) st[i] = (int)strtol(p, &p, 10);
        for (int i = 0; i < 80; i++) xy[i] = (uint32_t)strtoul(p, &p, 16);
        hipMemcpy(dxy, xy, sizeof(xy), hipMemcpyHostToDevice);
        hipMemcpy(dst, st, sizeof(st), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_serve, dim3(1), dim3(64), 0, 0, dxy, stop, dst, dw);
        uint32_t w[8];
        hipMemcpy(w, dw, sizeof(w), hipMemcpyDeviceToHost);
        hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost);
        printf("%d %d %d", st[0], st[1], st[2]);
        for (int i = 0; i < 8; i++) printf(" %x", w[i]);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
