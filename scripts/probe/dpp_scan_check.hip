// Checks the cross-lane helpers of afhip_lz4_kernels.h on the device: DPP prefix sum / prefix maximum, ds_bpermute wrap-around.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../aggfly_amd/csrc/afhip_lz4_kernels.h"
using namespace afhip;
__global__ void k(const int* in, int* out) {
    const int lane = threadIdx.x;
    const int v = in[lane];
    out[lane] = wave_scan_add(v);
    out[64 + lane] = wave_scan_max(v);
    out[128 + lane] = lane_get(v, lane + 70);      // wraps: lane + 6
    out[192 + lane] = lane_get(v, lane - 1);       // lane 0 reads lane 63
    out[256 + lane] = mod_small(in[64 + lane], in[128 + lane]);
}
__global__ void kwalk(const int* nextv_in, int* out) {
    __shared__ volatile int mark[64];
    const int lane = threadIdx.x;
    const int nextv = nextv_in[lane];
    int c0 = 0, c1 = 0;
    const uint64_t m0 = token_walk_serial(nextv, c0), m1 = token_walk_lifted(nextv, lane, mark, c1);
    if (lane == 0) { out[0] = (int)m0; out[1] = (int)(m0 >> 32); out[2] = c0; out[3] = (int)m1; out[4] = (int)(m1 >> 32); out[5] = c1; }
}
int main() {
    int h[192], *d, *o, r[320];
    srand(7);
    int bad = 0;
    for (int trial = 0; trial < 50; ++trial) {
        for (int i = 0; i < 64; ++i) { h[i] = rand() % 400; h[64 + i] = rand() % 4096; h[128 + i] = 1 + rand() % (trial % 2 ? 300 : 65535); }
        hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
        hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        int s = 0, m = 0;
        for (int i = 0; i < 64; ++i) {
            s += h[i]; m = h[i] > m ? h[i] : m;
            if (r[i] != s) { if (bad++ < 5) printf("scan_add lane %d: %d want %d\n", i, r[i], s); }
            if (r[64 + i] != m) { if (bad++ < 10) printf("scan_max lane %d: %d want %d\n", i, r[64 + i], m); }
            if (r[128 + i] != h[(i + 70) & 63]) { if (bad++ < 15) printf("bperm wrap lane %d\n", i); }
            if (r[192 + i] != h[(i - 1) & 63]) { if (bad++ < 20) printf("bperm -1 lane %d\n", i); }
            if (r[256 + i] != h[64 + i] % h[128 + i]) { if (bad++ < 25) printf("mod %d %% %d: %d\n", h[64 + i], h[128 + i], r[256 + i]); }
        }
        hipFree(d); hipFree(o);
    }
    int wbad = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        int nv[64], r6[6], *d, *o;
        // random sequences: a token at i leads to i + 3 .. i + 3 + spread, sometimes beyond the window / not fitting
        const int spread = 1 + rand() % 30;
        for (int i = 0; i < 64; ++i) { int nx = i + 3 + rand() % spread; nv[i] = (nx > 64 || rand() % 23 == 0) ? 65 : nx; }
        hipMalloc(&d, sizeof nv); hipMalloc(&o, sizeof r6);
        hipMemcpy(d, nv, sizeof nv, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kwalk, dim3(1), dim3(64), 0, 0, d, o);
        hipMemcpy(r6, o, sizeof r6, hipMemcpyDeviceToHost);
        unsigned long long want = 0; int c = 0;
        for (int pos = 0; pos < 64;) { int t = nv[pos]; if (t > 64) break; want |= 1ull << pos; pos = c = t; }
        const unsigned long long m0 = (unsigned)r6[0] | ((unsigned long long)(unsigned)r6[1] << 32), m1 = (unsigned)r6[3] | ((unsigned long long)(unsigned)r6[4] << 32);
        if (m0 != want || r6[2] != c) { if (wbad++ < 5) printf("serial walk: %llx c %d want %llx c %d\n", m0, r6[2], want, c); }
        if (m1 != want || r6[5] != c) { if (wbad++ < 10) printf("lifted walk: %llx c %d want %llx c %d\n", m1, r6[5], want, c); }
        hipFree(d); hipFree(o);
    }
    printf("token walk check: %d mismatches\n", wbad);
    bad += wbad;
    printf("cross-lane helper check: %d mismatches\n", bad);
    return bad != 0;
}
