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
    printf("cross-lane helper check: %d mismatches\n", bad);
    return bad != 0;
}
