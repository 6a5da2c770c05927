// What can a bare streaming read reach on this box?  A kernel with k_fused_temporal's access pattern and nothing else: every lane owns
// W consecutive dwords of a row, a workgroup owns one tile of columns and walks a chunk of rows with D row loads in flight, the values
// are added up as integers and one dword per lane is stored at the end.  Sweeps load width, depth, workgroup size, chunk count and the
// nontemporal bit over a cube of the configs[1] size (8760 rows x 2,476,800 bytes = 21.7 GB), prints GB/s per arm (median of 7).
//
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/read_bw.hip -o scripts/probe/_build/read_bw && scripts/probe/_build/read_bw
//
// The roofline denominator stays the 8 TB/s spec peak; this is the other reference point (profiles/r03_read_ceiling.txt).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int W> struct Vec;
template <> struct Vec<1> { typedef uint32_t t; };
template <> struct Vec<2> { typedef uint2 t; };
template <> struct Vec<4> { typedef uint4 t; };

__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint2 v) { return v.x + v.y; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x + v.y + v.z + v.w; }

template <int W, bool NT> __device__ __forceinline__ typename Vec<W>::t ld(const uint32_t* p) {
    typedef typename Vec<W>::t V;
    if constexpr (NT) {
        if constexpr (W == 1) return __builtin_nontemporal_load(p);
        else if constexpr (W == 2) { typedef uint32_t v2 __attribute__((ext_vector_type(2))); v2 r = __builtin_nontemporal_load((const v2*)p); return V{r.x, r.y}; }
        else { typedef uint32_t v4 __attribute__((ext_vector_type(4))); v4 r = __builtin_nontemporal_load((const v4*)p); return V{r.x, r.y, r.z, r.w}; }
    } else {
        return *(const V*)p;
    }
}

// grid.x = column tiles, grid.y = row chunks; row_dw = dwords per row (a multiple of W), rows [t0, t1) per chunk
template <int W, int D, bool NT>
__global__ void k_read(const uint32_t* __restrict__ base, size_t row_dw, int T, int rows_per_chunk, uint32_t* __restrict__ out) {
    const size_t col = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * W;
    if (col >= row_dw) return;                                   // (whole lanes only: row_dw % W == 0)
    const int t0 = blockIdx.y * rows_per_chunk, t1 = min(T, t0 + rows_per_chunk);
    const uint32_t* p = base + (size_t)t0 * row_dw + col;
    uint32_t acc = 0;
    int t = t0;
    for (; t + D <= t1; t += D) {
        typename Vec<W>::t v[D];
#pragma unroll
        for (int j = 0; j < D; ++j) v[j] = ld<W, NT>(p + (size_t)j * row_dw);
#pragma unroll
        for (int j = 0; j < D; ++j) acc += fold(v[j]);
        p += (size_t)D * row_dw;
    }
    for (; t < t1; ++t, p += row_dw) acc += fold(ld<W, NT>(p));
    out[(size_t)blockIdx.y * (row_dw / W) + col / W] = acc;
}

// the same walk with raw buffer loads on a uniform row pointer (the production kernel's ld_stream_row) and an explicit cache-policy
// operand: aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1.  REMAP: 0 = blockIdx.x is the tile; 1 = every XCD (blockIdx.x % 8) walks its own
// contiguous eighth of the tiles; 2 = tiles in groups of 8 per XCD (4 KB of a row per XCD at 8-byte lanes)
template <int AUX, int D, int REMAP>
__global__ void k_read_buf(const uint32_t* __restrict__ base, size_t row_dw, int T, int rows_per_chunk, uint32_t* __restrict__ out) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    uint32_t b = blockIdx.x;
    const uint32_t nb = gridDim.x;
    if constexpr (REMAP == 1) { const uint32_t per = (nb + 7) / 8; b = (b % 8) * per + b / 8; if (b >= nb) return; }
    if constexpr (REMAP == 2) { const uint32_t g = b / 64, r = b % 64; b = g * 64 + (r % 8) * 8 + r / 8; if (b >= nb) return; }
    const size_t lane = (size_t)b * blockDim.x + threadIdx.x;
    const size_t col = lane * 2;
    if (col >= row_dw) return;
    const int t0 = blockIdx.y * rows_per_chunk, t1 = min(T, t0 + rows_per_chunk);
    const uint32_t* row = base + (size_t)t0 * row_dw;
    const uint32_t voff = (uint32_t)(col * 4);
    uint32_t acc = 0;
    int t = t0;
    for (; t + D <= t1; t += D) {
        u2 v[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(row + (size_t)j * row_dw), 0, -1, 0x00020000);
            v[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, AUX);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) acc += v[j].x + v[j].y;
        row += (size_t)D * row_dw;
    }
    for (; t < t1; ++t, row += row_dw) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(row), 0, -1, 0x00020000);
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, AUX);
        acc += v.x + v.y;
    }
    out[(size_t)blockIdx.y * (row_dw / 2) + lane] = acc;
}

template <int AUX, int REMAP>
static void launch_buf(const uint32_t* base, size_t row_dw, int T, uint32_t* out, hipStream_t s) {
    const size_t lanes = row_dw / 2;
    dim3 grid((unsigned)((lanes + 63) / 64), 1);
    if (REMAP == 1) grid.x = (grid.x + 7) / 8 * 8;
    if (REMAP == 2) grid.x = (grid.x + 63) / 64 * 64;
    hipLaunchKernelGGL((k_read_buf<AUX, 4, REMAP>), grid, dim3(64), 0, s, base, row_dw, T, T, out);
}

__global__ void k_fill(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u);
}

struct Arm { int W, D, nt, wg, chunks; };

template <int W, int D, bool NT>
static void launch(const uint32_t* base, size_t row_dw, int T, int wg, int chunks, uint32_t* out, hipStream_t s) {
    const size_t lanes = row_dw / W;
    dim3 grid((unsigned)((lanes + wg - 1) / wg), (unsigned)chunks);
    const int rpc = (T + chunks - 1) / chunks;
    hipLaunchKernelGGL((k_read<W, D, NT>), grid, dim3(wg), 0, s, base, row_dw, T, rpc, out);
}

template <int W, int D>
static void launch_nt(bool nt, const uint32_t* base, size_t row_dw, int T, int wg, int chunks, uint32_t* out, hipStream_t s) {
    if (nt) launch<W, D, true>(base, row_dw, T, wg, chunks, out, s); else launch<W, D, false>(base, row_dw, T, wg, chunks, out, s);
}

template <int W>
static void launch_d(int D, bool nt, const uint32_t* base, size_t row_dw, int T, int wg, int chunks, uint32_t* out, hipStream_t s) {
    switch (D) {
        case 2: launch_nt<W, 2>(nt, base, row_dw, T, wg, chunks, out, s); break;
        case 4: launch_nt<W, 4>(nt, base, row_dw, T, wg, chunks, out, s); break;
        case 8: launch_nt<W, 8>(nt, base, row_dw, T, wg, chunks, out, s); break;
        case 16: launch_nt<W, 16>(nt, base, row_dw, T, wg, chunks, out, s); break;
        default: std::fprintf(stderr, "depth %d not built\n", D); std::exit(1);
    }
}

static void launch_arm(const Arm& a, const uint32_t* base, size_t row_dw, int T, uint32_t* out, hipStream_t s) {
    if (a.W == 1) launch_d<1>(a.D, a.nt, base, row_dw, T, a.wg, a.chunks, out, s);
    else if (a.W == 2) launch_d<2>(a.D, a.nt, base, row_dw, T, a.wg, a.chunks, out, s);
    else launch_d<4>(a.D, a.nt, base, row_dw, T, a.wg, a.chunks, out, s);
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? std::atoi(argv[1]) : 8760;
    const size_t row_bytes = argc > 2 ? (size_t)std::atoll(argv[2]) : (size_t)215 * 1440 * 8;
    const size_t row_dw = row_bytes / 4;
    if (row_dw % 4 != 0 || T < 16) { std::fprintf(stderr, "row bytes must be a multiple of 16, T >= 16\n"); return 1; }
    const size_t n = (size_t)T * row_dw;
    uint32_t *cube = nullptr, *out = nullptr;
    CK(hipMalloc(&cube, n * 4));
    CK(hipMalloc(&out, row_dw * 4 * 64));                        // one dword per lane and chunk, at most 64 chunks
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, cube, n);
    CK(hipDeviceSynchronize());
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Arm> arms;
    for (int W : {1, 2, 4})
        for (int D : {4, 8, 16})
            for (int wg : {64, 256})
                for (int nt : {1, 0})
                    arms.push_back({W, D, nt, wg, 1});
    for (int chunks : {2, 4, 8, 16}) { arms.push_back({2, 4, 1, 64, chunks}); arms.push_back({2, 8, 1, 256, chunks}); arms.push_back({4, 4, 1, 256, chunks}); }
    arms.push_back({2, 2, 1, 64, 1}); arms.push_back({4, 2, 1, 64, 1});
    std::printf("cube: %d rows x %zu bytes = %.3f GB\n", T, row_bytes, n * 4 / 1e9);
    std::printf("%-8s %-6s %-4s %-5s %-7s %9s %9s %9s\n", "dwords", "depth", "nt", "wg", "chunks", "ms_med", "ms_min", "GB/s_med");
    for (int pass = 0; pass < 2; ++pass) {                       // the whole sweep twice: the second pass shows what is clock ramp / order
        for (const Arm& a : arms) {
            std::vector<float> ms;
            for (int r = 0; r < 9; ++r) {
                CK(hipEventRecord(e0, s));
                launch_arm(a, cube, row_dw, T, out, s);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float m; CK(hipEventElapsedTime(&m, e0, e1));
                if (r >= 2) ms.push_back(m);
            }
            CK(hipGetLastError());
            std::sort(ms.begin(), ms.end());
            const float med = ms[ms.size() / 2];
            std::printf("%-8d %-6d %-4d %-5d %-7d %9.3f %9.3f %9.1f\n", a.W, a.D, a.nt, a.wg, a.chunks, med, ms[0], n * 4 / 1e6 / med);
            std::fflush(stdout);
        }
        struct B { const char* name; void (*fn)(const uint32_t*, size_t, int, uint32_t*, hipStream_t); };
        const B bufs[] = {{"buffer aux=0 (plain)", launch_buf<0, 0>}, {"buffer aux=1 (sc0)", launch_buf<1, 0>}, {"buffer aux=2 (nt)", launch_buf<2, 0>},
                          {"buffer aux=3 (sc0 nt)", launch_buf<3, 0>}, {"buffer aux=16 (sc1)", launch_buf<16, 0>}, {"buffer aux=17 (sc0 sc1)", launch_buf<17, 0>},
                          {"buffer aux=18 (sc1 nt)", launch_buf<18, 0>}, {"buffer aux=19 (sc0 sc1 nt)", launch_buf<19, 0>},
                          {"buffer nt, XCD-contiguous eighths", launch_buf<2, 1>}, {"buffer nt, 8 tiles per XCD", launch_buf<2, 2>}};
        for (const B& b : bufs) {                                // 8-byte lanes, depth 4, 64 threads, one chunk
            std::vector<float> ms;
            for (int r = 0; r < 9; ++r) {
                CK(hipEventRecord(e0, s));
                b.fn(cube, row_dw, T, out, s);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float m; CK(hipEventElapsedTime(&m, e0, e1));
                if (r >= 2) ms.push_back(m);
            }
            CK(hipGetLastError());
            std::sort(ms.begin(), ms.end());
            std::printf("%-40s %9.3f %9.3f %9.1f\n", b.name, ms[ms.size() / 2], ms[0], n * 4 / 1e6 / ms[ms.size() / 2]);
            std::fflush(stdout);
        }
        std::printf("-- pass %d done\n", pass + 1);
    }
    CK(hipFree(cube)); CK(hipFree(out));
    return 0;
}
