// afhip_loads.h — how k_fused_temporal (afhip_kernels.h) reads: EXEC-masked threshold adds, scalar (constant address space) table loads, the streaming row loads
// (global / buffer loads with the nt policy) and the row vectors they fill
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afhip {

// acc += w on the lanes where t0 < v < t1 (strict; a NaN v fails both compares): the two
// compares narrow EXEC directly (v_cmpx), the add runs under that mask and EXEC is put back —
// 3 VALU ops, where compare + compare + a 64-bit select (2 x v_cndmask) + add takes 5.
template <typename T>
__device__ __forceinline__ void add_if_between(double& acc, double w, T v, T t0, T t1) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long saved;
    if constexpr (sizeof(T) == 4) {
        asm("s_mov_b64 %[sv], exec\n\t"
            "v_cmpx_lt_f32_e32 %[t0], %[v]\n\t"
            "v_cmpx_gt_f32_e32 %[t1], %[v]\n\t"
            "v_add_f64 %[acc], %[acc], %[w]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [acc] "+v"(acc), [sv] "=&s"(saved)
            : [w] "v"(w), [v] "v"(v), [t0] "s"(t0), [t1] "s"(t1)
            : "vcc");
    } else {
        asm("s_mov_b64 %[sv], exec\n\t"
            "v_cmpx_lt_f64_e32 %[t0], %[v]\n\t"
            "v_cmpx_gt_f64_e32 %[t1], %[v]\n\t"
            "v_add_f64 %[acc], %[acc], %[w]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [acc] "+v"(acc), [sv] "=&s"(saved)
            : [w] "v"(w), [v] "v"(v), [t0] "s"(t0), [t1] "s"(t1)
            : "vcc");
    }
#else
    acc += (v > t0 && v < t1) ? w : 0.0;
#endif
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// The plan tables (bounds, emit flags, chunk list) are read-only for the whole launch and
// indexed by wave-uniform values.  Reading them through the constant address space makes
// hipcc use scalar loads (s_load, counted on lgkmcnt), so no compiler-issued vector load —
// and with it no compiler-inserted s_waitcnt vmcnt(0) — lands inside the streaming loop.
template <typename T>
__device__ __forceinline__ T ld_uniform(const T* p) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "scalar words only");
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(4))) T*)(uintptr_t)p;
#else
    return *p;
#endif
}

template <typename TIn, int VEC> struct RawVec;
template <> struct alignas(8) RawVec<double, 1> { double v[1]; };
template <> struct alignas(16) RawVec<double, 2> { double v[2]; };
template <> struct alignas(4) RawVec<float, 1> { float v[1]; };
template <> struct alignas(16) RawVec<float, 4> { float v[4]; };
template <> struct alignas(8) RawVec<float, 2> { float v[2]; };

// One lane's VEC cells of a row, read once: non-temporal loads keep the stream from
// displacing the plan tables and partials in L2 / Infinity Cache.
template <typename TIn, int VEC, int AUX>
__device__ __forceinline__ RawVec<TIn, VEC> ld_stream(const TIn* p) {
    RawVec<TIn, VEC> r;
    if constexpr (AUX != 0) {
        typedef TIn vec_t __attribute__((ext_vector_type(VEC)));
        if constexpr (VEC == 1) {
            r.v[0] = __builtin_nontemporal_load(p);
        } else {
            vec_t t = __builtin_nontemporal_load((const vec_t*)p);
#pragma unroll
            for (int i = 0; i < VEC; ++i) r.v[i] = t[i];
        }
    } else {
        r = *(const RawVec<TIn, VEC>*)p;
    }
    return r;
}

// The same through a buffer descriptor on a UNIFORM row pointer + the lane's 32-bit byte offset: the address needs no vector
// arithmetic at all (a global_load wants a 64-bit vector address: one v_lshl_add_u64 per load).
template <typename TIn, int VEC, int AUX>
__device__ __forceinline__ RawVec<TIn, VEC> ld_stream_row(const void* row, uint32_t voff) {
    RawVec<TIn, VEC> r;
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(row), 0, -1, 0x00020000);
    constexpr int aux = AUX != 0 ? 2 : 0;          // nt
    constexpr int bytes = (int)sizeof(TIn) * VEC;
    static_assert(bytes == 4 || bytes == 8 || bytes == 16, "one dword, two or four per lane");
    if constexpr (bytes == 4) {
        const uint32_t t = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, aux);
        __builtin_memcpy(&r, &t, 4);
    } else if constexpr (bytes == 8) {
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        const u2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, aux);
        __builtin_memcpy(&r, &t, 8);
    } else {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, aux);
        __builtin_memcpy(&r, &t, 16);
    }
#endif
    return r;
}

}  // namespace afhip
