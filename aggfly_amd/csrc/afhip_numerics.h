// afhip_numerics.h — device arithmetic helpers of k_fused_temporal (afhip_kernels.h): integer powers by a double-double chain, f64 division / reciprocal without the
// library's scaling code, operand-pinned fma / max / min forms
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afhip {

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double nan64() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ double inf64() { return __longlong_as_double(0x7ff0000000000000LL); }

// A volatile empty asm cannot be speculated, so a block that starts with it stays behind its branch.
#define KEEP_BRANCH() asm volatile("")

// x**e for a small integer e, evaluated as a double-double product chain so that the
// result is the correctly rounded power in all but ~1e-14 of cases — what libm's pow()
// behind np.power (dataset.py:543) returns.  Plain repeated multiplication differs from
// np.power in the last bit for 26-35 % of inputs at e = 3, 4.
// The exponent is wave-uniform: one scalar loop drives the N independent chains of a lane.
//
// The pair (hi, lo) is NOT renormalised between steps: (hi + lo) x = p + (err + lo x) with p = RN(hi x), err = hi x - p exactly
// (one fma) and the small term rounded once (a second fma) — |lo| stays within a few ulps of hi for every exponent lowered to this
// form (|e| <= 64), so its rounding errors are of order 2^-106 and only the final hi + lo rounds at 2^-53.  Three instructions per
// step, two for the first (lo = 0), instead of the six of a renormalising step: x^3 6 instructions, x^4 9 (were 13 and 19).
template <int N, bool NEG = true>
__device__ __forceinline__ void powi_dd_vec(double (&x)[N], int e) {
    if (e == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = 1.0;
        return;
    }
    const int n = e < 0 ? -e : e;
    if (n == 2) {                       // the chain's first step is RN(x * x) exactly: one multiply
        KEEP_BRANCH();
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = x[i] * x[i];
    } else if (n > 2) {
        double hi[N], lo[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { hi[i] = x[i] * x[i]; lo[i] = __fma_rn(x[i], x[i], -hi[i]); }
        auto step = [&]() {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double p = hi[i] * x[i];
                const double err = __fma_rn(hi[i], x[i], -p);
                lo[i] = __fma_rn(lo[i], x[i], err);
                hi[i] = p;
            }
        };
        step();                         // n >= 3; the cube and the fourth power (polynomials) are straight-line code: a loop
        if (n > 3) {                    // carries (hi, lo) through register copies
            KEEP_BRANCH();
            step();
            for (int it = 4; it < n; ++it) step();
        }
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = hi[i] + lo[i];
    }
    if constexpr (NEG) {                // (the lean short-group forms take non-negative exponents only: no division code in them)
        if (e < 0) {
            KEEP_BRANCH();
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = 1.0 / x[i];
        }
    }
}
__device__ __forceinline__ double powi_dd(double x, int e) {
    double v[1] = {x};
    powi_dd_vec<1>(v, e);
    return v[0];
}

// ---- f64 division / square root without the library's scaling and special-case code ----
// 1/x for normal x: v_rcp_f64 (measured 2^-24.4 on gfx950, scripts/probe/rcp_rsq_probe.py) + one Newton step: <= 10 ulp
// (2.2e-15).  Enough for sine_arc's quotients: their error only matters next to |x| = 1, where the closed forms are
// ill-conditioned by themselves.
__device__ __forceinline__ double rcp_newton1(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double e = __fma_rn(-x, y, 1.0);
    return __fma_rn(y, e, y);
}
// a / b given y ~ 1/b (Markstein): q0 = a*y, r = a - b*q0 exactly (fma), q = q0 + r*y.  With y the
// CORRECTLY rounded reciprocal (the host's 1.0/n for a group length n) q is the correctly rounded
// quotient — bit-identical to a true division (checked against exact rational arithmetic,
// DESIGN.md §5); with a faithful y it is within 1 ulp.  v_div_fixup restores the IEEE results for
// inf / NaN / zero operands.
__device__ __forceinline__ double div_by_finite(double a, double b, double y) {   // finite a, normal b
    const double q0 = a * y;
    const double r = __fma_rn(-q0, b, a);
    return __fma_rn(r, y, q0);
}
__device__ __forceinline__ double div_by(double a, double b, double y) {
    return __builtin_amdgcn_div_fixup(div_by_finite(a, b, y), b, a);
}
// d = a * b + c as ONE three-address v_fma_f64 with the (wave-uniform) factor b in a scalar register pair and the addend in a
// vector register: hipcc otherwise emits v_mov_b64 + v_fmac_f64 (the two-address form clobbers its addend).
__device__ __forceinline__ double fma_vsv(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
#else
    return __builtin_fma(a, b, c);
#endif
}
// a * b + c with the addend in scalar registers (three-address form: the compiler's v_fmac needs a v_mov of the constant first)
__device__ __forceinline__ double fma_vvs(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
#else
    return __builtin_fma(a, b, c);
#endif
}
// (k << n) + base in one instruction
__device__ __forceinline__ uint32_t lshl_add(uint32_t base, uint32_t k, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(k), "n"(n), "v"(base));
    return d;
#else
    return (k << n) + base;
#endif
}
// max(x, 0) / max(-x, 0) as one v_max_f64: the builtin first canonicalises an operand it cannot prove quiet (v_max x, x);
// a NaN operand gives 0 either way (callers replace the value of a NaN window afterwards)
__device__ __forceinline__ double max0(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_max_f64 %0, %1, 0" : "=v"(d) : "v"(x));
    return d;
#else
    return x > 0.0 ? x : 0.0;
#endif
}
// min(x, c) with the wave-uniform c in a scalar register pair, one instruction (no canonicalising v_max in front)
__device__ __forceinline__ double min_vs(double x, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(x), "s"(c));
    return d;
#else
    return x < c ? x : c;
#endif
}
__device__ __forceinline__ double max0_neg(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_max_f64 %0, -%1, 0" : "=v"(d) : "v"(x));
    return d;
#else
    return -x > 0.0 ? -x : 0.0;
#endif
}

}  // namespace afhip
