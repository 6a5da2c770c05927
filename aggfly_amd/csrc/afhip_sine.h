// afhip_sine.h — single-sine degree days (aggfly/aggregate/nb_kernels.py:202-251): the closed forms, the acos table (general windows) and the cubic arc
// table of (tmin, tmax) pairs, as k_fused_temporal evaluates them (afhip_kernels.h: sine_column)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "afhip_numerics.h"

namespace afhip {

// ---- single-sine degree days (nb_kernels.py:202-251) ----
// Both of the reference's closed forms are one function,
//   arc(d, x) = d * acos(x) + alpha * sqrt(1 - x^2),            alpha = (tmax - tmin) / 2:
//   cooling part, tmin < thr < tmax:  ((tavg - thr) * acos(z) + rng * sin(acos(z)) / 2) / pi,  z = (2 thr - tmax - tmin) / rng
//                                     = arc(tavg - thr, z) / pi                       since sin(acos z) = sqrt(1 - z^2)
//   heating part:  ((thr - tavg) * (atan(r / sqrt(1 - r^2)) + pi/2) + alpha * cos(atan(...))) / pi,  r = (thr - tavg) / alpha
//                                     = arc(thr - tavg, -r) / pi                      since atan(r / sqrt(1 - r^2)) = asin(r),
//                                       asin(r) + pi/2 = acos(-r) and cos(asin r) = sqrt(1 - r^2)
// (|x| > 1 gives NaN in both forms, as in the reference).
//
// acos comes from a table instead of a polynomial (round 2 evaluated a degree-14 asin: 25 of the arc's 44 fp64 instructions;
// C5 is bound by them).  With a = |x| and g = sqrt(1 - a^2) the smaller of the two, u = min(a, g) <= 0.7072, has
//   asin(u) = phi_k + asin(delta),   k = round(256 u),  phi_k = asin(k / 256),  delta = sin(asin u - phi_k) = u cos(phi_k) - w sin(phi_k),
//   w = max(a, g) = cos(asin u),  |delta| <= 0.0028   ->   asin(delta) = delta + delta^3 (1/6 + 3/40 delta^2)   (next term 5e-20)
// and theta = acos(a) is asin(u) when g is the smaller one, pi/2 - asin(u) otherwise.  Both cases are one table row
// (C, S, TH) per (half, k):  theta = TH + asin(u C + w S)   with
//   a <= g:  C = -cos(phi_k), S = +sin(phi_k), TH = pi/2 - phi_k          a > g:  C = +cos(phi_k), S = -sin(phi_k), TH = phi_k
// (host: afhip_api.hip:sine_table_host; 184 x 2 rows of 32 bytes, the two cases of a k next to each other, copied into LDS by
// every workgroup of a sine_dd plan).
// 11 fp64 + 3 integer instructions and two LDS reads.  Checked on the host against the reference's libm form by
// scripts/fit/arc_table_emulation.py (1.9e-13 absolute on values of order 1-30) and on the device by scripts/sine_accuracy.py.
constexpr int SINE_SCALE = 256;
constexpr int SINE_ROWS = 184;                  // rows per half: k <= 181 for u <= 0.70711; the last rows are guards
constexpr int SINE_TAB_BYTES = 2 * SINE_ROWS * 32;
struct alignas(32) SineRow { double C, S, TH, pad; };
typedef const __attribute__((address_space(3))) SineRow* sine_tab_t;

// sqrt(q) for q in {0} U [2^-53, 1]: adding DBL_MIN leaves every q > 0 as it is, keeps q < 0 negative (rsq -> NaN: |x| > 1) and
// makes the rsq of q = 0 finite, so that g = q * y = 0 needs no select.  v_rsq_f64 seed (2^-24.4 on gfx950,
// scripts/probe/rcp_rsq_probe.py) + one coupled Goldschmidt step: 3e-15 relative — the arc's other terms carry more.
__device__ __forceinline__ double sqrt_unit(double q) {
    const double y = __builtin_amdgcn_rsq(q + 2.2250738585072014e-308);
    const double g = q * y, h = 0.5 * y;
    const double r = __fma_rn(-h, g, 0.5);
    return __fma_rn(g, r, g);
}
// acos(a) for 0 <= a <= 1 given g = sqrt(1 - a^2).  CLAMP = false: the caller guarantees a <= 1 or NaN (pair mode: the arc is
// only evaluated where the threshold lies strictly inside the window, so |thr - tavg| < alpha; a NaN a gives k = 0), and the row
// index needs no bound; the generic form may be handed |r| up to 2 (the reference's heating form, NaN there) and keeps it.
template <bool CLAMP = true>
__device__ __forceinline__ double sine_theta(double a, double g, sine_tab_t tab) {
    static_assert(SINE_SCALE == 256, "the index trick below adds 2^44 = 2^52 / 256");
    const double u = __builtin_fmin(a, g), w = __builtin_fmax(a, g);
    const double t = u + 17592186044416.0;                                      // + 2^44 (ulp 2^-8): the sum's low word is round(256 u)
    uint32_t k = (uint32_t)__double2loint(t);
    if (CLAMP) k = k < (uint32_t)(SINE_ROWS - 1) ? k : (uint32_t)(SINE_ROWS - 1);   // NaN / out-of-range operands stay inside the table
    // the two cases of a k are neighbouring rows (row 2k: a <= g, row 2k + 1: a > g): row address = 64 k + base + (0 | 32)
    const uint32_t base = (uint32_t)(uintptr_t)tab;
    const uint32_t half = (a <= g) ? base : base + (uint32_t)sizeof(SineRow);       // (v_mov of the second base + v_cndmask)
    sine_tab_t row = (sine_tab_t)(uintptr_t)lshl_add(half, k, 6);
    const double C = row->C, S = row->S, TH = row->TH;
    const double delta = __fma_rn(u, C, w * S);
    const double t2 = delta * delta;
    const double p = fma_vsv(t2, 0.075, 0.16666666666666666);
    return __fma_rn(delta, __fma_rn(t2, p, 1.0), TH);            // TH + delta (1 + t2 p): four instructions from delta
}
__device__ __forceinline__ double sine_arc(double d, double x, double alpha, sine_tab_t tab) {
    const double HALF_PI = 1.57079632679489661923;
    const double a = fabs(x);
    const double g = sqrt_unit(__fma_rn(-a, a, 1.0));          // 1 - a^2 with ONE rounding
    const double th = sine_theta(a, g, tab);                   // acos(|x|)
    const double ac = HALF_PI - copysign(HALF_PI - th, x);     // acos(x)
    return __fma_rn(d, ac, alpha * g);
}
// (tmin, tmax) pairs: tavg is the mid-range, so z = r = (thr - tavg) / alpha =: d / alpha in both forms and, with a = |d| / alpha,
//   cooling part = max(tavg - thr, 0) + [tmin < thr < tmax] alpha F(a),     heating part = max(thr - tavg, 0) + [..] alpha F(a),
//   F(a) = (sqrt(1 - a^2) - a acos(a)) / pi        (F(-a) = F(a) + a folds the sign of z into the max() term;
// the max() term alone is the reference's value on either side of the window).
// F has ONE singularity on [0, 1], the (1 - a)^(3/2) branch point at a = 1:  F(a) = (1 - a)^(3/2) P2(a)  with P2 analytic for
// |1 - a| < 2 — and nearly constant on [0, 1] (0.300 .. 0.318).  P2 is a table of cubics (scripts/fit/sine_p2_fit.py: mpmath,
// Chebyshev nodes), so the arc is ONE square root and one 32-byte LDS row, where acos from sine_theta and g - a theta took 23 fp64 /
// integer instructions (and the degree-14 asin of round 2, 44).  It is better conditioned, too: no cancellation g - a theta next
// to a = 1.  Three layouts of the table were built in round 3: rows in a (centred cubics, 17 + rsq), rows in x = 4 (1 - a)
// (absolute cubics, 13 + rsq) and — the one in the tree — rows in th = 2 sqrt(1 - a), 512 on [0, 2] (11 + rsq; sine_pair_g).
constexpr int SINE_P2_N = 512;                                   // = AFHIP_SINE_P2_N of the generated table
constexpr int SINE_P2_BYTES = (SINE_P2_N + 1) * 32 + 32;         // (+ a pad row: multiple of 64 bytes)
struct alignas(32) SineP2Row { double c0, c1, c2, c3; };
typedef const __attribute__((address_space(3))) SineP2Row* sine_p2_t;
// max(x, DBL_MIN): 1 - a may come out 0 or a rounding error below it (a = |d| / alpha next to 1): t = sqrt(.) is then ~1e-154,
// F = 0 — the limit — without a NaN from rsq; one v_max_f64 (a NaN operand would give DBL_MIN too: the cubic still carries it)
__device__ __forceinline__ double max_tiny(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(x), "s"(2.2250738585072014e-308));
    return d;
#else
    return x > 2.2250738585072014e-308 ? x : 2.2250738585072014e-308;
#endif
}
// one threshold strictly inside a (tmin, tmax) pair: alpha F(a) = u sqrt(om) P2(1 - om), u = alpha - |thr - tavg| in (0, alpha],
// om = u / alpha = 1 - a — without the reciprocal of alpha: with rng = 2 alpha, u2 = 2 u and the seed z ~ rsq(u2 rng),
//   th = u2 z (3 - (u2 rng) z^2) = 2 sqrt(om)   (one Newton step on the seed, in product form),   th^2 = 4 om,
//   alpha F = (u2 th) H(th),   H(th) = P2(1 - th^2 / 4) / 4: cubic rows in th itself (scripts/fit/sine_p2_fit.py),
// returned as the two factors w = su2 th and p = H(th).  11 VALU + rsq; `su2` is u2 carrying the sign the caller wants on the
// product (only |su2| enters v and th); `rng` may come with either sign (the sine-only lean form hands over tmax - tmin of an UNORDERED
// pair: only |rng| is read).
__device__ __forceinline__ void sine_pair_g(double su2, double rng, sine_p2_t tab, double& w, double& p) {
    static_assert(SINE_P2_N == 512, "the index trick adds 2^44 = 2^52 / 256");
    double v;                                                    // |su2| rng + tiny: u may round to 0 (thr one ulp inside the window); rsq(0) = inf
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_f64 %0, |%1|, |%2|, %3" : "=v"(v) : "v"(su2), "v"(rng), "s"(1e-300));
#else
    v = __builtin_fma(__builtin_fabs(su2), __builtin_fabs(rng), 1e-300);
#endif
    const double z = __builtin_amdgcn_rsq(v);
    const double a = v * z;
    const double e = __fma_rn(-a, z, 3.0);
    const double th = (__builtin_fabs(su2) * z) * e;
    // the table is indexed by th itself (H(th) = G(th^2), 512 cubics on [0, 2]: no x = th * th; the first table of round 3 was in x)
    const double ti = th + 17592186044416.0;                     // + 2^44 (ulp 2^-8): the sum's low word is k = round(256 th)
    const uint32_t k = (uint32_t)__double2loint(ti);
    sine_p2_t row = (sine_p2_t)(uintptr_t)lshl_add((uint32_t)(uintptr_t)tab, k, 5);
    const double c0 = row->c0, c1 = row->c1, c2 = row->c2, c3 = row->c3;
    p = __fma_rn(__fma_rn(__fma_rn(c3, th, c2), th, c1), th, c0);
    w = su2 * th;
}
// cooling part for one threshold (nb_kernels.py:224-236); alpha = rng / 2, inv_rng ~ 1 / rng (faithful; only read when `inside`)
__device__ __forceinline__ double sine_cool(double thr, double thr2, bool inside, double tmin, double tmax, double tavg, double alpha,
                                            double inv_rng, sine_tab_t tab) {
    const double INV_PI = 0.31830988618379067154;
    if (thr <= tmin) return tavg - thr;
    if (inside) {
        const double z = (thr2 - tmax - tmin) * inv_rng;         // thr2 = 2 thr; inf / NaN operands give NaN here too
        return sine_arc(tavg - thr, z, alpha, tab) * INV_PI;
    }
    return 0.0;
}
// heating part (nb_kernels.py:238-249); inv_alpha ~ 2 / rng
__device__ __forceinline__ double sine_heat(double thr, bool inside, double tmin, double tmax, double tavg, double alpha, double inv_alpha,
                                            sine_tab_t tab) {
    const double INV_PI = 0.31830988618379067154;
    if (thr >= tmax) return thr - tavg;
    if (inside) {
        const double d = thr - tavg;
        return sine_arc(d, -(d * inv_alpha), alpha, tab) * INV_PI;
    }
    return 0.0;
}

}  // namespace afhip
