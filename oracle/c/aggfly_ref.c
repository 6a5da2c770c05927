/* CPU oracle, plain C.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Restates the reference's compiled hot loops so that mid-size parity checks and
 * the CPU-baseline timing do not run at Python speed:
 *
 *   ref_block_stat     <- _block_stat     aggfly/aggregate/nb_kernels.py:121-155
 *   ref_block_dd       <- _block_dd       aggfly/aggregate/nb_kernels.py:158-179
 *   ref_block_bins     <- _block_bins     aggfly/aggregate/nb_kernels.py:182-199
 *   ref_block_sine_dd  <- _block_sine_dd  aggfly/aggregate/nb_kernels.py:202-251
 *   ref_power          <- _power          aggfly/dataset/dataset.py:527-543
 *   ref_scatter_block  <- _scatter_block  aggfly/aggregate/spatial.py:181-186
 *
 * Same loop nest as the numba kernels: parallel over grid rows iy (numba prange,
 * `nb_kernels.py:125`), then ix, g, (d), k; float64 accumulators; result stored in the
 * input dtype (`nb_kernels.py:257-268`).  Built with -ffp-contract=off: numba compiles
 * with fastmath=False (`nb_kernels.py:120`), so no fused multiply-adds.
 *
 * The time-major strided walk (stride NY*NX per step) is the reference's own access
 * pattern (`nb_kernels.py:36-41` notes it loses to numpy on large blocks); it is kept,
 * because this file is also the "port" CPU baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DEFINE_KERNELS(T, SFX)                                                              \
void ref_block_stat_##SFX(const T* cube, int64_t nt, int64_t ny, int64_t nx,                 \
                          const int64_t* bounds, int64_t G, int code, T* out) {             \
    (void)nt;                                                                                \
    _Pragma("omp parallel for schedule(static)")                                             \
    for (int64_t iy = 0; iy < ny; ++iy)                                                      \
        for (int64_t ix = 0; ix < nx; ++ix)                                                  \
            for (int64_t g = 0; g < G; ++g) {                                                \
                int64_t lo = bounds[g], hi = bounds[g + 1];                                  \
                int64_t n = 0; double s = 0.0, mn = INFINITY, mx = -INFINITY; int hasnan = 0;\
                for (int64_t k = lo; k < hi; ++k) {                                          \
                    double v = (double)cube[(k * ny + iy) * nx + ix];                        \
                    if (isnan(v)) hasnan = 1;                                                \
                    else { s += v; n += 1; if (v < mn) mn = v; if (v > mx) mx = v; }         \
                }                                                                            \
                double r;                                                                    \
                if (hi == lo) r = NAN;                                                       \
                else if (code == 4) r = n > 0 ? s / (double)n : NAN;                         \
                else if (hasnan) r = NAN;                                                    \
                else if (code == 0) r = s / (double)n;                                       \
                else if (code == 1) r = s;                                                   \
                else if (code == 2) r = mn;                                                  \
                else r = mx;                                                                 \
                out[(g * ny + iy) * nx + ix] = (T)r;                                         \
            }                                                                                \
}                                                                                            \
void ref_block_dd_##SFX(const T* cube, int64_t nt, int64_t ny, int64_t nx,                   \
                        const int64_t* bounds, int64_t G, const double* ddargs, int64_t D,  \
                        T* out) {                                                            \
    (void)nt;                                                                                \
    _Pragma("omp parallel for schedule(static)")                                             \
    for (int64_t iy = 0; iy < ny; ++iy)                                                      \
        for (int64_t ix = 0; ix < nx; ++ix)                                                  \
            for (int64_t g = 0; g < G; ++g) {                                                \
                int64_t lo = bounds[g], hi = bounds[g + 1];                                  \
                for (int64_t d = 0; d < D; ++d) {                                            \
                    double t0 = ddargs[d * 3], t1 = ddargs[d * 3 + 1];                       \
                    double base = ddargs[d * 3 + 2] == 0 ? t0 : t1;                          \
                    double acc = 0.0; int hasnan = 0;                                        \
                    for (int64_t k = lo; k < hi; ++k) {                                      \
                        double v = (double)cube[(k * ny + iy) * nx + ix];                    \
                        if (isnan(v)) hasnan = 1;                                            \
                        else if (v > t0 && v < t1) {                                         \
                            double av = v - base; if (av < 0.0) av = -av; acc += av;         \
                        }                                                                    \
                    }                                                                        \
                    out[((g * ny + iy) * nx + ix) * D + d] = (T)((hasnan || hi == lo) ? NAN : acc); \
                }                                                                            \
            }                                                                                \
}                                                                                            \
void ref_block_bins_##SFX(const T* cube, int64_t nt, int64_t ny, int64_t nx,                 \
                          const int64_t* bounds, int64_t G, const double* ddargs, int64_t D,\
                          T* out) {                                                          \
    (void)nt;                                                                                \
    _Pragma("omp parallel for schedule(static)")                                             \
    for (int64_t iy = 0; iy < ny; ++iy)                                                      \
        for (int64_t ix = 0; ix < nx; ++ix)                                                  \
            for (int64_t g = 0; g < G; ++g) {                                                \
                int64_t lo = bounds[g], hi = bounds[g + 1];                                  \
                for (int64_t d = 0; d < D; ++d) {                                            \
                    double t0 = ddargs[d * 3], t1 = ddargs[d * 3 + 1];                       \
                    double c = 0.0;                                                          \
                    for (int64_t k = lo; k < hi; ++k) {                                      \
                        double v = (double)cube[(k * ny + iy) * nx + ix];                    \
                        if (v > t0 && v < t1) c += 1.0;                                      \
                    }                                                                        \
                    out[((g * ny + iy) * nx + ix) * D + d] = (T)(hi == lo ? NAN : c);        \
                }                                                                            \
            }                                                                                \
}                                                                                            \
void ref_block_sine_dd_##SFX(const T* cube, int64_t nt, int64_t ny, int64_t nx,              \
                             const int64_t* bounds, int64_t G, const double* ddargs,        \
                             int64_t D, T* out) {                                            \
    (void)nt;                                                                                \
    const double PI = 3.14159265358979323846;                                                \
    _Pragma("omp parallel for schedule(static)")                                             \
    for (int64_t iy = 0; iy < ny; ++iy)                                                      \
        for (int64_t ix = 0; ix < nx; ++ix)                                                  \
            for (int64_t g = 0; g < G; ++g) {                                                \
                int64_t lo = bounds[g], hi = bounds[g + 1];                                  \
                int64_t n = 0; double s = 0.0, tmax = -INFINITY, tmin = INFINITY; int hasnan = 0; \
                for (int64_t k = lo; k < hi; ++k) {                                          \
                    double v = (double)cube[(k * ny + iy) * nx + ix];                        \
                    if (isnan(v)) hasnan = 1;                                                \
                    else { s += v; n += 1; if (v > tmax) tmax = v; if (v < tmin) tmin = v; } \
                }                                                                            \
                for (int64_t d = 0; d < D; ++d) {                                            \
                    T* o = &out[((g * ny + iy) * nx + ix) * D + d];                          \
                    if (hasnan || n == 0) { *o = (T)NAN; continue; }                         \
                    double tavg = s / (double)n, kind = ddargs[d * 3 + 2], val = 0.0;        \
                    for (int j = 0; j < 2; ++j) {                                            \
                        double thr = ddargs[d * 3 + j], part;                                \
                        if (kind == 0) {                                                     \
                            if (thr <= tmin) part = tavg - thr;                              \
                            else if (thr < tmax && tmin < thr) {                             \
                                double rng = tmax - tmin;                                    \
                                double a = acos((2.0 * thr - tmax - tmin) / rng);            \
                                part = ((tavg - thr) * a + rng * sin(a) / 2.0) / PI;         \
                            } else part = 0.0;                                               \
                            val += (j == 0) ? part : -part;                                  \
                        } else {                                                             \
                            if (thr >= tmax) part = thr - tavg;                              \
                            else if (thr < tmax && tmin < thr) {                             \
                                double alpha = (tmax - tmin) / 2.0;                          \
                                double r = (thr - tavg) / alpha;                             \
                                double at = atan(r / sqrt(1.0 - r * r));                     \
                                part = (1.0 / PI) * ((thr - tavg) * (at + PI / 2.0) + alpha * cos(at)); \
                            } else part = 0.0;                                               \
                            val += (j == 0) ? -part : part;                                  \
                        }                                                                    \
                    }                                                                        \
                    *o = (T)val;                                                             \
                }                                                                            \
            }                                                                                \
}                                                                                            \
void ref_power_##SFX(const T* x, int64_t n, double e, T* out) {                              \
    _Pragma("omp parallel for schedule(static)")                                             \
    for (int64_t i = 0; i < n; ++i) out[i] = (T)pow((double)x[i], e);                        \
}

DEFINE_KERNELS(double, f64)
DEFINE_KERNELS(float, f32)

/* _scatter_block, spatial.py:181-186: out[r, t] += w[j] * block[cell[j], t], entries in
 * table order (np.add.at is a sequential unbuffered loop).  Parallel over time columns:
 * each output element still receives its adds in entry order, so results equal the
 * sequential loop's bit for bit. */
void ref_scatter_block(const double* block, int64_t n_cells, int64_t nt,
                       const int64_t* region_idx, const int64_t* cell_idx, const double* w,
                       int64_t nnz, int64_t n_regions, double* out) {
    (void)n_cells;
    memset(out, 0, (size_t)(n_regions * nt) * sizeof(double));
    #pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < nt; ++t)
        for (int64_t j = 0; j < nnz; ++j) {
            double contrib = w[j] * block[cell_idx[j] * nt + t];
            out[region_idx[j] * nt + t] += contrib;
        }
}
