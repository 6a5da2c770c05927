/* Calling the engine from plain C through include/aggfly_hip.h — no Python, no torch.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/c_abi_demo.c -Iinclude -I/opt/rocm/include \
 *       -Laggfly_amd -laggfly_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/aggfly_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo
 *   (the HIP runtime is only used here for hipMalloc / hipMemcpy of the demo's own buffers)
 *
 * A 48-step, 2x3-cell cube; daily mean -> square -> sum over the two days, and degree days
 * [10, 30); two regions.  The expected numbers are computed on the host in the same order.
 * The plan runs twice — on scratch the library allocates itself, and on a caller-owned workspace
 * sized by afhip_plan_run_workspace_bytes (nothing on the run path allocates then) — and the
 * drop-in for _block_dd (nb_kernels.py:158-179) is called with D = 20 threshold rows, more than
 * one pass over the cube holds: the library loops the passes, as the reference loops ddargs. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "aggfly_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, afhip_last_error()); return 1; } } while (0)

int main(void) {
    enum { T = 48, C = 6, K = 2, R = 2, P = 1 };
    if (afhip_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 2; }
    double cube[T * C];
    for (int k = 0; k < T; ++k)
        for (int c = 0; c < C; ++c) cube[k * C + c] = 12.0 + 9.0 * sin(0.26 * k + c) + 0.5 * c;
    cube[7 * C + 4] = NAN;                                   /* one missing value: cell 4 drops out */

    int64_t inner[] = {0, 24, 48}, outer[] = {0, 2};
    afhip_column cols[K] = {{0}};
    cols[0].inner = AFHIP_MEAN; cols[0].transform = AFHIP_TF_POW; cols[0].transform_arg = 2.0; cols[0].outer = AFHIP_SUM;
    cols[1].inner = AFHIP_DD; cols[1].inner_args[0] = 10; cols[1].inner_args[1] = 30; cols[1].inner_args[2] = 0; cols[1].outer = AFHIP_SUM;
    afhip_plan_desc d = {0};
    d.T = T; d.n_cells = C; d.dtype = AFHIP_F64; d.K = K; d.G1 = 2; d.inner_bounds = inner; d.P = P; d.outer_bounds = outer;
    d.columns = cols; d.exact_order = 1;
    afhip_plan* plan = NULL;
    CHECK(afhip_plan_create(&d, &plan));

    /* region 0 = cells {0,1,4}, region 1 = cells {2,3,4,5}; cell 4 is shared */
    int64_t indptr[] = {0, 3, 7}, ccols[] = {0, 1, 4, 2, 3, 4, 5};
    double w[] = {0.5, 0.25, 0.25, 0.3, 0.3, 0.2, 0.2};
    afhip_csr* csr = NULL;
    CHECK(afhip_csr_create(indptr, ccols, w, R, 7, C, &csr));

    double *d_cube, *d_num, *d_den, *d_res, num[K * R * P], den[R * P], res[K * R * P];
    if (hipMalloc((void**)&d_cube, sizeof cube) || hipMalloc((void**)&d_num, sizeof num) ||
        hipMalloc((void**)&d_den, sizeof den) || hipMalloc((void**)&d_res, sizeof res)) return 3;
    hipMemcpy(d_cube, cube, sizeof cube, hipMemcpyHostToDevice);
    CHECK(afhip_plan_run(plan, d_cube, csr, d_num, d_den, d_res, NULL, NULL, 0, NULL, NULL));      /* plan-owned scratch */
    hipDeviceSynchronize();
    double first[K * R * P];
    hipMemcpy(first, d_res, sizeof first, hipMemcpyDeviceToHost);
    /* the same run on a workspace this program owns (hipMalloc returns 256-byte aligned blocks) */
    const int64_t ws_bytes = afhip_plan_run_workspace_bytes(plan, csr);
    void* d_ws = NULL;
    if (ws_bytes <= 0 || hipMalloc(&d_ws, (size_t)ws_bytes)) return 3;
    hipMemset(d_res, 0, sizeof res);
    CHECK(afhip_plan_run(plan, d_cube, csr, d_num, d_den, d_res, NULL, d_ws, ws_bytes, NULL, NULL));
    if (afhip_plan_run(plan, d_cube, csr, d_num, d_den, d_res, NULL, d_ws, ws_bytes - 256, NULL, NULL) != AFHIP_E_INVALID) {
        fprintf(stderr, "a workspace that is too small was not refused\n"); return 1;
    }
    hipDeviceSynchronize();
    hipMemcpy(num, d_num, sizeof num, hipMemcpyDeviceToHost);
    hipMemcpy(den, d_den, sizeof den, hipMemcpyDeviceToHost);
    hipMemcpy(res, d_res, sizeof res, hipMemcpyDeviceToHost);

    /* host restatement, same operation order as the reference */
    double y[K][C];
    for (int c = 0; c < C; ++c) {
        double sq = 0, dd = 0; int bad = 0;
        for (int g = 0; g < 2; ++g) {
            double s = 0, a = 0; int nan_ = 0;
            for (int k = inner[g]; k < inner[g + 1]; ++k) {
                double v = cube[k * C + c];
                if (isnan(v)) { nan_ = 1; continue; }
                s += v;
                if (v > 10 && v < 30) a += fabs(v - 10);
            }
            if (nan_) bad = 1;
            double m = s / 24.0;
            sq += m * m; dd += a;
        }
        y[0][c] = bad ? NAN : sq; y[1][c] = bad ? NAN : dd;
    }
    int fails = 0;
    for (int r = 0; r < R; ++r) {
        double dn = 0, nm[K] = {0, 0};
        for (int j = indptr[r]; j < indptr[r + 1]; ++j) {
            int c = (int)ccols[j], ok = !isnan(y[0][c]) && !isnan(y[1][c]);
            dn += w[j] * (ok ? 1.0 : 0.0);
            for (int k = 0; k < K; ++k) nm[k] += w[j] * (ok ? y[k][c] : 0.0);
        }
        for (int k = 0; k < K; ++k) {
            double want = nm[k] / dn, got = res[k * R + r];
            printf("region %d column %d: gpu %.15g  host %.15g  (num %.15g den %.15g)\n", r, k, got, want, num[k * R + r], den[r]);
            if (fabs(got - want) > 1e-12 * fabs(want)) ++fails;
        }
    }
    for (int i = 0; i < K * R * P; ++i)
        if (!(first[i] == res[i])) { printf("plan-owned and caller-owned scratch disagree at %d\n", i); ++fails; }

    /* _block_dd with D = 20 rows (t0, t1, flag): out[G][cell][D] in the cube's dtype, two daily groups */
    enum { D = 20, G = 2 };
    double ddargs[D * 3], *d_out, out[G * C * D];
    for (int q = 0; q < D; ++q) { ddargs[3 * q] = -2.0 + 1.5 * q; ddargs[3 * q + 1] = 9.0 + 1.5 * q; ddargs[3 * q + 2] = q % 2; }
    if (hipMalloc((void**)&d_out, sizeof out)) return 3;
    CHECK(afhip_group_dd(d_cube, AFHIP_F64, T, C, inner, G, ddargs, D, d_out, NULL));
    hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost);
    for (int g = 0; g < G; ++g)
        for (int c = 0; c < C; ++c)
            for (int q = 0; q < D; ++q) {
                const double t0 = ddargs[3 * q], t1 = ddargs[3 * q + 1], base = ddargs[3 * q + 2] == 0.0 ? t0 : t1;
                double a = 0; int nan_ = 0;
                for (int k = inner[g]; k < inner[g + 1]; ++k) {
                    double v = cube[k * C + c];
                    if (isnan(v)) { nan_ = 1; break; }
                    if (v > t0 && v < t1) a += fabs(v - base);
                }
                const double got = out[(g * C + c) * D + q];
                if (nan_ ? !isnan(got) : !(got == a)) { printf("group_dd g=%d c=%d d=%d: gpu %.17g host %.17g\n", g, c, q, got, nan_ ? NAN : a); ++fails; }
            }
    printf("afhip_group_dd with D = %d rows: %s\n", D, fails ? "MISMATCH" : "bit-exact");
    afhip_plan_destroy(plan);
    afhip_csr_destroy(csr);
    printf(fails ? "MISMATCH\n" : "C ABI OK\n");
    return fails ? 1 : 0;
}
