/* aggfly_hip.h — C ABI of the MI355X (gfx950) engine for aggfly's aggregate_dataset() hot path.
 *
 * The reference (dylanhogan/aggfly v0.2.0) is pure Python; it has no FFI layer.  The
 * seams this library sits behind are its numba kernels and numpy block functions, which
 * are already C-shaped: contiguous arrays in, caller-owned output, no exceptions
 * (SURVEY.md §8b).  Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - `*_dev` pointers are device (HBM) addresses owned by the caller; tables marked
 *     HOST are read on the host during the call and may be freed afterwards;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     enqueued on it and the call returns without synchronising unless stated;
 *   - cubes are time-major: element (k, cell) at cube[k * n_cells + cell], cell =
 *     iy * NX + ix, exactly the (time, y, x) block the reference hands its kernels
 *     (aggfly/aggregate/nb_kernels.py:280);
 *   - every function returns 0 on success or a negative AFHIP_E_* code;
 *     afhip_last_error() gives the message for the calling thread.  Like the numba
 *     kernels, the GPU kernels themselves never raise: argument errors are reported
 *     before anything is launched.
 */
#ifndef AGGFLY_HIP_H
#define AGGFLY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFHIP_ABI_VERSION 4

/* status codes */
#define AFHIP_OK            0
#define AFHIP_E_INVALID    -1   /* bad argument (the Python host raises ValueError)      */
#define AFHIP_E_HIP        -2   /* a HIP runtime call failed                              */
#define AFHIP_E_UNSUPPORTED -3  /* valid plan the fused kernels cannot express            */
#define AFHIP_E_NOMEM      -4

/* element type of a climate cube */
#define AFHIP_F32 0
#define AFHIP_F64 1

/* reducer codes: 0..4 are _STAT_CODE of aggfly/aggregate/nb_kernels.py:33 */
#define AFHIP_MEAN     0
#define AFHIP_SUM      1
#define AFHIP_MIN      2
#define AFHIP_MAX      3
#define AFHIP_NANMEAN  4
#define AFHIP_DD       5   /* _block_dd      nb_kernels.py:158-179 */
#define AFHIP_BINS     6   /* _block_bins    nb_kernels.py:182-199 */
#define AFHIP_SINE_DD  7   /* _block_sine_dd nb_kernels.py:202-251 */
#define AFHIP_IDENTITY 8   /* outer level of a single-level plan: pass the inner value through */

/* element-wise transforms between the two temporal levels */
#define AFHIP_TF_NONE  0
#define AFHIP_TF_POW   1   /* Dataset.power / _power  aggfly/dataset/dataset.py:442-473,527-543 */
#define AFHIP_TF_HINGE 2   /* Dataset.spline hinge (x>knot)*(x-knot), dataset.py:475-481 (knot 20) */
#define AFHIP_TF_INTER 3   /* Dataset.interact / _interact: times a second array, dataset.py:483-518,547-563 */

/* The reference stores every step's output in its input dtype (nb_kernels.py:257-262), so on a
 * float32 cube its intermediates are float32.  This engine keeps float64 throughout unless a
 * column asks for the reference's roundings: */
#define AFHIP_ROUND_INNER 1   /* round the inner reducer's value to float32                  */
#define AFHIP_ROUND_HINGE 2   /* evaluate the hinge / inter transform in float32             */
#define AFHIP_ROUND_FINAL 4   /* round the column's final (outer) value to float32           */

const char* afhip_last_error(void);
int afhip_abi_version(void);
/* What this build of the library holds, as text into buf ("menu=full variants=426 arms=0 region_fused_twins=150 abi=4"); returns the
 * bytes needed.  menu: full = every kernel the planner can pick; arms = + the tuning arms a `tuning` hint can name (make MENU=arms). */
int afhip_build_info(char* buf, int buf_len);
/* Number of visible GPUs (hipGetDeviceCount); 0 when there is none. */
int afhip_device_count(void);
/* Name/arch/CU count of device `dev` into caller buffers (arch e.g. "gfx950"). */
int afhip_device_info(int dev, char* name, int name_len, char* arch, int arch_len, int* n_cus,
                      int64_t* hbm_bytes);

/* ------------------------------------------------------------------------------------
 * Grouped temporal reducers — drop-in for the numba kernels.
 *
 *   afhip_group_stat     replaces _block_stat(cube, bounds, code, out)        nb_kernels.py:121-155
 *   afhip_group_dd       replaces _block_dd(cube, bounds, ddargs, out)        nb_kernels.py:158-179
 *   afhip_group_bins     replaces _block_bins(cube, bounds, ddargs, out)      nb_kernels.py:182-199
 *   afhip_group_sine_dd  replaces _block_sine_dd(cube, bounds, ddargs, out)   nb_kernels.py:202-251
 *
 * cube_dev   [T, n_cells] of `dtype`;  bounds HOST int64[G+1], group g = steps
 *            [bounds[g], bounds[g+1]) (empty groups allowed -> NaN);
 * ddargs     HOST double[D*3] rows (t0, t1, flag) as in the reference;
 * out_dev    [G, n_cells] (stat) or [G, n_cells, D] of `dtype` — the reference's output
 *            layout and dtype (accumulation is float64, the store rounds to `dtype`,
 *            nb_kernels.py:257-268).
 * Numerics: stat / dd / bins are bit-identical to the reference's loops (same operation order, float64 accumulators, no contraction).
 * sine_dd evaluates the same closed forms by other means (table acos, rsq + Newton, cubic arc tables):
 *     |out - exact| <= 1e-10 |exact| + 1e-12 max(tmax - tmin, |exact|),   exact = the closed forms of nb_kernels.py:202-251 in exact arithmetic
 * (tests/golden/sine_dd_fixtures.json at 50 digits; the reference's own libm evaluation sits at 3.5e-9 relative / 3e-14 of the window's scale next to
 * the window's edges, the (tmin, tmax)-pair forms here at 1.7e-11 / 5.5e-15: profiles/r04_sine_accuracy.json).
 * D is unbounded, like the reference's loop over ddargs rows (nb_kernels.py:166,190,215): more rows than one pass over the
 * cube holds (16) run as consecutive passes inside the call, each writing its own columns of out_dev.
 * These calls build a temporary plan, own their scratch and synchronise `stream` before they return.
 * ---------------------------------------------------------------------------------- */
int afhip_group_stat(const void* cube_dev, int dtype, int64_t T, int64_t n_cells,
                     const int64_t* bounds, int64_t G, int code, void* out_dev, void* stream);
int afhip_group_dd(const void* cube_dev, int dtype, int64_t T, int64_t n_cells,
                   const int64_t* bounds, int64_t G, const double* ddargs, int64_t D,
                   void* out_dev, void* stream);
int afhip_group_bins(const void* cube_dev, int dtype, int64_t T, int64_t n_cells,
                     const int64_t* bounds, int64_t G, const double* ddargs, int64_t D,
                     void* out_dev, void* stream);
int afhip_group_sine_dd(const void* cube_dev, int dtype, int64_t T, int64_t n_cells,
                        const int64_t* bounds, int64_t G, const double* ddargs, int64_t D,
                        void* out_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * Region x cell weights, uploaded once as CSR.
 *
 * Replaces the COO triplets of _weight_triplets (aggfly/aggregate/spatial.py:157-178).
 * Rows are regions in sorted-region-id order; inside a row the entries keep the weights
 * table's order, so sums run in the order np.add.at visits them (spatial.py:185).
 * indptr HOST int64[R+1]; cols HOST int64[nnz] (cell positions, 0 <= col < n_cells);
 * w HOST double[nnz].  The handle owns device copies; free with afhip_csr_destroy.
 * ---------------------------------------------------------------------------------- */
typedef struct afhip_csr afhip_csr;
int afhip_csr_create(const int64_t* indptr, const int64_t* cols, const double* w, int64_t R,
                     int64_t nnz, int64_t n_cells, afhip_csr** out);
void afhip_csr_destroy(afhip_csr* csr);
/* Devices.  A handle (afhip_csr, afhip_plan) belongs to the device that was current (hipGetDevice) on the calling thread
 * when it was created; its tables and scratch live there.  Every entry point that takes a handle makes that device current
 * for the duration of the call and restores the caller's afterwards, so a worker thread that still sits on device 0 (the
 * reference calls its kernels from dask's thread pool, nb_kernels.py:271-305) drives a handle of device r correctly.  What
 * must match is the data: afhip_plan_run / _run_temporal refuse (AFHIP_E_INVALID) a cube or a CSR that lives on another
 * device than the plan.  Entry points without a handle (afhip_group_*, afhip_transform, afhip_place_box, ...) run on the
 * device that owns their array argument.  `stream` must be a stream of that device (or NULL).  The same holds for every other
 * device pointer handed to a plan: the second cube of afhip_plan_bind_inter, the outputs (num / den / res / cells) and a
 * caller-owned workspace are refused with AFHIP_E_INVALID when the runtime knows them to live on another device.
 * afhip_csr_device / afhip_plan_device: the device a handle was created on (-1 for NULL).
 *
 * Threads.  Entry points without a handle are re-entrant.  An afhip_csr is immutable after creation (the run tables it
 * caches on first use are built under its own lock): share ONE CSR between any number of threads and plans.  An afhip_plan is
 * NOT re-entrant: it owns scratch in HBM, per-launch event pairs and the second-cube bindings, so two calls must never be
 * inside the same plan at once — give every worker thread its own plan (plans are cheap: tables of a few KB) or serialise the
 * calls; `aggfly_amd/engine.py` holds a per-plan lock around bind + run.  Two plans on two streams run concurrently. */
int afhip_csr_device(const afhip_csr* csr);

/* Replaces _scatter_block(block, region_idx, cell_idx, w_vals, n_regions)
 * (spatial.py:181-186): out[r, t] = sum_j w[j] * block[col[j], t], float64, entries in
 * table order, products rounded before each add (no fused multiply-add).
 * block_dev [n_cells, nt] float64; out_dev [R, nt] float64. */
int afhip_scatter_block(const afhip_csr* csr, const double* block_dev, int64_t nt,
                        double* out_dev, void* stream);

/* Element-wise transforms on a whole array — what Dataset.power / Dataset.spline / Dataset.interact do per dask block
 * (`_power` np.power(block, exp) dataset.py:527-543; hinge (x > knot) * (x - knot) dataset.py:475-481; `_interact`
 * np.multiply(block, other) dataset.py:547-563).  x_dev [n] of x_dtype -> out_dev [n] of out_dtype (AFHIP_F32 / AFHIP_F64);
 * transform AFHIP_TF_POW (arg = exponent: integer exponents through the correctly rounded double-double chain, others
 * through pow), AFHIP_TF_HINGE (arg = knot) or AFHIP_TF_INTER (other_dev [n] of other_dtype; arg unused).  Arithmetic is
 * float64 and the store rounds to out_dtype, except that a float32 -> float32 hinge / product is evaluated in float32
 * like numpy's.  In-place (out_dev == x_dev with equal dtypes) is allowed. */
int afhip_transform(const void* x_dev, int x_dtype, int64_t n, int transform, double arg,
                    const void* other_dev, int other_dtype, void* out_dev, int out_dtype, void* stream);

/* Measuring aid (no counterpart in the reference): the streaming-read ceiling of THIS box for a time-major cube — a kernel with the
 * temporal kernels' access pattern and no arithmetic (8 bytes per lane, single-wave workgroups, four non-temporal row loads in
 * flight; rows too short to fill the card are read in time chunks, as the temporal kernels do) launched `launches` times back to back over
 * cube_dev [T rows of row_bytes bytes, row_bytes % 8 == 0]; ms_out[i] = the i-th launch's duration by HIP events.  Synchronises `stream`.  bench.py reports T * row_bytes / min(ms) beside the 8 TB/s spec peak. */
int afhip_read_probe(const void* cube_dev, int64_t T, int64_t row_bytes, int launches, float* ms_out, void* stream);

/* Ingestion helper (no counterpart in the reference, whose chunks are assembled by dask on the host,
 * aggfly/dataset/dataset.py:697-728): copies the part [st, st+nt) x [sy, sy+ny) x [sx, sx+nx) of a
 * decoded chunk [*, by, bx] (contiguous, in HBM) into the time-major cube [*, NY, NX] at (t0, y0, x0).
 * elem_size 2, 4 or 8 bytes; both pointers are device memory; nothing is retained. */
int afhip_place_box(const void* chunk_dev, void* cube_dev, int elem_size,
                    int64_t by, int64_t bx, int64_t st, int64_t sy, int64_t sx,
                    int64_t nt, int64_t ny, int64_t nx,
                    int64_t NY, int64_t NX, int64_t t0, int64_t y0, int64_t x0, void* stream);

/* Chunk decode in HBM ("or GPU-side decode"; the reference decodes on host threads inside its dask
 * graph, aggfly/dataset/dataset.py:697-728).  Blosc-1 chunks whose streams are LZ4 cross PCIe compressed; the host parses
 * the containers (afcodec_blosc_lz4_plan, include/aggfly_codec.h) into the two record lists below, which travel to HBM with
 * the compressed bytes.  afhip_lz4_decode_streams: one wave per stream (any length) resolves the sequences of a 64-byte
 * window in its lanes and writes dsize bytes straight to (to_out ? out_dev : tmp_dev) + dst_off — out_dev may be the data
 * cube itself; stored streams (csize == dsize) are copied.  max_dsize = the longest dsize of the list (checked >= 0, not
 * otherwise used since the decode left LDS).  A malformed stream writes nothing outside its own destination and adds 1 to
 * *errors_dev (read it after the next synchronisation).
 * afhip_unshuffle_blocks: Blosc's byte shuffle undone, tmp_dev -> out_dev.  All pointers are device memory. */
typedef struct afhip_lz4_stream { int64_t src_off, dst_off; int32_t csize, dsize, to_out, pad; } afhip_lz4_stream;
typedef struct afhip_shuffle_block { int64_t tmp_off, out_off; int32_t bsize, typesize; } afhip_shuffle_block;
int afhip_lz4_decode_streams(const void* comp_dev, const afhip_lz4_stream* streams_dev, int64_t n_streams, int32_t max_dsize,
                             void* tmp_dev, void* out_dev, int32_t* errors_dev, void* stream);
int afhip_unshuffle_blocks(const void* tmp_dev, void* out_dev, const afhip_shuffle_block* blocks_dev, int64_t n_blocks,
                           int32_t max_bsize, void* stream);

/* Replaces the body of SpatialAggregator.compute (spatial.py:110-133) for K names:
 * shared validity (all K non-NaN), den = W.valid, num_k = W.where(valid, x_k, 0),
 * res = num/den where den != 0 else NaN.
 * x_dev [K, n_cells, nt] float64; num_dev [K, R, nt]; den_dev [R, nt]; res_dev [K, R, nt]
 * (num_dev / den_dev may be NULL). */
int afhip_spatial_wavg(const afhip_csr* csr, const double* x_dev, int64_t K, int64_t nt,
                       double* num_dev, double* den_dev, double* res_dev, void* stream);

/* The divide of SpatialAggregator.compute on its own (spatial.py:127-133): res = num / den where den != 0 else NaN.
 * num_dev [K, R, P], den_dev [R, P], res_dev [K, R, P] float64.  For cell-axis sharding across GPUs: every rank's
 * numerators and denominators are summed first (one all-reduce), then divided once. */
int afhip_panel_divide(const double* num_dev, const double* den_dev, double* res_dev, int64_t K, int64_t R,
                       int64_t P, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused plan: one pass over the raw cube for all output columns.
 *
 * Replaces, for one aggregate_dataset() call, the lazy graph built by aggregate_time
 * (aggfly/aggregate/aggregate.py:101-162) + SpatialAggregator.compute and executed by the
 * single dask.compute at spatial.py:125.  A column is
 *
 *     inner reducer over inner groups of raw steps        ('aggregate', calc @ groupby)
 *       -> element-wise transform                         ('transform', power / spline)
 *       -> outer reducer over outer groups of inner values ('aggregate', calc @ groupby)
 *
 * A single-level spec uses outer = AFHIP_IDENTITY with outer_bounds = [0,1,..,G1].
 * ---------------------------------------------------------------------------------- */
typedef struct afhip_column {
    int32_t inner;          /* AFHIP_MEAN..AFHIP_SINE_DD                                   */
    int32_t transform;      /* AFHIP_TF_*                                                   */
    int32_t outer;          /* AFHIP_MEAN, SUM, MIN, MAX, DD, BINS or AFHIP_IDENTITY        */
    int32_t rounding;       /* AFHIP_ROUND_* bits: emulate the reference's float32 intermediates (0 = all float64) */
    double inner_args[3];   /* (t0, t1, flag) for DD / BINS / SINE_DD                       */
    double transform_arg;   /* exponent (POW) or knot (HINGE)                               */
    double outer_args[3];   /* (t0, t1, flag) for an outer DD / BINS                        */
} afhip_column;

typedef struct afhip_plan_desc {
    int64_t T;                     /* time steps in the cube                                */
    int64_t n_cells;               /* NY*NX                                                  */
    int32_t dtype;                 /* AFHIP_F32 / AFHIP_F64                                  */
    int32_t K;                     /* number of columns                                      */
    int64_t G1;                    /* inner groups                                           */
    const int64_t* inner_bounds;   /* HOST int64[G1+1] over time steps                       */
    int64_t P;                     /* outer groups (output periods)                          */
    const int64_t* outer_bounds;   /* HOST int64[P+1] over inner groups                      */
    const afhip_column* columns;   /* HOST [K]                                               */
    int32_t exact_order;           /* 1: never split an outer period across workgroups, so
                                      every per-cell sum runs in the reference's order       */
    int32_t tuning;                /* 0 = default; else a load-path arm pipe*1000+vec*100+depth (a hint:
                                      falls back to the default when that arm is not compiled) */
} afhip_plan_desc;

typedef struct afhip_plan afhip_plan;
int afhip_plan_create(const afhip_plan_desc* desc, afhip_plan** out);
void afhip_plan_destroy(afhip_plan* plan);
int afhip_plan_device(const afhip_plan* plan);
/* AFHIP_TF_INTER columns: bind column `column`'s second cube before the plan runs.  inter_dev [G1, n_cells] of `dtype`
 * (AFHIP_F32 / AFHIP_F64), time-major like the cube: the value the column's inner reducer gives inner group g at a cell
 * is multiplied by inter_dev[g * n_cells + cell] (np.multiply(block, other), dataset.py:563, on the inner level's
 * output, whose time axis is the inner groups).  The pointer is read by every later run until it is bound again; the
 * caller keeps the memory alive.  A run with an unbound AFHIP_TF_INTER column fails with AFHIP_E_INVALID. */
int afhip_plan_bind_inter(afhip_plan* plan, int column, const void* inter_dev, int dtype);
/* Bytes of device scratch a run needs.  afhip_plan_workspace_bytes: the temporal stage alone (per-chunk partials; what
 * afhip_plan_run_temporal takes).  afhip_plan_run_workspace_bytes: a whole afhip_plan_run against `csr` (partials + the
 * cell-major panel + the [rows][P][K+1] sums of that table).  A caller-owned workspace must be 256-byte aligned. */
int64_t afhip_plan_workspace_bytes(const afhip_plan* plan);
int64_t afhip_plan_run_workspace_bytes(const afhip_plan* plan, const afhip_csr* csr);
/* Human-readable lowering (kernel variant, chunks, slots) into buf; returns bytes needed. */
int afhip_plan_describe(const afhip_plan* plan, char* buf, int buf_len);

/* Temporal stage only: cells_dev[K, P, n_cells] float64 = the per-cell, per-period value
 * of every column (NaN where the reference's temporal stage yields NaN).  This is what
 * aggregate_time returns before the spatial step (aggregate.py:160-162).
 * workspace_dev / workspace_bytes: caller-owned scratch of at least afhip_plan_workspace_bytes(plan) (AFHIP_E_INVALID when
 * smaller), or NULL / 0: the plan allocates and caches its own (hipMalloc; a plan that outgrows it keeps the old block
 * until afhip_plan_destroy, so that no run ever waits for the device in a hipFree). */
int afhip_plan_run_temporal(afhip_plan* plan, const void* cube_dev, double* cells_dev,
                            void* workspace_dev, int64_t workspace_bytes, void* stream);

/* Whole path: temporal stage, shared validity, CSR weighted sums, divide.
 * num_dev [K, R, P], den_dev [R, P], res_dev [K, R, P] float64 (num/den may be NULL);
 * cells_dev optional as above (NULL to skip materialising it).
 * Routes (results equal to rounding; exact_order plans always take the first): with cells_dev, or exact_order — slot merge into a
 * cell-major panel, then weighted sums in table order; without — one gather over the period partials (no panel), or, for plans with
 * several output periods whose table allows it, weighted sums per region formed inside the temporal kernel at every period end (the
 * per-cell period values are then never written); bin-count plans gather their packed records.  The CSR handle caches the tables of
 * the last route on first use (it is entered through a const pointer but guarded by its own lock).
 * workspace_dev / workspace_bytes: caller-owned scratch of at least afhip_plan_run_workspace_bytes(plan, csr), or NULL / 0 for
 * plan-owned scratch as above.  Nothing on the run path allocates, frees or synchronises when the workspace is the caller's
 * (the Python host hands every plan a block of torch's caching allocator: where a process's first hipMalloc'ed scratch lands
 * decides 7-9 % of the bin-count kernel on some boxes, profiles/r03_plan_order_probe.txt; afhip_plan_describe names which it was).
 * If kernel_ms is not NULL the call records HIP events on `stream` around the temporal
 * kernel and around the whole sequence, synchronises the stream, and writes
 * kernel_ms[0] = temporal kernel ms, kernel_ms[1] = whole sequence ms. */
int afhip_plan_run(afhip_plan* plan, const void* cube_dev, const afhip_csr* csr,
                   double* num_dev, double* den_dev, double* res_dev, double* cells_dev,
                   void* workspace_dev, int64_t workspace_bytes, void* stream, float* kernel_ms);

/* Per-launch device timing of the dominant (temporal) kernel without host syncs in the
 * timed region: _begin arms up to max_launches HIP event pairs; every afhip_plan_run then
 * records one pair around the temporal kernel on its stream; _end waits for the recorded
 * events, writes one duration (ms) per launch into ms_out[0..cap) and returns how many
 * launches were recorded (-1 on error).  Used by bench.py for roofline.achieved. */
int afhip_plan_profile_begin(afhip_plan* plan, int64_t max_launches);
int64_t afhip_plan_profile_end(afhip_plan* plan, float* ms_out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* AGGFLY_HIP_H */
