/* aggfly_codec.h — C ABI of libaggfly_codec.so: host-side chunk codecs of the ingestion path
 * (SURVEY.md §8f row N2).  Plain C, no GPU code; every function is re-entrant (OpenMP inside the
 * *_files / *_many calls only).
 *
 * Replaces, for this path, the numcodecs calls the reference makes inside its dask graph when a Zarr
 * store is opened (aggfly/dataset/dataset.py:697-728: zarr chunk decode per task;
 * benchmarks/bench_read_scheduler.py:4-8: GIL-limited to ~2 cores warm).
 *
 * Return values: >= 0 = bytes produced (decode / encode), or AFCODEC_OK for the batch calls;
 * < 0 = error code, text in afcodec_last_error() (thread-local).
 */
#ifndef AGGFLY_CODEC_H
#define AGGFLY_CODEC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AFCODEC_OK 0
#define AFCODEC_E_FORMAT (-1)      /* not a well-formed container / truncated */
#define AFCODEC_E_UNSUPPORTED (-2) /* codec or library not available (snappy; liblz4 / libzstd missing) */
#define AFCODEC_E_SIZE (-3)        /* destination too small, sizes out of range, out of memory */
#define AFCODEC_E_CODEC (-4)       /* the inner codec failed */
#define AFCODEC_MISSING (-100)     /* results[i] of the *_files calls: the chunk file does not exist */

const char* afcodec_last_error(void);
int afcodec_have(int codec); /* Blosc codec ids: 0 blosclz, 1 lz4, 3 zlib, 4 zstd -> 1 if usable */

/* Blosc-1 container (format version 2; codecs blosclz / lz4 / lz4hc / zlib / zstd; byte- and bit-shuffle). */
int afcodec_blosc_info(const void* chunk, int64_t size, int64_t* nbytes, int64_t* blocksize, int32_t* typesize, int32_t* flags);
int64_t afcodec_blosc_decode(const void* chunk, int64_t csize, void* dst, int64_t dstsize);
/* the blocks of ONE (large) chunk spread over an OpenMP team */
int64_t afcodec_blosc_decode_mt(const void* chunk, int64_t csize, void* dst, int64_t dstsize, int nthreads);
int afcodec_blosc_decode_many(int64_t n, const void* const* chunks, const int64_t* csizes, void* const* dsts,
                              const int64_t* dstsizes, int nthreads, int64_t* results);
int afcodec_blosc_decode_files(int64_t n, const char* const* paths, void* const* dsts, const int64_t* dstsizes,
                               int nthreads, int64_t* results);
int64_t afcodec_blosc_bound(int64_t nbytes, int64_t blocksize);
int64_t afcodec_blosc_encode_lz4(const void* src, int64_t nbytes, int typesize, int shuffle, int64_t blocksize,
                                 void* dst, int64_t cap);

/* Plan of a GPU-side decode (libaggfly_hip: afhip_lz4_decode_streams / afhip_unshuffle_blocks, include/aggfly_hip.h): the
 * containers of n Blosc-1 chunks — chunk i = comp_size[i] bytes at base + comp_off[i], its decoded bytes wanted at offset
 * out_off[i] (capacity out_size[i]) of the output buffer — are parsed on the host and turned into
 *   streams [*n_streams]  afhip_lz4_stream records: every LZ4 stream (a block, or one byte plane of a split block) with its
 *                         offset in the compressed bytes (relative to base: the device copy keeps the same offsets) and
 *                         its destination (the output, or the shuffled scratch of *tmp_bytes bytes); stored streams and
 *                         stored chunks appear with csize == dsize;
 *   blocks  [*n_blocks]   afhip_shuffle_block records: blocks whose byte shuffle is undone from the scratch into the output.
 * Nothing is decoded here.  results[i] = the chunk's decoded size, or < 0: AFCODEC_E_UNSUPPORTED marks a chunk the GPU
 * route does not take (another codec than LZ4, bit shuffle) — decode it on the host; the call then returns that code too.
 * *max_dsize = the longest stream (sizes the kernel's LDS ring). */
int afcodec_blosc_lz4_plan(const void* base, int64_t n, const int64_t* comp_off, const int64_t* comp_size, const int64_t* out_off,
                           const int64_t* out_size, void* streams, int64_t cap_streams, int64_t* n_streams, void* blocks,
                           int64_t cap_blocks, int64_t* n_blocks, int64_t* tmp_bytes, int32_t* max_dsize, int64_t* results);

/* Chunk files of one codec kind (0 raw, 1 Blosc-1, 2 Zstandard frame, 3 zlib or gzip stream, 4 numcodecs LZ4;
 * kind + 16 * element_size adds a byte-unshuffle after the codec: HDF5 / netCDF-4 shuffle + deflate chunks): read and
 * decoded paths[i] -> dsts[i], one chunk per OpenMP thread. */
int afcodec_decode_files(int kind, int64_t n, const char* const* paths, void* const* dsts, const int64_t* dstsizes,
                         int nthreads, int64_t* results);

/* The same for byte ranges [offsets[i], offsets[i] + lengths[i]) of the files (inner chunks of Zarr v3
 * shards); lengths[i] < 0 or offsets == NULL = the whole file. */
int afcodec_decode_ranges(int kind, int64_t n, const char* const* paths, const int64_t* offsets, const int64_t* lengths,
                          void* const* dsts, const int64_t* dstsizes, int nthreads, int64_t* results);

/* Byte ranges of files packed back to back into one buffer (what the decode-in-HBM route uploads): range i lands at
 * dst + out_off[i] (out_off has n + 1 entries, steps rounded up to `align`); results[i] = its size, -100 = missing file.
 * Sizes are taken from the files here; lengths[i] < 0 or offsets == NULL = the whole file. */
int afcodec_read_packed(int64_t n, const char* const* paths, const int64_t* offsets, const int64_t* lengths, void* dst, int64_t cap,
                        int64_t align, int nthreads, int64_t* out_off, int64_t* results);

/* numcodecs' LZ4 codec (Zarr v2 compressor id "lz4"): int32 decoded size + one raw LZ4 block. */
int64_t afcodec_lz4_decode(const void* src, int64_t n, void* dst, int64_t cap);

/* Plain Zstandard frames (Zarr compressor / codec "zstd"). */
int64_t afcodec_zstd_decode(const void* src, int64_t n, void* dst, int64_t cap);
int64_t afcodec_zstd_bound(int64_t n);
int64_t afcodec_zstd_encode(const void* src, int64_t n, int level, void* dst, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
