/* blosc1.c — host-side chunk codec of the ingestion path.
 *
 * Zarr stores in the wild are compressed with Blosc-1 (numcodecs' default compressor is
 * Blosc(cname="lz4", shuffle=SHUFFLE)); the reference decodes them through numcodecs inside its
 * dask graph (aggfly/dataset/dataset.py:697-728), GIL-limited to ~2 cores
 * (benchmarks/bench_read_scheduler.py:4-8).  This file decodes Blosc-1 chunks natively, one
 * chunk per host thread, straight into the time-major staging buffer that is then copied to HBM.
 *
 * Blosc-1 container (c-blosc 1.x, format version 2), restated from the published format:
 *   header  [0] version  [1] versionlz  [2] flags  [3] typesize  [4:8] nbytes  [8:12] blocksize
 *           [12:16] cbytes (all little-endian)
 *   flags   0x01 byte shuffle | 0x02 stored uncompressed ("memcpyed") | 0x04 bit shuffle |
 *           0x10 blocks are not split into per-byte streams | bits 5-7: codec (0 blosclz, 1 lz4 /
 *           lz4hc, 2 snappy, 3 zlib, 4 zstd)
 *   then    int32 bstarts[nblocks]: offset of every block's data from the start of the chunk
 *   block   nsplits streams (typesize streams when split, else 1), each: int32 csize, data;
 *           csize == stream length means the stream is stored raw
 *   the (un)shuffle acts per block; the last, shorter block is never split.
 * Pinned against chunks produced by the real c-blosc 1.21 (tests/golden/blosc_fixtures.json).
 *
 * LZ4 / Zstandard come from the system's liblz4.so.1 / libzstd.so.1 (dlopen, stable C ABIs:
 * LZ4_decompress_safe, LZ4_compress_default, ZSTD_decompress, ZSTD_isError), zlib is linked.
 */
#include <dlfcn.h>
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
#include <zlib.h>
#include <emmintrin.h>

/* AFCODEC_NT_COPY=0 switches the non-temporal stores of the reader and of the byte unshuffle off (A/B knob) */
static int nt_store_default(void) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("AFCODEC_NT_COPY"); v = e ? atoi(e) != 0 : 1; }
    return v;
}

#define AFCODEC_OK 0
#define AFCODEC_E_FORMAT (-1)
#define AFCODEC_E_UNSUPPORTED (-2)
#define AFCODEC_E_SIZE (-3)
#define AFCODEC_E_CODEC (-4)

static __thread char g_err[256];
static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
const char* afcodec_last_error(void) { return g_err; }

/* ---- lazily bound system codecs ---- */
typedef int (*lz4_dec_fn)(const char*, char*, int, int);
typedef int (*lz4_enc_fn)(const char*, char*, int, int);
typedef int (*lz4_bound_fn)(int);
typedef size_t (*zstd_dec_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_iserr_fn)(size_t);
typedef size_t (*zstd_enc_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*zstd_bound_fn)(size_t);
static lz4_dec_fn p_lz4_dec;
static lz4_enc_fn p_lz4_enc;
static lz4_bound_fn p_lz4_bound;
static zstd_dec_fn p_zstd_dec;
static zstd_iserr_fn p_zstd_iserr;
static zstd_enc_fn p_zstd_enc;
static zstd_bound_fn p_zstd_bound;
static int g_lz4_tried, g_zstd_tried;

static int need_lz4(void) {
    if (!g_lz4_tried) {
        void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            p_lz4_dec = (lz4_dec_fn)dlsym(h, "LZ4_decompress_safe");
            p_lz4_enc = (lz4_enc_fn)dlsym(h, "LZ4_compress_default");
            p_lz4_bound = (lz4_bound_fn)dlsym(h, "LZ4_compressBound");
        }
        g_lz4_tried = 1;
    }
    return (p_lz4_dec && p_lz4_enc && p_lz4_bound) ? 0 : fail(AFCODEC_E_UNSUPPORTED, "liblz4.so.1 not available");
}
static int need_zstd(void) {
    if (!g_zstd_tried) {
        void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            p_zstd_dec = (zstd_dec_fn)dlsym(h, "ZSTD_decompress");
            p_zstd_iserr = (zstd_iserr_fn)dlsym(h, "ZSTD_isError");
            p_zstd_enc = (zstd_enc_fn)dlsym(h, "ZSTD_compress");
            p_zstd_bound = (zstd_bound_fn)dlsym(h, "ZSTD_compressBound");
        }
        g_zstd_tried = 1;
    }
    return (p_zstd_dec && p_zstd_iserr) ? 0 : fail(AFCODEC_E_UNSUPPORTED, "libzstd.so.1 not available");
}
int afcodec_have(int codec) { /* 1 lz4, 3 zlib, 4 zstd, 0 blosclz */
    if (codec == 1) return need_lz4() == 0;
    if (codec == 4) return need_zstd() == 0;
    return codec == 0 || codec == 3;
}

static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* ---- BloscLZ (the FastLZ-derived codec shipped inside c-blosc 1.x) ---- */
static int blosclz_decode(const uint8_t* ip, int length, uint8_t* out, int maxout) {
    const uint8_t* ip_limit = ip + length;
    uint8_t* op = out;
    uint8_t* op_limit = out + maxout;
    if (length == 0) return 0;
    uint32_t ctrl = (*ip++) & 31u;
    for (;;) {
        if (ctrl >= 32) {                                   /* match: length in the top 3 bits, distance below */
            int32_t len = (int32_t)(ctrl >> 5) - 1;
            int32_t ofs = (int32_t)((ctrl & 31u) << 8);
            const uint8_t* ref = op - ofs;
            uint8_t code;
            if (len == 7 - 1) {
                do {
                    if (ip + 1 >= ip_limit) return -1;
                    code = *ip++;
                    len += code;
                } while (code == 255);
            } else if (ip + 1 >= ip_limit) {
                return -1;
            }
            code = *ip++;
            len += 3;
            ref -= code;
            if (code == 255 && ofs == (31 << 8)) {          /* far match: 16-bit distance beyond 8191 */
                if (ip + 1 >= ip_limit) return -1;
                ofs = (*ip++) << 8;
                ofs += *ip++;
                ref = op - ofs - 8191;
            }
            if (op + len > op_limit) return -1;
            if (ref - 1 < out) return -1;
            const int last = ip >= ip_limit;
            if (!last) ctrl = *ip++;
            ref--;
            for (int32_t i = 0; i < len; ++i) op[i] = ref[i];      /* byte-wise: overlapping runs repeat */
            op += len;
            if (last) break;
        } else {                                            /* literal run of ctrl + 1 bytes */
            ctrl++;
            if (op + ctrl > op_limit || ip + ctrl > ip_limit) return -1;
            memcpy(op, ip, ctrl);
            op += ctrl;
            ip += ctrl;
            if (ip >= ip_limit) break;
            ctrl = *ip++;
        }
    }
    return (int)(op - out);
}

/* ---- shuffle filters, per block ---- */
/* 4- and 8-byte elements: 16 elements a step with SSE2 byte / word / dword interleaves (every x86-64 CPU has them).  The planes
 * come out of a cache-resident scratch block and the elements go to a destination nobody reads back soon (page-locked staging
 * the GPU fetches by DMA): with a 16-byte-aligned destination the stores are non-temporal, so the destination lines are not
 * read first. */
static void unshuffle_bytes(int ts, int64_t bsize, const uint8_t* src, uint8_t* dst) {
    const int64_t n = bsize / ts, rem = bsize % ts;
    if (ts == 4) {
        const uint8_t *s0 = src, *s1 = src + n, *s2 = src + 2 * n, *s3 = src + 3 * n;
        const int nt = (((uintptr_t)dst) & 15) == 0 && nt_store_default();
        int64_t i = 0;
        for (; i + 16 <= n; i += 16) {
            const __m128i a = _mm_loadu_si128((const __m128i*)(s0 + i)), b = _mm_loadu_si128((const __m128i*)(s1 + i));
            const __m128i c = _mm_loadu_si128((const __m128i*)(s2 + i)), d = _mm_loadu_si128((const __m128i*)(s3 + i));
            const __m128i ab0 = _mm_unpacklo_epi8(a, b), ab1 = _mm_unpackhi_epi8(a, b), cd0 = _mm_unpacklo_epi8(c, d), cd1 = _mm_unpackhi_epi8(c, d);
            const __m128i o0 = _mm_unpacklo_epi16(ab0, cd0), o1 = _mm_unpackhi_epi16(ab0, cd0), o2 = _mm_unpacklo_epi16(ab1, cd1), o3 = _mm_unpackhi_epi16(ab1, cd1);
            __m128i* out = (__m128i*)(dst + 4 * i);
            if (nt) { _mm_stream_si128(out, o0); _mm_stream_si128(out + 1, o1); _mm_stream_si128(out + 2, o2); _mm_stream_si128(out + 3, o3); }
            else { _mm_storeu_si128(out, o0); _mm_storeu_si128(out + 1, o1); _mm_storeu_si128(out + 2, o2); _mm_storeu_si128(out + 3, o3); }
        }
        for (; i < n; ++i) { dst[4 * i] = s0[i]; dst[4 * i + 1] = s1[i]; dst[4 * i + 2] = s2[i]; dst[4 * i + 3] = s3[i]; }
        if (nt) _mm_sfence();
    } else if (ts == 8) {
        const int nt = (((uintptr_t)dst) & 15) == 0 && nt_store_default();
        int64_t i = 0;
        for (; i + 16 <= n; i += 16) {
            __m128i p[8], t[8], u[8];
            for (int j = 0; j < 8; ++j) p[j] = _mm_loadu_si128((const __m128i*)(src + (int64_t)j * n + i));
            for (int j = 0; j < 4; ++j) { t[2 * j] = _mm_unpacklo_epi8(p[2 * j], p[2 * j + 1]); t[2 * j + 1] = _mm_unpackhi_epi8(p[2 * j], p[2 * j + 1]); }
            /* t[0], t[1]: bytes 0-1 of elements 0-7 / 8-15;  t[2], t[3]: bytes 2-3;  t[4], t[5]: bytes 4-5;  t[6], t[7]: bytes 6-7 */
            u[0] = _mm_unpacklo_epi16(t[0], t[2]); u[1] = _mm_unpackhi_epi16(t[0], t[2]);      /* bytes 0-3 of elements 0-3 / 4-7 */
            u[2] = _mm_unpacklo_epi16(t[1], t[3]); u[3] = _mm_unpackhi_epi16(t[1], t[3]);      /*              elements 8-11 / 12-15 */
            u[4] = _mm_unpacklo_epi16(t[4], t[6]); u[5] = _mm_unpackhi_epi16(t[4], t[6]);      /* bytes 4-7 */
            u[6] = _mm_unpacklo_epi16(t[5], t[7]); u[7] = _mm_unpackhi_epi16(t[5], t[7]);
            __m128i* out = (__m128i*)(dst + 8 * i);
            for (int q = 0; q < 4; ++q) {
                const __m128i lo = _mm_unpacklo_epi32(u[q], u[q + 4]), hi = _mm_unpackhi_epi32(u[q], u[q + 4]);   /* elements 4q, 4q+1 / 4q+2, 4q+3 */
                if (nt) { _mm_stream_si128(out + 2 * q, lo); _mm_stream_si128(out + 2 * q + 1, hi); }
                else { _mm_storeu_si128(out + 2 * q, lo); _mm_storeu_si128(out + 2 * q + 1, hi); }
            }
        }
        for (; i < n; ++i)
            for (int j = 0; j < 8; ++j) dst[8 * i + j] = src[(int64_t)j * n + i];
        if (nt) _mm_sfence();
    } else {
        for (int64_t i = 0; i < n; ++i)
            for (int j = 0; j < ts; ++j) dst[i * ts + j] = src[(int64_t)j * n + i];
    }
    if (rem) memcpy(dst + bsize - rem, src + bsize - rem, (size_t)rem);
}
static void shuffle_bytes(int ts, int64_t bsize, const uint8_t* src, uint8_t* dst) {
    const int64_t n = bsize / ts, rem = bsize % ts;
    for (int j = 0; j < ts; ++j)
        for (int64_t i = 0; i < n; ++i) dst[(int64_t)j * n + i] = src[i * ts + j];
    if (rem) memcpy(dst + bsize - rem, src + bsize - rem, (size_t)rem);
}
/* Bit shuffle: bit b (byte j = b / 8, LSB-first) of element e is stored in bit row b at bit e.
 * c-blosc 1.x bit-shuffles a block only when it holds a multiple of 8 elements; any other block
 * (typically the short last one) is stored unshuffled (pinned by the 1.21 fixtures). */
static void unshuffle_bits(int ts, int64_t bsize, const uint8_t* src, uint8_t* dst) {
    const int64_t n = bsize / ts;
    if (n % 8 != 0) { memcpy(dst, src, (size_t)bsize); return; }
    const int64_t nrow = n / 8;
    for (int64_t g = 0; g < nrow; ++g) {                    /* 8 elements at a time */
        for (int j = 0; j < ts; ++j) {
            uint64_t x = 0;                                 /* byte k of x = bit row (8j + k), byte g */
            for (int k = 0; k < 8; ++k) x |= (uint64_t)src[((int64_t)j * 8 + k) * nrow + g] << (8 * k);
            /* 8x8 bit-matrix transpose (Hacker's Delight 7-3) */
            uint64_t t;
            t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAULL;  x = x ^ t ^ (t << 7);
            t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCULL; x = x ^ t ^ (t << 14);
            t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ULL; x = x ^ t ^ (t << 28);
            for (int m = 0; m < 8; ++m) dst[(g * 8 + m) * ts + j] = (uint8_t)(x >> (8 * m));
        }
    }
    const int64_t off = n * ts;
    if (bsize > off) memcpy(dst + off, src + off, (size_t)(bsize - off));
}

int64_t afcodec_zstd_decode(const void* src, int64_t n, void* dst, int64_t cap);
int64_t afcodec_blosc_decode(const void* chunk, int64_t csize, void* dstv, int64_t dstsize);

/* ---- chunk header ---- */
int afcodec_blosc_info(const void* chunk, int64_t size, int64_t* nbytes, int64_t* blocksize, int32_t* typesize, int32_t* flags) {
    const uint8_t* c = (const uint8_t*)chunk;
    if (!c || size < 16) return fail(AFCODEC_E_FORMAT, "blosc chunk shorter than its 16-byte header");
    if (c[0] != 2) return fail(AFCODEC_E_UNSUPPORTED, "blosc format version is not 2 (Blosc-1)");
    if (nbytes) *nbytes = le32(c + 4);
    if (blocksize) *blocksize = le32(c + 8);
    if (typesize) *typesize = c[3];
    if (flags) *flags = c[2];
    if ((int64_t)le32(c + 12) > size) return fail(AFCODEC_E_SIZE, "blosc chunk is truncated (cbytes > size)");
    return AFCODEC_OK;
}

static int decode_stream(int codec, const uint8_t* src, int32_t csize, uint8_t* dst, int32_t want) {
    int got = -1;
    switch (codec) {
        case 0: got = blosclz_decode(src, csize, dst, want); break;
        case 1: got = p_lz4_dec((const char*)src, (char*)dst, csize, want); break;
        case 3: { uLongf n = (uLongf)want; got = uncompress(dst, &n, src, (uLong)csize) == Z_OK ? (int)n : -1; break; }
        case 4: { size_t n = p_zstd_dec(dst, (size_t)want, src, (size_t)csize); got = p_zstd_iserr(n) ? -1 : (int)n; break; }
        default: break;
    }
    return got == want ? 0 : -1;
}

static __thread uint8_t* t_scratch;
static __thread int64_t t_scratch_cap;

typedef struct {
    const uint8_t* c;
    uint8_t* dst;
    int64_t nbytes, blocksize, cbytes, nblocks, leftover;
    int ts, flags, codec, dont_split, want_shuffle, stored;
} blosc_ctx;

/* Parses and validates the header; > 0 = nothing left to do (empty chunk), < 0 = error. */
static int blosc_open(blosc_ctx* x, const void* chunk, int64_t csize, void* dstv, int64_t dstsize) {
    int32_t ts, flags;
    x->c = (const uint8_t*)chunk;
    x->dst = (uint8_t*)dstv;
    int rc = afcodec_blosc_info(chunk, csize, &x->nbytes, &x->blocksize, &ts, &flags);
    if (rc) return rc;
    x->ts = ts; x->flags = flags;
    if (x->nbytes > dstsize) return fail(AFCODEC_E_SIZE, "destination smaller than the chunk's nbytes");
    if (x->nbytes == 0) return 1;
    x->cbytes = le32(x->c + 12);
    x->stored = (flags & 0x02) != 0;
    if (x->stored) {
        if (x->cbytes < 16 + x->nbytes) return fail(AFCODEC_E_FORMAT, "stored chunk shorter than nbytes");
        return AFCODEC_OK;
    }
    x->codec = (flags >> 5) & 7;
    if (x->codec == 1) { if ((rc = need_lz4())) return rc; }
    else if (x->codec == 4) { if ((rc = need_zstd())) return rc; }
    else if (x->codec != 0 && x->codec != 3) return fail(AFCODEC_E_UNSUPPORTED, "blosc codec not supported (snappy)");
    if (x->blocksize <= 0 || ts <= 0 || x->blocksize > x->nbytes) return fail(AFCODEC_E_FORMAT, "bad blocksize / typesize");
    x->nblocks = (x->nbytes + x->blocksize - 1) / x->blocksize;
    x->leftover = x->nbytes % x->blocksize;
    if (16 + 4 * x->nblocks > x->cbytes) return fail(AFCODEC_E_FORMAT, "block table beyond the chunk");
    x->dont_split = (flags >> 4) & 1;
    x->want_shuffle = (flags & 0x01) && ts > 1;
    return AFCODEC_OK;
}

/* One block of an opened chunk; blocks are independent, so a big chunk decodes on many threads. */
static int blosc_block(const blosc_ctx* x, int64_t b) {
    const uint8_t* c = x->c;
    const int ts = x->ts;
    uint8_t* tmp = NULL;
    if (x->want_shuffle || (x->flags & 0x04)) {
        /* per-thread scratch, kept between calls: a fresh malloc of a block (>= 128 KiB -> mmap/munmap)
         * per chunk page-faults every time and serialises the threads on the process's mmap lock */
        if (t_scratch_cap < x->blocksize) {
            free(t_scratch);
            t_scratch = (uint8_t*)malloc((size_t)x->blocksize);
            t_scratch_cap = t_scratch ? x->blocksize : 0;
        }
        tmp = t_scratch;
        if (!tmp) return fail(AFCODEC_E_SIZE, "out of memory");
    }
    const int last_short = (b == x->nblocks - 1) && x->leftover > 0;
    const int64_t bsize = last_short ? x->leftover : x->blocksize;
    const int do_shuf = x->want_shuffle;
    const int do_bits = !do_shuf && (x->flags & 0x04) && bsize >= ts;
    uint8_t* out = x->dst + b * x->blocksize;
    uint8_t* into = (do_shuf || do_bits) ? tmp : out;
    int nsplits = 1;
    if (!x->dont_split && ts <= 16 && x->blocksize / ts >= 128 && !last_short) nsplits = ts;
    const int64_t neblock = bsize / nsplits;
    const int64_t start = (int64_t)(int32_t)le32(c + 16 + 4 * b);
    if (start < 16 + 4 * x->nblocks || start >= x->cbytes) return fail(AFCODEC_E_FORMAT, "block offset out of range");
    const uint8_t* src = c + start;
    for (int j = 0; j < nsplits; ++j) {
        if (src + 4 > c + x->cbytes) return fail(AFCODEC_E_FORMAT, "stream header beyond the chunk");
        const int32_t sz = (int32_t)le32(src);
        src += 4;
        if (sz < 0 || src + sz > c + x->cbytes) return fail(AFCODEC_E_FORMAT, "stream beyond the chunk");
        if (sz == neblock) memcpy(into + j * neblock, src, (size_t)neblock);
        else if (decode_stream(x->codec, src, sz, into + j * neblock, (int32_t)neblock))
            return fail(AFCODEC_E_CODEC, "block failed to decompress to its recorded size");
        src += sz;
    }
    if (do_shuf) unshuffle_bytes(ts, bsize, tmp, out);
    else if (do_bits) unshuffle_bits(ts, bsize, tmp, out);
    return AFCODEC_OK;
}

/* ---- plan of a GPU-side decode (include/aggfly_codec.h: afcodec_blosc_lz4_plan) ----
 * Parses the containers of n chunks and lists, for the kernels of libaggfly_hip (afhip_lz4_decode_streams,
 * afhip_unshuffle_blocks), every LZ4 stream with its place in the compressed bytes and in the output, and every block whose
 * byte shuffle has to be undone.  Nothing is decoded here. */
typedef struct { int64_t src_off, dst_off; int32_t csize, dsize, to_out, pad; } lz4_stream_t;
typedef struct { int64_t tmp_off, out_off; int32_t bsize, typesize; } shuf_block_t;
#define GPU_STREAM_MAX 65536

int afcodec_blosc_lz4_plan(const void* base, int64_t n, const int64_t* comp_off, const int64_t* comp_size, const int64_t* out_off,
                           const int64_t* out_size, void* streams_v, int64_t cap_streams, int64_t* n_streams, void* blocks_v,
                           int64_t cap_blocks, int64_t* n_blocks, int64_t* tmp_bytes, int32_t* max_dsize, int64_t* results) {
    lz4_stream_t* streams = (lz4_stream_t*)streams_v;
    shuf_block_t* blocks = (shuf_block_t*)blocks_v;
    int64_t ns = 0, nb = 0, tmp = 0;
    int32_t maxd = 0;
    int rc_all = AFCODEC_OK;
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* c = (const uint8_t*)base + comp_off[i];
        const int64_t csz = comp_size[i];
        results[i] = 0;
        int64_t nbytes, blocksize; int32_t ts, flags;
        int rc = afcodec_blosc_info(c, csz, &nbytes, &blocksize, &ts, &flags);
        if (rc) { results[i] = rc; rc_all = rc; continue; }
        if (nbytes > out_size[i]) { results[i] = rc_all = fail(AFCODEC_E_SIZE, "destination smaller than the chunk's nbytes"); continue; }
        results[i] = nbytes;
        if (nbytes == 0) continue;
        const int64_t cbytes = le32(c + 12);
        if (flags & 0x02) {                                             /* stored chunk: plain copies, 64 KiB per wave */
            if (cbytes < 16 + nbytes) { results[i] = rc_all = fail(AFCODEC_E_FORMAT, "stored chunk shorter than nbytes"); continue; }
            for (int64_t o = 0; o < nbytes; o += GPU_STREAM_MAX) {
                const int32_t len = (int32_t)((nbytes - o) < GPU_STREAM_MAX ? (nbytes - o) : GPU_STREAM_MAX);
                if (ns >= cap_streams) return fail(AFCODEC_E_SIZE, "stream list too small");
                streams[ns++] = (lz4_stream_t){comp_off[i] + 16 + o, out_off[i] + o, len, len, 1, 0};
            }
            continue;
        }
        const int codec = (flags >> 5) & 7;
        if (codec != 1 || (flags & 0x04)) { results[i] = AFCODEC_E_UNSUPPORTED; rc_all = fail(AFCODEC_E_UNSUPPORTED, "not an LZ4 chunk with byte shuffle or none: decode on the host"); continue; }
        if (blocksize <= 0 || ts <= 0 || blocksize > nbytes) { results[i] = rc_all = fail(AFCODEC_E_FORMAT, "bad blocksize / typesize"); continue; }
        const int64_t nblocks = (nbytes + blocksize - 1) / blocksize, leftover = nbytes % blocksize;
        if (16 + 4 * nblocks > cbytes) { results[i] = rc_all = fail(AFCODEC_E_FORMAT, "block table beyond the chunk"); continue; }
        const int dont_split = (flags >> 4) & 1, want_shuffle = (flags & 0x01) && ts > 1;
        const int64_t ns0 = ns, nb0 = nb, tmp0 = tmp;
        int bad = 0;
        for (int64_t b = 0; b < nblocks && !bad; ++b) {
            const int last_short = (b == nblocks - 1) && leftover > 0;
            const int64_t bsize = last_short ? leftover : blocksize;
            int nsplits = 1;
            if (!dont_split && ts <= 16 && blocksize / ts >= 128 && !last_short) nsplits = ts;
            const int64_t neblock = bsize / nsplits;
            const int64_t start = (int64_t)(int32_t)le32(c + 16 + 4 * b);
            if (start < 16 + 4 * nblocks || start >= cbytes) { bad = AFCODEC_E_FORMAT; break; }
            int64_t dst_base;
            if (want_shuffle) {
                if (nb >= cap_blocks) return fail(AFCODEC_E_SIZE, "block list too small");
                blocks[nb++] = (shuf_block_t){tmp, out_off[i] + b * blocksize, (int32_t)bsize, ts};
                dst_base = tmp;
                tmp += (bsize + 15) & ~(int64_t)15;
            } else {
                dst_base = out_off[i] + b * blocksize;
            }
            const uint8_t* src = c + start;
            for (int j = 0; j < nsplits; ++j) {
                if (src + 4 > c + cbytes) { bad = AFCODEC_E_FORMAT; break; }
                const int32_t sz = (int32_t)le32(src);
                src += 4;
                if (sz < 0 || src + sz > c + cbytes || sz > neblock + neblock / 255 + 16) { bad = AFCODEC_E_FORMAT; break; }
                if (ns >= cap_streams) return fail(AFCODEC_E_SIZE, "stream list too small");
                streams[ns++] = (lz4_stream_t){comp_off[i] + (src - c), dst_base + j * neblock, sz, (int32_t)neblock, want_shuffle ? 0 : 1, 0};
                if (neblock > maxd) maxd = (int32_t)neblock;
                src += sz;
            }
        }
        if (bad) {
            ns = ns0; nb = nb0; tmp = tmp0;
            results[i] = bad;
            rc_all = fail(bad, "malformed block table or stream header");
        }
    }
    *n_streams = ns; *n_blocks = nb; *tmp_bytes = tmp; *max_dsize = maxd;
    return rc_all;
}

/* Decodes one chunk into dst (capacity dstsize); returns the number of bytes written or < 0. */
int64_t afcodec_blosc_decode(const void* chunk, int64_t csize, void* dstv, int64_t dstsize) {
    blosc_ctx x;
    int rc = blosc_open(&x, chunk, csize, dstv, dstsize);
    if (rc < 0) return rc;
    if (rc > 0) return 0;
    if (x.stored) { memcpy(x.dst, x.c + 16, (size_t)x.nbytes); return x.nbytes; }
    for (int64_t b = 0; b < x.nblocks; ++b)
        if ((rc = blosc_block(&x, b))) return rc;
    return x.nbytes;
}

/* The same with the blocks of the chunk spread over an OpenMP team: for chunks that are few and large
 * (the reference's own converter writes ~256 MB chunks — one per thread would leave most threads idle). */
int64_t afcodec_blosc_decode_mt(const void* chunk, int64_t csize, void* dstv, int64_t dstsize, int nthreads) {
    blosc_ctx x;
    int rc = blosc_open(&x, chunk, csize, dstv, dstsize);
    if (rc < 0) return rc;
    if (rc > 0) return 0;
    if (x.stored) { memcpy(x.dst, x.c + 16, (size_t)x.nbytes); return x.nbytes; }
    if (nthreads < 2 || x.nblocks < 2) {
        for (int64_t b = 0; b < x.nblocks; ++b)
            if ((rc = blosc_block(&x, b))) return rc;
        return x.nbytes;
    }
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : bad)
    for (int64_t b = 0; b < x.nblocks; ++b)
        if (blosc_block(&x, b)) bad += 1;
    return bad ? fail(AFCODEC_E_CODEC, "one or more blocks of the chunk failed to decode") : x.nbytes;
}

/* Many chunks at once, one per OpenMP thread (ctypes releases the GIL around this call). */
int afcodec_blosc_decode_many(int64_t n, const void* const* chunks, const int64_t* csizes, void* const* dsts,
                              const int64_t* dstsizes, int nthreads, int64_t* results) {
    int bad = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : bad)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = afcodec_blosc_decode(chunks[i], csizes[i], dsts[i], dstsizes[i]);
        results[i] = r;
        if (r < 0) bad += 1;
    }
    return bad ? fail(AFCODEC_E_CODEC, "one or more chunks failed to decode (see results[])") : AFCODEC_OK;
}

/* zlib / gzip streams through zlib's inflate (wbits 15 + 32: header auto-detected). */
static int64_t inflate_any(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return fail(AFCODEC_E_CODEC, "inflateInit2 failed");
    zs.next_in = (Bytef*)src; zs.avail_in = (uInt)n;
    zs.next_out = dst; zs.avail_out = (uInt)cap;
    const int rc = inflate(&zs, Z_FINISH);
    const int64_t got = (int64_t)zs.total_out;
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) return fail(AFCODEC_E_CODEC, "zlib / gzip stream failed to inflate into the chunk size");
    return got;
}

/* numcodecs' LZ4 codec (Zarr v2 compressor id "lz4"): int32 decoded size, then one raw LZ4 block. */
int64_t afcodec_lz4_decode(const void* srcv, int64_t n, void* dst, int64_t cap) {
    const uint8_t* src = (const uint8_t*)srcv;
    int rc = need_lz4();
    if (rc) return rc;
    if (n < 4) return fail(AFCODEC_E_FORMAT, "lz4 chunk shorter than its size header");
    const int64_t want = (int64_t)(int32_t)le32(src);
    if (want < 0 || want > cap) return fail(AFCODEC_E_SIZE, "lz4 chunk decodes to more than its destination");
    const int got = p_lz4_dec((const char*)src + 4, (char*)dst, (int)(n - 4), (int)want);
    if (got != want) return fail(AFCODEC_E_CODEC, "lz4 block failed to decode to its recorded size");
    return want;
}

/* One chunk file -> dst, by codec kind: 0 raw bytes, 1 Blosc-1, 2 Zstandard frame, 3 zlib or gzip, 4 numcodecs LZ4.
 * kind >> 4 = element size of a byte-unshuffle stage applied AFTER the codec (HDF5's shuffle + deflate filter
 * pair, numcodecs' Shuffle filter): the codec then decodes into per-thread scratch and the planes are woven
 * into dst. */
static int64_t decode_kind_plain(int kind, const uint8_t* buf, int64_t sz, void* dst, int64_t cap);
static __thread uint8_t* t_plane;
static __thread int64_t t_plane_cap;
static int64_t decode_kind(int kind, const uint8_t* buf, int64_t sz, void* dst, int64_t cap) {
    const int es = kind >> 4;
    if (es <= 1) return decode_kind_plain(kind & 15, buf, sz, dst, cap);
    if (t_plane_cap < cap) {
        free(t_plane);
        t_plane = (uint8_t*)malloc((size_t)cap);
        t_plane_cap = t_plane ? cap : 0;
    }
    if (!t_plane) return fail(AFCODEC_E_SIZE, "out of memory");
    const int64_t got = decode_kind_plain(kind & 15, buf, sz, t_plane, cap);
    if (got < 0) return got;
    unshuffle_bytes(es, got, t_plane, (uint8_t*)dst);
    return got;
}
static int64_t decode_kind_plain(int kind, const uint8_t* buf, int64_t sz, void* dst, int64_t cap) {
    switch (kind) {
        case 0:
            if (sz > cap) return fail(AFCODEC_E_SIZE, "raw chunk larger than its destination");
            memcpy(dst, buf, (size_t)sz);
            return sz;
        case 1: return afcodec_blosc_decode(buf, sz, dst, cap);
        case 2: return afcodec_zstd_decode(buf, sz, dst, cap);
        case 3: return inflate_any(buf, sz, (uint8_t*)dst, cap);
        case 4: return afcodec_lz4_decode(buf, sz, dst, cap);
        default: return fail(AFCODEC_E_UNSUPPORTED, "unknown codec kind");
    }
}

/* dst <- src with non-temporal 16-byte stores (SSE2: every x86-64 CPU); head and tail up to alignment by memcpy */
static void nt_memcpy(uint8_t* dst, const uint8_t* src, size_t n) {
    size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
    if (head > n) head = n;
    memcpy(dst, src, head);
    dst += head; src += head; n -= head;
    const size_t body = n & ~(size_t)63;
    for (size_t i = 0; i < body; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
        _mm_stream_si128((__m128i*)(dst + i), a); _mm_stream_si128((__m128i*)(dst + i + 16), b);
        _mm_stream_si128((__m128i*)(dst + i + 32), c); _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    memcpy(dst + body, src + body, n - body);
    _mm_sfence();
}

/* len bytes of fd at file offset off -> dst; -> bytes read.  nt: through a cache-resident bounce buffer, then non-temporal
 * stores — the destination lines are not read first (a third less DRAM traffic than the kernel's copy into the destination),
 * which leaves an upload reading the same page-locked memory more of the host's bandwidth: store -> HBM 59 -> 67 GB/s on a
 * 3.4 GB store, 43 -> 47 on 0.86 GB (profiles/r02_nt_copy_ab.txt).  AFCODEC_NT_COPY=0: the kernel copies straight into dst. */
enum { BOUNCE = 128 << 10 };
static int nt_copy_default(void) { return nt_store_default(); }
static int64_t read_piece(int fd, uint8_t* dst, int64_t len, int64_t off, int nt) {
    int64_t done = 0;
    if (nt) {
        static __thread uint8_t bounce[BOUNCE] __attribute__((aligned(64)));
        while (done < len) {
            const int64_t want = len - done < BOUNCE ? len - done : BOUNCE;
            const ssize_t got = pread(fd, bounce, (size_t)want, (off_t)(off + done));
            if (got <= 0) break;
            nt_memcpy(dst + done, bounce, (size_t)got);
            done += got;
        }
    } else {
        while (done < len) {
            const ssize_t got = pread(fd, dst + done, (size_t)(len - done), (off_t)(off + done));
            if (got <= 0) break;
            done += got;
        }
    }
    return done;
}

/* Is a failed open / stat an ABSENT chunk?  Only "no such file" is (a Zarr chunk that was never written reads as the fill
 * value); EMFILE, EACCES, EIO ... are failures of this process or the medium and must never be mistaken for one: a batch
 * larger than RLIMIT_NOFILE used to report existing chunks as absent, and the caller silently filled them with NaN. */
static int errno_means_absent(int e) { return e == ENOENT || e == ENOTDIR; }

/* The byte range of request i: offset and size checked against the file, WITHOUT keeping a descriptor (batches hold thousands
 * of files; descriptors are opened per 1 MiB piece below, so at most one per thread is open at a time).
 * -> size >= 0, -100 for an absent file (or an empty path), AFCODEC_E_FORMAT for an unreadable file or a range beyond its end. */
static int64_t stat_range(const char* path, const int64_t* offsets, const int64_t* lengths, int64_t i, int64_t* off_out) {
    *off_out = 0;
    if (!path || !path[0]) return -100;
    int64_t off = 0, sz = -1;
    if (offsets && lengths && lengths[i] >= 0) { off = offsets[i]; sz = lengths[i]; }
    struct stat stt;
    if (stat(path, &stt) != 0) return errno_means_absent(errno) ? -100 : AFCODEC_E_FORMAT;
    if (off < 0 || off > (int64_t)stt.st_size) return AFCODEC_E_FORMAT;
    if (sz < 0) sz = (int64_t)stt.st_size - off;
    if (off + sz > (int64_t)stt.st_size) return AFCODEC_E_FORMAT;
    *off_out = off;
    return sz;
}

/* One piece of a file through the calling thread's one-entry descriptor cache: consecutive pieces of a file (the dynamic schedule
 * hands a thread runs of them) share ONE open — at most one descriptor per thread is ever held, so a batch of any size stays
 * under the descriptor limit (every file held open at once ran into EMFILE on large batches and reported existing chunks as
 * absent), without an open / close per MiB.  After the open the file is fstat'ed again: one that shrank since the sizes were
 * taken (replaced while the batch was being read) is reported as such (-2), not as a short read.
 * -> bytes read; -1: could not be opened / read, -2: the file changed. */
typedef struct { int64_t file; int fd; } fd_cache;
static int64_t read_piece_of(fd_cache* fc, int64_t file, const char* path, uint8_t* dst, int64_t len, int64_t off, int nt_copy) {
    if (fc->file != file) {
        if (fc->fd >= 0) close(fc->fd);
        fc->file = file;
        fc->fd = open(path, O_RDONLY);
    }
    if (fc->fd < 0) return -1;
    struct stat stt;
    if (fstat(fc->fd, &stt) != 0) return -1;
    if (off + len > (int64_t)stt.st_size) return -2;
    return read_piece(fc->fd, dst, len, off, nt_copy);
}

/* Reads and decodes byte ranges of chunk files -> dsts[i] on an OpenMP team (no Python between chunks):
 * [offsets[i], offsets[i] + lengths[i]) of paths[i]; lengths[i] < 0 (or offsets == NULL) = the whole file.
 * Ranges serve the inner chunks of Zarr v3 shards.  A missing file leaves results[i] = -100 (the caller
 * fills the Zarr fill value). */
int afcodec_decode_ranges(int kind, int64_t n, const char* const* paths, const int64_t* offsets, const int64_t* lengths,
                          void* const* dsts, const int64_t* dstsizes, int nthreads, int64_t* results) {
    int bad = 0;
    if (nthreads < 1) nthreads = 1;
    if ((kind & 15) == 2 && need_zstd()) return AFCODEC_E_UNSUPPORTED;
    if (kind == 0 && n > 0) {
        /* raw bytes (stores without a compressor; the compressed chunk files of the decode-in-HBM route): the files are
         * cut into 1 MiB pieces and the PIECES are spread over the team — a batch of 14 files of 12.5 MB read one file
         * per thread moved 22 GB/s out of the page cache, 227 files of 1.6 MB 60 GB/s */
        enum { PIECE = 1 << 20 };
        int64_t* first = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));   /* first piece of file i */
        int64_t* offs = (int64_t*)malloc((size_t)n * sizeof(int64_t));
        if (!first || !offs) { free(first); free(offs); return fail(AFCODEC_E_SIZE, "out of memory"); }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 8)
        for (int64_t i = 0; i < n; ++i) {
            int64_t sz = stat_range(paths[i], offsets, lengths, i, &offs[i]);
            if (sz >= 0 && sz > dstsizes[i]) sz = AFCODEC_E_SIZE;
            results[i] = sz;
        }
        first[0] = 0;
        for (int64_t i = 0; i < n; ++i) first[i + 1] = first[i] + (results[i] > 0 ? (results[i] + PIECE - 1) / PIECE : 0);
        const int64_t npieces = first[n];
        int changed = 0;
#pragma omp parallel num_threads(nthreads)
        {
            fd_cache fc = {-1, -1};
#pragma omp for schedule(dynamic, 1)
            for (int64_t q = 0; q < npieces; ++q) {
                int64_t lo = 0, hi = n;                               /* the file of piece q: first[lo] <= q < first[lo + 1] */
                while (hi - lo > 1) { const int64_t mid = (lo + hi) / 2; if (first[mid] <= q) lo = mid; else hi = mid; }
                while (first[lo + 1] <= q) ++lo;                      /* (files without pieces share a boundary) */
                const int64_t at = (q - first[lo]) * PIECE, len = results[lo] - at < PIECE ? results[lo] - at : PIECE;
                int64_t o;
#pragma omp atomic read
                o = offs[lo];
                if (o < 0) continue;                                  /* an earlier piece of this file failed */
                const int64_t done = read_piece_of(&fc, lo, paths[lo], (uint8_t*)dsts[lo] + at, len, o + at, nt_copy_default());
                if (done != len) {
#pragma omp atomic write
                    offs[lo] = done == -2 ? -2 : -1;                  /* marks the file as failed (-2: it changed under the read) */
                }
            }
            if (fc.fd >= 0) close(fc.fd);
        }
        for (int64_t i = 0; i < n; ++i) {
            if (results[i] >= 0 && offs[i] < 0) { changed += offs[i] == -2; results[i] = AFCODEC_E_FORMAT; }
            if (results[i] < 0 && results[i] != -100) bad += 1;
        }
        free(first); free(offs);
        if (bad) {
            for (int64_t i = 0; i < n; ++i)
                if (results[i] == AFCODEC_E_SIZE) return fail(AFCODEC_E_CODEC, "raw chunk larger than its destination (see results[])");
            if (changed) return fail(AFCODEC_E_CODEC, "a chunk file shrank between stat and read: the store is being rewritten (see results[])");
            return fail(AFCODEC_E_CODEC, "one or more chunk files could not be read (see results[])");
        }
        return AFCODEC_OK;
    }
    if (kind == 1 && n * 2 <= nthreads) {
        /* fewer Blosc chunks than half the team: one chunk at a time, its blocks over the whole team, decoded
         * straight out of the page cache (mmap: no read() copy of a ~200 MB file in front of the decode) */
        const long page = sysconf(_SC_PAGESIZE);
        for (int64_t i = 0; i < n; ++i) {
            const int fd = open(paths[i], O_RDONLY);
            if (fd < 0) {
                if (errno_means_absent(errno)) { results[i] = -100; continue; }
                results[i] = fail(AFCODEC_E_FORMAT, "chunk file could not be opened"); bad += 1; continue;
            }
            int64_t off = 0, sz = -1;
            if (offsets && lengths && lengths[i] >= 0) { off = offsets[i]; sz = lengths[i]; }
            struct stat stt;
            int64_t r;
            if (fstat(fd, &stt) != 0) {
                r = fail(AFCODEC_E_FORMAT, "chunk file could not be read");
            } else {
                if (sz < 0) sz = (int64_t)stt.st_size - off;
                const int64_t a0 = off - off % page, span = off + sz - a0;
                void* m = (sz > 0 && off + sz <= (int64_t)stt.st_size) ? mmap(NULL, (size_t)span, PROT_READ, MAP_PRIVATE, fd, (off_t)a0) : MAP_FAILED;
                if (m == MAP_FAILED) {
                    r = fail(AFCODEC_E_FORMAT, "chunk file could not be mapped");
                } else {
                    r = afcodec_blosc_decode_mt((const uint8_t*)m + (off - a0), sz, dsts[i], dstsizes[i], nthreads);
                    munmap(m, (size_t)span);
                }
            }
            close(fd);
            results[i] = r;
            if (r < 0) bad += 1;
        }
        return bad ? fail(AFCODEC_E_CODEC, "one or more chunks failed to decode (see results[])") : AFCODEC_OK;
    }
#pragma omp parallel num_threads(nthreads) reduction(+ : bad)
    {
        uint8_t* buf = NULL;
        int64_t cap = 0;
#pragma omp for schedule(dynamic, 1)
        for (int64_t i = 0; i < n; ++i) {
            FILE* f = fopen(paths[i], "rb");
            if (!f) {
                if (errno_means_absent(errno)) { results[i] = -100; continue; }
                results[i] = fail(AFCODEC_E_FORMAT, "chunk file could not be opened"); bad += 1; continue;
            }
            int64_t off = 0, sz = -1;
            if (offsets && lengths && lengths[i] >= 0) { off = offsets[i]; sz = lengths[i]; }
            if (sz < 0) {
                fseek(f, 0, SEEK_END);
                sz = ftell(f);
            }
            int64_t r;
            if (fseek(f, (long)off, SEEK_SET) != 0) {
                r = fail(AFCODEC_E_FORMAT, "chunk range beyond the file");
            } else if (kind == 0 && sz <= dstsizes[i]) {     /* raw, no unshuffle: straight into the destination */
                r = (int64_t)fread(dsts[i], 1, (size_t)sz, f) == sz ? sz : fail(AFCODEC_E_FORMAT, "chunk file could not be read");
            } else {
                if (sz > cap) { free(buf); buf = (uint8_t*)malloc((size_t)sz + 64); cap = buf ? sz : 0; }
                if (!buf || (int64_t)fread(buf, 1, (size_t)sz, f) != sz) r = fail(AFCODEC_E_FORMAT, "chunk file could not be read");
                else r = decode_kind(kind, buf, sz, dsts[i], dstsizes[i]);
            }
            fclose(f);
            results[i] = r;
            if (r < 0) bad += 1;
        }
        free(buf);
    }
    return bad ? fail(AFCODEC_E_CODEC, "one or more chunks failed to decode (see results[])") : AFCODEC_OK;
}
/* Byte ranges of files packed back to back into ONE buffer (the compressed chunk files of the decode-in-HBM route):
 * range i lands at dst + out_off[i], out_off[i + 1] = out_off[i] + its size rounded up to `align`; results[i] = its size,
 * -100 for a missing file (or an empty path), which takes no room.  The sizes are found here (fstat on the team) — the
 * caller does not stat the files first — and the bytes are read in 1 MiB pieces spread over the team. */
int afcodec_read_packed(int64_t n, const char* const* paths, const int64_t* offsets, const int64_t* lengths, void* dst, int64_t cap,
                        int64_t align, int nthreads, int64_t* out_off, int64_t* results) {
    enum { PIECE = 1 << 20 };
    const int nt_copy = nt_copy_default();
    if (n < 0 || !dst || !out_off || !results || align < 1) return fail(AFCODEC_E_SIZE, "read_packed: bad arguments");
    if (nthreads < 1) nthreads = 1;
    out_off[0] = 0;
    if (n == 0) return AFCODEC_OK;
    int64_t* first = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
    int64_t* foff = (int64_t*)malloc((size_t)n * sizeof(int64_t));
    if (!first || !foff) { free(first); free(foff); return fail(AFCODEC_E_SIZE, "out of memory"); }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 8)
    for (int64_t i = 0; i < n; ++i) results[i] = stat_range(paths[i], offsets, lengths, i, &foff[i]);
    int bad = 0;
    first[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t sz = results[i] > 0 ? results[i] : 0;
        if (results[i] < 0 && results[i] != -100) bad += 1;
        out_off[i + 1] = out_off[i] + (sz + align - 1) / align * align;
        first[i + 1] = first[i] + (sz + PIECE - 1) / PIECE;
    }
    int too_big = !bad && out_off[n] > cap;
    if (!bad && !too_big) {
        const int64_t npieces = first[n];
#pragma omp parallel num_threads(nthreads)
        {
            fd_cache fc = {-1, -1};
#pragma omp for schedule(dynamic, 1)
            for (int64_t q = 0; q < npieces; ++q) {
                int64_t lo = 0, hi = n;                               /* the file of piece q: first[lo] <= q < first[lo + 1] */
                while (hi - lo > 1) { const int64_t mid = (lo + hi) / 2; if (first[mid] <= q) lo = mid; else hi = mid; }
                while (first[lo + 1] <= q) ++lo;
                const int64_t at = (q - first[lo]) * PIECE, len = results[lo] - at < PIECE ? results[lo] - at : PIECE;
                int64_t o;
#pragma omp atomic read
                o = foff[lo];
                if (o < 0) continue;
                const int64_t done = read_piece_of(&fc, lo, paths[lo], (uint8_t*)dst + out_off[lo] + at, len, o + at, nt_copy);
                if (done != len) {
#pragma omp atomic write
                    foff[lo] = done == -2 ? -2 : -1;                  /* marks the file as failed (-2: it changed under the read) */
                }
            }
            if (fc.fd >= 0) close(fc.fd);
        }
    }
    int changed = 0;
    for (int64_t i = 0; i < n; ++i)
        if (results[i] >= 0 && foff[i] < 0) { changed += foff[i] == -2; results[i] = AFCODEC_E_FORMAT; bad += 1; }
    free(first); free(foff);
    if (too_big) return fail(AFCODEC_E_SIZE, "read_packed: the files do not fit the buffer");
    if (changed) return fail(AFCODEC_E_CODEC, "a chunk file shrank between stat and read: the store is being rewritten (see results[])");
    return bad ? fail(AFCODEC_E_CODEC, "one or more chunk files could not be read (see results[])") : AFCODEC_OK;
}

int afcodec_decode_files(int kind, int64_t n, const char* const* paths, void* const* dsts, const int64_t* dstsizes,
                         int nthreads, int64_t* results) {
    return afcodec_decode_ranges(kind, n, paths, NULL, NULL, dsts, dstsizes, nthreads, results);
}
int afcodec_blosc_decode_files(int64_t n, const char* const* paths, void* const* dsts, const int64_t* dstsizes,
                               int nthreads, int64_t* results) {
    return afcodec_decode_files(1, n, paths, dsts, dstsizes, nthreads, results);
}

/* Encoder for the writer side (dataset_to_zarr, synthetic stores of the ingestion benchmark):
 * LZ4, byte shuffle on request, blocks never split.  Readable by any Blosc-1 decoder. */
int64_t afcodec_blosc_bound(int64_t nbytes, int64_t blocksize) {
    if (blocksize <= 0) blocksize = 1 << 16;                     /* the smallest automatic block */
    /* the encoder rounds the block size DOWN to whole elements (and clamps it to the buffer): it may need up to twice the
     * blocks of the nominal size, plus the short last one */
    const int64_t nblocks = 2 * ((nbytes + blocksize - 1) / blocksize) + 2;
    return 16 + nblocks * (4 + 4 * 16) + nbytes + nbytes / 255 + 64;      /* block table + up to 16 stream headers per block */
}
static int64_t blosc_need(int64_t nbytes, int64_t blocksize) {           /* exact need for the block size actually used */
    const int64_t nblocks = (nbytes + blocksize - 1) / blocksize;
    return 16 + nblocks * (4 + 4 * 16) + nbytes + nbytes / 255 + 64;
}
int64_t afcodec_blosc_encode_lz4(const void* srcv, int64_t nbytes, int typesize, int shuffle, int64_t blocksize,
                                 void* dstv, int64_t cap) {
    const uint8_t* src = (const uint8_t*)srcv;
    uint8_t* dst = (uint8_t*)dstv;
    int rc = need_lz4();
    if (rc) return rc;
    if (nbytes < 0 || nbytes > 0x7fffffffLL - 64 || typesize < 1 || typesize > 255) return fail(AFCODEC_E_SIZE, "bad nbytes / typesize");
    /* like c-blosc at level 5: 64 KiB per byte plane (blocks of 64 KiB x typesize, split into one LZ4 stream per plane) */
    if (blocksize <= 0) { blocksize = (int64_t)65536 * (typesize <= 16 ? typesize : 1); if (blocksize > (1 << 20)) blocksize = 1 << 20; }
    blocksize -= blocksize % typesize;
    if (blocksize < typesize) blocksize = typesize;
    if (blocksize > nbytes && nbytes > 0) blocksize = nbytes;
    /* the clamp may have left a block that is not a whole number of elements: round down again (c-blosc's compute_blocksize
     * does), so that a split block divides evenly into its byte planes and the remainder becomes the short, unsplit last
     * block — a 1001-byte buffer of 4-byte elements used to lose its last byte */
    if (blocksize > typesize) blocksize -= blocksize % typesize;
    if (cap < blosc_need(nbytes, blocksize)) return fail(AFCODEC_E_SIZE, "destination smaller than afcodec_blosc_bound()");
    const int do_shuf = shuffle && typesize > 1;
    dst[0] = 2; dst[1] = 1; dst[3] = (uint8_t)typesize;
    put32(dst + 4, (uint32_t)nbytes);
    if (nbytes < 128) {                                      /* tiny buffers are stored */
        dst[2] = 0x02 | (do_shuf ? 0x01 : 0) | (1 << 5) | 0x10;
        put32(dst + 8, (uint32_t)(nbytes ? nbytes : typesize));
        memcpy(dst + 16, src, (size_t)nbytes);
        put32(dst + 12, (uint32_t)(16 + nbytes));
        return 16 + nbytes;
    }
    /* split blocks (flag 0x10 clear), as c-blosc writes LZ4 chunks: a full block of a shuffled buffer is typesize
     * streams, one per byte plane; the short last block is one stream (the reader's rule in blosc_block) */
    dst[2] = (uint8_t)((do_shuf ? 0x01 : 0) | (do_shuf ? 0 : 0x10) | (1 << 5));
    put32(dst + 8, (uint32_t)blocksize);
    const int64_t nblocks = (nbytes + blocksize - 1) / blocksize;
    const int64_t leftover = nbytes % blocksize;
    uint8_t* tmp = do_shuf ? (uint8_t*)malloc((size_t)blocksize) : NULL;
    if (do_shuf && !tmp) return fail(AFCODEC_E_SIZE, "out of memory");
    int64_t pos = 16 + 4 * nblocks;
    for (int64_t b = 0; b < nblocks; ++b) {
        const int last_short = (b == nblocks - 1) && leftover > 0;
        const int64_t bsize = last_short ? leftover : blocksize;
        const uint8_t* in = src + b * blocksize;
        if (do_shuf) { shuffle_bytes(typesize, bsize, in, tmp); in = tmp; }
        put32(dst + 16 + 4 * b, (uint32_t)pos);
        int nsplits = 1;
        if (do_shuf && typesize <= 16 && blocksize / typesize >= 128 && !last_short) nsplits = typesize;
        const int64_t ne = bsize / nsplits;
        for (int j = 0; j < nsplits; ++j) {
            int csz = p_lz4_enc((const char*)in + j * ne, (char*)dst + pos + 4, (int)ne, (int)(ne - 1));   /* 0 if it does not shrink */
            if (csz <= 0) { memcpy(dst + pos + 4, in + j * ne, (size_t)ne); csz = (int)ne; }
            put32(dst + pos, (uint32_t)csz);
            pos += 4 + csz;
        }
    }
    free(tmp);
    put32(dst + 12, (uint32_t)pos);
    return pos;
}

/* Plain Zstandard frames (Zarr compressor id "zstd"). */
int64_t afcodec_zstd_decode(const void* src, int64_t n, void* dst, int64_t cap) {
    int rc = need_zstd();
    if (rc) return rc;
    const size_t got = p_zstd_dec(dst, (size_t)cap, src, (size_t)n);
    if (p_zstd_iserr(got)) return fail(AFCODEC_E_CODEC, "zstd frame failed to decode");
    return (int64_t)got;
}

/* Zstandard frame of src (writer side of Zarr v3 stores); cap >= afcodec_zstd_bound(n). */
int64_t afcodec_zstd_bound(int64_t n) {
    if (need_zstd() || !p_zstd_bound) return n + n / 128 + 512;
    return (int64_t)p_zstd_bound((size_t)n);
}
int64_t afcodec_zstd_encode(const void* src, int64_t n, int level, void* dst, int64_t cap) {
    int rc = need_zstd();
    if (rc) return rc;
    if (!p_zstd_enc) return fail(AFCODEC_E_UNSUPPORTED, "libzstd has no ZSTD_compress");
    const size_t got = p_zstd_enc(dst, (size_t)cap, src, (size_t)n, level);
    if (p_zstd_iserr(got)) return fail(AFCODEC_E_CODEC, "zstd compression failed");
    return (int64_t)got;
}
