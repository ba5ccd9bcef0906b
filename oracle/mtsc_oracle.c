/*
 * mtsc_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked, imported or called by the product).
 *
 * CPU restatement, in plain C, of the arithmetic on mtscomp's per-chunk hot path:
 *
 *   compress   : Writer._compress_chunk            /root/reference/mtscomp.py:375-397
 *                  diff_along_axis                 /root/reference/mtscomp.py:143-159
 *                  chunkd.tobytes(order=...)       /root/reference/mtscomp.py:394
 *                  zlib.compress(bytes)            /root/reference/mtscomp.py:394   (-> system libz)
 *   decompress : Reader.read_chunk                 /root/reference/mtscomp.py:602-635
 *                  zlib.decompress                 /root/reference/mtscomp.py:619   (-> system libz)
 *                  reshape(order=...) + cumsum     /root/reference/mtscomp.py:630-632, 162-169
 *                  np.ascontiguousarray            /root/reference/mtscomp.py:635
 *
 * The DEFLATE/INFLATE arithmetic is NOT in /root/reference: it is the third-party dependency
 * zlib (CPython's `zlib` module -> system libz; 1.2.11 in this image, un-pinned by the reference).
 * This file restates zlib 1.2.11's published algorithm (deflate.c: deflate_fast / deflate_slow /
 * longest_match / fill_window; trees.c: _tr_tally / _tr_flush_block / build_tree / gen_bitlen /
 * gen_codes / scan_tree / send_tree / compress_block; adler32.c; inflate.c semantics per RFC 1950/1951)
 * from the description in SURVEY.md Appendix A.  It is pinned by
 *   (1) tests/test_oracle_*.py: byte-for-byte differential tests against the stdlib `zlib` module
 *       (the same libz the reference calls) on fuzzed and corner-case inputs, and
 *   (2) tests/golden/: .cbin/.ch fixtures produced by importing /root/reference/mtscomp.py
 *       (oracle/gen_golden.py is the generating script).
 *
 * Two formulations of the LZ77 parse are provided:
 *   orc_deflate()        sequential, follows deflate.c statement by statement (absolute positions);
 *   orc_match_tables() + orc_parse_tables()
 *                        the parse-independent per-position candidate tables + state machine of
 *                        SURVEY.md Appendix A.3 -- this is the shape the HIP kernels use, so the GPU
 *                        intermediates can be compared stage by stage.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_E_BUF (-1)      /* output buffer too small */
#define ORC_E_DATA (-3)     /* corrupt stream (Z_DATA_ERROR) */
#define ORC_E_TRUNC (-5)    /* truncated stream (Z_BUF_ERROR) */
#define ORC_E_ARG (-2)
#define ORC_E_MEM (-4)

/* ------------------------------------------------------------------------------------------- */
/* adler32 (RFC 1950)                                                                            */
/* ------------------------------------------------------------------------------------------- */
uint32_t orc_adler32(const uint8_t *buf, long n)
{
    uint32_t a = 1, b = 0;
    long i = 0;
    while (i < n) {
        long m = n - i; if (m > 5552) m = 5552;      /* NMAX: largest m with no u32 overflow */
        for (long k = 0; k < m; k++) { a += buf[i + k]; b += a; }
        a %= 65521u; b %= 65521u;
        i += m;
    }
    return (b << 16) | a;
}

/* ------------------------------------------------------------------------------------------- */
/* delta / cumsum transforms (numpy semantics: integer arithmetic wraps in the array's dtype)    */
/*   flags: bit0 = do_time_diff, bit1 = do_spatial_diff, bit2 = chunk_order 'F'                  */
/* ------------------------------------------------------------------------------------------- */
static uint64_t ld(const uint8_t *p, int sz)
{
    uint64_t v = 0; memcpy(&v, p, (size_t)sz); return v;   /* little-endian host */
}
static void st(uint8_t *p, int sz, uint64_t v) { memcpy(p, &v, (size_t)sz); }

/* raw: C-order (nt, nc) items of `sz` bytes.  out: the byte stream handed to zlib.compress. */
/* float items (flags & 8): np.diff / np.cumsum in the item type, every operation rounded in that type, in numpy's order:
   r[t] = r[t-1] + d[t] one after the other (mtscomp.py:150-155, :162-169 on float arrays, tests.py:212-237) */
#define ORC_FLOAT_TRANSFORMS(T, NAME)                                                                               \
    static void NAME##_delta(const T *x, long nt, long nc, int flags, T *out)                                       \
    {                                                                                                               \
        long n = nt * nc;                                                                                           \
        T *d = (T *)malloc(sizeof(T) * (size_t)(n ? n : 1));                                                        \
        for (long t = 0; t < nt; t++)                                                                               \
            for (long c = 0; c < nc; c++) {                                                                         \
                volatile T v = x[t * nc + c];                                                                       \
                if ((flags & 1) && t > 0) v = x[t * nc + c] - x[(t - 1) * nc + c];                                  \
                d[t * nc + c] = v;                                                                                  \
            }                                                                                                       \
        if (flags & 2)                                                                                              \
            for (long t = 0; t < nt; t++)                                                                           \
                for (long c = nc - 1; c > 0; c--) { volatile T v = d[t * nc + c] - d[t * nc + c - 1]; d[t * nc + c] = v; } \
        if (flags & 4) { for (long c = 0; c < nc; c++) for (long t = 0; t < nt; t++) out[c * nt + t] = d[t * nc + c]; } \
        else memcpy(out, d, sizeof(T) * (size_t)n);                                                                 \
        free(d);                                                                                                    \
    }                                                                                                               \
    static void NAME##_cumsum(const T *s, long nt, long nc, int flags, T *out)                                      \
    {                                                                                                               \
        long n = nt * nc;                                                                                           \
        if (flags & 4) { for (long c = 0; c < nc; c++) for (long t = 0; t < nt; t++) out[t * nc + c] = s[c * nt + t]; } \
        else memcpy(out, s, sizeof(T) * (size_t)n);                                                                 \
        if (flags & 2)                                                                                              \
            for (long t = 0; t < nt; t++)                                                                           \
                for (long c = 1; c < nc; c++) { volatile T v = out[t * nc + c] + out[t * nc + c - 1]; out[t * nc + c] = v; } \
        if (flags & 1)                                                                                              \
            for (long t = 1; t < nt; t++)                                                                           \
                for (long c = 0; c < nc; c++) { volatile T v = out[t * nc + c] + out[(t - 1) * nc + c]; out[t * nc + c] = v; } \
    }
ORC_FLOAT_TRANSFORMS(float, orc_f32)
ORC_FLOAT_TRANSFORMS(double, orc_f64)

int orc_delta_transpose(const uint8_t *raw, long nt, long nc, int sz, int flags, uint8_t *out)
{
    if (flags & 8) {
        if (sz == 4) orc_f32_delta((const float *)raw, nt, nc, flags, (float *)out);
        else if (sz == 8) orc_f64_delta((const double *)raw, nt, nc, flags, (double *)out);
        else return ORC_E_ARG;
        return ORC_OK;
    }
    if (sz != 1 && sz != 2 && sz != 4 && sz != 8) return ORC_E_ARG;
    long n = nt * nc;
    uint64_t *d = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    if (!d) return ORC_E_MEM;
    /* time diff: d[0,:] = x[0,:]; d[t,:] = x[t,:] - x[t-1,:]          (mtscomp.py:150-155) */
    for (long t = 0; t < nt; t++)
        for (long c = 0; c < nc; c++) {
            uint64_t x = ld(raw + (t * nc + c) * sz, sz);
            if ((flags & 1) && t > 0) x -= ld(raw + ((t - 1) * nc + c) * sz, sz);
            d[t * nc + c] = x;
        }
    /* spatial diff applied AFTER the time diff: d[:,0] kept, d[:,c] -= d[:,c-1] (mtscomp.py:382) */
    if (flags & 2)
        for (long t = 0; t < nt; t++)
            for (long c = nc - 1; c > 0; c--) d[t * nc + c] -= d[t * nc + c - 1];
    /* tobytes(order): 'F' = channel-major, 'C' = as is                        (mtscomp.py:394) */
    if (flags & 4) {
        for (long c = 0; c < nc; c++)
            for (long t = 0; t < nt; t++) st(out + (c * nt + t) * sz, sz, d[t * nc + c]);
    } else {
        for (long i = 0; i < n; i++) st(out + i * sz, sz, d[i]);
    }
    free(d);
    return ORC_OK;
}

/* stream: inflated bytes.  out: C-order (nt, nc) array, the value read_chunk returns. */
int orc_cumsum_transpose(const uint8_t *stream, long nt, long nc, int sz, int flags, uint8_t *out)
{
    if (flags & 8) {
        if (sz == 4) orc_f32_cumsum((const float *)stream, nt, nc, flags, (float *)out);
        else if (sz == 8) orc_f64_cumsum((const double *)stream, nt, nc, flags, (double *)out);
        else return ORC_E_ARG;
        return ORC_OK;
    }
    if (sz != 1 && sz != 2 && sz != 4 && sz != 8) return ORC_E_ARG;
    long n = nt * nc;
    uint64_t *d = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    if (!d) return ORC_E_MEM;
    /* reshape((nt,nc), order)                                                 (mtscomp.py:630) */
    if (flags & 4) {
        for (long c = 0; c < nc; c++)
            for (long t = 0; t < nt; t++) d[t * nc + c] = ld(stream + (c * nt + t) * sz, sz);
    } else {
        for (long i = 0; i < n; i++) d[i] = ld(stream + i * sz, sz);
    }
    /* cumsum along space first, then along time                           (mtscomp.py:631-632) */
    if (flags & 2)
        for (long t = 0; t < nt; t++)
            for (long c = 1; c < nc; c++) d[t * nc + c] += d[t * nc + c - 1];
    if (flags & 1)
        for (long t = 1; t < nt; t++)
            for (long c = 0; c < nc; c++) d[t * nc + c] += d[(t - 1) * nc + c];
    for (long i = 0; i < n; i++) st(out + i * sz, sz, d[i]);     /* truncation = dtype wrap */
    free(d);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* DEFLATE constants (zlib 1.2.11 deflate.h / trees.c)                                           */
/* ------------------------------------------------------------------------------------------- */
#define MIN_MATCH 3
#define MAX_MATCH 258
#define WSIZE 32768
#define MIN_LOOKAHEAD (MAX_MATCH + MIN_MATCH + 1)       /* 262 */
#define MAX_DIST (WSIZE - MIN_LOOKAHEAD)                /* 32506 */
#define TOO_FAR 4096
#define HASH_MASK 0x7fff
#define LIT_BUFSIZE 16384                               /* 1 << (memLevel 8 + 6) */
#define L_CODES 286
#define D_CODES 30
#define BL_CODES 19
#define HEAP_SIZE (2 * L_CODES + 1)
#define END_BLOCK 256
#define REP_3_6 16
#define REPZ_3_10 17
#define REPZ_11_138 18

static const int extra_lbits[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const int extra_dbits[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const int extra_blbits[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
static const uint8_t bl_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

typedef struct { int good, lazy, nice, chain, slow; } level_cfg;
static const level_cfg LEVELS[10] = {
    {0, 0, 0, 0, 0},
    {4, 4, 8, 4, 0}, {4, 5, 16, 8, 0}, {4, 6, 32, 32, 0},
    {4, 4, 16, 16, 1}, {8, 16, 32, 32, 1}, {8, 16, 128, 128, 1},
    {8, 32, 128, 256, 1}, {32, 128, 258, 1024, 1}, {32, 258, 258, 4096, 1}};

static int base_length[29], base_dist[30];
static uint8_t length_code[256], dist_code[512];
static uint8_t static_llen[288];
static int tables_ready = 0;

static void init_tables(void)
{
    if (tables_ready) return;
    int length = 0, code, n, dist = 0;
    for (code = 0; code < 28; code++) {
        base_length[code] = length;
        for (n = 0; n < (1 << extra_lbits[code]); n++) length_code[length++] = (uint8_t)code;
    }
    length_code[length - 1] = (uint8_t)code;           /* length 258 -> code 28 (symbol 285) */
    base_length[28] = 0;
    for (code = 0; code < 16; code++) {
        base_dist[code] = dist;
        for (n = 0; n < (1 << extra_dbits[code]); n++) dist_code[dist++] = (uint8_t)code;
    }
    dist >>= 7;
    for (; code < D_CODES; code++) {
        base_dist[code] = dist << 7;
        for (n = 0; n < (1 << (extra_dbits[code] - 7)); n++) dist_code[256 + dist++] = (uint8_t)code;
    }
    for (n = 0; n <= 143; n++) static_llen[n] = 8;
    for (; n <= 255; n++) static_llen[n] = 9;
    for (; n <= 279; n++) static_llen[n] = 7;
    for (; n <= 287; n++) static_llen[n] = 8;
    tables_ready = 1;
}
static inline int d_code(int dist) { return dist < 256 ? dist_code[dist] : dist_code[256 + (dist >> 7)]; }

static unsigned bi_reverse(unsigned code, int len)
{
    unsigned res = 0;
    do { res |= code & 1; code >>= 1; res <<= 1; } while (--len > 0);
    return res >> 1;
}

/* ------------------------------------------------------------------------------------------- */
/* bit writer (LSB first)                                                                        */
/* ------------------------------------------------------------------------------------------- */
typedef struct { uint8_t *out; long cap, pos; uint64_t acc; int nbits; int err; } bitw;

static void bw_put(bitw *w, unsigned v, int n)
{
    w->acc |= (uint64_t)v << w->nbits; w->nbits += n;
    while (w->nbits >= 8) {
        if (w->pos < w->cap) w->out[w->pos] = (uint8_t)w->acc; else w->err = 1;
        w->pos++; w->acc >>= 8; w->nbits -= 8;
    }
}
static void bw_align(bitw *w) { if (w->nbits) bw_put(w, 0, 8 - w->nbits); }
static void bw_byte(bitw *w, unsigned v) { bw_put(w, v & 0xff, 8); }

/* ------------------------------------------------------------------------------------------- */
/* trees.c: one block                                                                            */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    uint16_t freq[HEAP_SIZE];
    uint16_t len[HEAP_SIZE + 1];
    uint16_t code[HEAP_SIZE];
    uint16_t dad[HEAP_SIZE];
    int max_code;
} tree_t;

typedef struct {
    tree_t lt, dt, bl;
    long opt_len, static_len;
    int heap[HEAP_SIZE]; int heap_len, heap_max;
    uint8_t depth[HEAP_SIZE];
    uint16_t bl_count[16];
} blk_t;

#define SMALLER(t, n, m) ((t)->freq[n] < (t)->freq[m] || ((t)->freq[n] == (t)->freq[m] && b->depth[n] <= b->depth[m]))

static void pqdownheap(blk_t *b, tree_t *t, int k)
{
    int v = b->heap[k], j = k << 1;
    while (j <= b->heap_len) {
        if (j < b->heap_len && SMALLER(t, b->heap[j + 1], b->heap[j])) j++;
        if (SMALLER(t, v, b->heap[j])) break;
        b->heap[k] = b->heap[j]; k = j; j <<= 1;
    }
    b->heap[k] = v;
}

/* kind: 0 literal/length, 1 distance, 2 bit-length tree */
static void build_tree(blk_t *b, tree_t *t, int kind)
{
    const int elems = kind == 0 ? L_CODES : kind == 1 ? D_CODES : BL_CODES;
    const int max_length = kind == 2 ? 7 : 15;
    const int *extra = kind == 0 ? extra_lbits : kind == 1 ? extra_dbits : extra_blbits;
    const int base = kind == 0 ? 257 : 0;
    int n, m, max_code = -1, node;
    b->heap_len = 0; b->heap_max = HEAP_SIZE;
    for (n = 0; n < elems; n++) {
        if (t->freq[n] != 0) { b->heap[++b->heap_len] = max_code = n; b->depth[n] = 0; }
        else t->len[n] = 0;
    }
    while (b->heap_len < 2) {
        node = b->heap[++b->heap_len] = (max_code < 2 ? ++max_code : 0);
        t->freq[node] = 1; b->depth[node] = 0; b->opt_len--;
        if (kind == 0) b->static_len -= static_llen[node];
        else if (kind == 1) b->static_len -= 5;
    }
    t->max_code = max_code;
    for (n = b->heap_len / 2; n >= 1; n--) pqdownheap(b, t, n);
    node = elems;
    do {
        n = b->heap[1]; b->heap[1] = b->heap[b->heap_len--]; pqdownheap(b, t, 1);
        m = b->heap[1];
        b->heap[--b->heap_max] = n; b->heap[--b->heap_max] = m;
        t->freq[node] = (uint16_t)(t->freq[n] + t->freq[m]);
        b->depth[node] = (uint8_t)((b->depth[n] >= b->depth[m] ? b->depth[n] : b->depth[m]) + 1);
        t->dad[n] = t->dad[m] = (uint16_t)node;
        b->heap[1] = node++;
        pqdownheap(b, t, 1);
    } while (b->heap_len >= 2);
    b->heap[--b->heap_max] = b->heap[1];

    /* gen_bitlen */
    int h, bits, xbits, overflow = 0;
    for (bits = 0; bits <= 15; bits++) b->bl_count[bits] = 0;
    t->len[b->heap[b->heap_max]] = 0;
    for (h = b->heap_max + 1; h < HEAP_SIZE; h++) {
        n = b->heap[h];
        bits = t->len[t->dad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        t->len[n] = (uint16_t)bits;
        if (n > max_code) continue;
        b->bl_count[bits]++;
        xbits = 0; if (n >= base) xbits = extra[n - base];
        long f = t->freq[n];
        b->opt_len += f * (bits + xbits);
        if (kind == 0) b->static_len += f * (static_llen[n] + xbits);
        else if (kind == 1) b->static_len += f * (5 + xbits);
    }
    if (overflow > 0) {
        do {
            bits = max_length - 1;
            while (b->bl_count[bits] == 0) bits--;
            b->bl_count[bits]--; b->bl_count[bits + 1] += 2; b->bl_count[max_length]--;
            overflow -= 2;
        } while (overflow > 0);
        for (bits = max_length; bits != 0; bits--) {
            n = b->bl_count[bits];
            while (n != 0) {
                m = b->heap[--h];
                if (m > max_code) continue;
                if (t->len[m] != (unsigned)bits) {
                    b->opt_len += ((long)bits - (long)t->len[m]) * (long)t->freq[m];
                    t->len[m] = (uint16_t)bits;
                }
                n--;
            }
        }
    }
    /* gen_codes */
    unsigned next_code[16], code = 0;
    for (bits = 1; bits <= 15; bits++) { code = (code + b->bl_count[bits - 1]) << 1; next_code[bits] = code; }
    for (n = 0; n <= max_code; n++) {
        int l = t->len[n];
        if (l == 0) continue;
        t->code[n] = (uint16_t)bi_reverse(next_code[l]++, l);
    }
}

static void scan_tree(blk_t *b, tree_t *t, int max_code)
{
    int n, prevlen = -1, curlen, nextlen = t->len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    t->len[max_code + 1] = 0xffff;
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = t->len[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) b->bl.freq[curlen] += count;
        else if (curlen != 0) { if (curlen != prevlen) b->bl.freq[curlen]++; b->bl.freq[REP_3_6]++; }
        else if (count <= 10) b->bl.freq[REPZ_3_10]++;
        else b->bl.freq[REPZ_11_138]++;
        count = 0; prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

static void send_tree(blk_t *b, bitw *w, tree_t *t, int max_code)
{
    int n, prevlen = -1, curlen, nextlen = t->len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
#define SEND_BL(c) bw_put(w, b->bl.code[c], b->bl.len[c])
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = t->len[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) { do { SEND_BL(curlen); } while (--count != 0); }
        else if (curlen != 0) {
            if (curlen != prevlen) { SEND_BL(curlen); count--; }
            SEND_BL(REP_3_6); bw_put(w, (unsigned)(count - 3), 2);
        } else if (count <= 10) { SEND_BL(REPZ_3_10); bw_put(w, (unsigned)(count - 3), 3); }
        else { SEND_BL(REPZ_11_138); bw_put(w, (unsigned)(count - 11), 7); }
        count = 0; prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
#undef SEND_BL
}

/* a token: dist == 0 -> literal byte `lc`; else match of length lc+3 at distance dist */
typedef struct { uint16_t dist; uint16_t lc; } tok_t;

static void compress_block(bitw *w, const tok_t *tk, long ntok,
                           const uint16_t *lcode, const uint16_t *llen,
                           const uint16_t *dcode, const uint16_t *dlen)
{
    for (long i = 0; i < ntok; i++) {
        unsigned dist = tk[i].dist, lc = tk[i].lc;
        if (dist == 0) { bw_put(w, lcode[lc], llen[lc]); continue; }
        unsigned code = length_code[lc];
        bw_put(w, lcode[code + 257], llen[code + 257]);
        int extra = extra_lbits[code];
        if (extra) bw_put(w, lc - (unsigned)base_length[code], extra);
        dist--;
        code = (unsigned)d_code((int)dist);
        bw_put(w, dcode[code], dlen[code]);
        extra = extra_dbits[code];
        if (extra) bw_put(w, dist - (unsigned)base_dist[code], extra);
    }
    bw_put(w, lcode[END_BLOCK], llen[END_BLOCK]);
}

/* per-block report, filled when the caller asks for it (debug/parity of GPU stages) */
typedef struct {
    long tok_start, ntok;       /* token range of this block */
    long in_start, in_len;      /* input byte range covered */
    long opt_len, static_len;   /* in bits, as trees.c computes them (before the +3) */
    int btype, last;            /* 0 stored, 1 fixed, 2 dynamic */
    long bit_start, bit_end;    /* bit offsets in the zlib stream (incl. the 2-byte header) */
} orc_block_info;

/* _tr_flush_block */
static void flush_block(bitw *w, const uint8_t *in, long in_start, long in_len, int buf_ok,
                        const tok_t *tk, long ntok, int last, orc_block_info *info)
{
    static uint16_t s_lcode[288], s_llen[288], s_dcode[30], s_dlen[30];
    static int s_ready = 0;
    if (!s_ready) {
        uint16_t cnt[16] = {0};
        for (int n = 0; n < 288; n++) { s_llen[n] = static_llen[n]; cnt[s_llen[n]]++; }
        unsigned next_code[16], code = 0;
        for (int bits = 1; bits <= 15; bits++) { code = (code + cnt[bits - 1]) << 1; next_code[bits] = code; }
        for (int n = 0; n < 288; n++) s_lcode[n] = (uint16_t)bi_reverse(next_code[s_llen[n]]++, s_llen[n]);
        for (int n = 0; n < 30; n++) { s_dlen[n] = 5; s_dcode[n] = (uint16_t)bi_reverse((unsigned)n, 5); }
        s_ready = 1;
    }
    blk_t *b = (blk_t *)calloc(1, sizeof(blk_t));
    /* tally (the _tr_tally calls, replayed) */
    for (long i = 0; i < ntok; i++) {
        if (tk[i].dist == 0) b->lt.freq[tk[i].lc]++;
        else { b->lt.freq[length_code[tk[i].lc] + 257]++; b->dt.freq[d_code(tk[i].dist - 1)]++; }
    }
    b->lt.freq[END_BLOCK] = 1;
    build_tree(b, &b->lt, 0);
    build_tree(b, &b->dt, 1);
    /* build_bl_tree */
    scan_tree(b, &b->lt, b->lt.max_code);
    scan_tree(b, &b->dt, b->dt.max_code);
    build_tree(b, &b->bl, 2);
    int max_blindex;
    for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--)
        if (b->bl.len[bl_order[max_blindex]] != 0) break;
    b->opt_len += 3 * (max_blindex + 1) + 5 + 5 + 4;
    long opt_lenb = (b->opt_len + 3 + 7) >> 3, static_lenb = (b->static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
    long bit_start = w->pos * 8 + w->nbits;
    int btype;
    if (in_len + 4 <= opt_lenb && buf_ok) {
        btype = 0;
        bw_put(w, (unsigned)last, 3); bw_align(w);
        bw_byte(w, (unsigned)in_len); bw_byte(w, (unsigned)in_len >> 8);
        bw_byte(w, ~(unsigned)in_len); bw_byte(w, (~(unsigned)in_len) >> 8);
        for (long i = 0; i < in_len; i++) bw_byte(w, in[in_start + i]);
    } else if (static_lenb == opt_lenb) {
        btype = 1;
        bw_put(w, (1u << 1) + (unsigned)last, 3);
        compress_block(w, tk, ntok, s_lcode, s_llen, s_dcode, s_dlen);
    } else {
        btype = 2;
        bw_put(w, (2u << 1) + (unsigned)last, 3);
        bw_put(w, (unsigned)(b->lt.max_code + 1 - 257), 5);
        bw_put(w, (unsigned)(b->dt.max_code + 1 - 1), 5);
        bw_put(w, (unsigned)(max_blindex + 1 - 4), 4);
        for (int rank = 0; rank < max_blindex + 1; rank++) bw_put(w, b->bl.len[bl_order[rank]], 3);
        send_tree(b, w, &b->lt, b->lt.max_code);
        send_tree(b, w, &b->dt, b->dt.max_code);
        compress_block(w, tk, ntok, b->lt.code, b->lt.len, b->dt.code, b->dt.len);
    }
    if (last) bw_align(w);
    if (info) {
        info->ntok = ntok; info->in_start = in_start; info->in_len = in_len;
        info->opt_len = b->opt_len; info->static_len = b->static_len;
        info->btype = btype; info->last = last;
        info->bit_start = bit_start; info->bit_end = w->pos * 8 + w->nbits;
    }
    free(b);
}

/* ------------------------------------------------------------------------------------------- */
/* deflate.c on absolute positions                                                               */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *in; long n;
    int32_t *head, *prev;
    level_cfg cfg;
} lz_t;

static inline unsigned hash3(const uint8_t *p) { return ((p[0] << 10) ^ (p[1] << 5) ^ p[2]) & HASH_MASK; }

static inline long insert_string(lz_t *s, long pos)
{
    unsigned h = hash3(s->in + pos);
    long hh = s->head[h];
    s->prev[pos] = (int32_t)hh; s->head[h] = (int32_t)pos;
    return hh;
}

static int longest_match(lz_t *s, long p, long cur, int prev_length, long *match_start)
{
    int chain = s->cfg.chain, best = prev_length, nice = s->cfg.nice;
    long lookahead = s->n - p;
    if (prev_length >= s->cfg.good) chain >>= 2;
    if ((long)nice > lookahead) nice = (int)lookahead;
    long limit = p > MAX_DIST ? p - MAX_DIST : 0;
    int maxlen = lookahead < MAX_MATCH ? (int)lookahead : MAX_MATCH;
    const uint8_t *scan = s->in + p;
    do {
        const uint8_t *m = s->in + cur;
        int len = 0;
        while (len < maxlen && m[len] == scan[len]) len++;
        if (len > best) { *match_start = cur; best = len; if (len >= nice) break; }
    } while ((cur = s->prev[cur]) > limit && --chain != 0);
    return (long)best <= lookahead ? best : (int)lookahead;
}

/* fill_window's slide bookkeeping (only observable through the `buf != NULL` test of
   _tr_flush_block): wbase = absolute position of window[0], wend = end of data read so far. */
typedef struct { long wbase, wend, n; } win_t;
static void fill_window(win_t *w, long p)
{
    long lookahead = w->wend - p;
    if (lookahead >= MIN_LOOKAHEAD) return;
    do {
        long more = 2L * WSIZE - lookahead - (p - w->wbase);
        if (p - w->wbase >= WSIZE + MAX_DIST) { w->wbase += WSIZE; more += WSIZE; }
        if (w->wend >= w->n) break;                            /* avail_in == 0 */
        long r = w->n - w->wend; if (r > more) r = more;
        w->wend += r; lookahead += r;
    } while (lookahead < MIN_LOOKAHEAD && w->wend < w->n);
}

/* Number of window slides that have happened once the loop top at absolute position q has run
   (closed form used by the HIP path; cross-checked against fill_window() in the tests). */
long orc_slides_at(long q, long n)
{
    long k = 0;
    for (;;) {
        long edge = (k + 2) * (long)WSIZE;                     /* (k+1)-th slide threshold */
        long theta = edge - 261 - (n < edge ? 1 : 0);
        if (q >= theta) k++; else break;
    }
    return k;
}

typedef struct {
    bitw w;
    tok_t *tk; long ntok, tok_cap;           /* all tokens of the stream (kept for reports) */
    long blk_tok0;                            /* first token of the open block */
    long block_start;
    orc_block_info *binfo; long nblk, blk_cap;
    long *tokpos;                             /* input position of each token (optional) */
} emit_t;

static void do_flush(emit_t *e, const uint8_t *in, long strstart, int last, const win_t *win)
{
    orc_block_info info; memset(&info, 0, sizeof info);
    info.tok_start = e->blk_tok0;
    flush_block(&e->w, in, e->block_start, strstart - e->block_start, e->block_start >= win->wbase,
                e->tk + e->blk_tok0, e->ntok - e->blk_tok0, last, &info);
    if (e->binfo && e->nblk < e->blk_cap) e->binfo[e->nblk] = info;
    e->nblk++;
    e->blk_tok0 = e->ntok; e->block_start = strstart;
}
static inline int tally(emit_t *e, unsigned dist, unsigned lc, long pos)
{
    if (e->tokpos) e->tokpos[e->ntok] = pos;
    e->tk[e->ntok].dist = (uint16_t)dist; e->tk[e->ntok].lc = (uint16_t)lc; e->ntok++;
    return (e->ntok - e->blk_tok0) == LIT_BUFSIZE - 1;
}

/*
 * zlib.compress(in, level) restated.  Returns the stream length or a negative error.
 * Optional reports: tokens (dist,lc pairs, capacity n+1), tokpos, block info.
 */
long orc_deflate(const uint8_t *in, long n, int level, uint8_t *out, long out_cap,
                 uint16_t *tokens_out /* 2*ntok */, long *tokpos_out, long *ntok_out,
                 orc_block_info *binfo, long binfo_cap, long *nblk_out)
{
    init_tables();
    if (level == -1) level = 6;
    if (level < 1 || level > 9) return ORC_E_ARG;
    lz_t s; s.in = in; s.n = n; s.cfg = LEVELS[level];
    s.head = (int32_t *)calloc(HASH_MASK + 1, sizeof(int32_t));
    s.prev = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
    emit_t e; memset(&e, 0, sizeof e);
    e.tk = (tok_t *)malloc(sizeof(tok_t) * (size_t)(n + 1));
    e.tokpos = tokpos_out;
    e.binfo = binfo; e.blk_cap = binfo_cap;
    e.w.out = out; e.w.cap = out_cap;
    if (!s.head || !s.prev || !e.tk) { free(s.head); free(s.prev); free(e.tk); return ORC_E_MEM; }
    win_t win = {0, 0, n};

    /* zlib header (deflate.c: deflate(), INIT_STATE) */
    unsigned lf = level < 2 ? 0 : level < 6 ? 1 : level == 6 ? 2 : 3;
    unsigned header = (0x78u << 8) | (lf << 6);
    header += 31 - (header % 31);
    bw_byte(&e.w, header >> 8); bw_byte(&e.w, header);

    long p = 0;
    if (s.cfg.slow) {
        /* deflate_slow */
        int match_length = MIN_MATCH - 1, match_available = 0, prev_length;
        long match_start = 0, prev_match;
        for (;;) {
            fill_window(&win, p);
            if (n - p == 0) break;
            long hash_head = 0;
            if (n - p >= MIN_MATCH) hash_head = insert_string(&s, p);
            prev_length = match_length; prev_match = match_start;
            match_length = MIN_MATCH - 1;
            if (hash_head != 0 && prev_length < s.cfg.lazy && p - hash_head <= MAX_DIST) {
                match_length = longest_match(&s, p, hash_head, prev_length, &match_start);
                if (match_length <= 5 && (match_length == MIN_MATCH && p - match_start > TOO_FAR))
                    match_length = MIN_MATCH - 1;
            }
            if (prev_length >= MIN_MATCH && match_length <= prev_length) {
                long max_insert = n - MIN_MATCH;
                int bflush = tally(&e, (unsigned)(p - 1 - prev_match), (unsigned)(prev_length - MIN_MATCH), p - 1);
                prev_length -= 2;
                do { if (++p <= max_insert) insert_string(&s, p); } while (--prev_length != 0);
                match_available = 0; match_length = MIN_MATCH - 1;
                p++;
                if (bflush) do_flush(&e, in, p, 0, &win);
            } else if (match_available) {
                int bflush = tally(&e, 0, in[p - 1], p - 1);
                if (bflush) do_flush(&e, in, p, 0, &win);
                p++;
            } else { match_available = 1; p++; }
        }
        if (match_available) tally(&e, 0, in[p - 1], p - 1);
    } else {
        /* deflate_fast */
        for (;;) {
            fill_window(&win, p);
            if (n - p == 0) break;
            long hash_head = 0, match_start = 0;
            int match_length = MIN_MATCH - 1, bflush;
            if (n - p >= MIN_MATCH) hash_head = insert_string(&s, p);
            if (hash_head != 0 && p - hash_head <= MAX_DIST)
                match_length = longest_match(&s, p, hash_head, MIN_MATCH - 1, &match_start);
            if (match_length >= MIN_MATCH) {
                bflush = tally(&e, (unsigned)(p - match_start), (unsigned)(match_length - MIN_MATCH), p);
                long lookahead = n - p - match_length;
                if (match_length <= s.cfg.lazy /* max_insert_length */ && lookahead >= MIN_MATCH) {
                    match_length--;
                    do { p++; insert_string(&s, p); } while (--match_length != 0);
                    p++;
                } else p += match_length;
            } else {
                bflush = tally(&e, 0, in[p], p);
                p++;
            }
            if (bflush) do_flush(&e, in, p, 0, &win);
        }
    }
    do_flush(&e, in, p, 1, &win);                       /* FLUSH_BLOCK(s, 1) under Z_FINISH */
    uint32_t ad = orc_adler32(in, n);
    bw_byte(&e.w, ad >> 24); bw_byte(&e.w, ad >> 16); bw_byte(&e.w, ad >> 8); bw_byte(&e.w, ad);

    long total = e.w.pos;
    int err = e.w.err;
    if (tokens_out) for (long i = 0; i < e.ntok; i++) { tokens_out[2 * i] = e.tk[i].dist; tokens_out[2 * i + 1] = e.tk[i].lc; }
    if (ntok_out) *ntok_out = e.ntok;
    if (nblk_out) *nblk_out = e.nblk;
    free(s.head); free(s.prev); free(e.tk);
    return err ? ORC_E_BUF : total;
}

long orc_compress_bound(long n) { return n + (n >> 12) + (n >> 14) + (n >> 25) + 13; }

/* ------------------------------------------------------------------------------------------- */
/* parse-independent candidate tables + state machine (levels 4..9)         */
/*   t_full[p], t_quarter[p] = (len << 16) | dist, or 0 when no candidate of length >= 3:        */
/*   best match over the first `chain` / `chain>>2` hash-chain predecessors of p.                */
/* ------------------------------------------------------------------------------------------- */
int orc_match_tables(const uint8_t *in, long n, int level, uint32_t *t_full, uint32_t *t_quarter)
{
    init_tables();
    if (level == -1) level = 6;
    if (level < 4 || level > 9) return ORC_E_ARG;
    level_cfg cfg = LEVELS[level];
    int32_t *head = (int32_t *)calloc(HASH_MASK + 1, sizeof(int32_t));
    int32_t *prev = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
    if (!head || !prev) { free(head); free(prev); return ORC_E_MEM; }
    for (long p = 0; p < n; p++) {
        t_full[p] = 0; t_quarter[p] = 0;
        if (n - p < MIN_MATCH) continue;
        unsigned h = hash3(in + p);
        long cur = head[h]; prev[p] = (int32_t)cur; head[h] = (int32_t)p;
        if (cur == 0 || p - cur > MAX_DIST) continue;
        long lookahead = n - p;
        int nice = cfg.nice; if ((long)nice > lookahead) nice = (int)lookahead;
        int maxlen = lookahead < MAX_MATCH ? (int)lookahead : MAX_MATCH;
        long limit = p > MAX_DIST ? p - MAX_DIST : 0;
        int best = 2, budget = cfg.chain, qbudget = cfg.chain >> 2, k = 0, stopped = 0, q_done = 0;
        long bstart = 0;
#define SNAP() (best >= 3 ? ((uint32_t)best << 16) | (uint32_t)(p - bstart) : 0)
        do {
            int len = 0;
            while (len < maxlen && in[cur + len] == in[p + len]) len++;
            k++;
            if (len > best) { best = len; bstart = cur; if (len >= nice) stopped = 1; }
            /* the walk with budget chain>>2 sees exactly the first qbudget candidates */
            if (!q_done && (k == qbudget || stopped)) { t_quarter[p] = SNAP(); q_done = 1; }
            if (stopped) break;
        } while ((cur = prev[cur]) > limit && --budget != 0);
        if (!q_done) t_quarter[p] = SNAP();
        t_full[p] = SNAP();
#undef SNAP
    }
    free(head); free(prev);
    return ORC_OK;
}

/* deflate_slow driven by the tables only.  tokens: (dist, lc) pairs; tokpos: start byte of each. */
long orc_parse_tables(const uint8_t *in, long n, int level, const uint32_t *t_full,
                      const uint32_t *t_quarter, uint16_t *tokens_out, long *tokpos_out)
{
    if (level == -1) level = 6;
    if (level < 4 || level > 9) return ORC_E_ARG;
    level_cfg cfg = LEVELS[level];
    long ntok = 0, p = 0;
    while (p < n) {
        /* base state at p (no pending match): look the candidate up with prev_length = 2 */
        uint32_t c = t_full[p];
        int len = (int)(c >> 16); long dist = c & 0xffff;
        if (len == MIN_MATCH && dist > TOO_FAR) len = 0;
        if (len < MIN_MATCH) {                            /* literal b[p] */
            if (tokpos_out) tokpos_out[ntok] = p;
            tokens_out[2 * ntok] = 0; tokens_out[2 * ntok + 1] = in[p]; ntok++;
            p++; continue;
        }
        /* lazy evaluation chain */
        for (;;) {
            long q = p + 1;
            int better = 0; long ndist = 0;
            if (q < n && len < cfg.lazy) {
                uint32_t d = len >= cfg.good ? t_quarter[q] : t_full[q];
                int l2 = (int)(d >> 16);
                if (l2 > len) { better = l2; ndist = d & 0xffff; }
            }
            if (!better) break;
            if (tokpos_out) tokpos_out[ntok] = p;
            tokens_out[2 * ntok] = 0; tokens_out[2 * ntok + 1] = in[p]; ntok++;
            p = q; len = better; dist = ndist;
        }
        if (tokpos_out) tokpos_out[ntok] = p;
        tokens_out[2 * ntok] = (uint16_t)dist; tokens_out[2 * ntok + 1] = (uint16_t)(len - MIN_MATCH); ntok++;
        p += len;
    }
    return ntok;
}

/* ------------------------------------------------------------------------------------------- */
/* INFLATE (RFC 1950 wrapper + RFC 1951), semantics of zlib.decompress(buf)  (mtscomp.py:619)    */
/*   Trailing bytes after the adler32 are ignored.  *consumed = bytes of `in` used.              */
/* ------------------------------------------------------------------------------------------- */
typedef struct { const uint8_t *in; long n, pos; uint64_t acc; int nbits; int trunc; } bitr;

static inline void br_need(bitr *r, int k)
{
    while (r->nbits < k) {
        uint64_t b = 0;
        if (r->pos < r->n) b = r->in[r->pos]; else r->trunc = 1;
        r->pos++;
        r->acc |= b << r->nbits; r->nbits += 8;
    }
}
static inline unsigned br_get(bitr *r, int k)
{
    if (k == 0) return 0;
    br_need(r, k);
    unsigned v = (unsigned)(r->acc & ((1ull << k) - 1));
    r->acc >>= k; r->nbits -= k;
    return v;
}

typedef struct { uint16_t count[16]; uint16_t sym[288]; } huff_t;

/* returns 0 ok (complete), 1 incomplete, -1 over-subscribed */
static int huff_build(huff_t *h, const uint8_t *lens, int n)
{
    int offs[16], left = 1;
    for (int i = 0; i < 16; i++) h->count[i] = 0;
    for (int i = 0; i < n; i++) h->count[lens[i]]++;
    if (h->count[0] == n) return 0;                 /* no codes: complete as far as inflate cares */
    for (int len = 1; len < 16; len++) { left <<= 1; left -= h->count[len]; if (left < 0) return -1; }
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h->count[len];
    for (int i = 0; i < n; i++) if (lens[i]) h->sym[offs[lens[i]]++] = (uint16_t)i;
    return left > 0 ? 1 : 0;
}
static int huff_maxlen(const huff_t *h)
{
    for (int len = 15; len >= 1; len--) if (h->count[len]) return len;
    return 0;
}
static int huff_decode(bitr *r, const huff_t *h)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; len++) {
        code |= (int)br_get(r, 1);
        int count = h->count[len];
        if (code - count < first) return h->sym[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}

long orc_inflate(const uint8_t *in, long n, uint8_t *out, long out_cap, long *consumed)
{
    init_tables();
    static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const uint16_t dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    bitr r = {in, n, 0, 0, 0, 0};
    if (n < 2) return ORC_E_TRUNC;
    unsigned cmf = in[0], flg = in[1];
    if (((cmf << 8) | flg) % 31 != 0 || (cmf & 15) != 8 || (cmf >> 4) > 7 || (flg & 0x20)) return ORC_E_DATA;
    r.pos = 2;
    long op = 0;
    int last;
    do {
        last = (int)br_get(&r, 1);
        unsigned type = br_get(&r, 2);
        if (r.trunc) return ORC_E_TRUNC;
        if (type == 0) {
            r.acc = 0; r.nbits = 0;                              /* skip to byte boundary */
            if (r.pos + 4 > n) return ORC_E_TRUNC;
            unsigned len = in[r.pos] | (in[r.pos + 1] << 8), nlen = in[r.pos + 2] | (in[r.pos + 3] << 8);
            r.pos += 4;
            if ((len ^ 0xffff) != nlen) return ORC_E_DATA;
            if (r.pos + (long)len > n) return ORC_E_TRUNC;
            if (op + (long)len > out_cap) return ORC_E_BUF;
            memcpy(out + op, in + r.pos, len); op += len; r.pos += len;
        } else if (type == 1 || type == 2) {
            huff_t hl, hd; uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 288; i++) lens[i] = static_llen[i];
                huff_build(&hl, lens, 288);
                for (int i = 0; i < 30; i++) lens[i] = 5;
                huff_build(&hd, lens, 30);
            } else {
                int nlen = (int)br_get(&r, 5) + 257, ndist = (int)br_get(&r, 5) + 1, ncode = (int)br_get(&r, 4) + 4;
                if (r.trunc) return ORC_E_TRUNC;
                if (nlen > 286 || ndist > 30) return ORC_E_DATA;
                uint8_t cl[19]; memset(cl, 0, sizeof cl);
                for (int i = 0; i < ncode; i++) cl[bl_order[i]] = (uint8_t)br_get(&r, 3);
                if (r.trunc) return ORC_E_TRUNC;
                huff_t hc;
                if (huff_build(&hc, cl, 19) != 0) return ORC_E_DATA;      /* must be complete */
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = huff_decode(&r, &hc);
                    if (r.trunc) return ORC_E_TRUNC;
                    if (sym < 0) return ORC_E_DATA;
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (idx == 0) return ORC_E_DATA; val = lens[idx - 1]; rep = 3 + (int)br_get(&r, 2); }
                        else if (sym == 17) rep = 3 + (int)br_get(&r, 3);
                        else rep = 11 + (int)br_get(&r, 7);
                        if (r.trunc) return ORC_E_TRUNC;
                        if (idx + rep > nlen + ndist) return ORC_E_DATA;
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                if (lens[256] == 0) return ORC_E_DATA;                     /* missing end-of-block */
                /* inftrees.c: an incomplete code is accepted only when its longest code is 1 bit */
                int e1 = huff_build(&hl, lens, nlen);
                if (e1 < 0 || (e1 > 0 && huff_maxlen(&hl) != 1)) return ORC_E_DATA;
                int e2 = huff_build(&hd, lens + nlen, ndist);
                if (e2 < 0 || (e2 > 0 && huff_maxlen(&hd) != 1)) return ORC_E_DATA;
            }
            for (;;) {
                int sym = huff_decode(&r, &hl);
                if (r.trunc) return ORC_E_TRUNC;
                if (sym < 0) return ORC_E_DATA;
                if (sym < 256) { if (op >= out_cap) return ORC_E_BUF; out[op++] = (uint8_t)sym; }
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return ORC_E_DATA;
                    int len = lbase[sym] + (int)br_get(&r, extra_lbits[sym]);
                    int ds = huff_decode(&r, &hd);
                    if (r.trunc) return ORC_E_TRUNC;
                    if (ds < 0 || ds >= 30) return ORC_E_DATA;
                    long dist = dbase[ds] + (long)br_get(&r, extra_dbits[ds]);
                    if (r.trunc) return ORC_E_TRUNC;
                    if (dist > op) return ORC_E_DATA;                      /* too far back */
                    if (op + len > out_cap) return ORC_E_BUF;
                    for (int k = 0; k < len; k++) { out[op] = out[op - dist]; op++; }
                }
            }
        } else return ORC_E_DATA;
    } while (!last);
    /* bytes not yet consumed stay in the bit accumulator: give whole bytes back */
    r.pos -= r.nbits >> 3; r.nbits = 0; r.acc = 0;
    if (r.pos + 4 > n) return ORC_E_TRUNC;
    uint32_t want = ((uint32_t)in[r.pos] << 24) | ((uint32_t)in[r.pos + 1] << 16) | ((uint32_t)in[r.pos + 2] << 8) | in[r.pos + 3];
    r.pos += 4;
    if (want != orc_adler32(out, op)) return ORC_E_DATA;
    if (consumed) *consumed = r.pos;
    return op;
}

/* ------------------------------------------------------------------------------------------- */
/* whole-chunk restatements                                                                      */
/* ------------------------------------------------------------------------------------------- */
/* Writer._compress_chunk (mtscomp.py:375-397): returns compressed length */
long orc_compress_chunk(const uint8_t *raw, long nt, long nc, int sz, int flags, int level,
                        uint8_t *out, long out_cap)
{
    long n = nt * nc * sz;
    uint8_t *stream = (uint8_t *)malloc((size_t)(n ? n : 1));
    if (!stream) return ORC_E_MEM;
    int rc = orc_delta_transpose(raw, nt, nc, sz, flags, stream);
    long r = rc;
    if (rc == ORC_OK) r = orc_deflate(stream, n, level, out, out_cap, 0, 0, 0, 0, 0, 0);
    free(stream);
    return r;
}

/* Reader.read_chunk (mtscomp.py:602-635): 0 ok; <0: corrupt (IOError in the reference);
   1: valid stream of the wrong size (AssertionError in the reference, mtscomp.py:628) */
int orc_decompress_chunk(const uint8_t *cbuf, long clen, long nt, long nc, int sz, int flags, uint8_t *out)
{
    long n = nt * nc * sz;
    uint8_t *stream = (uint8_t *)malloc((size_t)(n + 1));
    if (!stream) return ORC_E_MEM;
    long got = orc_inflate(cbuf, clen, stream, n + 1, 0);
    int rc;
    if (got == ORC_E_BUF) rc = 1;
    else if (got < 0) rc = (int)got;
    else if (got != n) rc = 1;
    else rc = orc_cumsum_transpose(stream, nt, nc, sz, flags, out);
    free(stream);
    return rc;
}
