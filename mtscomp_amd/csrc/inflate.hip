// General RFC 1950/1951 INFLATE on gfx950 -- replaces `zlib.decompress(cbuffer)`
// (/root/reference/mtscomp.py:619).  Accepts any valid stream (stored / fixed / dynamic blocks, any
// encoder), verifies the adler32 trailer, ignores trailing bytes, reports corruption per chunk
// (mtscomp.py:620-621).  orc_inflate() in oracle/mtsc_oracle.c is the oracle.
//
// Two passes per chunk:
//   H  k_inf_decode   bitstream -> token list (literal | (length, distance)); Huffman decoding by
//                     canonical-code compare chains held in registers (no big lookup tables), symbol
//                     tables lane-interleaved in LDS.
//   Z  k_inf_lz       token list -> bytes: one wave per chunk, 64 tokens per step; the 32 KiB history
//                     lives in an LDS ring; copies that only need older bytes run in parallel,
//                     the few that depend on the current step are resolved in lane order.
// followed by the adler32 reduction over the produced stream.
#include "common.h"

namespace mts {

// ------------------------------------------------------------------------------------------------
// bit reader (lane private): 64-bit buffer, one word prefetched
// ------------------------------------------------------------------------------------------------
struct BitIn {
    const u32 *w;      // 4-byte aligned base
    u64 nwords;        // words that may be read
    u64 widx;          // index of `nextw`
    u64 buf;
    u32 cnt;           // valid bits in buf
    u32 nextw;
    u64 pos;           // bits consumed so far (relative to w)
    u64 end;           // first bit past the compressed data
    __device__ __forceinline__ void init(const u8 *base, u64 byte_off, u64 byte_len, u64 start_bit)
    {
        const u64 a = (u64)(base + byte_off);
        w = (const u32 *)(a & ~(u64)3);
        const u64 bit0 = (a & 3) * 8;
        end = bit0 + 8 * byte_len;
        nwords = (end + 31) >> 5;
        seek(bit0 + start_bit);
    }
    __device__ __forceinline__ u32 word(u64 i) const { return i < nwords ? w[i] : 0u; }
    __device__ __forceinline__ void seek(u64 p)
    {
        pos = p;
        const u64 i = p >> 5;
        buf = (u64)word(i) >> (p & 31);
        cnt = 32 - (u32)(p & 31);
        widx = i + 1;
        nextw = word(widx);
        refill();
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            buf |= (u64)nextw << cnt;
            cnt += 32;
            widx++;
            nextw = word(widx);
        }
    }
    __device__ __forceinline__ u32 peek() const { return (u32)buf; }      // >= 32 valid bits after refill()
    __device__ __forceinline__ void skip(u32 n) { buf >>= n; cnt -= n; pos += n; }
    __device__ __forceinline__ u32 get(u32 n)
    {
        refill();
        const u32 v = (u32)buf & (n >= 32 ? 0xffffffffu : ((1u << n) - 1));
        skip(n);
        return v;
    }
};

// ------------------------------------------------------------------------------------------------
// canonical Huffman decode by compare chain.
//   lc[l] (l = 1..MAXL): low 16 bits = lim[l] = exclusive upper bound, left-aligned to MAXL bits, of
//   the code prefixes of length <= l;  high 16 bits = number of codes of length l.
//   v = next MAXL bits, first bit most significant.  Returns the index into the (len, symbol)-sorted
//   symbol table, or -1 for a prefix no code owns.
// ------------------------------------------------------------------------------------------------
template <int MAXL>
__device__ __forceinline__ int chain_decode(u32 v, const u32 (&lc)[MAXL + 1], u32 &len)
{
    u32 l_ = 1, lo = 0, so = 0;
#pragma unroll
    for (int l = 1; l < MAXL; l++) {
        const u32 lim = lc[l] & 0xffff;
        const bool ge = v >= lim;
        l_ += ge;
        lo = ge ? lim : lo;
        so += ge ? (lc[l] >> 16) : 0;
    }
    len = l_;
    if (v >= (lc[MAXL] & 0xffff)) return -1;
    return (int)(so + ((v - lo) >> (MAXL - l_)));
}

// per-lane LDS scratch, lane interleaved (element k of lane L at k*64 + L): 1 KiB per lane
constexpr int INF_LENS = 320;        // u8 code lengths
constexpr int INF_CNT = 32;          // u16: [0..15] counts, [16..31] offsets
constexpr int INF_LSYM = 288;        // u16
constexpr int INF_DSYM = 32;         // u16
constexpr int INF_LDS_PER_WAVE = 64 * (INF_LENS + 2 * (INF_CNT + INF_LSYM + INF_DSYM));

struct LaneLds {
    u8 *lens; u16 *cnt; u16 *lsym; u16 *dsym; int lane;
    __device__ __forceinline__ u8 &len(int k) { return lens[k * 64 + lane]; }
    __device__ __forceinline__ u16 &c(int k) { return cnt[k * 64 + lane]; }
    __device__ __forceinline__ u16 &ls(int k) { return lsym[k * 64 + lane]; }
    __device__ __forceinline__ u16 &ds(int k) { return dsym[k * 64 + lane]; }
};

// Build the compare chain + sorted symbol table for lens[first .. first+nsym).
// returns 0 complete, 1 incomplete, -1 over-subscribed; *maxlen = longest code
template <int MAXL, bool DIST>
__device__ int build_chain(LaneLds &L, int first, int nsym, u32 (&lc)[MAXL + 1], int &maxlen)
{
    for (int l = 0; l < 16; l++) L.c(l) = 0;
    for (int s = 0; s < nsym; s++) L.c(L.len(first + s))++;
    int left = 1, ml = 0;
    u32 off = 0, firstc = 0;
    lc[0] = 0;
#pragma unroll
    for (int l = 1; l <= MAXL; l++) {
        const u32 cn = L.c(l);
        left = (left << 1) - (int)cn;
        if (cn) ml = l;
        L.c(16 + l) = (u16)off;
        off += cn;
        // first code of length l = (first code of length l-1 + count[l-1]) << 1
        const u32 lim = (firstc + cn) << (MAXL - l);
        lc[l] = (lim & 0xffff) | (cn << 16);      // lim <= 2^MAXL <= 32768 unless over-subscribed (rejected below)
        firstc = (firstc + cn) << 1;
    }
    maxlen = ml;
    if (left < 0) return -1;
    for (int s = 0; s < nsym; s++) {
        const int l = L.len(first + s);
        if (l) {
            const int o = L.c(16 + l);
            L.c(16 + l) = (u16)(o + 1);
            if (DIST) L.ds(o) = (u16)s; else L.ls(o) = (u16)s;
        }
    }
    return left > 0 ? 1 : 0;
}

#define INF_OK 0
#define INF_CORRUPT (-1)
#define INF_TOOLONG (-2)

// decodes one deflate block starting at br.pos (just after nothing: reads BFINAL/BTYPE itself).
// EMIT: write tokens to tk[ntok...].  out_base + nout = bytes produced before each token.
// token: literal = byte; match = 1<<31 | (len-3) << 16 | (dist-1)
template <bool EMIT>
__device__ int decode_block(BitIn &br, LaneLds &L, const u8 *cbytes, u32 *tk, u32 &ntok, u64 &nout, u64 out_limit,
                            bool &last)
{
    u32 hdr = br.get(3);
    last = hdr & 1;
    const u32 type = hdr >> 1;
    if (br.pos > br.end) return INF_CORRUPT;
    if (type == 3) return INF_CORRUPT;
    if (type == 0) {
        br.seek((br.pos + 7) & ~7ull);
        const u32 len = br.get(16), nlen = br.get(16);
        if (br.pos > br.end) return INF_CORRUPT;
        if ((len ^ 0xffff) != nlen) return INF_CORRUPT;
        if (br.pos + 8ull * len > br.end) return INF_CORRUPT;
        if (nout + len > out_limit) return INF_TOOLONG;
        // bytes are at absolute bit position br.pos (byte aligned) relative to br.w
        const u8 *src = (const u8 *)br.w + (br.pos >> 3);
        if (EMIT) for (u32 i = 0; i < len; i++) tk[ntok + i] = src[i];
        ntok += len; nout += len;
        br.seek(br.pos + 8ull * len);
        (void)cbytes;
        return INF_OK;
    }
    u32 LC[16], DC[16];
    int nlen_codes, ndist_codes;
    if (type == 1) {
        for (int i = 0; i < 144; i++) L.len(i) = 8;
        for (int i = 144; i < 256; i++) L.len(i) = 9;
        for (int i = 256; i < 280; i++) L.len(i) = 7;
        for (int i = 280; i < 288; i++) L.len(i) = 8;
        for (int i = 0; i < 30; i++) L.len(288 + i) = 5;
        nlen_codes = 288; ndist_codes = 30;
    } else {
        nlen_codes = (int)br.get(5) + 257;
        ndist_codes = (int)br.get(5) + 1;
        const int ncode = (int)br.get(4) + 4;
        if (nlen_codes > 286 || ndist_codes > 30) return INF_CORRUPT;
        const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        // the 19 code-length-code lengths sit at the top of the length slots (dead once CC is built);
        // its sorted symbols go through the distance-symbol slots (19 <= 32 entries)
        constexpr int CL0 = INF_LENS - 19;
        for (int i = 0; i < 19; i++) L.len(CL0 + i) = 0;
        for (int i = 0; i < ncode; i++) L.len(CL0 + order[i]) = (u8)br.get(3);
        if (br.pos > br.end) return INF_CORRUPT;
        u32 CC[8];
        int ml;
        if (build_chain<7, true>(L, CL0, 19, CC, ml) != 0) return INF_CORRUPT;
        int idx = 0;
        const int total = nlen_codes + ndist_codes;
        int prev = 0;
        while (idx < total) {
            br.refill();
            const u32 v = __brev(br.peek()) >> 25;
            u32 cl;
            const int si = chain_decode<7>(v, CC, cl);
            if (si < 0) return INF_CORRUPT;
            const int sym = L.ds(si);
            br.skip(cl);
            if (sym < 16) { L.len(idx) = (u8)sym; prev = sym; idx++; }
            else {
                int rep, val = 0;
                if (sym == 16) { if (idx == 0) return INF_CORRUPT; val = prev; rep = 3 + (int)br.get(2); }
                else if (sym == 17) rep = 3 + (int)br.get(3);
                else rep = 11 + (int)br.get(7);
                if (idx + rep > total) return INF_CORRUPT;
                for (int k = 0; k < rep; k++) L.len(idx + k) = (u8)val;
                idx += rep;
                prev = val;
            }
            if (br.pos > br.end) return INF_CORRUPT;
        }
        if (L.len(256) == 0) return INF_CORRUPT;                 // missing end-of-block code
    }
    int ml;
    // inftrees.c: an incomplete code is accepted only when its longest code has 1 bit
    // (the fixed distance code is incomplete by definition: 30 of 32 five-bit codes)
    int e = build_chain<15, false>(L, 0, nlen_codes, LC, ml);
    if (type == 2 && (e < 0 || (e > 0 && ml != 1))) return INF_CORRUPT;
    e = build_chain<15, true>(L, nlen_codes, ndist_codes, DC, ml);
    if (type == 2 && (e < 0 || (e > 0 && ml > 1))) return INF_CORRUPT;
    for (;;) {
        br.refill();
        u32 cl;
        int si = chain_decode<15>(__brev(br.peek()) >> 17, LC, cl);
        if (si < 0) return INF_CORRUPT;
        u32 sym = L.ls(si);
        br.skip(cl);
        if (sym < 256) {
            if (nout + 1 > out_limit) return INF_TOOLONG;
            if (EMIT) tk[ntok] = sym;
            ntok++; nout++;
        } else if (sym == 256) {
            if (br.pos > br.end) return INF_CORRUPT;
            break;
        } else {
            sym -= 257;
            if (sym >= 29) return INF_CORRUPT;
            u32 eb, lbase;
            if (sym < 8) { eb = 0; lbase = 3 + sym; }
            else if (sym == 28) { eb = 0; lbase = 258; }
            else { eb = (sym - 4) >> 2; lbase = 3 + ((4 + (sym & 3)) << eb); }
            const u32 length = lbase + br.get(eb);
            br.refill();
            si = chain_decode<15>(__brev(br.peek()) >> 17, DC, cl);
            if (si < 0) return INF_CORRUPT;
            const u32 dsym = L.ds(si);
            br.skip(cl);
            if (dsym >= 30) return INF_CORRUPT;
            u32 dbase;
            if (dsym < 4) { eb = 0; dbase = 1 + dsym; }
            else { eb = (dsym - 2) >> 1; dbase = 1 + ((2 + (dsym & 1)) << eb); }
            const u32 dist = dbase + br.get(eb);
            if (br.pos > br.end) return INF_CORRUPT;
            if ((u64)dist > nout) return INF_CORRUPT;                 // too far back
            if (nout + length > out_limit) return INF_TOOLONG;
            if (EMIT) tk[ntok] = 0x80000000u | ((length - 3) << 16) | (dist - 1);
            ntok++; nout += length;
        }
        if (br.pos > br.end) return INF_CORRUPT;
    }
    return INF_OK;
}

// H: one lane per chunk, sequential over the chunk's blocks
__global__ __launch_bounds__(64) void k_inf_decode(const u8 *__restrict__ cdata, const InfChunk *__restrict__ chunks,
                                                   int n_chunks, u32 *__restrict__ tokens, InfResult *__restrict__ res)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int ci = blockIdx.x * 64 + threadIdx.x;
    LaneLds L;
    L.lane = threadIdx.x;
    L.lens = smem;
    L.cnt = (u16 *)(smem + 64 * INF_LENS);
    L.lsym = L.cnt + 64 * INF_CNT;
    L.dsym = L.lsym + 64 * INF_LSYM;
    if (ci >= n_chunks) return;
    const InfChunk ch = chunks[ci];
    InfResult r;
    r.status = MTS_CHUNK_OK; r.n_out = 0; r.ntok = 0; r.adler_stored = 0; r.end_bit = 0;
    const u8 *cb = cdata + ch.c_off;
    bool ok = ch.c_len >= 2;
    if (ok) {
        const u32 cmf = cb[0], flg = cb[1];
        ok = ((cmf << 8) | flg) % 31 == 0 && (cmf & 15) == 8 && (cmf >> 4) <= 7 && !(flg & 0x20);
    }
    if (!ok) { r.status = MTS_CHUNK_CORRUPT; res[ci] = r; return; }
    BitIn br;
    br.init(cdata, ch.c_off, ch.c_len, 16);
    u32 *tk = tokens + ch.tok_off;
    u32 ntok = 0;
    u64 nout = 0;
    bool last = false;
    int st = INF_OK;
    while (!last && st == INF_OK) st = decode_block<true>(br, L, cdata, tk, ntok, nout, (u64)ch.n_expect, last);
    if (st == INF_TOOLONG) {
        // More output than the header promises.  zlib.decompress would carry on: a stream that is valid
        // to its end is a size mismatch (AssertionError, mtscomp.py:628), anything else is corruption
        // (IOError, :621).  Re-walk the stream without emitting to tell the two apart (rare path).
        br.init(cdata, ch.c_off, ch.c_len, 16);
        u32 nt2 = 0; u64 no2 = 0;
        last = false; st = INF_OK;
        while (!last && st == INF_OK) st = decode_block<false>(br, L, cdata, tk, nt2, no2, ~0ull, last);
        const u64 tb = (br.pos + 7) & ~7ull;
        r.status = (st == INF_OK && tb + 32 <= br.end) ? MTS_CHUNK_BADSIZE : MTS_CHUNK_CORRUPT;
    }
    else if (st != INF_OK) r.status = MTS_CHUNK_CORRUPT;
    else {
        // trailer: adler32, big endian, at the next byte boundary
        const u64 tb = (br.pos + 7) & ~7ull;
        if (tb + 32 > br.end) r.status = MTS_CHUNK_CORRUPT;
        else {
            const u8 *t = (const u8 *)br.w + (tb >> 3);
            r.adler_stored = ((u32)t[0] << 24) | ((u32)t[1] << 16) | ((u32)t[2] << 8) | t[3];
            r.end_bit = tb + 32;
            if (nout != ch.n_expect) r.status = MTS_CHUNK_BADSIZE;
        }
    }
    r.n_out = (u32)nout; r.ntok = ntok;
    res[ci] = r;
}

// Z: tokens -> bytes
__global__ __launch_bounds__(64) void k_inf_lz(const u32 *__restrict__ tokens, const InfChunk *__restrict__ chunks,
                                               const InfResult *__restrict__ res, u8 *__restrict__ stream)
{
    const int ci = blockIdx.x;
    const InfResult r = res[ci];
    if (r.status != MTS_CHUNK_OK) return;
    const InfChunk ch = chunks[ci];
    __shared__ __attribute__((aligned(16))) u8 ring[65536];
    const int lane = threadIdx.x;
    const u32 *tk = tokens + ch.tok_off;
    u8 *out = stream + ch.stream_off;
    const u32 ntok = r.ntok;
    u32 base = 0, flushed = 0;
    constexpr u32 M = 65535;
    for (u32 t0 = 0; t0 < ntok; t0 += 64) {
        const bool act = t0 + lane < ntok;
        const u32 t = act ? tk[t0 + lane] : 0;
        const bool cp = act && (t >> 31);
        const u32 len = !act ? 0 : cp ? ((t >> 16) & 0xff) + 3 : 1;
        u32 x = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const u32 y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
        const u32 total = __shfl(x, 63, 64);
        const u32 dst = base + x - len;
        const u32 dist = (t & 0x7fff) + 1;
        const u32 src = dst - dist;
        if (act && !cp) ring[dst & M] = (u8)t;
        const bool indep = cp && (src + (len < dist ? len : dist) <= base);
        if (indep) for (u32 k = 0; k < len; k++) ring[(dst + k) & M] = ring[(src + k) & M];
        __builtin_amdgcn_wave_barrier();
        u64 dep = __ballot(cp && !indep);
        while (dep) {
            const int Ld = __ffsll((long long)dep) - 1;
            if (lane == Ld) for (u32 k = 0; k < len; k++) ring[(dst + k) & M] = ring[(src + k) & M];
            dep &= dep - 1;
            __builtin_amdgcn_wave_barrier();
        }
        base += total;
        while (flushed + 16384 <= base) {
            __builtin_amdgcn_wave_barrier();
            for (u32 i = 0; i < 16; i++) {
                const u32 o = flushed + i * 1024 + lane * 16;
                *(uint4 *)(out + o) = *(const uint4 *)(ring + (o & M));
            }
            flushed += 16384;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (u32 o = flushed + lane; o < base; o += 64) out[o] = ring[o & M];
}

// adler check + final status
__global__ __launch_bounds__(64) void k_inf_finish(const InfChunk *__restrict__ chunks, InfResult *__restrict__ res,
                                                   const u64 *__restrict__ adler_acc, int n_chunks, int *__restrict__ status_out)
{
    const int ci = blockIdx.x * 64 + threadIdx.x;
    if (ci >= n_chunks) return;
    InfResult r = res[ci];
    if (r.status == MTS_CHUNK_OK) {
        const u32 a = (u32)((1 + adler_acc[2 * ci]) % 65521u), b = (u32)((chunks[ci].n_expect + adler_acc[2 * ci + 1]) % 65521u);
        if (((b << 16) | a) != r.adler_stored) r.status = MTS_CHUNK_CORRUPT;
    }
    res[ci].status = r.status;
    status_out[ci] = r.status;
}

size_t inflate_scratch_bytes(int n_chunks, u64 total_cbytes)
{
    (void)total_cbytes;
    return (size_t)n_chunks * 32 + 4096;
}

int launch_inflate(hipStream_t st, const u8 *d_cdata, const InfChunk *d_chunks, int n_chunks, u8 *d_stream,
                   u32 *d_tokens, InfResult *d_res, u64 *d_adler_acc, u32 max_n, int *d_status_out, void *d_scratch,
                   size_t scratch_bytes, void *engine)
{
    if (n_chunks == 0) return MTS_OK;
    static bool attr_set = false;
    if (!attr_set) {
        MTS_HIP(hipFuncSetAttribute((const void *)k_inf_decode, hipFuncAttributeMaxDynamicSharedMemorySize, INF_LDS_PER_WAVE));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_inf_decode, dim3((n_chunks + 63) / 64), dim3(64), INF_LDS_PER_WAVE, st, d_cdata, d_chunks, n_chunks,
                       d_tokens, d_res);
    inflate_mark(engine, st, "inflate_huffman");
    hipLaunchKernelGGL(k_inf_lz, dim3(n_chunks), dim3(64), 0, st, d_tokens, d_chunks, d_res, d_stream);
    MTS_HIP(hipGetLastError());
    inflate_mark(engine, st, "inflate_lz");
    // adler32 of the produced streams: per-chunk offset/length tables live in d_scratch
    u64 *d_so = (u64 *)d_scratch;
    u32 *d_nn = (u32 *)(d_so + n_chunks);
    (void)scratch_bytes;
    int rc = launch_adler_stream(st, d_stream, d_so, d_nn, n_chunks, max_n, d_adler_acc);
    if (rc) return rc;
    hipLaunchKernelGGL(k_inf_finish, dim3((n_chunks + 63) / 64), dim3(64), 0, st, d_chunks, d_res, d_adler_acc, n_chunks,
                       d_status_out);
    MTS_HIP(hipGetLastError());
    inflate_mark(engine, st, "adler32");
    return MTS_OK;
}

}  // namespace mts
