// General RFC 1950/1951 INFLATE on gfx950 -- replaces `zlib.decompress(cbuffer)`
// (/root/reference/mtscomp.py:619).  Accepts any valid stream (stored / fixed / dynamic blocks, any
// encoder), verifies the adler32 trailer, ignores trailing bytes, reports corruption per chunk
// (mtscomp.py:620-621).  orc_inflate() in oracle/mtsc_oracle.c is the oracle.
//
// Fast path (dynamic-Huffman streams, i.e. what zlib writes for this kind of data):
//   k_inf_scan/validate  every bit offset is tested for a well-formed dynamic block header -> candidate block starts
//   k_inf_passA          wave per candidate: 64 lanes decode 64 consecutive sub-sequences, re-synchronise, count tokens
//   k_inf_chain          wave per chunk: accept candidates only where the previous block really ends
//   k_inf_passB          wave per accepted block: decode again from the recorded starts, write tokens
//   k_inf_wave           one workgroup per chunk: finishes whatever the fast path declined (same sub-sequence decode, block after block)
//   k_inf_lz(_seg) ...   tokens -> bytes through an LDS ring with per-byte dataflow (see "Z" below); a chunk is cut
//                        into segments resolved on symbolic 16-bit cells when there are fewer chunks than CUs
// followed by the adler32 reduction over the produced stream.
#include <stdlib.h>

#include <vector>

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
    const u32 *lw = nullptr;   // optional LDS copy of words [lw0, lw0 + lwn) (staged by the caller with coalesced loads)
    u64 lw0 = 0;
    u32 lwn = 0;
    __device__ __forceinline__ void init(const u8 *base, u64 byte_off, u64 byte_len, u64 start_bit)
    {
        const u64 a = (u64)(base + byte_off);
        w = (const u32 *)(a & ~(u64)3);
        const u64 bit0 = (a & 3) * 8;
        end = bit0 + 8 * byte_len;
        nwords = (end + 31) >> 5;
        seek(bit0 + start_bit);
    }
    __device__ __forceinline__ u32 word(u64 i) const
    {
        const u64 o = i - lw0;
        if (o < lwn) return lw[o];
        return i < nwords ? w[i] : 0u;
    }
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

// words [wb0, wb0 + count) of a stream into LDS (zero past the data), by `nthr` threads of which this is thread `t`.  The loads
// are unconditional and eight per thread are in flight: written as  dst[k] = br.word(wb0 + k)  every load sat under the
// bounds check's branch and was waited for where the branch ends -- a round trip per word and thread (a third of pass A's time
// per round went there).
__device__ __forceinline__ void stage_stream_words(u32 *dst, const BitIn &br, u64 wb0, u32 count, u32 t, u32 nthr)
{
    const __attribute__((address_space(1))) u32 *g = (const __attribute__((address_space(1))) u32 *)(u64)br.w;
    const u64 nw = br.nwords;
    if (nw == 0) { for (u32 k = t; k < count; k += nthr) dst[k] = 0; return; }
    for (u32 k0 = 0; k0 < count; k0 += 8 * nthr) {
        u32 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { const u64 i = wb0 + k0 + (u32)j * nthr + t; v[j] = g[i < nw ? i : nw - 1]; }
#pragma unroll
        for (int j = 0; j < 8; j++) { const u32 k = k0 + (u32)j * nthr + t; if (k < count) dst[k] = wb0 + k < nw ? v[j] : 0u; }
    }
}

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

// the same with the chain in memory (LDS, wave-shared: every lane reads the same word)
template <int MAXL>
__device__ __forceinline__ int chain_decode_mem(u32 v, const u32 *lc, u32 &len)
{
    u32 l_ = 1, lo = 0, so = 0;
#pragma unroll
    for (int l = 1; l < MAXL; l++) {
        const u32 w = lc[l], lim = w & 0xffff;
        const bool ge = v >= lim;
        l_ += ge;
        lo = ge ? lim : lo;
        so += ge ? (w >> 16) : 0;
    }
    len = l_;
    if (v >= (lc[MAXL] & 0xffff)) return -1;
    return (int)(so + ((v - lo) >> (MAXL - l_)));
}

// the same for a prefix whose code is known to be longer than FROM bits (a lookup table indexed by FROM bits had no entry for
// it): the first FROM steps of the chain are known to pass, and what they add up to -- the number of codes of up to FROM
// bits -- is in lc[0] (build_chain puts it there)
template <int MAXL, int FROM>
__device__ __forceinline__ int chain_decode_mem_from(u32 v, const u32 *lc, u32 &len)
{
    u32 l_ = FROM + 1, lo = lc[FROM] & 0xffff, so = lc[0];
#pragma unroll
    for (int l = FROM + 1; l < MAXL; l++) {
        const u32 w = lc[l], lim = w & 0xffff;
        const bool ge = v >= lim;
        l_ += ge;
        lo = ge ? lim : lo;
        so += ge ? (w >> 16) : 0;
    }
    len = l_;
    if (v >= (lc[MAXL] & 0xffff)) return -1;
    return (int)(so + ((v - lo) >> (MAXL - l_)));
}

// Huffman scratch in LDS.  Element k of a user sits at k*stride + lane: stride 64 = one private table set
// per lane of a wave (64 independent streams), stride 1 / lane 0 = one table set shared by a wave.
constexpr int INF_LENS = 320;        // u8 code lengths
constexpr int INF_CNT = 32;          // u16: [0..15] counts, [16..31] offsets
constexpr int INF_LSYM = 288;        // u16
constexpr int INF_DSYM = 32;         // u16
constexpr int INF_SET_BYTES = INF_LENS + 2 * (INF_CNT + INF_LSYM + INF_DSYM);      // 1024

struct LaneLds {
    u8 *lens; u16 *cnt; u16 *lsym; u16 *dsym; int lane; int stride;
    __device__ __forceinline__ void bind(u8 *base, int stride_, int lane_)
    {
        stride = stride_; lane = lane_;
        lens = base;
        cnt = (u16 *)(base + stride_ * INF_LENS);
        lsym = cnt + stride_ * INF_CNT;
        dsym = lsym + stride_ * INF_LSYM;
    }
    __device__ __forceinline__ u8 &len(int k) { return lens[k * stride + lane]; }
    __device__ __forceinline__ u16 &c(int k) { return cnt[k * stride + lane]; }
    __device__ __forceinline__ u16 &ls(int k) { return lsym[k * stride + lane]; }
    __device__ __forceinline__ u16 &ds(int k) { return dsym[k * stride + lane]; }
};

#ifndef MTS_LUT_LBITS
#define MTS_LUT_LBITS 8
#endif
constexpr int CHAIN_LUT_LBITS = MTS_LUT_LBITS, CHAIN_LUT_DBITS = 8;      // (lit/len index bits, pass A on the recordings: 11: 4.50 ms, 10: 3.80, 9: 3.46, 8: 3.22, 7: 3.45 --
                                                                          //  the table's LDS sets how many waves a CU holds; longer codes walk the chain's tail)        // index bits of the lookup tables (= LUT_LBITS, LUT_DBITS below)
// Build the compare chain + sorted symbol table for lens[first .. first+nsym).
// returns 0 complete, 1 incomplete, -1 over-subscribed; maxlen = longest code
template <int MAXL, bool DIST, class CHAIN>
__device__ int build_chain(LaneLds &L, int first, int nsym, CHAIN &&lc, int &maxlen)
{
    for (int l = 0; l < 16; l++) L.c(l) = 0;
    for (int s = 0; s < nsym; s++) L.c(L.len(first + s))++;
    int left = 1, ml = 0;
    u32 off = 0, firstc = 0, short_codes = 0;
#pragma unroll
    for (int l = 1; l <= MAXL; l++) {
        const u32 cn = L.c(l);
        left = (left << 1) - (int)cn;
        if (cn) ml = l;
        L.c(16 + l) = (u16)off;
        off += cn;
        if (l <= (DIST ? CHAIN_LUT_DBITS : CHAIN_LUT_LBITS)) short_codes += cn;
        // first code of length l = (first code of length l-1 + count[l-1]) << 1
        const u32 lim = (firstc + cn) << (MAXL - l);
        lc[l] = (lim & 0xffff) | (cn << 16);      // lim <= 2^MAXL <= 32768 unless over-subscribed (rejected below)
        firstc = (firstc + cn) << 1;
    }
    lc[0] = short_codes;         // (slot 0 is no code length: chain_decode_mem_from starts from it)
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

// Code tables of a fixed (type 1) or dynamic (type 2) block; br stands right after the 3 header bits.
template <class CHAIN>
__device__ int parse_tables(BitIn &br, LaneLds &L, u32 type, CHAIN &&LC, CHAIN &&DC)
{
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
    return INF_OK;
}

// ------------------------------------------------------------------------------------------------
// The same for a table set that a whole wave shares (pass A: stride 1, every lane used to run parse_tables() redundantly --
// 316 code lengths decoded one after the other by 64 lanes in step, and three counting / placing loops over the symbols:
// 0.40 ms of pass A's 3.03).  Here the wave works together:
//   * the code-length symbol that WOULD start at every bit position of the header (64 positions a step, 512 at a time, on
//     demand) goes to LDS as symbol | code length << 5 | extra-bit value << 8;
//   * the walk from the first position hops over that table (one LDS read and a dozen instructions per symbol instead of
//     a refill, a 7-step compare chain and a table look-up) and notes runs (first index, count, value);
//   * the runs are filled in by the lanes, and the two code tables are counted and put in order with ballots (a symbol's
//     place = start of its length + symbols of that length before it: the same order as the serial loops give).
// Same verdicts in the same order as parse_tables().  `scratch`: PARSE_WAVE_WORDS words of LDS; br must read through an LDS
// copy (br.lw) that covers the header.
// ------------------------------------------------------------------------------------------------
constexpr int PARSE_WAVE_POS = 2560 + 64;                     // a dynamic header is < 2560 bits long (316 lengths of <= 7 bits + 74)
constexpr int PARSE_WAVE_WORDS = PARSE_WAVE_POS / 2 + 320;   // the position table (u16), the runs

template <int MAXL, bool DIST, class CHAIN>
__device__ int build_chain_wave(LaneLds &L, int first, int nsym, CHAIN &&lc, int &maxlen, int lane)
{
    constexpr int R = DIST ? 1 : 5;                            // rounds of 64 symbols (<= 32 distance / code-length, <= 288 literal/length symbols)
    const u64 lt = ((u64)1 << lane) - 1;
    u32 mylen[R], cn[MAXL + 1];
#pragma unroll
    for (int l = 0; l <= MAXL; l++) cn[l] = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int s_ = r * 64 + lane;
        mylen[r] = s_ < nsym ? (u32)L.len(first + s_) : 0u;
#pragma unroll
        for (int l = 1; l <= MAXL; l++) cn[l] += (u32)__popcll(__ballot(mylen[r] == (u32)l));
    }
    int left = 1, ml = 0;
    u32 off = 0, firstc = 0, short_codes = 0, base[MAXL + 1];
#pragma unroll
    for (int l = 1; l <= MAXL; l++) {
        left = (left << 1) - (int)cn[l];
        if (cn[l]) ml = l;
        base[l] = off;
        off += cn[l];
        if (l <= (DIST ? CHAIN_LUT_DBITS : CHAIN_LUT_LBITS)) short_codes += cn[l];
        const u32 lim = (firstc + cn[l]) << (MAXL - l);
        lc[l] = (lim & 0xffff) | (cn[l] << 16);
        firstc = (firstc + cn[l]) << 1;
    }
    lc[0] = short_codes;
    maxlen = ml;
    if (left < 0) return -1;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int s_ = r * 64 + lane;
#pragma unroll
        for (int l = 1; l <= MAXL; l++) {
            const u64 m = __ballot(mylen[r] == (u32)l);
            if (mylen[r] == (u32)l) {
                const u32 o = base[l] + (u32)__popcll(m & lt);
                if (DIST) L.ds(o) = (u16)s_; else L.ls(o) = (u16)s_;
            }
            base[l] += (u32)__popcll(m);
        }
    }
    return left > 0 ? 1 : 0;
}

template <class CHAIN>
__device__ int parse_tables_wave(BitIn &br, LaneLds &L, u32 type, CHAIN &&LC, CHAIN &&DC, u32 *scratch, int lane)
{
    if (type != 2 || br.lw == nullptr) return parse_tables(br, L, type, LC, DC);
    const int nlen_codes = (int)br.get(5) + 257;
    const int ndist_codes = (int)br.get(5) + 1;
    const int ncode = (int)br.get(4) + 4;
    if (nlen_codes > 286 || ndist_codes > 30) return INF_CORRUPT;
    const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    constexpr int CL0 = INF_LENS - 19;
    for (int i = 0; i < 19; i++) L.len(CL0 + i) = 0;
    for (int i = 0; i < ncode; i++) L.len(CL0 + order[i]) = (u8)br.get(3);
    if (br.pos > br.end) return INF_CORRUPT;
    u32 CC[8];
    int ml;
    if (build_chain<7, true>(L, CL0, 19, CC, ml) != 0) return INF_CORRUPT;
    u16 *dec = (u16 *)scratch;
    u32 *runs = scratch + PARSE_WAVE_POS / 2;
    const u64 p0 = br.pos;
    const int total = nlen_codes + ndist_codes;
    u32 limit = 0, p = 0;
    int idx = 0, prev = 0, nruns = 0;
    while (idx < total) {
        if (limit + 512 > (u32)PARSE_WAVE_POS) return INF_CORRUPT;          // (cannot happen: see PARSE_WAVE_POS)
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int r = 0; r < 8; r++) {
            const u32 q = limit + (u32)r * 64 + (u32)lane;
            const u64 ap = p0 + q, wi = (ap >> 5) - br.lw0;
            u32 bits = 0;
            if (wi + 1 < br.lwn) bits = __builtin_amdgcn_alignbit(br.lw[wi + 1], br.lw[wi], (u32)(ap & 31));
            u32 cl;
            const int si = chain_decode<7>(__brev(bits) >> 25, CC, cl);     // (the code is complete: every 7 bits are a symbol)
            const u32 sym = L.ds(si < 0 ? 0 : si);
            const u32 xb = sym < 16 ? 0u : sym == 16 ? 2u : sym == 17 ? 3u : 7u;
            dec[q] = (u16)(sym | (cl << 5) | (((bits >> cl) & ((1u << xb) - 1u)) << 8));
        }
        limit += 512;
        __builtin_amdgcn_wave_barrier();
        while (idx < total && p < limit) {
            const u32 d = dec[p];
            const u32 sym = d & 31, x = d >> 8;
            u32 adv = (d >> 5) & 7;
            int rep = 1, val = (int)sym;
            if (sym >= 16) {
                if (sym == 16) { if (idx == 0) return INF_CORRUPT; val = prev; rep = 3 + (int)x; adv += 2; }
                else if (sym == 17) { val = 0; rep = 3 + (int)x; adv += 3; }
                else { val = 0; rep = 11 + (int)x; adv += 7; }
                if (idx + rep > total) return INF_CORRUPT;
            }
            runs[nruns++] = (u32)idx | ((u32)rep << 9) | ((u32)val << 17);
            idx += rep; prev = val; p += adv;
            if (p0 + p > br.end) return INF_CORRUPT;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < nruns; r += 64) {
        const u32 e = runs[r];
        const int i0 = (int)(e & 511), rep = (int)((e >> 9) & 255), val = (int)(e >> 17);
        for (int k = 0; k < rep; k++) L.len(i0 + k) = (u8)val;
    }
    __builtin_amdgcn_wave_barrier();
    br.seek(p0 + p);
    if (L.len(256) == 0) return INF_CORRUPT;                     // missing end-of-block code
    int e = build_chain_wave<15, false>(L, 0, nlen_codes, LC, ml, lane);
    if (e < 0 || (e > 0 && ml != 1)) return INF_CORRUPT;
    e = build_chain_wave<15, true>(L, nlen_codes, ndist_codes, DC, ml, lane);
    if (e < 0 || (e > 0 && ml > 1)) return INF_CORRUPT;
    return INF_OK;
}

constexpr u32 LZ_PIECE = 8;          // longest copy handed to one lane of the LZ resolver
constexpr u32 LZ_WINDOW_BYTES = 32768; // the deflate window (= LZ_WIN of the segmented resolver)
__device__ __forceinline__ u32 lz_pieces(u32 tok, u32 olen) { return (tok >> 31) ? (olen + LZ_PIECE - 1) / LZ_PIECE : 1; }
__device__ __forceinline__ void lz_emit_pieces(u32 *tk, u32 tok, u32 olen)
{
    if (!(tok >> 31) || olen <= LZ_PIECE) { tk[0] = tok; return; }
    const u32 dist = (tok & 0x7fff) + 1;
    u32 k = 0;
    if (dist >= olen) {
        // the copy does not overlap itself: every piece reads at the copy's own distance.  (Kept apart from the general
        // case below because that one divides by the distance, and in a wave the set-up of the division ran for every
        // token longer than a piece -- three quarters of all steps have one.)
        const u32 base = 0x80000000u | (dist - 1);
        if (olen <= 2 * LZ_PIECE) {
            const u32 l0 = olen - LZ_PIECE < 3 ? olen - 3 : LZ_PIECE;
            tk[0] = base | ((l0 - 3) << 16);
            tk[1] = base | ((olen - l0 - 3) << 16);
            return;
        }
        u32 left = olen;
        while (left > 0) {
            u32 l = left > LZ_PIECE ? LZ_PIECE : left;
            if (left > LZ_PIECE && left - LZ_PIECE < 3) l = left - 3;
            tk[k++] = base | ((l - 3) << 16);
            left -= l;
        }
        return;
    }
    // pieces of LZ_PIECE bytes; a short remainder (< 3 bytes is not encodable as len-3 >= 0) is merged
    // into the last piece by making the last two pieces share the remainder.
    // A copy that overlaps itself (dist < length) repeats a pattern of `dist` bytes, so byte o of it equals the byte any
    // multiple of dist further back, as long as that is not before the pattern: the piece at offset o reads at distance
    // dist * (o / dist + 1), i.e. from [start - dist, start + 8) -- the bytes before the copy and its first piece -- instead
    // of from the piece before it.  A run (dist 1, 33 pieces per 258 bytes) is then 2 dependent steps deep, not 33.
    u32 left = olen, o = 0, next_mult = dist;                            // next_mult = dist * (o / dist + 1)
    while (left > 0) {
        u32 l = left > LZ_PIECE ? LZ_PIECE : left;
        if (left > LZ_PIECE && left - LZ_PIECE < 3) l = left - 3;          // leave >= 3 for the last piece
        if (next_mult <= o) {
            // (never further back than the 32 KiB a resolver segment knows of what precedes it: its pieces of a copy that
            //  began in the segment before must find their source inside that window)
            const u32 m = o / dist + 1, mmax = LZ_WINDOW_BYTES / dist;
            next_mult = dist * (m < mmax ? m : mmax);
        }
        tk[k++] = 0x80000000u | ((l - 3) << 16) | (next_mult - 1);
        left -= l; o += l;
    }
}

// One token.  Returns 0 literal/match decoded, 1 end of block, <0 error.
// token: literal = byte; match = 1<<31 | (len-3) << 16 | (dist-1);  *olen = bytes it produces
__device__ __forceinline__ int decode_token(BitIn &br, LaneLds &L, const u32 (&LC)[16], const u32 (&DC)[16], u32 &tok, u32 &olen)
{
    br.refill();
    u32 cl;
    int si = chain_decode<15>(__brev(br.peek()) >> 17, LC, cl);
    if (si < 0) return INF_CORRUPT;
    u32 sym = L.ls(si);
    br.skip(cl);
    if (sym < 256) { tok = sym; olen = 1; return 0; }
    if (sym == 256) return 1;
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
    tok = 0x80000000u | ((length - 3) << 16) | (dist - 1);
    olen = length;
    return 0;
}

// Wave-shared lookup tables for the kernels in which all 64 lanes decode the SAME block: the next LUT_LBITS
// (LUT_DBITS) stream bits index a 32-bit entry that already holds what the symbol means:
//   [3:0] code length (0 = the code is longer than the index, or invalid: those go through the compare chain)
//   [5:4] 0 literal, 1 length / distance, 2 end of block, 3 invalid symbol
//   [23:8] literal byte, or base length / base distance      [27:24] number of extra bits
// The tables are filled by decoding every index with the compare chain once.
constexpr int LUT_LBITS = CHAIN_LUT_LBITS, LUT_DBITS = CHAIN_LUT_DBITS;
constexpr int LUT_BYTES = 4 * ((1 << LUT_LBITS) + (1 << LUT_DBITS)) + 2 * 16 * 4;      // tables + the two compare chains
__device__ __forceinline__ u32 lut_len_entry(u32 sym, u32 cl)
{
    if (sym < 256) return (sym << 8) | cl;
    if (sym == 256) return (2u << 4) | cl;
    sym -= 257;
    if (sym >= 29) return (3u << 4) | cl;
    u32 eb, lbase;
    if (sym < 8) { eb = 0; lbase = 3 + sym; }
    else if (sym == 28) { eb = 0; lbase = 258; }
    else { eb = (sym - 4) >> 2; lbase = 3 + ((4 + (sym & 3)) << eb); }
    return (eb << 24) | (lbase << 8) | (1u << 4) | cl;
}
__device__ __forceinline__ u32 lut_dist_entry(u32 dsym, u32 cl)
{
    if (dsym >= 30) return (3u << 4) | cl;
    u32 eb, dbase;
    if (dsym < 4) { eb = 0; dbase = 1 + dsym; }
    else { eb = (dsym - 2) >> 1; dbase = 1 + ((2 + (dsym & 1)) << eb); }
    return (eb << 24) | (dbase << 8) | (1u << 4) | cl;
}
// lutl: lit/len table, then the distance table, then the chains LC[16], DC[16] (already there: parse_tables wrote them)
__device__ __forceinline__ void build_luts(LaneLds &L, u32 *lutl, u32 *lutd, int lane)
{
    const u32 *LC = lutd + (1 << LUT_DBITS), *DC = LC + 16;
    for (u32 e = lane; e < (1u << LUT_LBITS); e += 64) {
        u32 cl;
        const int si = chain_decode_mem<15>(__brev(e) >> 17, LC, cl);
        lutl[e] = (si >= 0 && cl <= (u32)LUT_LBITS) ? lut_len_entry(L.ls(si), cl) : 0u;
    }
    for (u32 e = lane; e < (1u << LUT_DBITS); e += 64) {
        u32 cl;
        const int si = chain_decode_mem<15>(__brev(e) >> 17, DC, cl);
        lutd[e] = (si >= 0 && cl <= (u32)LUT_DBITS) ? lut_dist_entry(L.ds(si), cl) : 0u;
    }
}
// One token through the tables; READER = BitIn or BitL.  Returns 0 literal/match decoded, 1 end of block, <0 error.
// Written without a literal / match branch: in a wave both kinds are always present, and a divergent wave pays for
// both sides one after the other.  A literal lane simply looks at a distance entry it does not use and skips 0 bits.
template <class READER>
__device__ __forceinline__ int decode_token_lut(READER &br, LaneLds &L, const u32 *lutl, const u32 *lutd, u32 &tok, u32 &olen)
{
    const u32 *LC = lutd + (1 << LUT_DBITS), *DC = LC + 16;
    br.refill();
    u32 p = br.peek();
    u32 e = lutl[p & ((1u << LUT_LBITS) - 1)];
    if (!(e & 15)) {                                             // (rare) longer than the table index
        u32 cl;
        const int si = chain_decode_mem_from<15, LUT_LBITS>(__brev(p) >> 17, LC, cl);
        if (si < 0) return INF_CORRUPT;
        e = lut_len_entry(L.ls(si), cl);
    }
    const u32 type = (e >> 4) & 3, cl = e & 15, eb = e >> 24, val = (e >> 8) & 0xffff;
    const bool is_match = type == 1;
    const u32 length = val + ((p >> cl) & ((1u << eb) - 1));      // code + extra bits: <= 15 + 5 of the >= 32 valid bits (eb = 0 unless a length)
    br.skip(cl + eb);
    br.refill();
    p = br.peek();
    u32 d = lutd[p & ((1u << LUT_DBITS) - 1)];
    if (is_match && !(d & 15)) {
        u32 dl;
        const int si = chain_decode_mem_from<15, LUT_DBITS>(__brev(p) >> 17, DC, dl);
        if (si < 0) return INF_CORRUPT;
        d = lut_dist_entry(L.ds(si), dl);
    }
    const u32 dcl = d & 15, deb = d >> 24;
    const u32 dist = ((d >> 8) & 0xffff) + ((p >> dcl) & ((1u << deb) - 1));      // <= 15 + 13 bits
    br.skip(is_match ? dcl + deb : 0u);
    tok = is_match ? 0x80000000u | ((length - 3) << 16) | (dist - 1) : val & 0xff;
    olen = is_match ? length : 1u;
    if (type >= 2) return type == 2 ? 1 : INF_CORRUPT;
    if (is_match && ((d >> 4) & 3) == 3) return INF_CORRUPT;
    return 0;
}

// Lean bit reader for the wave-shared kernels: the words come from an LDS copy of the piece of stream the wave
// works on (staged with coalesced loads, zero padded past the data), positions are 32-bit and relative to the first
// staged word.  No range checks: the caller stages enough slack for the two words the reader runs ahead.
struct BitL {
    const u32 *w;
    u32 widx;          // index of `nextw`
    u64 buf;
    u32 cnt;           // valid bits in buf
    u32 nextw;
    u32 pos;           // bits consumed (relative to word 0)
    __device__ __forceinline__ void seek(u32 p)
    {
        pos = p;
        const u32 i = p >> 5;
        buf = (u64)w[i] >> (p & 31);
        cnt = 32 - (p & 31);
        widx = i + 1;
        nextw = w[widx];
        refill();
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            buf |= (u64)nextw << cnt;
            cnt += 32;
            widx++;
            nextw = w[widx];
        }
    }
    __device__ __forceinline__ u32 peek() const { return (u32)buf; }
    __device__ __forceinline__ void skip(u32 n) { buf >>= n; cnt -= n; pos += n; }
    __device__ __forceinline__ u32 get(u32 n)      // n <= 13
    {
        refill();
        const u32 v = (u32)buf & ((1u << n) - 1);
        skip(n);
        return v;
    }
};
// Window reader (round 6, pass A): no state but the position.  Every token reads the 64 bits at its position from the LDS copy
// (three words, two funnel shifts) instead of carrying a 64-bit buffer, its fill count and a prefetched word through the loop:
// the buffered reader spends two 64-bit shifts (slow-class opcodes, profiles/r6_valu_issue_table.txt), two refills under
// branches and their selects per token -- pass A is bound by vector issue (NOTEBOOK.md, round 6), not by the LDS.
struct BitW {
    const u32 *w;
    u32 pos;           // bits consumed (relative to word 0)
    u32 p, p2;         // the bits [pos, pos + 32) and [pos + 32, pos + 64) of the token being decoded
    __device__ __forceinline__ void seek(u32 q) { pos = q; }
    __device__ __forceinline__ void load()
    {
        const u32 i = pos >> 5, sh = pos & 31;
        const u32 a = w[i], b = w[i + 1], c = w[i + 2];
        p = __builtin_amdgcn_alignbit(b, a, sh);
        p2 = __builtin_amdgcn_alignbit(c, b, sh);
    }
};
// decode_token_lut on the window reader: the same tables, verdicts and token words
__device__ __forceinline__ int decode_token_w(BitW &br, LaneLds &L, const u32 *lutl, const u32 *lutd, u32 &tok, u32 &olen)
{
    const u32 *LC = lutd + (1 << LUT_DBITS), *DC = LC + 16;
    br.load();
    const u32 p = br.p;
    u32 e = lutl[p & ((1u << LUT_LBITS) - 1)];
    if (!(e & 15)) {                                             // (rare) longer than the table index
        u32 cl;
        const int si = chain_decode_mem_from<15, LUT_LBITS>(__brev(p) >> 17, LC, cl);
        if (si < 0) return INF_CORRUPT;
        e = lut_len_entry(L.ls(si), cl);
    }
    const u32 type = (e >> 4) & 3, cl = e & 15, eb = e >> 24, val = (e >> 8) & 0xffff;
    const bool is_match = type == 1;
    const u32 length = val + __builtin_amdgcn_ubfe(p, cl, eb);    // code + extra bits: <= 15 + 5 of the 32
    const u32 n1 = cl + eb;
    const u32 q = __builtin_amdgcn_alignbit(br.p2, p, n1);        // the 32 bits behind them (n1 <= 20)
    u32 d = lutd[q & ((1u << LUT_DBITS) - 1)];
    if (is_match && !(d & 15)) {
        u32 dl;
        const int si = chain_decode_mem_from<15, LUT_DBITS>(__brev(q) >> 17, DC, dl);
        if (si < 0) return INF_CORRUPT;
        d = lut_dist_entry(L.ds(si), dl);
    }
    const u32 dcl = d & 15, deb = d >> 24;
    const u32 dist = ((d >> 8) & 0xffff) + __builtin_amdgcn_ubfe(q, dcl, deb);      // <= 15 + 13 bits
    br.pos += n1 + (is_match ? dcl + deb : 0u);
    tok = is_match ? 0x80000000u | ((length - 3) << 16) | (dist - 1) : val & 0xff;
    olen = is_match ? length : 1u;
    if (type >= 2) return type == 2 ? 1 : INF_CORRUPT;
    if (is_match && ((d >> 4) & 3) == 3) return INF_CORRUPT;
    return 0;
}

// One whole deflate block, sequentially by one lane, starting at br.pos (reads BFINAL/BTYPE itself).
// EMIT: write tokens to tk[ntok...].  nout = bytes produced so far in the stream (distance check).
template <bool EMIT>
__device__ int decode_block(BitIn &br, LaneLds &L, u32 *tk, u32 &ntok, u64 &nout, u64 out_limit, bool &last)
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
        const u8 *src = (const u8 *)br.w + (br.pos >> 3);        // byte aligned
        if (EMIT) for (u32 i = 0; i < len; i++) tk[ntok + i] = src[i];
        ntok += len; nout += len;
        br.seek(br.pos + 8ull * len);
        return INF_OK;
    }
    u32 LC[16], DC[16];
    const int rc = parse_tables(br, L, type, LC, DC);
    if (rc != INF_OK) return rc;
    for (;;) {
        u32 tok, olen;
        const int t = decode_token(br, L, LC, DC, tok, olen);
        if (t < 0) return INF_CORRUPT;
        if (br.pos > br.end) return INF_CORRUPT;
        if (t == 1) break;
        if ((tok >> 31) && (u64)((tok & 0x7fff) + 1) > nout) return INF_CORRUPT;       // too far back
        if (nout + olen > out_limit) return INF_TOOLONG;
        if (EMIT) lz_emit_pieces(tk + ntok, tok, olen);
        ntok += lz_pieces(tok, olen); nout += olen;
    }
    return INF_OK;
}

// ================================================================================================
// Fast path, step 1: find every bit offset that carries a valid dynamic-block header
// ================================================================================================
// A DEFLATE stream has no sync points, so block starts are not known without decoding.  They can be
// guessed, though: test EVERY bit offset for a well-formed dynamic header (BTYPE=10, HLIT/HDIST in range,
// a complete code-length code, then lit/len + distance code lengths that decode to complete codes with
// an end-of-block symbol).  Random bits pass the cheap part ~0.08 % of the time and the full check
// almost never; true block starts always pass.  Candidates are only hints: the chain walk below
// accepts a block only if the previous block really ends there.
__device__ __forceinline__ u32 wave_excl_scan(u32 v, u32 &total)
{
    const int lane = threadIdx.x & 63;
    u32 x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
    total = x;                     // inclusive value: lane 63 holds the wave total
    return x - v;
}

__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 x);
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_SPAN_BITS = SCAN_THREADS * 32;
constexpr int SCAN_SURV_CAP = 512;
constexpr int SCAN_TAIL = 88;            // words staged past the span: a dynamic header is < 2560 bits long

// Full validation of the dynamic header at absolute bit `o` (BFINAL bit), as a state machine: vh_setup reads the counts
// and the code-length code, vh_step decodes one code-length symbol (0 = go on, 1 = well formed, -1 = not a header).  A lane
// whose candidate is decided takes the next one, so the few real headers (316 symbols) do not hold 63 lanes that were
// done after ~50.
struct VhState {
    BitIn br;
    // the 7-bit code-length code as a compare chain in bytes: lim_l = first code (left-justified to 7 bits) that is longer than l
    // bits, 128 where there is none.  lima / limb: lim_1..4 / lim_5, lim_6, 128, 128; lo64: byte l = lim_l (byte 0 = 0);
    // so64: byte l = number of codes of up to l bits
    u32 lima, limb;
    u64 lo64, so64;
    u64 t0, t1;                           // sorted symbol k at bits 5*(k%12) of t[k/12]
    int nlen, total, idx, prev, len256, maxl, maxd;
    u32 kl, kd;                           // Kraft sums scaled by 2^15
};

__device__ bool vh_setup(VhState &s, const u32 *w, u64 nwords, u64 end, u64 o, const u32 *lw, u32 lwn)
{
    BitIn &br = s.br;
    br.w = w; br.nwords = nwords; br.end = end;
    br.lw = lw; br.lw0 = o >> 5; br.lwn = lwn;            // the first words of the header, staged by the caller
    br.seek(o + 3);
    const int nlen = (int)br.get(5) + 257, ndist = (int)br.get(5) + 1, ncode = (int)br.get(4) + 4;
    if (nlen > 286 || ndist > 30) return false;
    // code-length code: lengths packed 3 bits each by symbol
    const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    u64 cl = 0;
    for (int i = 0; i < ncode; i++) cl |= (u64)br.get(3) << (3 * order[i]);
    // compare chain of the 7-bit code + its sorted symbols packed 5 bits each (two u64).  Counts per length live in one
    // u64, 5 bits each (at most 19 codes), so nothing is indexed by a run-time value.
    u64 cntp = 0;
#pragma unroll
    for (int k = 0; k < 19; k++) cntp += 1ull << (5 * ((u32)(cl >> (3 * k)) & 7));
    u32 firstc = 0, off = 0;
    u64 offp = 0;                         // first sorted slot of each length, 5 bits each
    int left = 1;
    u64 lim64 = 0, so64 = 0;
#pragma unroll
    for (int l = 1; l <= 7; l++) {
        const u32 c = (u32)(cntp >> (5 * l)) & 31;
        left = (left << 1) - (int)c;
        offp |= (u64)off << (5 * l); off += c;
        if (l <= 6) {                                            // (a complete code: lim_7 = 128, never reached by a 7-bit value)
            lim64 |= (u64)min((firstc + c) << (7 - l), 128u) << (8 * l);
            so64 |= (u64)off << (8 * l);
        }
        firstc = (firstc + c) << 1;
    }
    if (left != 0) return false;
    s.lo64 = lim64; s.so64 = so64;
    s.lima = (u32)(lim64 >> 8);
    s.limb = (u32)(lim64 >> 40) | 0x80800000u;
    u64 t0 = 0, t1 = 0;
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const u32 l = (u32)(cl >> (3 * k)) & 7;
        if (l) {
            const u32 o2 = (u32)(offp >> (5 * l)) & 31;
            offp += 1ull << (5 * l);
            if (o2 < 12) t0 |= (u64)k << (5 * o2); else t1 |= (u64)k << (5 * (o2 - 12));
        }
    }
    s.t0 = t0; s.t1 = t1;
    s.nlen = nlen; s.total = nlen + ndist;
    s.idx = 0; s.prev = 0; s.len256 = 0; s.maxl = 0; s.maxd = 0; s.kl = 0; s.kd = 0;
    return true;
}

__device__ __forceinline__ int vh_step(VhState &s)
{
    BitIn &br = s.br;
    br.refill();
    const u32 pk = br.peek();                                    // >= 32 valid bits: a code (<= 7) and a repeat's extra bits (<= 7)
    // how many of the six limits the next 7 bits reach, all at once: byte k of (v + 128) - lim is 128 or more where v >= lim_k
    const u32 v7 = __brev(pk) >> 25, vb = __builtin_amdgcn_perm(v7, v7, 0u) | 0x80808080u;     // (v7 in every byte)
    const u32 nge = (u32)__popc((vb - s.lima) & 0x80808080u) + (u32)__popc((vb - s.limb) & 0x80808080u);
    const u32 clen = 1 + nge;
    const int si = (int)(((u32)(s.so64 >> (8 * nge)) & 0xffu) + ((v7 - ((u32)(s.lo64 >> (8 * nge)) & 0xffu)) >> (6 - nge)));
    const int sym = (int)((si < 12 ? s.t0 >> (5 * si) : s.t1 >> (5 * (si - 12))) & 31);
    // The repeat codes 16, 17, 18 without a branch each (the lanes of a wave are on different symbols, so every branch was
    // everybody's): extra bits 2 / 3 / 7 and base 3 / 3 / 11 by t = 1, 2, 3 out of two constants, taken from the same 32 bits
    const u32 t4 = 4u * (u32)(max(sym, 15) - 15);                // 0 for a length, 4 / 8 / 12 for 16 / 17 / 18
    const u32 xb = (0x7320u >> t4) & 15u;
    const int rep = (int)(((0xb331u >> t4) & 15u) + __builtin_amdgcn_ubfe(pk, clen, xb));
    br.skip(clen + xb);
    const int idx = s.idx, nlen = s.nlen;
    const int val = sym < 16 ? sym : sym == 16 ? s.prev : 0;
    if (sym >= 16 && ((sym == 16 && idx == 0) || idx + rep > s.total)) return -1;
    s.prev = val;
    if (val) {
        const int nl = idx >= nlen ? 0 : (idx + rep <= nlen ? rep : nlen - idx);
        const int nd = rep - nl;
        s.kl += (u32)nl << (15 - val);                           // nl * 2^(15 - val) (as a shift; the stage's time is the same either way: 3.13 = 3.15 ms)
        s.kd += (u32)nd << (15 - val);
        if (nl && val > s.maxl) s.maxl = val;
        if (nd && val > s.maxd) s.maxd = val;
        if (idx <= 256 && 256 < idx + rep) s.len256 = val;
        if (s.kl > 32768u || s.kd > 32768u) return -1;          // over-subscribed already: random bits get here after ~50 symbols
    }
    s.idx = idx + rep;
    if (br.pos > br.end) return -1;
    if (s.idx < s.total) return 0;
    if (s.len256 == 0) return -1;
    if (s.kl > 32768u || (s.kl < 32768u && s.maxl != 1)) return -1;
    if (s.kd > 32768u || (s.kd < 32768u && s.maxd > 1)) return -1;
    return 1;
}

struct InfFast {                 // per-chunk bookkeeping of the fast path (device arrays, one entry per chunk)
    u64 cand_off;                // first slot of this chunk in the candidate arrays
    u32 cand_cap;
    u32 true_off;                // first slot in the true-block array
    u32 true_cap;
    u32 pad;
};
struct CandRes {                 // result of decoding one candidate block (pass A)
    u64 end_bit;                 // position after the end-of-block symbol
    u32 ntok, nout, nsub;
    u32 ok;                      // 1 = decoded to a clean end of block
    u32 bfinal;
    u32 rows;                    // 1 = pass A kept the block's tokens (rows of ROWCAP per sub-sequence): pass B only moves them
};
struct TrueBlk {
    u64 start_bit;
    u32 cand;                    // candidate slot, or 0xffffffff = decode sequentially (stored/fixed/unseen block)
    u32 tok_off;                 // chunk relative
    u32 ntok;
    u32 chunk;
};
constexpr int PASSA_STAGE_WORDS = 64 * 1024 / 32 + 16;   // a round of pass A: 64 sub-sequences + the reader's look-ahead and a token's overshoot
#ifndef MTS_PASSB_STAGE_WORDS
#define MTS_PASSB_STAGE_WORDS 2304
#endif
constexpr int PASSB_STAGE_WORDS = MTS_PASSB_STAGE_WORDS;   // LDS copy of the 64 sub-sequences a wave decodes in one step: 64 x (1024 bits + a token's overshoot) <= 2144 words
                                          // (a step that needs more reads the stream from memory); 3072 words left room for 8 waves per CU, 2304 for 10
constexpr int SUBCAP = 512;      // sub-sequences recorded per candidate block
constexpr u32 SUB_BITS = 1024;   // bits per sub-sequence, at most (measured 1024 .. 3072: shorter is faster, the lanes' serial chains dominate)
#ifndef MTS_SUB_TARGET
#define MTS_SUB_TARGET 1024
#endif
constexpr u32 SUB_TARGET = MTS_SUB_TARGET;      // pass A cuts a block into rounds of 64 EQUAL sub-sequences of about this many bits (see k_inf_passA)
constexpr u32 SUB_MIN = 256;
// Pass A's counting decode keeps the tokens it sees: ROWCAP per sub-sequence (a row), 64 rows per round, taken from a pool
// with one atomic per round.  A block whose rows all fit (a sub-sequence of <= 1024 bits holds ~100 tokens of ~10 bits on the
// recordings) is not decoded a third time: pass B moves its rows to where the block's tokens belong.  Anything else -- a
// row too short for its sub-sequence (runs: 33 pieces per 258-byte copy), the pool used up -- goes through pass B's decode.
#ifndef MTS_INF_ROWCAP
#define MTS_INF_ROWCAP 192
#endif
constexpr u32 ROWCAP = MTS_INF_ROWCAP;
constexpr int ROUNDS_MAX = (SUBCAP + 63) / 64;          // rounds of a block (all but the last have 64 sub-sequences)
static inline u64 rows_rounds_of(u64 c_len) { return c_len * 8 / (64 * 640) + 32; }      // pool share of a chunk, in rounds

#ifndef MTS_SCAN_SPANS
#define MTS_SCAN_SPANS 1
#endif
#ifndef MTS_PA_LEAN
#define MTS_PA_LEAN 1      // pass A on the window reader (BitW): 2.82 -> 2.51 ms for the 60-chunk recording, same tokens (round 6)
#endif
constexpr int SCAN_SPANS = MTS_SCAN_SPANS;   // spans per workgroup (2, 4, 8: the stage takes 4.2 ms instead of 3.25 -- the Kraft table made once per
                                             // workgroup does not pay for the fewer, longer workgroups)
constexpr u64 SCAN_FINAL_ZONE_BITS = 8ull * 256 * 1024;      // candidates with BFINAL = 1 are looked at in the last 256 KiB of a stream only
constexpr int SCAN_L1_CAP = 14336;       // filter-1 survivors kept per workgroup (expected ~7200 of 32768)

__global__ __launch_bounds__(SCAN_THREADS) void k_inf_scan(const u8 *__restrict__ cdata, const InfChunk *__restrict__ chunks,
                                                           const InfFast *__restrict__ fast, u64 *__restrict__ cand_pos,
                                                           u32 *__restrict__ cand_cnt, u64 *__restrict__ surv_list,
                                                           u32 *__restrict__ surv_cnt, u32 surv_cap, u64 final_zone_bits)
{
    const int ci = blockIdx.y;
    const InfChunk ch = chunks[ci];
    if (ch.c_len < 8) return;
    const u64 a = (u64)(cdata + ch.c_off);
    const u32 *w = (const u32 *)(a & ~(u64)3);
    const u64 bit0 = (a & 3) * 8, end = bit0 + 8 * ch.c_len, nwords = (end + 31) >> 5;
    if ((u64)blockIdx.x * SCAN_SPANS * SCAN_SPAN_BITS >= end) return;
    __shared__ u32 sw[SCAN_THREADS + SCAN_TAIL];
    __shared__ u16 l1[SCAN_L1_CAP];
    __shared__ u32 surv[SCAN_SURV_CAP];
    __shared__ u32 wsum[16];
    __shared__ u32 nsurv;
    __shared__ u8 ktab[4096];                                      // Kraft sum (in 1/128) of four 3-bit code lengths, saturated at 255 (anything over 128 is
                                                                   // no code: a byte per entry is indexed by the field itself, two bytes needed a shift more per look-up)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 4096; i += SCAN_THREADS) {
        u32 k = 0;
        for (int f = 0; f < 4; f++) k += (0x80u >> ((i >> (3 * f)) & 7)) & 0x7f;
        ktab[i] = (u8)(k < 255 ? k : 255);
    }
    __shared__ u32 gbase_s;
    // a workgroup takes SCAN_SPANS consecutive spans (the Kraft table above is made once for them)
    for (int sp = 0; sp < SCAN_SPANS; sp++) {
    const u64 span0 = ((u64)blockIdx.x * SCAN_SPANS + sp) * SCAN_SPAN_BITS;        // absolute bit of the span
    if (span0 >= end) break;
    const u64 word0 = span0 >> 5;
    __syncthreads();                                             // (everybody is done with the span before)
    for (int i = tid; i < SCAN_THREADS + SCAN_TAIL; i += SCAN_THREADS) sw[i] = word0 + i < nwords ? w[word0 + i] : 0;
    if (tid == 0) nsurv = 0;
    __syncthreads();
    // phase A1: the 3 header fields at each of this thread's 32 offsets, all 32 at once (bit k of S(j) is
    // stream bit o_k + j):  BTYPE == 2 <=> bit1 = 0, bit2 = 1;  HLIT <= 29 <=> not (bits 4..7 all set);
    // HDIST <= 29 <=> not (bits 9..12 all set)
    const u32 w0 = sw[tid], w1 = sw[tid + 1];
#define S_(j) __builtin_amdgcn_alignbit(w1, w0, j)
    u32 m = ~S_(1) & S_(2);
    m &= ~(S_(4) & S_(5) & S_(6) & S_(7));
    m &= ~(S_(9) & S_(10) & S_(11) & S_(12));
#undef S_
    // BFINAL: only the LAST block of a stream has it set, and that block starts within a block's length of the end (16383 symbols
    // of at most 48 bits at zlib's default memLevel; twice that at memLevel 9).  Away from the end a candidate with BFINAL = 1 is
    // never a block the chain walk can accept: dropping it here halves the survivors of this filter, i.e. the Kraft sums below and
    // the full validations after them.  A final block longer than SCAN_FINAL_ZONE
    // is still decoded -- unannounced, by the block-after-block decoder the chain hands the rest of the chunk to.
    if (span0 + SCAN_SPAN_BITS + final_zone_bits <= end) m &= ~w0;
    // keep only offsets inside the stream
    {
        const u64 o0 = span0 + (u64)tid * 32;
        const u64 first_ok = bit0 + 16;
        if (o0 + 32 <= first_ok || o0 + 100 > end) {
            u32 keep = 0;
            for (int k = 0; k < 32; k++) if (o0 + k >= first_ok && o0 + k + 100 <= end) keep |= 1u << k;
            m &= keep;
        } else if (o0 < first_ok) m &= ~0u << (u32)(first_ok - o0);
    }
    // compact the survivors of the workgroup
    const u32 cntm = __popc(m);
    const u32 incl = wave_incl_scan_dpp(cntm), ex = incl - cntm;
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 base = 0, n1 = 0;
    for (int q = 0; q < 16; q++) { if (q < wave) base += wsum[q]; n1 += wsum[q]; }
    n1 = n1 < (u32)SCAN_L1_CAP ? n1 : (u32)SCAN_L1_CAP;
    {
        u32 slot = base + ex, mm = m;
        while (mm) {
            const int k = __ffs(mm) - 1;
            mm &= mm - 1;
            if (slot < (u32)SCAN_L1_CAP) l1[slot] = (u16)(tid * 32 + k);
            slot++;
        }
    }
    __syncthreads();
    // phase A2: Kraft sum of the code-length code (3-bit lengths from bit 17), all lanes busy
    for (u32 it = tid; it < n1; it += SCAN_THREADS) {
        const u32 rel = l1[it];
        const u32 wi = rel >> 5, k = rel & 31;
        const u32 x0 = sw[wi], x1 = sw[wi + 1], x2 = sw[wi + 2], x3 = sw[wi + 3];
        const u32 f0 = __builtin_amdgcn_alignbit(x1, x0, k), f1 = __builtin_amdgcn_alignbit(x2, x1, k),
                  f2 = __builtin_amdgcn_alignbit(x3, x2, k);
        const u32 ncode = ((f0 >> 13) & 15) + 4;
        // bits 17.. of the stream at this offset: g0 = fields 0..9 (+2 bits of field 10), g1 = rest
        const u32 g0 = __builtin_amdgcn_alignbit(f1, f0, 17), g1 = __builtin_amdgcn_alignbit(f2, f1, 17);
        // fields at and after ncode do not count; the rest is summed four fields (12 bits) per table look-up
        const u32 nb = 3 * ncode;                                 // 12 .. 57 bits
        const u32 h0 = nb >= 32 ? g0 : g0 & ((1u << nb) - 1u);
        const u32 h1 = nb > 32 ? g1 & ((1u << (nb - 32)) - 1u) : 0u;
        const u32 kraft = ktab[h0 & 0xfff] + ktab[(h0 >> 12) & 0xfff] + ktab[((h0 >> 24) | (h1 << 8)) & 0xfff] +
                          ktab[(h1 >> 4) & 0xfff] + ktab[(h1 >> 16) & 0xfff];
        if (kraft == 128) {
            const u32 slot = atomicAdd(&nsurv, 1u);
            if (slot < SCAN_SURV_CAP) surv[slot] = rel;
        }
    }
    __syncthreads();
    // the survivors go to a global list; their full validation is a separate, fully occupied launch
    // (it is a long serial decode per survivor: latency bound, so it wants many waves in flight)
    const u32 ns = min(nsurv, (u32)SCAN_SURV_CAP);
    if (tid == 0) gbase_s = ns ? atomicAdd(surv_cnt, ns) : 0;
    __syncthreads();
    if ((u32)tid < ns && gbase_s + tid < surv_cap) surv_list[gbase_s + tid] = ((u64)ci << 40) | (span0 + surv[tid]);
    }
    (void)fast; (void)cand_pos; (void)cand_cnt;
}

constexpr int VAL_STAGE = 16;            // words of a candidate header staged in LDS
constexpr int VAL_STEPS = 8;             // code-length symbols between two looks at the idle lanes
constexpr u32 VAL_REFILL = 16;           // idle lanes that make a refill worth its setup (global loads + the code-length code)
#ifndef MTS_VAL_WAVES
#define MTS_VAL_WAVES 4    // (5, 6, 8 waves per SIMD by spilling registers: 3.41, 3.91, 5.02 ms for the stage against 3.24)
#endif
__global__ __launch_bounds__(256, MTS_VAL_WAVES) void k_inf_validate(const u8 *__restrict__ cdata, const InfChunk *__restrict__ chunks,
                                                      const InfFast *__restrict__ fast, const u64 *__restrict__ surv_list,
                                                      const u32 *__restrict__ surv_cnt, u32 surv_cap,
                                                      u64 *__restrict__ cand_pos, u32 *__restrict__ cand_cnt)
{
    const u32 n = min(surv_cnt[0], surv_cap);
    const int lane = threadIdx.x & 63;
    const u64 lt = ((u64)1 << lane) - 1;
    // every wave owns a contiguous slice of the survivor list (a shared counter would be one hot atomic)
    const u32 n_waves = gridDim.x * 4, wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    const u32 per = (n + n_waves - 1) / n_waves;
    u32 next = min(n, wave_id * per);
    const u32 stop = min(n, next + per);
    __shared__ u32 stage[4][64][VAL_STAGE + 1];                    // (odd row pitch: lanes read their own rows)
    VhState s;
    bool active = false;
    u32 ci = 0;
    u64 o = 0;
    for (;;) {
        const u64 idle = __ballot(!active);
        if ((u32)__popcll(idle) >= VAL_REFILL) {
            const u32 cnt = (u32)__popcll(idle);
            const u32 base = next;
            next = min(stop, next + cnt);
            if (!active) {
                const u32 my = base + (u32)__popcll(idle & lt);
                if (my < stop) {
                    const u64 v = surv_list[my];
                    ci = (u32)(v >> 40);
                    o = v & ((1ull << 40) - 1);
                    const InfChunk ch = chunks[ci];
                    const u64 a = (u64)(cdata + ch.c_off);
                    const u64 bit0 = (a & 3) * 8, end = bit0 + 8 * ch.c_len;
                    // the first VAL_STAGE words from the header on go to this lane's LDS row with four loads in flight: a
                    // candidate that is not a header is decided inside them (~44 symbols of ~7 bits), and the bit reader
                    // would otherwise fetch them one dependent load after the other
                    const u32 *w = (const u32 *)(a & ~(u64)3);
                    const u64 nwords = (end + 31) >> 5, i0 = o >> 5;
                    u32 *row = &stage[threadIdx.x >> 6][lane][0];
                    if (i0 + VAL_STAGE <= nwords) {
                        typedef u32 w4 __attribute__((ext_vector_type(4), aligned(4)));
                        const w4 q0 = *(const w4 *)(w + i0), q1 = *(const w4 *)(w + i0 + 4), q2 = *(const w4 *)(w + i0 + 8), q3 = *(const w4 *)(w + i0 + 12);
                        row[0] = q0.x; row[1] = q0.y; row[2] = q0.z; row[3] = q0.w; row[4] = q1.x; row[5] = q1.y; row[6] = q1.z; row[7] = q1.w;
                        row[8] = q2.x; row[9] = q2.y; row[10] = q2.z; row[11] = q2.w; row[12] = q3.x; row[13] = q3.y; row[14] = q3.z; row[15] = q3.w;
                    } else {
                        for (int k = 0; k < VAL_STAGE; k++) row[k] = i0 + k < nwords ? w[i0 + k] : 0u;
                    }
                    active = vh_setup(s, w, nwords, end, o, row, VAL_STAGE);
                }
            }
            if (!__any(active)) {
                if (base + cnt >= stop) break;
                continue;
            }
        }
        for (int k = 0; k < VAL_STEPS; k++) {
            if (active) {
                const int r = vh_step(s);
                if (r) {
                    active = false;
                    if (r > 0) {
                        const InfFast f = fast[ci];
                        const u32 slot = atomicAdd(&cand_cnt[ci], 1u);
                        if (slot < f.cand_cap) cand_pos[f.cand_off + slot] = o;
                    }
                }
            }
        }
    }
}

// sort each chunk's candidates by position (rank sort; the lists are short)
__global__ __launch_bounds__(256) void k_inf_sortc(const InfFast *__restrict__ fast, u64 *__restrict__ cand_pos,
                                                   u64 *__restrict__ cand_tmp, u32 *__restrict__ cand_cnt)
{
    const int ci = blockIdx.x;
    const InfFast f = fast[ci];
    const u32 n = min(cand_cnt[ci], f.cand_cap);
    u64 *p = cand_pos + f.cand_off, *t = cand_tmp + f.cand_off;
    for (u32 i = threadIdx.x; i < n; i += 256) t[i] = p[i];
    __syncthreads();
    for (u32 i = threadIdx.x; i < n; i += 256) {
        const u64 v = t[i];
        u32 r = 0;
        for (u32 j = 0; j < n; j++) r += t[j] < v;
        p[r] = v;                                   // positions are distinct
    }
}

// ================================================================================================
// Fast path, step 2: wave-per-block self-synchronising Huffman decode
// ================================================================================================
// The 64 lanes of a wave decode 64 consecutive SUB_BITS-bit sub-sequences of one block at once.  Only
// lane 0 starts on a token boundary; the others start mid-token, decode garbage for a while and -- because
// Huffman streams re-synchronise -- end on true token boundaries.  Each lane then restarts from the exit
// of its left neighbour; this repeats until every lane's start equals its neighbour's exit, i.e. until
// the 64 decodes chain exactly like a sequential decode.  Pass A (this kernel) records, per sub-sequence,
// the start bit and the running token count; pass B re-decodes from those starts and writes the tokens.
enum { SPAN_CONT = 0, SPAN_EOB = 1, SPAN_ERR = 2 };

template <int MODE>      // 0: find the exit only, 1: count, 2: emit `want` tokens to tk
__device__ __forceinline__ void decode_span(BitIn &br, LaneLds &L, const u32 *lutl, const u32 *lutd, u64 stop, u32 &ntok, u32 &nout,
                                            int &flag, u32 *tk, u32 want)
{
    flag = SPAN_CONT;
    ntok = 0; nout = 0;
    for (;;) {
        if (MODE == 2) { if (ntok >= want) break; }
        else if (br.pos >= stop) break;
        u32 tok, olen;
        const int t = decode_token_lut(br, L, lutl, lutd, tok, olen);
        if (t < 0 || br.pos > br.end) { flag = SPAN_ERR; break; }
        if (t == 1) { flag = SPAN_EOB; break; }
        // a copy longer than LZ_PIECE bytes is written as several copies with the same distance (byte k of
        // a copy reads dst - dist whatever piece it is in), so no lane of the resolver owns a long run
        const u32 np = lz_pieces(tok, olen);
        if (MODE == 2) lz_emit_pieces(tk + ntok, tok, olen);
        ntok += np; nout += olen;
    }
}

// the same on the lean reader; `end` = first bit past the compressed data (relative, like the positions)
template <int MODE>
__device__ __forceinline__ void decode_span_fast(BitL &br, LaneLds &L, const u32 *lutl, const u32 *lutd, u32 stop, u32 end, u32 &ntok,
                                                 u32 &nout, int &flag, u32 *tk, u32 want)
{
    flag = SPAN_CONT;
    ntok = 0; nout = 0;
    for (;;) {
        if (MODE == 2) { if (ntok >= want) break; }
        else if (br.pos >= stop) break;
        u32 tok, olen;
        const int t = decode_token_lut(br, L, lutl, lutd, tok, olen);
        if (t < 0 || br.pos > end) { flag = SPAN_ERR; break; }
        if (t == 1) { flag = SPAN_EOB; break; }
        const u32 np = lz_pieces(tok, olen);
        if (MODE == 2) lz_emit_pieces(tk + ntok, tok, olen);
        ntok += np; nout += olen;
    }
}

// counting decode that also keeps the tokens while they fit into `cap` (pass A's rows)
__device__ __forceinline__ void decode_span_rows(BitL &br, LaneLds &L, const u32 *lutl, const u32 *lutd, u32 stop, u32 end, u32 &ntok,
                                                 u32 &nout, int &flag, u32 *row, u32 cap, bool &ovf)
{
    flag = SPAN_CONT;
    ntok = 0; nout = 0; ovf = false;
    for (;;) {
        if (br.pos >= stop) break;
        u32 tok, olen;
        const int t = decode_token_lut(br, L, lutl, lutd, tok, olen);
        if (t < 0 || br.pos > end) { flag = SPAN_ERR; break; }
        if (t == 1) { flag = SPAN_EOB; break; }
        const u32 np = lz_pieces(tok, olen);
        if (ntok + np <= cap) lz_emit_pieces(row + ntok, tok, olen);
        else ovf = true;
        ntok += np; nout += olen;
    }
}

// the two span decoders pass A uses, on the window reader
__device__ __forceinline__ void decode_span_exit_w(BitW &br, LaneLds &L, const u32 *lutl, const u32 *lutd, u32 stop, u32 end, int &flag)
{
    flag = SPAN_CONT;
    for (;;) {
        if (br.pos >= stop) break;
        u32 tok, olen;
        const int t = decode_token_w(br, L, lutl, lutd, tok, olen);
        if (t < 0 || br.pos > end) { flag = SPAN_ERR; break; }
        if (t == 1) { flag = SPAN_EOB; break; }
    }
}
__device__ __forceinline__ void decode_span_rows_w(BitW &br, LaneLds &L, const u32 *lutl, const u32 *lutd, u32 stop, u32 end, u32 &ntok,
                                                   u32 &nout, int &flag, u32 *row, u32 cap, bool &ovf)
{
    flag = SPAN_CONT;
    ntok = 0; nout = 0; ovf = false;
    for (;;) {
        if (br.pos >= stop) break;
        u32 tok, olen;
        const int t = decode_token_w(br, L, lutl, lutd, tok, olen);
        if (t < 0 || br.pos > end) { flag = SPAN_ERR; break; }
        if (t == 1) { flag = SPAN_EOB; break; }
        const u32 np = lz_pieces(tok, olen);
        if (ntok + np <= cap) lz_emit_pieces(row + ntok, tok, olen);
        else ovf = true;
        ntok += np; nout += olen;
    }
}

// position of the first lane (>= 1 bit set) in a ballot, or 64
__device__ __forceinline__ int first_lane(u64 m) { return m ? __ffsll((long long)m) - 1 : 64; }

__global__ __launch_bounds__(64) void k_inf_passA(const u8 *__restrict__ cdata, const InfChunk *__restrict__ chunks,
                                                  const InfFast *__restrict__ fast, const u32 *__restrict__ slot_chunk,
                                                  const u64 *__restrict__ cand_pos, const u32 *__restrict__ cand_cnt,
                                                  CandRes *__restrict__ cres, uint2 *__restrict__ subs, u32 *__restrict__ rows,
                                                  u32 *__restrict__ row_ctr, u32 rounds_cap, u32 *__restrict__ round_row)
{
    const u32 slot = blockIdx.x;
    const u32 ci = slot_chunk[slot];
    const InfFast f = fast[ci];
    const u32 k = slot - (u32)f.cand_off;
    if (k >= min(cand_cnt[ci], f.cand_cap)) return;
    const InfChunk ch = chunks[ci];
    __shared__ __attribute__((aligned(16))) u8 tabs[INF_SET_BYTES];
    LaneLds L;
    L.bind(tabs, 1, 0);
    const int lane = threadIdx.x;
    BitIn br;
    br.init(cdata, ch.c_off, ch.c_len, 0);
    const u64 o = cand_pos[slot];
    // the header (< 2560 bits) is parsed out of an LDS copy: read from memory it was a dependent load per 32 bits
    __shared__ u32 stage[PASSA_STAGE_WORDS];
    stage_stream_words(stage, br, o >> 5, SCAN_TAIL + 8, (u32)threadIdx.x, 64);
    __builtin_amdgcn_wave_barrier();
    br.lw = stage; br.lw0 = o >> 5; br.lwn = SCAN_TAIL + 8;
    br.seek(o);
    CandRes r;
    r.end_bit = 0; r.ntok = 0; r.nout = 0; r.nsub = 0; r.ok = 0; r.bfinal = 0; r.rows = 0;
    const u32 hdr = br.get(3);
    r.bfinal = hdr & 1;
    __shared__ u32 lut_s[LUT_BYTES / 4];
    u32 *lutl = lut_s, *lutd = lut_s + (1 << LUT_LBITS);
    u32 *LC = lutd + (1 << LUT_DBITS), *DC = LC + 16;               // the compare chains live in LDS (wave-shared)
    // the header is parsed by the wave together (parse_tables_wave; its scratch: the staged words behind the header's)
    static_assert(SCAN_TAIL + 8 + PARSE_WAVE_WORDS <= PASSA_STAGE_WORDS, "the header's words and the parse's scratch fit the stage");
    const int rc = parse_tables_wave(br, L, hdr >> 1, LC, DC, stage + SCAN_TAIL + 8, lane);
    if (rc != INF_OK) { if (lane == 0) cres[slot] = r; return; }
    __builtin_amdgcn_wave_barrier();
    build_luts(L, lutl, lutd, lane);
    __builtin_amdgcn_wave_barrier();
    uint2 *sub = subs + (u64)slot * SUBCAP;
    // Sub-sequence length.  A block of B bits takes ceil(B / (64 * SUB_BITS)) rounds whatever is left for the last one (the
    // ~26 KB blocks of zlib's level 6 on the recordings: 3.25 rounds' worth, i.e. a fourth round with 16 of 64 lanes busy).
    // The candidates of a chunk are sorted and the next one is -- almost always -- where this block ends, so the length is
    // known in advance and the rounds are made equal: the same number of rounds, each as short as the block allows.  A
    // wrong guess (a validated header inside a block) can only make the sub-sequences shorter than they need be; if the
    // block then has more of them than the records hold, it is done again with full-length ones.
    u32 sub_bits = SUB_BITS;
    {
        const u32 ncand_c = min(cand_cnt[ci], f.cand_cap);
        const u64 next_o = k + 1 < ncand_c ? cand_pos[slot + 1] : 8ull * ch.c_len;
        const u64 est = next_o > br.pos ? next_o - br.pos : 0;
        const u64 rounds = (est + 64ull * SUB_TARGET - 1) / (64ull * SUB_TARGET);
        if (rounds >= 1 && rounds * 64 + 64 <= (u64)SUBCAP - 1) {
            const u64 per = (est + 64 * rounds - 1) / (64 * rounds);
            sub_bits = (u32)((per + 31) & ~31ull);
            sub_bits = sub_bits < SUB_MIN ? SUB_MIN : sub_bits > SUB_BITS ? SUB_BITS : sub_bits;
        }
    }
    const u64 base0 = br.pos;
    u64 base = base0;
    u32 tot_tok = 0, tot_out = 0, nsub = 0;
    bool done = false, fail = false, rows_ok = true;
    // a round = 64 sub-sequences: that piece of the stream (+ slack for the reader's look-ahead and the last token's
    // overshoot) is staged in LDS; positions inside a round are relative to its first staged word
    // (reading the stream straight from memory instead -- no LDS copy, 20 waves per CU -- was measured at 6.55 ms against 3.22)
    br.lw = nullptr; br.lwn = 0;                                     // (the header's copy makes way for the rounds')
#if MTS_PA_LEAN
    BitW bl;
#else
    BitL bl;
#endif
    bl.w = stage;
    while (!done && !fail) {
        const u64 wb0 = base >> 5;
        const u32 bofs = (u32)(base & 31);
        __builtin_amdgcn_wave_barrier();
        const u32 stage_words = 64 * sub_bits / 32 + 16;
        stage_stream_words(stage, br, wb0, stage_words, (u32)lane, 64);
        __builtin_amdgcn_wave_barrier();
        const u64 end_rel64 = br.end - (wb0 << 5);
        const u32 end_rel = end_rel64 < 0x7fffffffull ? (u32)end_rel64 : 0x7fffffffu;
        const u32 stop = bofs + (u32)(lane + 1) * sub_bits;
        u32 start = bofs + (u32)lane * sub_bits, ex;
        u32 nt, no; int fl;
        // the round's rows
        u32 rb = 0;
        if (lane == 0) rb = atomicAdd(row_ctr, 1u);
        rb = (u32)__builtin_amdgcn_readfirstlane((int)rb);
        const bool have_rows = rb < rounds_cap;
        u32 *row = rows + ((u64)(have_rows ? rb : 0) * 64 + lane) * ROWCAP;
        bool ovf = false;
        // speculative pass: exits only
        bl.seek(start);
#if MTS_PA_LEAN
        decode_span_exit_w(bl, L, lutl, lutd, stop, end_rel, fl);
#else
        decode_span_fast<0>(bl, L, lutl, lutd, stop, end_rel, nt, no, fl, nullptr, 0);
#endif
        ex = bl.pos;
        bool counted = false;
        for (int it = 0; it < 66; it++) {
            // true start of lane i = exit of lane i-1; lanes after the first EOB/ERR lane are void
            const u32 pex = __shfl_up(ex, 1, 64);
            const u32 want_start = lane == 0 ? bofs : pex;
            const u64 stopm = __ballot(fl != SPAN_CONT);
            const int fstop = first_lane(stopm);              // lanes > fstop are void
            const bool valid = lane <= fstop;
            const bool redo = valid && (!counted || want_start != start);
            // (a lane that is void now may become valid later only if an earlier lane changes: handled
            //  because `counted` stays false for lanes that never ran the counting pass)
            if (!__any(redo)) break;
            if (redo) {
                start = want_start;
                bl.seek(start);
#if MTS_PA_LEAN
                decode_span_rows_w(bl, L, lutl, lutd, stop, end_rel, nt, no, fl, row, have_rows ? ROWCAP : 0u, ovf);
#else
                decode_span_rows(bl, L, lutl, lutd, stop, end_rel, nt, no, fl, row, have_rows ? ROWCAP : 0u, ovf);
#endif
                ex = bl.pos;
                counted = true;
            }
        }
        // consistent chain: lanes 0..fstop hold the true sub-sequences of this round
        const u64 stopm = __ballot(fl != SPAN_CONT);
        const int fstop = first_lane(stopm);
        const bool valid = lane <= fstop && counted;
        // verify (a lane could still be inconsistent if the iteration cap was hit)
        const u32 pex = __shfl_up(ex, 1, 64);
        const bool bad = valid && lane > 0 && pex != start;
        if (__any(bad) || __any(lane <= fstop && !counted)) { fail = true; break; }
        // prefix sums of the token counts
        u32 x = valid ? nt : 0, y = valid ? no : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 xx = __shfl_up(x, off, 64), yy = __shfl_up(y, off, 64);
            if (lane >= off) { x += xx; y += yy; }
        }
        const int nvalid = fstop < 64 ? fstop + 1 : 64;
        if (nsub + nvalid > SUBCAP - 1) {
            if (sub_bits < SUB_BITS) { sub_bits = SUB_BITS; base = base0; tot_tok = 0; tot_out = 0; nsub = 0; rows_ok = true; continue; }      // (the guess was too short)
            fail = true; break;
        }
        if (!have_rows || __any(valid && ovf)) rows_ok = false;
        if (lane == 0) round_row[(u64)slot * ROUNDS_MAX + (nsub >> 6)] = rb;       // (every round before this one had 64 sub-sequences)
        if (valid) sub[nsub + lane] = make_uint2((u32)((wb0 << 5) + start - o), tot_tok + x - nt);
        const u32 rt = __shfl(x, nvalid - 1, 64), ro = __shfl(y, nvalid - 1, 64);
        tot_tok += rt; tot_out += ro;
        nsub += nvalid;
        if (fstop < 64) {
            const int ffl = __shfl(fl, fstop, 64);
            const u32 fex = __shfl(ex, fstop, 64);
            if (ffl == SPAN_EOB) { done = true; r.end_bit = (wb0 << 5) + fex; }
            else fail = true;
        } else base = (wb0 << 5) + (u32)__shfl(ex, 63, 64);
    }
    if (lane == 0) {
        if (done && !fail) {
            sub[nsub] = make_uint2(0, tot_tok);
            r.ok = 1; r.ntok = tot_tok; r.nout = tot_out; r.nsub = nsub; r.rows = rows_ok ? 1u : 0u;
        }
        cres[slot] = r;
    }
}

// ================================================================================================
// Fast path, step 3: chain the blocks of each chunk (one wave per chunk)
// ================================================================================================
constexpr int CHAIN_NEED_SEQ = 1;     // value of seq_flag: the wave decoder finishes the chunk
constexpr u64 CHAIN_INLINE_BYTES = 4096;   // longest unannounced Huffman block (bytes of output) the chain walk counts itself
constexpr int CHAIN_LDS_CAND = 1024;      // candidates of a chunk the chain walk holds in LDS (a 23 MB chunk has ~320 blocks)

__global__ __launch_bounds__(64) void k_inf_chain(const u8 *__restrict__ cdata, const InfChunk *__restrict__ chunks,
                                                  const InfFast *__restrict__ fast, const u64 *__restrict__ cand_pos,
                                                  const u32 *__restrict__ cand_cnt, const CandRes *__restrict__ cres,
                                                  TrueBlk *__restrict__ tblk, u32 *__restrict__ true_cnt,
                                                  InfResult *__restrict__ res, int *__restrict__ seq_flag)
{
    const int ci = blockIdx.x;
    const InfChunk ch = chunks[ci];
    const InfFast f = fast[ci];
    const int lane = threadIdx.x;
    __shared__ __attribute__((aligned(16))) u8 tabs[INF_SET_BYTES];
    InfResult r;
    r.status = MTS_CHUNK_OK; r.n_out = 0; r.ntok = 0; r.adler_stored = 0; r.end_bit = 0;
    // zlib header
    const u8 *cb = cdata + ch.c_off;
    bool ok = ch.c_len >= 2;
    if (ok) {
        const u32 cmf = cb[0], flg = cb[1];
        ok = ((cmf << 8) | flg) % 31 == 0 && (cmf & 15) == 8 && (cmf >> 4) <= 7 && !(flg & 0x20);
    }
    if (!ok) { if (lane == 0) { r.status = MTS_CHUNK_CORRUPT; res[ci] = r; seq_flag[ci] = 0; true_cnt[ci] = 0; } return; }
    BitIn br;
    br.init(cdata, ch.c_off, ch.c_len, 16);
    const u32 ncand = min(cand_cnt[ci], f.cand_cap);
    const bool cand_overflow = cand_cnt[ci] > f.cand_cap;
    const u64 *cp = cand_pos + f.cand_off;
    u32 cur = 0;                   // candidates below `cur` are behind the walk
    u64 pos = br.pos;
    u32 ntok = 0, ntrue = 0;
    u64 nout = 0;
    bool last = false, need_seq = cand_overflow, enough = false;      // enough: a partial decode has the bytes it wants
    LaneLds L;
    L.bind(tabs, 1, 0);
    // The walk goes from block to block: where a block ends is where the next one must have been announced.  Position and result
    // of 64 candidates at a time are kept in the lanes' registers (round 5): the search for the candidate at `pos` and the look at
    // its result are a ballot and a few shuffles instead of two dependent trips to memory per block (0.23 ms for the 320 blocks of
    // a chunk, whatever the number of chunks: 8 % of a cold read).
    // Round 5, second half: the whole walk out of LDS.  All of a chunk's candidates (a few hundred) are loaded at once, every
    // candidate finds the one that starts where it ends (a binary search over the sorted positions, all candidates at a time),
    // and the walk is a hop from record to record: ~0.1 us per block instead of 0.5 (a lone wave's dependent instructions: the
    // ballot, eight v_readlane and the reload every 64 candidates).  Where the walk does not find a block announced -- or a
    // chunk has more candidates than the LDS holds -- the loop below takes over where it stands.
    {
        __shared__ uint4 c_rec[CHAIN_LDS_CAND];                  // end bit (x, y), tokens (z), bytes | ok << 30 | bfinal << 31 (w)
        __shared__ u64 c_pos[CHAIN_LDS_CAND];
        __shared__ u16 c_succ[CHAIN_LDS_CAND];
        if (ncand <= (u32)CHAIN_LDS_CAND && ncand > 0 && !need_seq) {
            for (u32 i = lane; i < ncand; i += 64) {
                const CandRes c = cres[f.cand_off + i];
                c_pos[i] = cp[i];
                c_rec[i] = make_uint4((u32)c.end_bit, (u32)(c.end_bit >> 32), c.ntok, (c.nout & 0x3fffffffu) | (c.ok ? 1u << 30 : 0u) | (c.bfinal ? 1u << 31 : 0u));
            }
            __builtin_amdgcn_wave_barrier();
            auto find = [&](u64 p) -> u32 {                         // the candidate at exactly p, or 0xffff
                u32 lo = 0, hi = ncand;
                while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (c_pos[mid] < p) lo = mid + 1; else hi = mid; }
                return (lo < ncand && c_pos[lo] == p) ? lo : 0xffffu;
            };
            for (u32 i = lane; i < ncand; i += 64) {
                const uint4 e = c_rec[i];
                c_succ[i] = (u16)(((e.w >> 30) & 1) ? find((u64)e.x | ((u64)e.y << 32)) : 0xffffu);
            }
            __builtin_amdgcn_wave_barrier();
            u32 i = find(pos);
            while (i != 0xffffu) {
                const uint4 e = c_rec[i];
                if (!((e.w >> 30) & 1)) break;                       // announced, but pass A could not finish it: the loop below looks at it
                const u32 b_ntok = e.z, b_nout = e.w & 0x3fffffffu;
                if (ntrue >= f.true_cap || (u64)ntok + b_ntok > (u64)ch.n_expect + 1 || nout + b_nout > ch.n_expect) { need_seq = true; break; }
                if (lane == 0) {
                    TrueBlk tb;
                    tb.start_bit = pos; tb.cand = (u32)(f.cand_off + i); tb.tok_off = ntok; tb.ntok = b_ntok; tb.chunk = (u32)ci;
                    tblk[f.true_off + ntrue] = tb;
                }
                ntrue++;
                ntok += b_ntok; nout += b_nout;
                pos = (u64)e.x | ((u64)e.y << 32);
                cur = i + 1;
                last = (e.w >> 31) != 0;
                if (last) break;
                if (ch.n_need && nout >= ch.n_need) { enough = true; break; }
                i = c_succ[i];
            }
        }
    }
    u32 hb = 0;                    // the candidates held: hb + lane
    bool have = false;
    u64 hv = ~0ull, h_end = 0;
    u32 h_ntok = 0, h_nout = 0, h_ok = 0, h_bfinal = 0;
    while (!last && !need_seq && !enough) {
        // find a candidate at exactly `pos` (wave-wide search forward)
        int found = -1, fl = 0;
        for (;;) {
            if (!have) {
                hb = cur;
                const u32 i = hb + lane;
                hv = i < ncand ? cp[i] : ~0ull;
                h_ok = 0;
                if (i < ncand) { const CandRes c = cres[f.cand_off + i]; h_end = c.end_bit; h_ntok = c.ntok; h_nout = c.nout; h_ok = c.ok; h_bfinal = c.bfinal; }
                have = true;
            }
            const u64 ge = __ballot(hv >= pos);
            if (ge == 0) { cur = hb + 64; have = false; if (cur >= ncand) break; continue; }
            fl = first_lane(ge);
            const u64 fv = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)hv, fl) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(hv >> 32), fl) << 32);
            cur = hb + (u32)fl;
            if (fv == pos) found = (int)cur;
            break;
        }
        u32 b_ntok = 0, b_nout = 0, b_cand = 0xffffffffu;
        u64 b_end = 0;
        bool b_ok = false;
        // (fl is the same in every lane: v_readlane, not a trip through the LDS crossbar)
        auto rl = [&](u32 x) -> u32 { return (u32)__builtin_amdgcn_readlane((int)x, fl); };
        if (found >= 0 && rl(h_ok)) {
            b_ok = true; b_ntok = rl(h_ntok); b_nout = rl(h_nout); b_end = (u64)rl((u32)h_end) | ((u64)rl((u32)(h_end >> 32)) << 32);
            b_cand = (u32)(f.cand_off + found); last = rl(h_bfinal);
        }
        if (!b_ok) {
            // A block the scan did not announce.  A stored block is its length field; anything else (fixed Huffman, a
            // dynamic header the scan missed or pass A could not finish) is counted here, every lane running the same
            // decode, only while it is SHORT (zlib closes a stream with a small fixed block now and then): a long one, or
            // an error, hands the rest of the chunk to the wave decoder, which takes over at this block.
            br.seek(pos);
            const u32 hdr = br.get(3);
            if ((hdr >> 1) == 0 && br.pos <= br.end) {
                br.seek((br.pos + 7) & ~7ull);
                const u32 len = br.get(16), nlen = br.get(16);
                if (br.pos > br.end || (len ^ 0xffff) != nlen || br.pos + 8ull * len > br.end) { need_seq = true; break; }
                b_ntok = len; b_nout = len; b_end = br.pos + 8ull * len; last = hdr & 1;
            } else {
                br.seek(pos);
                u32 nt = 0; u64 no = nout; bool lst = false;
                const int st = decode_block<false>(br, L, nullptr, nt, no, nout + CHAIN_INLINE_BYTES, lst);
                if (st != INF_OK) { need_seq = true; break; }
                b_ntok = nt; b_nout = (u32)(no - nout); b_end = br.pos; last = lst;
            }
        }
        if (ntrue >= f.true_cap || (u64)ntok + b_ntok > (u64)ch.n_expect + 1 || nout + b_nout > ch.n_expect) { need_seq = true; break; }
        if (lane == 0) {
            TrueBlk tb;
            tb.start_bit = pos; tb.cand = b_cand; tb.tok_off = ntok; tb.ntok = b_ntok; tb.chunk = (u32)ci;
            tblk[f.true_off + ntrue] = tb;
        }
        ntrue++;
        ntok += b_ntok; nout += b_nout;
        pos = b_end;
        if (ch.n_need && nout >= ch.n_need) enough = true;
    }
    if (enough) {
        // whole blocks up to the one that reaches the wanted prefix (the stream buffer has room for the whole chunk)
        r.end_bit = pos; r.n_out = (u32)nout; r.ntok = ntok;
    } else if (need_seq && ch.n_need) {
        // the bytes given end before the wanted prefix does (or the fast path cannot follow them): the caller comes back with the whole chunk
        r.status = MTS_CHUNK_NEEDMORE;
        need_seq = false;
        ntrue = 0;
    } else if (!need_seq) {
        const u64 tb = (pos + 7) & ~7ull;
        if (tb + 32 > br.end) r.status = MTS_CHUNK_CORRUPT;                   // no room for the check value
        else {
            const u8 *t = (const u8 *)br.w + (tb >> 3);
            r.adler_stored = ((u32)t[0] << 24) | ((u32)t[1] << 16) | ((u32)t[2] << 8) | t[3];
            r.end_bit = tb + 32;
            if (nout != ch.n_expect) r.status = MTS_CHUNK_BADSIZE;            // a valid stream of another length
        }
        r.n_out = (u32)nout; r.ntok = ntok;
    } else {
        // the wave decoder resumes where the walk stopped; the blocks accepted so far keep their tokens (pass B)
        r.end_bit = pos; r.n_out = (u32)nout; r.ntok = ntok;
    }
    if (lane == 0) {
        res[ci] = r;
        seq_flag[ci] = need_seq ? CHAIN_NEED_SEQ : 0;
        true_cnt[ci] = ntrue;
    }
}

// ================================================================================================
// Fast path, step 4a: the accepted blocks whose tokens pass A kept -- rows to their place in the chunk's token array
// ================================================================================================
// Row j holds the tokens of sub-sequence j, which belong at sub[j].y of the block.  A thread takes one token of the block at a
// time, finds the row it is in (the row starts are ascending; the threads go through the block in order, so the search
// starts at the row of the workgroup's first token of the step before) and moves it: reads run along rows, writes are
// consecutive.
constexpr int MOVE_THREADS = 256;
__global__ __launch_bounds__(MOVE_THREADS) void k_inf_move_rows(const InfChunk *__restrict__ chunks, const InfFast *__restrict__ fast,
                                                                const u32 *__restrict__ tslot_chunk, const TrueBlk *__restrict__ tblk,
                                                                const u32 *__restrict__ true_cnt, const CandRes *__restrict__ cres,
                                                                const uint2 *__restrict__ subs, const u32 *__restrict__ rows,
                                                                const u32 *__restrict__ round_row, u32 *__restrict__ tokens)
{
    const u32 slot = blockIdx.x;
    const u32 ci = tslot_chunk[slot];
    const InfFast f = fast[ci];
    if (slot - f.true_off >= true_cnt[ci]) return;
    const TrueBlk tb = tblk[slot];
    if (tb.cand == 0xffffffffu || !cres[tb.cand].rows) return;
    __shared__ u32 ys[SUBCAP + 1], rid[SUBCAP];
    const u32 nsub = cres[tb.cand].nsub, total = tb.ntok;
    const uint2 *sub = subs + (u64)tb.cand * SUBCAP;
    for (u32 j = threadIdx.x; j < nsub; j += MOVE_THREADS) {
        ys[j] = sub[j].y;
        rid[j] = round_row[(u64)tb.cand * ROUNDS_MAX + (j >> 6)] * 64 + (j & 63);
    }
    if (threadIdx.x == 0) ys[nsub] = 0xffffffffu;               // (ends every search)
    __syncthreads();
    u32 *tk = tokens + chunks[ci].tok_off + tb.tok_off;
    constexpr u32 U = 16;                                       // tokens a thread has in flight (4: 0.72 ms for row move + pass B, 8: 0.71, 16: 0.69)
    u32 jw = 0;                                                 // a row at or before the one of the workgroup's first token of this step
    for (u32 t0 = 0; t0 < total; t0 += U * MOVE_THREADS) {
        while (ys[jw + 1] <= t0) jw++;                          // (uniform)
        u32 v[U], j = jw;
#pragma unroll
        for (u32 u = 0; u < U; u++) {
            const u32 t = min(t0 + u * MOVE_THREADS + threadIdx.x, total - 1);
            while (ys[j + 1] <= t) j++;
            v[u] = rows[(u64)rid[j] * ROWCAP + (t - ys[j])];
        }
#pragma unroll
        for (u32 u = 0; u < U; u++) {
            const u32 t = t0 + u * MOVE_THREADS + threadIdx.x;
            if (t < total) tk[t] = v[u];
        }
    }
}

// ================================================================================================
// Fast path, step 4: emit the tokens of every accepted block (one wave per block)
// ================================================================================================
#ifndef MTS_PASSB_WAVES
#define MTS_PASSB_WAVES 1
#endif
constexpr int PASSB_WAVES = MTS_PASSB_WAVES;      // waves that share a block's tables and take its steps in turn.  Measured: 1: 2.90 ms, 2: 2.98, 4: 3.76 (more waves
                                                  // per CU, but they wait for the first one's tables and for each other's last, short step)
__global__ __launch_bounds__(64 * PASSB_WAVES) void k_inf_passB(const u8 *__restrict__ cdata, const InfChunk *__restrict__ chunks,
                                                  const InfFast *__restrict__ fast, const u32 *__restrict__ tslot_chunk,
                                                  const TrueBlk *__restrict__ tblk, const u32 *__restrict__ true_cnt,
                                                  const CandRes *__restrict__ cres, const uint2 *__restrict__ subs,
                                                  u32 *__restrict__ tokens, InfResult *__restrict__ res, int use_rows)
{
    const u32 slot = blockIdx.x;
    const u32 ci = tslot_chunk[slot];
    const InfFast f = fast[ci];
    if (slot - f.true_off >= true_cnt[ci]) return;
    const TrueBlk tb = tblk[slot];
    const InfChunk ch = chunks[ci];
    __shared__ __attribute__((aligned(16))) u8 tabs[INF_SET_BYTES];
    LaneLds L;
    L.bind(tabs, 1, 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    BitIn br;
    br.init(cdata, ch.c_off, ch.c_len, 0);
    br.seek(tb.start_bit);
    u32 *tk = tokens + ch.tok_off + tb.tok_off;
    if (tb.cand == 0xffffffffu) {
        if (wave) return;
        // a block the chain walk counted itself: stored bytes become literal tokens 64 at a time; a short Huffman block is
        // decoded by lane 0
        const u32 hdr0 = br.get(3);
        if ((hdr0 >> 1) == 0) {
            br.seek((br.pos + 7) & ~7ull);
            const u32 len = br.get(16);
            const u8 *src = (const u8 *)br.w + ((br.pos + 16) >> 3);
            if (len != tb.ntok) { if (lane == 0) res[ci].status = MTS_CHUNK_CORRUPT; return; }
            for (u32 i = lane; i < len; i += 64) tk[i] = src[i];
            return;
        }
        br.seek(tb.start_bit);
        u32 nt = 0; u64 no = 1ull << 40; bool lst;
        if (lane == 0) {
            const int st = decode_block<true>(br, L, tk, nt, no, ~0ull, lst);
            if (st != INF_OK || nt != tb.ntok) res[ci].status = MTS_CHUNK_CORRUPT;
        }
        return;
    }
    __shared__ u32 stage_s[PASSB_WAVES][PASSB_STAGE_WORDS];
    if (use_rows && cres[tb.cand].rows) return;                 // pass A kept this block's tokens: k_inf_move_rows puts them in place
    const u32 hdr = br.get(3);
    __shared__ u32 lut_s[LUT_BYTES / 4];
    u32 *lutl = lut_s, *lutd = lut_s + (1 << LUT_LBITS);
    u32 *LC = lutd + (1 << LUT_DBITS), *DC = LC + 16;               // the compare chains live in LDS (wave-shared)
    // the first wave parses the header (every lane of it: identical control flow, identical LDS writes) and fills the tables
    __shared__ int hdr_rc;
    if (wave == 0) {
        const int rc = parse_tables(br, L, hdr >> 1, LC, DC);
        if (lane == 0) hdr_rc = rc;
        __builtin_amdgcn_wave_barrier();
        if (rc == INF_OK) build_luts(L, lutl, lutd, lane);
    }
    __syncthreads();
    if (hdr_rc != INF_OK) { if (threadIdx.x == 0) res[ci].status = MTS_CHUNK_CORRUPT; return; }
    const u32 nsub = cres[tb.cand].nsub;
    const uint2 *sub = subs + (u64)tb.cand * SUBCAP;
    // every lane reads its own sub-sequence word by word: straight from memory that is one dependent, uncoalesced
    // load per ~3 tokens.  The 64 sub-sequences of a step are one contiguous piece of the stream (~16 KiB):
    // it is copied to LDS with coalesced loads first.
    u32 *stage = stage_s[wave];
    for (u32 j0 = 64 * (u32)wave; j0 < nsub; j0 += 64 * PASSB_WAVES) {
        const u32 j = j0 + lane;
        const u32 jl = min(j0 + 64, nsub);
        // (the entry after the last sub-sequence is a terminator without a position: the round then ends where the block does)
        const u64 round_end = jl < nsub ? tb.start_bit + sub[jl].x : cres[tb.cand].end_bit;
        const u64 wlo = (tb.start_bit + sub[j0].x) >> 5, whi = (round_end >> 5) + 3;
        __builtin_amdgcn_wave_barrier();
        const bool staged = whi - wlo < (u64)PASSB_STAGE_WORDS;
        if (staged) {
            const u32 nw = (u32)(whi - wlo) + 1;
            stage_stream_words(stage, br, wlo, nw, (u32)lane, 64);
        }
        __builtin_amdgcn_wave_barrier();
        if (j < nsub) {
            const uint2 a = sub[j], b = sub[j + 1];
            u32 nt, no; int fl;
            if (staged) {
                BitL bl;
                bl.w = stage;
                const u64 end_rel64 = br.end - (wlo << 5);
                bl.seek((u32)(tb.start_bit + a.x - (wlo << 5)));
                decode_span_fast<2>(bl, L, lutl, lutd, 0, end_rel64 < 0x7fffffffull ? (u32)end_rel64 : 0x7fffffffu, nt, no, fl, tk + a.y, b.y - a.y);
            } else {
                br.seek(tb.start_bit + a.x);
                decode_span<2>(br, L, lutl, lutd, 0, nt, no, fl, tk + a.y, b.y - a.y);
            }
            if (nt != b.y - a.y) res[ci].status = MTS_CHUNK_CORRUPT;
        }
    }
}

// ================================================================================================
// Block-after-block decoder: one workgroup per chunk, authoritative for whatever the fast path declined
// ================================================================================================
// Streams the block-start scan cannot open up (fixed-Huffman or stored blocks from end to end, more candidates than the
// tables hold) and chunks whose block chain broke (damage) come here.  The blocks are taken in order, but each one is
// decoded by all 1024 threads the way pass A does it with 64: consecutive sub-sequences of the block, every thread
// restarting from its left neighbour's exit until the decodes chain like a sequential one -- and, once the chain stands,
// decoded once more from the true starts to write the tokens (or, past the promised size, only to check them).  Wave 0
// reads the block headers and builds the code tables.  A 23 MB chunk of fixed-Huffman blocks takes ~0.1 s instead of the
// minutes of a single lane; a damaged chunk is refused within the block that is damaged (milliseconds).
// With seq_flag the kernel resumes after the blocks the chain walk accepted (res[ci]: bit position, tokens, bytes so far).
constexpr int WV_NT = 1024;
constexpr u32 WV_SUB = 512;                                     // bits per sub-sequence
constexpr int WV_STAGE_WORDS = WV_NT * WV_SUB / 32 + 16;        // a round of the stream + the readers' look-ahead
struct WvCtl { u64 pos; u32 type, last, len; int st; };

__global__ __launch_bounds__(WV_NT) void k_inf_wave(const u8 *__restrict__ cdata, const InfChunk *__restrict__ chunks,
                                                    int n_chunks, u32 *__restrict__ tokens, InfResult *__restrict__ res,
                                                    const int *__restrict__ seq_flag)
{
    const int ci = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ci >= n_chunks) return;
    if (seq_flag && seq_flag[ci] == 0) return;
    const InfChunk ch = chunks[ci];
    __shared__ __attribute__((aligned(16))) u8 tabs[INF_SET_BYTES];
    __shared__ u32 lut_s[LUT_BYTES / 4];
    __shared__ u32 stage[WV_STAGE_WORDS];
    __shared__ u32 s_ex[WV_NT];
    __shared__ u32 s_wt[WV_NT / 64], s_wo[WV_NT / 64], s_wstop[WV_NT / 64];
    __shared__ int s_any;
    __shared__ WvCtl ctl;
    LaneLds L;
    L.bind(tabs, 1, 0);
    u32 *lutl = lut_s, *lutd = lut_s + (1 << LUT_LBITS);
    u32 *LC = lutd + (1 << LUT_DBITS), *DC = LC + 16;
    InfResult r;
    r.status = MTS_CHUNK_OK; r.n_out = 0; r.ntok = 0; r.adler_stored = 0; r.end_bit = 0;
    BitIn br;
    u32 ntok = 0;
    u64 nout = 0;
    if (seq_flag) {
        const InfResult pr = res[ci];
        if (pr.status != MTS_CHUNK_OK) return;                    // (pass B found an accepted block inconsistent: the verdict stands)
        br.init(cdata, ch.c_off, ch.c_len, 0);
        br.seek(pr.end_bit);
        ntok = pr.ntok; nout = pr.n_out;
    } else {
        const u8 *cb = cdata + ch.c_off;
        bool ok = ch.c_len >= 2;
        if (ok) {
            const u32 cmf = cb[0], flg = cb[1];
            ok = ((cmf << 8) | flg) % 31 == 0 && (cmf & 15) == 8 && (cmf >> 4) <= 7 && !(flg & 0x20);
        }
        if (!ok) { r.status = MTS_CHUNK_CORRUPT; if (tid == 0) res[ci] = r; return; }
        br.init(cdata, ch.c_off, ch.c_len, 16);
    }
    u32 *tk = tokens + ch.tok_off;
    const u64 limit = ch.n_expect;
    bool last = false, emit = true, fixed_ready = false;
    int st = INF_OK;
    u64 pos = br.pos;                                              // (uniform) where the next block starts
    BitL bl;
    bl.w = stage;
    while (!last && st == INF_OK) {
        // ---- wave 0: block header, stored-block fields, code tables ----
        __syncthreads();
        if (wave == 0) {
            WvCtl c;
            c.st = INF_OK; c.len = 0;
            br.seek(pos);
            const u32 hdr = br.get(3);
            c.last = hdr & 1; c.type = hdr >> 1;
            if (br.pos > br.end || c.type == 3) c.st = INF_CORRUPT;
            else if (c.type == 0) {
                br.seek((br.pos + 7) & ~7ull);
                const u32 len = br.get(16), nlen = br.get(16);
                if (br.pos > br.end || (len ^ 0xffff) != nlen || br.pos + 8ull * len > br.end) c.st = INF_CORRUPT;
                c.len = len;
            } else if (!(c.type == 1 && fixed_ready)) {
                if (parse_tables(br, L, c.type, LC, DC) != INF_OK) c.st = INF_CORRUPT;
            }
            c.pos = br.pos;
            if (lane == 0) ctl = c;
        }
        __syncthreads();
        const WvCtl c = ctl;
        st = c.st; last = c.last;
        if (st != INF_OK) break;
        if (c.type == 0) {
            if (nout + c.len > limit) emit = false;               // more than the header promises: from here on the stream is only checked
            const u8 *src = (const u8 *)br.w + (c.pos >> 3);
            if (emit) for (u32 i = tid; i < c.len; i += WV_NT) tk[ntok + i] = src[i];
            ntok += emit ? c.len : 0; nout += c.len;
            pos = c.pos + 8ull * c.len;
            continue;
        }
        if (!(c.type == 1 && fixed_ready)) {
            for (u32 e = tid; e < (1u << LUT_LBITS); e += WV_NT) {
                u32 cl;
                const int si = chain_decode_mem<15>(__brev(e) >> 17, LC, cl);
                lutl[e] = (si >= 0 && cl <= (u32)LUT_LBITS) ? lut_len_entry(L.ls(si), cl) : 0u;
            }
            for (u32 e = tid; e < (1u << LUT_DBITS); e += WV_NT) {
                u32 cl;
                const int si = chain_decode_mem<15>(__brev(e) >> 17, DC, cl);
                lutd[e] = (si >= 0 && cl <= (u32)LUT_DBITS) ? lut_dist_entry(L.ds(si), cl) : 0u;
            }
            fixed_ready = c.type == 1;
        }
        u64 base = c.pos;
        bool done = false;
        while (!done && st == INF_OK) {
            const u64 wb0 = base >> 5;
            const u32 bofs = (u32)(base & 31);
            __syncthreads();
            stage_stream_words(stage, br, wb0, (u32)WV_STAGE_WORDS, (u32)tid, (u32)WV_NT);
            __syncthreads();
            const u64 end_rel64 = br.end - (wb0 << 5);
            const u32 end_rel = end_rel64 < 0x7fffffffull ? (u32)end_rel64 : 0x7fffffffu;
            const u32 stop = bofs + (u32)(tid + 1) * WV_SUB;
            u32 start = bofs + (u32)tid * WV_SUB, ex;
            u32 nt, no; int fl;
            bl.seek(start);
            decode_span_fast<0>(bl, L, lutl, lutd, stop, end_rel, nt, no, fl, nullptr, 0);
            ex = bl.pos;
            bool counted = false;
            int fstop = WV_NT;                                       // first thread whose span ends the block (or fails): threads behind it are void
            for (int it = 0; it < WV_NT + 2; it++) {
                s_ex[tid] = ex;
                const u64 stopm = __ballot(fl != SPAN_CONT);
                if (lane == 0) s_wstop[wave] = stopm ? (u32)(wave * 64 + first_lane(stopm)) : (u32)WV_NT;
                if (tid == 0) s_any = 0;
                __syncthreads();
                fstop = WV_NT;
#pragma unroll
                for (int w = 0; w < WV_NT / 64; w++) fstop = min(fstop, (int)s_wstop[w]);
                const u32 want_start = tid == 0 ? bofs : s_ex[tid - 1];
                const bool redo = tid <= fstop && (!counted || want_start != start);
                if (__any(redo) && lane == 0) s_any = 1;
                __syncthreads();
                if (!s_any) break;
                if (redo) {
                    start = want_start;
                    bl.seek(start);
                    decode_span_fast<1>(bl, L, lutl, lutd, stop, end_rel, nt, no, fl, nullptr, 0);
                    ex = bl.pos;
                    counted = true;
                }
                __syncthreads();                                     // (s_ex / s_any are rewritten at the top)
            }
            const bool valid = tid <= fstop && counted;
            // token / byte offsets of every thread: wave scans + the wave totals
            u32 x = valid ? nt : 0, y = valid ? no : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 xx = __shfl_up(x, off, 64), yy = __shfl_up(y, off, 64);
                if (lane >= off) { x += xx; y += yy; }
            }
            __syncthreads();
            if (lane == 63) { s_wt[wave] = x; s_wo[wave] = y; }
            __syncthreads();
            u32 pre_t = 0, pre_o = 0, rt = 0, ro = 0;
#pragma unroll
            for (int w = 0; w < WV_NT / 64; w++) { const u32 a = s_wt[w], bb = s_wo[w]; if (w < wave) { pre_t += a; pre_o += bb; } rt += a; ro += bb; }
            if (nout + ro > limit) emit = false;
            // the true sub-sequences once more: tokens out (or only checked), every copy against the bytes before it
            bool bad = false;
            if (valid) {
                u32 *out = tk + ntok + pre_t + (x - nt);
                u64 have = nout + pre_o + (y - no);
                bl.seek(start);
                u32 k = 0;
                while (k < nt) {
                    u32 tok, olen;
                    if (decode_token_lut(bl, L, lutl, lutd, tok, olen) != 0) { bad = true; break; }
                    if ((tok >> 31) && (u64)((tok & 0x7fff) + 1) > have) { bad = true; break; }       // too far back
                    if (emit) lz_emit_pieces(out + k, tok, olen);
                    k += lz_pieces(tok, olen); have += olen;
                }
            }
            if (tid == 0) s_any = 0;
            __syncthreads();
            if (__any(bad) && lane == 0) s_any = 1;
            // the thread that ended the round says how
            if (fstop < WV_NT && tid == fstop) { s_wstop[0] = (u32)fl; s_wstop[1] = ex; }
            if (fstop == WV_NT && tid == WV_NT - 1) s_wstop[1] = ex;
            __syncthreads();
            if (s_any) { st = INF_CORRUPT; break; }
            ntok += emit ? rt : 0; nout += ro;
            if (fstop < WV_NT) {
                if ((int)s_wstop[0] == SPAN_EOB) { done = true; pos = (wb0 << 5) + s_wstop[1]; }
                else st = INF_CORRUPT;
            } else base = (wb0 << 5) + s_wstop[1];
        }
    }
    if (st != INF_OK) r.status = MTS_CHUNK_CORRUPT;
    else {
        // trailer: adler32, big endian, at the next byte boundary.  A stream that is valid to its end but of another size
        // is a size mismatch (AssertionError, mtscomp.py:628); anything else is corruption (IOError, :621)
        const u64 tb = (pos + 7) & ~7ull;
        if (tb + 32 > br.end) r.status = MTS_CHUNK_CORRUPT;
        else {
            const u8 *t = (const u8 *)br.w + (tb >> 3);
            r.adler_stored = ((u32)t[0] << 24) | ((u32)t[1] << 16) | ((u32)t[2] << 8) | t[3];
            r.end_bit = tb + 32;
            if (!emit || nout != limit) r.status = MTS_CHUNK_BADSIZE;
        }
    }
    r.n_out = (u32)(nout < 0xffffffffull ? nout : 0xffffffffull); r.ntok = ntok;
    if (tid == 0) res[ci] = r;
}

// ================================================================================================
// Z: tokens -> bytes (LZ77 resolution)
// ================================================================================================
// One workgroup per chunk.  Tokens are taken in groups of 64 (one per lane); LZ_WORKERS waves take the
// groups round robin, LZ_FLUSHERS waves stream finished bytes to HBM.  The 64 KiB of history a copy can
// reach lives in LDS as a byte ring plus one "written" bit per byte.  A copy piece (3..8 bytes) reads
// the bits and the bytes of its source and commits when every byte it needs is written, so every piece
// waits for exactly the bytes it depends on (no group-level barrier, no serial commit): the critical
// path is the copy-of-copy depth of the data, not the number of tokens.
//
// The DS unit of gfx950 executes a misaligned access one lane per clock (measured: 64 cycles per
// ds_read_b64 against 4-8 aligned), so every access here is dword aligned: a source window is three
// aligned dwords funnel-shifted in registers, and a destination is written by OR-ing the shifted bytes
// into ring slots that are known to be zero (LDS atomics, no return).  The flushers re-zero a ring slot
// (bytes and bits) LZ_ZLAG granules after streaming it out, when no copy can reach it any more; a
// worker runs at most LZ_AHEAD bytes past what the flushers have processed, which is exactly the
// zeroed part of the ring.  Ordering relies on the LDS executing one wave's instructions in issue
// order: writers OR data then bits, readers load bits then data.
constexpr int LZ_THREADS = 1024;
constexpr int LZ_FLUSHERS = 2;                           // waves streaming finished bytes to HBM
constexpr int LZ_WORKERS = LZ_THREADS / 64 - LZ_FLUSHERS;
constexpr u32 LZ_RING = 65536;                           // positions held (power of two)
constexpr u32 LZ_REACH = 32768 + 258;                    // furthest back a copy reads from its own start
constexpr u32 LZ_FLUSH = 4096;                           // flush granule (bytes)
constexpr u32 LZ_ZLAG = 10;                              // a granule's ring slot is zeroed when the granule LZ_ZLAG later is flushed
constexpr u32 LZ_AHEAD = LZ_RING - LZ_ZLAG * LZ_FLUSH;   // a worker may write this far past flushed()
static_assert(LZ_REACH <= (LZ_ZLAG - 1) * LZ_FLUSH, "a slot must be out of every copy's reach before it is zeroed");
constexpr u32 LZ_BITS_OFF = LZ_RING;                     // LDS layout: bytes, bits (+ one pad dword), control words
constexpr u32 LZ_CTL_OFF = LZ_RING + LZ_RING / 8 + 16;
constexpr int LZ_LDS = LZ_CTL_OFF + 256;
constexpr u32 LZ_EDGE = LZ_RING - 12;                    // windows starting above this ring offset go byte by byte
constexpr u32 LZ_SPIN_MAX = 1u << 20;                     // bound on every wait loop (a stuck kernel must end)

// A chunk is one serial dependency chain for the resolver, and one workgroup resolves ~2.6 GB/s: with fewer
// chunks than CUs most of the chip would idle.  So a chunk's groups are cut into up to LZ_MAXSEG segments
// resolved by different workgroups (k_inf_lz_seg).  A segment does not know the 32 KiB before its first byte,
// so it works on 16-bit cells: 0..255 = a byte, 256 + i = "byte i of my unknown window".  The cells leave
// as they are; k_inf_windows then makes the real windows one after the other (window k = the last 32 KiB
// of segment k-1, translated with window k-1) and k_inf_translate turns every cell into a byte.
#ifndef MTS_LZ_MAXSEG
#define MTS_LZ_MAXSEG 32
#endif
constexpr int LZ_MAXSEG = MTS_LZ_MAXSEG;
constexpr int LZ_STATUS_RETRY = 1000;                    // internal chunk status: a resolver wait expired, run the chunk again with one worker wave
constexpr u32 LZ_WIN = 32768;
struct LzPlan {
    u32 nseg;                          // 1: the whole chunk by k_inf_lz (bytes)
    u32 g0[LZ_MAXSEG + 1];             // first group of segment k (g0[nseg] = number of groups)
    u32 b0[LZ_MAXSEG + 1];             // first output byte of segment k (b0[nseg] = output size)
};
constexpr u32 LZ2_CTL_OFF = 2 * LZ_RING + 16;            // segment kernel: 16-bit cells (+ pad), control words
constexpr int LZ2_LDS = LZ2_CTL_OFF + 256;

// Output offset of every 64-token group, as a two-level scan:
//   k_inf_gsum   one thread per group sums the bytes its 64 tokens produce; a 256-thread workgroup scans its
//                256 groups (a "tile") -> gbase[g] = offset inside the tile, tile_tot[tile]
//   k_inf_gscan  one wave per chunk scans the tile totals -> tile_base[tile]
// group g starts at tile_base[g >> 8] + gbase[g].
constexpr int GS_TILE = 256;

__global__ __launch_bounds__(GS_TILE) void k_inf_gsum(const u32 *__restrict__ tokens, const InfChunk *__restrict__ chunks,
                                                      const InfResult *__restrict__ res, const u64 *__restrict__ gb_off,
                                                      const u64 *__restrict__ tb_off, u32 *__restrict__ gbase,
                                                      u32 *__restrict__ tile_base)
{
    const int ci = blockIdx.y;
    const InfResult r = res[ci];
    if (r.status != MTS_CHUNK_OK) return;
    const u32 ntok = r.ntok, ngroups = (ntok + 63) / 64;
    const u32 g = blockIdx.x * GS_TILE + threadIdx.x;
    if (blockIdx.x * GS_TILE >= ngroups && !(blockIdx.x == 0)) return;
    const u32 *tk = tokens + chunks[ci].tok_off;
    u32 sum = 0;
    {
        // A wave takes the 64 groups of its threads TOGETHER (round 5): 16 loads of 16 bytes per lane over consecutive addresses -- a
        // group's 64 tokens are the 16 lanes of a row in one of them, its sum a row reduction on the DPP network --, instead of a
        // thread reading its own group's 256 bytes with the lanes 256 bytes apart (64 lines per load instruction: the kernel spent
        // 70 % of its cycles waiting to issue).
        const int lane_ = threadIdx.x & 63;
        const u32 g0w = g - (u32)lane_;                              // the wave's first group
        const uint4 *q = (const uint4 *)(tk + (size_t)g0w * 64);      // (tok_off and 64 g are multiples of 4 tokens)
        const u32 nq = ntok > g0w * 64 ? (ntok - g0w * 64) / 4 : 0;   // whole 16-byte pieces of the chunk's tokens from there on
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const u32 i = (u32)it * 64 + (u32)lane_;
            u32 part = 0;
            if (i < nq) {
                const uint4 v = q[i];
                part = ((v.x >> 31) ? ((v.x >> 16) & 0xff) + 3 : 1) + ((v.y >> 31) ? ((v.y >> 16) & 0xff) + 3 : 1) +
                       ((v.z >> 31) ? ((v.z >> 16) & 0xff) + 3 : 1) + ((v.w >> 31) ? ((v.w >> 16) & 0xff) + 3 : 1);
            }
            // sum of the row's 16 lanes into its last lane
            part += (u32)__builtin_amdgcn_update_dpp(0, (int)part, 0x111, 0xf, 0xf, false);     // row_shr:1
            part += (u32)__builtin_amdgcn_update_dpp(0, (int)part, 0x112, 0xf, 0xf, false);     // row_shr:2
            part += (u32)__builtin_amdgcn_update_dpp(0, (int)part, 0x114, 0xf, 0xf, false);     // row_shr:4
            part += (u32)__builtin_amdgcn_update_dpp(0, (int)part, 0x118, 0xf, 0xf, false);     // row_shr:8
            // group 4 it + r is row r's: to lane 4 it + r
            const u32 got = (u32)__shfl((int)part, (lane_ & 3) * 16 + 15, 64);
            if ((lane_ >> 2) == it) sum = got;
        }
        // the chunk's last tokens, fewer than four
        if (g < ngroups) {
            const u32 t0 = g * 64, t1 = min(t0 + 64, ntok);
            for (u32 i = max(t0, (g0w * 64 + nq * 4)); i < t1; i++) { const u32 t = tk[i]; sum += (t >> 31) ? ((t >> 16) & 0xff) + 3 : 1; }
        } else sum = 0;
    }
    __shared__ u32 wtot[GS_TILE / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32 x = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    u32 add = 0, tot = 0;
    for (int w = 0; w < GS_TILE / 64; w++) { if (w < wave) add += wtot[w]; tot += wtot[w]; }
    if (g < ngroups) gbase[gb_off[ci] + g] = add + x - sum;
    if (threadIdx.x == 0) tile_base[tb_off[ci] + blockIdx.x] = tot;          // totals; scanned in place next
}

__global__ __launch_bounds__(64) void k_inf_gscan(const InfResult *__restrict__ res, const u64 *__restrict__ tb_off,
                                                  u32 *__restrict__ tile_base)
{
    const int ci = blockIdx.x;
    const InfResult r = res[ci];
    if (r.status != MTS_CHUNK_OK) return;
    const u32 ngroups = (r.ntok + 63) / 64, ntiles = (ngroups + GS_TILE - 1) / GS_TILE;
    u32 *tb = tile_base + tb_off[ci];
    const int lane = threadIdx.x;
    u32 carry = 0;
    for (u32 t0 = 0; t0 < ntiles; t0 += 64) {
        const u32 v = t0 + lane < ntiles ? tb[t0 + lane] : 0;
        u32 x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const u32 y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
        if (t0 + lane < ntiles) tb[t0 + lane] = carry + x - v;
        carry += __shfl(x, 63, 64);
    }
}

// inclusive wave scan on the DPP network (row shifts, then the two row broadcasts of gfx9)
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 x)
{
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return x;
}

typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
// LDS accesses of the resolver, in program order (see the note on ordering above); addresses are LDS byte
// offsets, dword aligned
__device__ __forceinline__ void lz_load_window(u32 bits_addr, u32 data_addr, u32 &b0, u32 &b1, u32 &d0, u32 &d1, u32 &d2)
{
    u32x2 b, d;
    asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %4 offset1:1\n\tds_read_b32 %2, %4 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(b), "=&v"(d), "=&v"(d2) : "v"(bits_addr), "v"(data_addr) : "memory");
    b0 = b.x; b1 = b.y; d0 = d.x; d1 = d.y;
}
__device__ __forceinline__ void lz_or_window(u32 data_addr, u32 w0, u32 w1, u32 w2, u32 bits_addr, u32 m0, u32 m1)
{
    asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %0, %2 offset:4\n\tds_or_b32 %0, %3 offset:8\n\tds_or_b32 %4, %5\n\tds_or_b32 %4, %6 offset:4"
                 :: "v"(data_addr), "v"(w0), "v"(w1), "v"(w2), "v"(bits_addr), "v"(m0), "v"(m1) : "memory");
}
__device__ __forceinline__ void lz_or_byte(u32 data_addr, u32 w, u32 bits_addr, u32 m)
{
    asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %2, %3" :: "v"(data_addr), "v"(w), "v"(bits_addr), "v"(m) : "memory");
}
__device__ __forceinline__ void lz_load_byte(u32 bits_addr, u32 data_addr, u32 &bits, u32 &data)
{
    asm volatile("ds_read_b32 %0, %2\n\tds_read_u8 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(bits), "=&v"(data) : "v"(bits_addr), "v"(data_addr) : "memory");
}
__device__ __forceinline__ void lds_zero16(u32 addr)
{
    const u32x4 z = {0, 0, 0, 0};
    asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(z) : "memory");
}

// control words of the resolver, addressed as LDS byte offsets (explicit DS instructions: a volatile
// pointer into dynamic LDS would be accessed with flat loads)
__device__ __forceinline__ u32 lds_ld(u32 addr)
{
    u32 v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st(u32 addr, u32 v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
// minimum over lanes 0..15 (row 0 of the wave), wave-uniform result
__device__ __forceinline__ u32 row0_min_dpp(u32 x)
{
    x = min(x, (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x111, 0xf, 0xf, false));
    x = min(x, (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x112, 0xf, 0xf, false));
    x = min(x, (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x114, 0xf, 0xf, false));
    x = min(x, (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x118, 0xf, 0xf, false));
    return (u32)__builtin_amdgcn_readlane((int)x, 15);
}

__global__ __launch_bounds__(LZ_THREADS) void k_inf_lz(const u32 *__restrict__ tokens, const InfChunk *__restrict__ chunks,
                                                       InfResult *__restrict__ res, const u64 *__restrict__ gb_off,
                                                       const u64 *__restrict__ tb_off, const u32 *__restrict__ gbase,
                                                       const u32 *__restrict__ tile_base, u8 *__restrict__ stream, int n_workers,
                                                       u64 *__restrict__ prof, const LzPlan *__restrict__ plan, int pass)
{
    const int ci = blockIdx.x;
    const InfResult r = res[ci];
    if (pass == 2) { if (threadIdx.x == 0 && r.status == MTS_CHUNK_OK && plan[ci].nseg == 1) res[ci].status = LZ_STATUS_RETRY; return; }   // (test hook)
    if (r.status != (pass ? LZ_STATUS_RETRY : MTS_CHUNK_OK)) return;    // pass 1: only the chunks whose first run gave up waiting
    if (plan[ci].nseg != 1) return;                                  // cut into segments: k_inf_lz_seg
    const InfChunk ch = chunks[ci];
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lds_data = (u32)(uintptr_t)smem;                       // LDS byte offsets of the byte ring, the bit ring, ...
    const u32 lds_bits = lds_data + LZ_BITS_OFF;
    const u32 lds_prog = lds_data + LZ_CTL_OFF;                      // [w] = first byte of the group worker w is on
    const u32 lds_flnext = lds_prog + 64;                            // [k] = first byte of the next granule of flusher k
    const u32 lds_bad = lds_prog + 80;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 *tk = tokens + ch.tok_off;
    const u32 *gbl = gbase + gb_off[ci];
    const u32 *tbs = tile_base + tb_off[ci];
    u8 *out = stream + ch.stream_off;
    const u32 ntok = r.ntok, ngroups = (ntok + 63) / 64, nout = r.n_out;
    if (ntok == 0) return;                                           // nothing to resolve (k_inf_finish checks the size)
    for (u32 i = threadIdx.x * 16; i < LZ_CTL_OFF; i += LZ_THREADS * 16) lds_zero16(lds_data + i);     // bytes and bits start at zero
    if (threadIdx.x < 24) lds_st(lds_prog + 4 * threadIdx.x, (threadIdx.x >= 16 && threadIdx.x < 16 + LZ_FLUSHERS) ? (threadIdx.x - 16) * LZ_FLUSH : 0);
    __syncthreads();
    // the next group's token and its output range are fetched one group ahead; the loaded words are only
    // combined when the group starts, so the loads stay in flight while the current group is resolved
    u32 t_n = 0, tb0_n = 0, gl0_n = 0, tb1_n = 0, gl1_n = 0;
    const int LZW = n_workers;          // decoding waves actually used (<= LZ_WORKERS)
    auto prefetch = [&](u32 h) {
        const u32 h0 = h < ngroups ? h : ngroups - 1, h1 = h + 1 < ngroups ? h + 1 : ngroups - 1;
        const u32 i2 = h0 * 64 + lane;
        t_n = tk[i2 < ntok ? i2 : ntok - 1];
        tb0_n = tbs[h0 / GS_TILE]; gl0_n = gbl[h0]; tb1_n = tbs[h1 / GS_TILE]; gl1_n = gbl[h1];
    };
    // every worker publishes the base of its first group before anyone looks at the progress words
    if (wave < LZW) {
        prefetch(wave);
        if (lane == 0) lds_st(lds_prog + 4 * wave, (u32)wave < ngroups ? tb0_n + gl0_n : 0xffffffffu);
    }
    __syncthreads();
    // bytes below this are final: the smallest start offset among the groups the workers are still on
    // (each worker publishes the base of its current group in prog[w]; 0xffffffff when it has no more)
    auto safe_bytes = [&]() -> u32 {
        const u32 v = row0_min_dpp(lane < LZW ? lds_ld(lds_prog + 4 * lane) : 0xffffffffu);
        return v < nout ? v : nout;
    };
    // every granule below this has been streamed out, and the ring slot LZ_ZLAG granules behind it zeroed
    auto flushed = [&]() -> u32 {
        u32 v = lds_ld(lds_flnext);
#pragma unroll
        for (int k = 1; k < LZ_FLUSHERS; k++) v = min(v, lds_ld(lds_flnext + 4 * k));
        return v;
    };
    if (wave >= LZ_WORKERS) {
        // ---- flushers: granule q belongs to flusher q % LZ_FLUSHERS; 16 bytes per lane and step ----
        const u32 me = wave - LZ_WORKERS;
        u32 idle = 0;
        for (u32 q = me;; q += LZ_FLUSHERS) {
            const u32 lo = q * LZ_FLUSH;
            if (lo >= nout) break;
            const u32 hi = lo + LZ_FLUSH < nout ? lo + LZ_FLUSH : nout;
            while (safe_bytes() < hi) {
                if (++idle > LZ_SPIN_MAX || lds_ld(lds_bad)) { if (lane == 0) { if (!lds_ld(lds_bad)) lds_st(lds_bad, 3); for (int k = 0; k < LZ_FLUSHERS; k++) lds_st(lds_flnext + 4 * k, 0xffffffffu); } return; }
                __builtin_amdgcn_s_sleep(2);
            }
            idle = 0;
            for (u32 o = lo + lane * 16; o < lo + LZ_FLUSH; o += 1024) {
                if (o < hi) {
                    u32x4 c;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(lds_data + (o & (LZ_RING - 1))) : "memory");
                    if (o + 16 <= hi) *(u32x4 *)(out + o) = c;
                    else { const u32 cw[4] = {c.x, c.y, c.z, c.w}; for (u32 k = 0; o + k < hi; k++) out[o + k] = (u8)(cw[k >> 2] >> (8 * (k & 3))); }
                }
            }
            // every running group starts at or above `hi`, so nothing reaches the granule LZ_ZLAG back any more:
            // clear its slot (bytes and bits) for the lap that comes next
            if (q >= LZ_ZLAG) {
                const u32 zo = ((q - LZ_ZLAG) * LZ_FLUSH) & (LZ_RING - 1);
#pragma unroll
                for (u32 o = lane * 16; o < LZ_FLUSH; o += 1024) lds_zero16(lds_data + zo + o);
                if (lane < (int)(LZ_FLUSH / 8 / 16)) lds_zero16(lds_bits + zo / 8 + lane * 16);
            }
            if (lane == 0) lds_st(lds_flnext + 4 * me, lo + LZ_FLUSHERS * LZ_FLUSH);
        }
        if (lane == 0) lds_st(lds_flnext + 4 * me, 0xffffffffu);
        return;
    }
    // ---- workers ----
    if (wave >= LZW) return;
    u64 pc_vm = 0, pc_sleeps = 0, pc_wait = 0, pc_pre = 0, pc_loop = 0, pc_iter = 0, pc_groups = 0, pc_t0 = prof ? __builtin_readcyclecounter() : 0;
    u32 fl_seen = 0;                     // last value read of flushed(): it only grows
    for (u32 g = wave; g < ngroups; g += LZW) {
        if (lds_ld(lds_bad)) break;                                 // the chunk is lost already (a copy from before the data, an expired wait)
        u64 c0_ = prof ? __builtin_readcyclecounter() : 0;
        const u32 t = t_n, base = tb0_n + gl0_n, next = g + 1 < ngroups ? tb1_n + gl1_n : nout;
        const u32 i = g * 64 + lane;
        const bool act = i < ntok;
        if (lane == 0) lds_st(lds_prog + 4 * wave, base);            // groups below `base` owned by this worker are done
        prefetch(g + LZW);
        u64 cv_ = 0;
        if (prof) { asm volatile("" :: "v"(next), "v"(t)); cv_ = __builtin_readcyclecounter(); pc_vm += cv_ - c0_; }
        // ring safety: this group writes up to `next`; only the slots below flushed() + LZ_AHEAD are zeroed
        if (next > LZ_AHEAD) {
            const u32 lim = next - LZ_AHEAD;
            for (u32 waits = 0; fl_seen < lim; waits++) {
                if (prof && waits) pc_sleeps++;
                fl_seen = flushed();
                if (fl_seen >= lim) break;
                if (waits > LZ_SPIN_MAX) { if (!lds_ld(lds_bad)) lds_st(lds_bad, 4); break; }                       // never hang
                __builtin_amdgcn_s_sleep(1);
            }
        }
        u64 c1_ = prof ? __builtin_readcyclecounter() : 0;
        const bool cp = act && (t >> 31);
        const u32 len = !act ? 0 : cp ? ((t >> 16) & 0xff) + 3 : 1;
        const u32 dst = base + wave_incl_scan_dpp(len) - len;
        const u32 dist = (t & 0x7fff) + 1;
        const u32 src = dst - dist;
        const u32 dofs = dst & (LZ_RING - 1), sofs = src & (LZ_RING - 1);
        u32 pv = cp ? 1u : 0u;                // the copy is pending (as a number: see k_inf_lz_seg)
        if (cp && (dist > dst || len > 8)) { pv = 0; lds_st(lds_bad, 1); }      // distance too far back (or a piece this kernel never makes): corrupt
        if (act && !cp) lz_or_byte(lds_data + (dofs & ~3u), (t & 0xff) << (8 * (dofs & 3)), lds_bits + ((dofs >> 5) << 2), 1u << (dofs & 31));
        // a piece whose source or destination window would run past the ring end goes byte by byte
        const bool slow = sofs > LZ_EDGE || dofs > LZ_EDGE;
        const u64 slowm = ballot64(slow);
        const u32 need = len < dist ? len : dist;                       // source bytes that are not this piece's own output
        const u32 nbits = (1u << need) - 1;
        const u32 ba = lds_bits + ((sofs >> 5) << 2), da = lds_data + (sofs & ~3u);
        const u32 wa = lds_data + (dofs & ~3u), wba = lds_bits + ((dofs >> 5) << 2);
        const u64 wbits = (u64)((1u << len) - 1) << (dofs & 31);
        const u64 lmask = len >= 8 ? ~0ull : (1ull << (8 * len)) - 1;
        const u32 wsh = 8 * (dofs & 3);
        u32 idle = 0, k = 0, spins = 0;       // idle: rounds in a row in which no lane of the wave moved (wave-uniform)
        u64 c2_ = prof ? __builtin_readcyclecounter() : 0;
        u64 pm = ballot64(pv != 0);
        while (pm) {
            if (prof) pc_iter++;
            if (pv != 0 && !slow) {
                u32 b0, b1, x0, x1, x2;
                lz_load_window(ba, da, b0, b1, x0, x1, x2);
                if ((__builtin_amdgcn_alignbit(b1, b0, sofs & 31) & nbits) == nbits) {
                    u64 v = (u64)__builtin_amdgcn_alignbyte(x1, x0, sofs & 3) | ((u64)__builtin_amdgcn_alignbyte(x2, x1, sofs & 3) << 32);
                    if (dist < 8) {                                   // overlapping copy: the piece repeats its first `dist` bytes
                        v &= (1ull << (8 * dist)) - 1;
                        v |= v << (8 * dist);
                        if (dist < 4) v |= v << (16 * dist);
                        if (dist < 2) v |= v << 32;
                    }
                    v &= lmask;
                    const u64 w01 = v << wsh;
                    const u32 w2 = (u32)(((v >> 32) << wsh) >> 32);
                    lz_or_window(wa, (u32)w01, (u32)(w01 >> 32), w2, wba, (u32)wbits, (u32)(wbits >> 32));
                    pv = 0;
                }
            } else if (pv != 0) {
                // byte-wise path (window across the ring end): one byte per round
                const u32 so = (src + k) & (LZ_RING - 1), dd = (dst + k) & (LZ_RING - 1);
                u32 bw, dv;
                lz_load_byte(lds_bits + ((so >> 5) << 2), lds_data + so, bw, dv);
                if ((bw >> (so & 31)) & 1) {
                    lz_or_byte(lds_data + (dd & ~3u), dv << (8 * (dd & 3)), lds_bits + ((dd >> 5) << 2), 1u << (dd & 31));
                    if (++k == len) pv = 0;
                    spins = 0;
                } else if ((++spins & 63u) == 0) {                    // (its own bound and look at the lost-chunk flag, as in k_inf_lz_seg)
                    if (lds_ld(lds_bad)) pv = 0;
                    else if (spins > (1u << 22)) { pv = 0; lds_st(lds_bad, 2); }
                }
            }
            const u64 pn = ballot64(pv != 0);
            // nothing of this wave's could commit: its sources are another wave's work.  Step back for a moment -- sixteen
            // waves polling the LDS at full rate leave the one wave that can make progress a sixteenth of it (a chain of
            // dependent copies, e.g. 7-byte matches at distance 8 through int64 data, then runs 100x slower than on one wave)
            // What it waits for may never come (damage): after 2^22 idle rounds, ~1 s, the wave gives up -- never hang the GPU.
            u64 nx = pn;
            if (pn != pm || (pn & slowm)) idle = 0;                      // somebody is done (or goes byte by byte)
            else if (__builtin_amdgcn_readfirstlane((int)lds_ld(lds_bad))) nx = 0;
            else if (++idle > (1u << 22)) { nx = 0; lds_st(lds_bad, 2); }
            else __builtin_amdgcn_s_sleep(4);
            pm = nx;
        }
        if (prof) { const u64 c3_ = __builtin_readcyclecounter(); pc_wait += c1_ - c0_; pc_pre += c2_ - c1_; pc_loop += c3_ - c2_; pc_groups++; }
    }
    if (prof && lane == 0) {
        u64 *q = prof + ((u64)ci * 16 + wave) * 8;
        q[0] = __builtin_readcyclecounter() - pc_t0; q[1] = pc_wait; q[2] = pc_pre; q[3] = pc_loop; q[4] = pc_iter; q[5] = pc_groups; q[6] = pc_vm; q[7] = pc_sleeps;
    }
    // no more groups for this worker
    if (lane == 0) lds_st(lds_prog + 4 * wave, 0xffffffffu);
    if (wave == 0) {
        // wait for the flushers before reporting (same workgroup: they are resident)
        for (u32 waits = 0; flushed() < nout && waits < 4 * LZ_SPIN_MAX; waits++) __builtin_amdgcn_s_sleep(8);
        // a copy that reaches before the data is corruption; an expired wait is not a verdict: the chunk is run once more,
        // with a single worker wave (nothing to wait for but the flushers), and only then given up
        if (lane == 0 && (lds_ld(lds_bad) || flushed() < nout)) res[ci].status = (pass || lds_ld(lds_bad) == 1) ? MTS_CHUNK_CORRUPT : LZ_STATUS_RETRY;
    }
}

// ---- segment resolver (16-bit cells) -------------------------------------------------------------
// window of 8 cells (16 bytes at a 2-byte boundary) = five aligned dwords.  A cell holds its value + 1: zero is "not written
// yet" (the ring is zeroed a lap ahead, as OR-ing into it needs anyway), so that the cells are their own "written" marks -- no
// bit per position to read, set and clear as the byte resolver keeps: a piece is 3 + 5 LDS instructions instead of 4 + 7, in
// a kernel that is bound by what the CU's LDS takes.  The flushers take the 1 off again.
__device__ __forceinline__ void lz2_load_window(u32 data_addr, u32 &x0, u32 &x1, u32 &x2, u32 &x3, u32 &x4)
{
    u32x2 p, q;
    asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %3 offset0:2 offset1:3\n\t"
                 "ds_read_b32 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(p), "=&v"(q), "=&v"(x4) : "v"(data_addr) : "memory");
    x0 = p.x; x1 = p.y; x2 = q.x; x3 = q.y;
}
__device__ __forceinline__ void lz2_or_window(u32 data_addr, u32 o0, u32 o1, u32 o2, u32 o3, u32 o4)
{
    asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %0, %2 offset:4\n\tds_or_b32 %0, %3 offset:8\n\tds_or_b32 %0, %4 offset:12\n\t"
                 "ds_or_b32 %0, %5 offset:16"
                 :: "v"(data_addr), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "v"(o4) : "memory");
}
__device__ __forceinline__ void lz2_or_cell(u32 data_addr, u32 w) { asm volatile("ds_or_b32 %0, %1" :: "v"(data_addr), "v"(w) : "memory"); }
__device__ __forceinline__ u32 lz2_load_cell(u32 data_addr)
{
    u32 data;
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(data) : "v"(data_addr) : "memory");
    return data;
}
// some 16-bit half of y is zero
__device__ __forceinline__ u32 lz2_zero_half(u32 y) { return (y - 0x00010001u) & ~y & 0x80008000u; }
// the smaller of the two low halves, and of the two high halves
__device__ __forceinline__ u32 lz2_pk_min(u32 a, u32 b) { u32 d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// how a chunk is cut (one thread per chunk)
__global__ __launch_bounds__(64) void k_inf_plan(const InfResult *__restrict__ res, const u64 *__restrict__ gb_off,
                                                 const u64 *__restrict__ tb_off, const u32 *__restrict__ gbase,
                                                 const u32 *__restrict__ tile_base, int n_chunks, int nseg_req, LzPlan *__restrict__ plan)
{
    const int ci = blockIdx.x * 64 + threadIdx.x;
    if (ci >= n_chunks) return;
    const InfResult r = res[ci];
    LzPlan *p = plan + ci;
    p->nseg = 1;
    for (int k = 0; k <= LZ_MAXSEG; k++) { p->g0[k] = 0; p->b0[k] = 0; }
    if (r.status == MTS_CHUNK_OK && r.ntok) {
        const u32 ngroups = (r.ntok + 63) / 64;
        const u32 *gbl = gbase + gb_off[ci];
        const u32 *tbs = tile_base + tb_off[ci];
        auto gb = [&](u32 g) -> u32 { return g >= ngroups ? r.n_out : tbs[g / GS_TILE] + gbl[g]; };
        p->g0[1] = ngroups; p->b0[1] = r.n_out;
        // the largest number of segments (the requested one, halved until it fits) such that every segment but the last
        // produces a whole window and every segment has work
        for (int ns = nseg_req; ns > 1; ns >>= 1) {
            const u32 per = (ngroups + ns - 1) / ns;
            bool ok = per > 0;
            u32 prev_g = 0, prev_b = 0;
            for (int k = 1; k <= ns && ok; k++) {
                const u32 g = min((u32)k * per, ngroups), b = gb(g);
                if (g <= prev_g) ok = false;
                if (k < ns && b - prev_b < LZ_WIN + 512) ok = false;
                prev_g = g; prev_b = b;
            }
            if (!ok) continue;
            p->nseg = (u32)ns;
            for (int k = 0; k <= ns; k++) { const u32 g = min((u32)k * per, ngroups); p->g0[k] = g; p->b0[k] = gb(g); }
            break;
        }
    }
}
#ifndef MTS_LZ2_STATS
#define MTS_LZ2_STATS 0
#endif
#if MTS_LZ2_STATS
__device__ unsigned long long g_lz2_stats[16];     // (instrumented builds: tools/lz2_stats.py)
#define LZ2_CLK() __builtin_readcyclecounter()
#endif
__global__ __launch_bounds__(LZ_THREADS) void k_inf_lz_seg(const u32 *__restrict__ tokens, const InfChunk *__restrict__ chunks,
                                                           InfResult *__restrict__ res, const u64 *__restrict__ gb_off,
                                                           const u64 *__restrict__ tb_off, const u32 *__restrict__ gbase,
                                                           const u32 *__restrict__ tile_base, const LzPlan *__restrict__ plan,
                                                           u16 *__restrict__ sym, int n_workers, int pass, u8 *__restrict__ stream,
                                                           u8 *__restrict__ gflag, u64 *__restrict__ seg_adler)
{
    const int ci = blockIdx.y, seg = blockIdx.x;
    const InfResult r = res[ci];
    if (pass == 2) { if (threadIdx.x == 0 && seg == 0 && r.status == MTS_CHUNK_OK && plan[ci].nseg >= 2) res[ci].status = LZ_STATUS_RETRY; return; }   // (test hook)
    if (r.status != (pass ? LZ_STATUS_RETRY : MTS_CHUNK_OK)) return;
    const LzPlan *pl = plan + ci;
    if (pl->nseg < 2 || (u32)seg >= pl->nseg) return;
    const InfChunk ch = chunks[ci];
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lds_data = (u32)(uintptr_t)smem;                       // LDS byte offsets of the cell ring, ...
    const u32 lds_prog = lds_data + LZ2_CTL_OFF;                     // [w] = first byte of the group worker w is on
    const u32 lds_flnext = lds_prog + 64;                            // [k] = first byte of the next granule of flusher k
    const u32 lds_bad = lds_prog + 80;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;      // (the group numbers are scalars then)
    const u32 *tk = tokens + ch.tok_off;
    const u32 *gbl = gbase + gb_off[ci];
    const u32 *tbs = tile_base + tb_off[ci];
    u16 *out = sym + ch.stream_off;                                  // cells of this chunk (same offsets as the stream, in cells)
    const u32 ntok = r.ntok, ngroups = (ntok + 63) / 64;
    const u32 g_lo = pl->g0[seg], g_hi = pl->g0[seg + 1], B = pl->b0[seg], Bend = pl->b0[seg + 1];
    for (u32 i = threadIdx.x * 16; i < LZ2_CTL_OFF; i += LZ_THREADS * 16) lds_zero16(lds_data + i);     // cells and bits start at zero
    const u32 q0 = B / LZ_FLUSH;
    if (threadIdx.x < 24) {
        u32 v = 0;
        if (threadIdx.x >= 16 && threadIdx.x < 16 + LZ_FLUSHERS) { const u32 k = threadIdx.x - 16; v = (q0 + ((k - q0) & (LZ_FLUSHERS - 1))) * LZ_FLUSH; }
        lds_st(lds_prog + 4 * threadIdx.x, v);
    }
    __syncthreads();
    // the unknown window: value 256 + i at position B - LZ_WIN + i (segment 0 has none)
    if (seg > 0) {
        for (u32 i = threadIdx.x; i < LZ_WIN; i += LZ_THREADS) {
            const u32 ro = (B - LZ_WIN + i) & (LZ_RING - 1);
            lz2_or_cell(lds_data + ((2 * ro) & ~3u), (256 + i + 1) << (16 * (ro & 1)));
        }
    }
    u32 t_n = 0, tb0_n = 0, gl0_n = 0, tb1_n = 0, gl1_n = 0;
    const int LZW = n_workers;
    auto prefetch = [&](u32 h) {
        const u32 h0 = h < g_hi ? h : g_hi - 1, h1 = h + 1 < ngroups ? h + 1 : ngroups - 1;
        const u32 i2 = h0 * 64 + lane;
        t_n = tk[i2 < ntok ? i2 : ntok - 1];
        tb0_n = tbs[h0 / GS_TILE]; gl0_n = gbl[h0]; tb1_n = tbs[h1 / GS_TILE]; gl1_n = gbl[h1];
    };
    if (wave < LZW) {
        prefetch(g_lo + wave);
        if (lane == 0) lds_st(lds_prog + 4 * wave, g_lo + (u32)wave < g_hi ? tb0_n + gl0_n : 0xffffffffu);
    }
    __syncthreads();
    auto safe_bytes = [&]() -> u32 {
        const u32 v = row0_min_dpp(lane < LZW ? lds_ld(lds_prog + 4 * lane) : 0xffffffffu);
        return v < Bend ? v : Bend;
    };
    auto flushed = [&]() -> u32 {
        u32 v = lds_ld(lds_flnext);
#pragma unroll
        for (int k = 1; k < LZ_FLUSHERS; k++) v = min(v, lds_ld(lds_flnext + 4 * k));
        return v;
    };
    if (wave >= LZ_WORKERS) {
        // ---- flushers: granule q belongs to flusher q % LZ_FLUSHERS; 8 cells (16 bytes) per lane and step ----
        // A granule that lies inside the segment and holds no symbolic cell (two thirds of them on the recordings: what
        // derives from the unknown window thins out quickly) leaves as BYTES, straight into the stream, with its part of
        // the adler32 sums taken here; k_inf_translate only goes through the granules flagged in gflag.
        const u32 me = wave - LZ_WORKERS;
        u32 idle = 0;
        u8 *bytes_out = stream + ch.stream_off;
        u8 *gf = gflag + (ch.stream_off >> 12) + ci;
        const u64 nn = ch.n_expect;
        u64 sa = 0, sb = 0;
        for (u32 q = q0 + ((me - q0) & (LZ_FLUSHERS - 1));; q += LZ_FLUSHERS) {
            const u32 lo = q * LZ_FLUSH;
            if (lo >= Bend) break;
            const u32 hi = lo + LZ_FLUSH < Bend ? lo + LZ_FLUSH : Bend;
            while (safe_bytes() < hi) {
                if (++idle > LZ_SPIN_MAX || lds_ld(lds_bad)) { if (lane == 0) { if (!lds_ld(lds_bad)) lds_st(lds_bad, 3); for (int k = 0; k < LZ_FLUSHERS; k++) lds_st(lds_flnext + 4 * k, 0xffffffffu); } return; }
                __builtin_amdgcn_s_sleep(2);
            }
            idle = 0;
            const bool whole = lo >= B && hi == lo + LZ_FLUSH;
            u32x4 c[LZ_FLUSH / 512];
            u32 high = 0;
#pragma unroll
            for (u32 st = 0; st < LZ_FLUSH / 512; st++) {
                const u32 o = lo + st * 512 + lane * 8;
                c[st] = u32x4{0, 0, 0, 0};
                if (o < hi && o + 8 > B) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c[st]) : "v"(lds_data + 2 * (o & (LZ_RING - 1))) : "memory");
                // cells hold value + 1 (see lz2_load_window); every cell of [max(lo, B), hi) is written by now.  (Cells outside
                // that range -- first and last granule only -- are zero: the borrow spoils their neighbour, which the cell-by-cell
                // branch below therefore takes from the cell itself)
                if (whole || (o >= B && o + 8 <= hi)) { c[st].x -= 0x00010001u; c[st].y -= 0x00010001u; c[st].z -= 0x00010001u; c[st].w -= 0x00010001u; }
                high |= c[st].x | c[st].y | c[st].z | c[st].w;
            }
            if (whole && !any64((high & 0xff00ff00u) != 0)) {
#pragma unroll
                for (u32 st = 0; st < LZ_FLUSH / 512; st++) {
                    const u32 o = lo + st * 512 + lane * 8;
                    const u32 cw[4] = {c[st].x, c[st].y, c[st].z, c[st].w};
                    u32 d[2] = {0, 0}, s1 = 0, s2 = 0;
#pragma unroll
                    for (u32 k = 0; k < 8; k++) {
                        const u32 v = (cw[k >> 1] >> (16 * (k & 1))) & 0xff;
                        d[k >> 2] |= v << (8 * (k & 3));
                        s1 += v; s2 += k * v;
                    }
                    *(uint2 *)(bytes_out + o) = make_uint2(d[0], d[1]);
                    sa += s1; sb += (nn - o) * (u64)s1 - s2;
                }
            } else {
                if (lane == 0) gf[q] = 1;
#pragma unroll
                for (u32 st = 0; st < LZ_FLUSH / 512; st++) {
                    const u32 o = lo + st * 512 + lane * 8;
                    if (o < hi && o + 8 > B) {
                        if (o >= B && o + 8 <= hi) *(u32x4 *)(out + o) = c[st];
                        else { const u32 cw[4] = {c[st].x, c[st].y, c[st].z, c[st].w}; for (u32 k = 0; k < 8; k++) if (o + k >= B && o + k < hi) out[o + k] = (u16)((cw[k >> 1] >> (16 * (k & 1))) - 1u); }
                    }
                }
            }
            if (q >= LZ_ZLAG) {
                const u32 zo = ((q - LZ_ZLAG) * LZ_FLUSH) & (LZ_RING - 1);
#pragma unroll
                for (u32 o = lane * 16; o < 2 * LZ_FLUSH; o += 1024) lds_zero16(lds_data + 2 * zo + o);
            }
            if (lane == 0) lds_st(lds_flnext + 4 * me, lo + LZ_FLUSHERS * LZ_FLUSH);
        }
        // this flusher's share of the byte sums of the granules it wrote as bytes: a slot of its own, plainly stored, so that a
        // second run of the segment (the retry pass runs every segment of its chunks again) replaces it
        for (int d = 32; d; d >>= 1) {
            sa += ((u64)(u32)__shfl_xor((u32)(sa >> 32), d) << 32) | (u32)__shfl_xor((u32)sa, d);
            sb += ((u64)(u32)__shfl_xor((u32)(sb >> 32), d) << 32) | (u32)__shfl_xor((u32)sb, d);
        }
        if (lane == 0) {
            u64 *slot = seg_adler + (((size_t)ci * LZ_MAXSEG + seg) * LZ_FLUSHERS + me) * 2;
            slot[0] = sa % 65521u; slot[1] = sb % 65521u;
            lds_st(lds_flnext + 4 * me, 0xffffffffu);
        }
        return;
    }
    // ---- workers ----
    if (wave >= LZW) return;
    u32 fl_seen = 0;
#if MTS_LZ2_STATS
    u64 sg = 0, sit = 0, snm = 0, c_wait = 0, c_pre = 0, c_loop = 0, spend = 0, sslow = 0, sfirst = 0, scp = 0, c_ld = 0;
    const u64 c_start = LZ2_CLK();
#endif
    for (u32 g = g_lo + wave; g < g_hi; g += LZW) {          // (a lost chunk is noticed by whoever waits: no look at lds_bad per group)
#if MTS_LZ2_STATS
        const u64 c0_ = LZ2_CLK();
#endif
        const u32 t = t_n, base = tb0_n + gl0_n, next = g + 1 < ngroups ? tb1_n + gl1_n : r.n_out;
        const u32 i = g * 64 + lane;
        const bool act = i < ntok;
        if (lane == 0) lds_st(lds_prog + 4 * wave, base);
        prefetch(g + LZW);
        if (next > LZ_AHEAD) {
            const u32 lim = next - LZ_AHEAD;
            for (u32 waits = 0; fl_seen < lim; waits++) {
                fl_seen = flushed();
                if (fl_seen >= lim) break;
                if (waits > LZ_SPIN_MAX) { if (!lds_ld(lds_bad)) lds_st(lds_bad, 4); break; }                       // never hang
                __builtin_amdgcn_s_sleep(1);
            }
        }
#if MTS_LZ2_STATS
        const u64 c1_ = LZ2_CLK();
#endif
        const bool cp = act && (t >> 31);
        const u32 len = !act ? 0 : cp ? ((t >> 16) & 0xff) + 3 : 1;
        const u32 dst = base + wave_incl_scan_dpp(len) - len;
        const u32 dist = (t & 0x7fff) + 1;
        const u32 src = dst - dist;
        const u32 dofs = dst & (LZ_RING - 1), sofs = src & (LZ_RING - 1);
        // (the copy's state as a number: a ballot of `pv != 0` is one compare, the ballot of a flag kept in a mask is two
        //  instructions more -- and this loop is made of instructions)
        u32 pv = cp ? 1u : 0u;
        if (cp && (dist > dst || len > 8)) { pv = 0; lds_st(lds_bad, 1); }
        if (act && !cp) lz2_or_cell(lds_data + ((2 * dofs) & ~3u), ((t & 0xff) + 1u) << (16 * (dofs & 1)));
        // cell by cell: windows that would run past the ring end, and overlapping copies (dist < len)
        const bool slow = sofs > LZ_EDGE || dofs > LZ_EDGE || dist < len;
        const u32 da = lds_data + ((2 * sofs) & ~3u), wa = lds_data + ((2 * dofs) & ~3u);
        // cells at or beyond len are not this piece's: dropped from what is written, and taken as written in the test.
        // n0 .. n3: ones in the cells that are NOT the piece's (cell c at bits 16 c of the 128)
        const u64 nlo = len >= 4 ? 0ull : ~0ull << (16 * len), nhi = len >= 8 ? 0ull : ~0ull << (16 * max(len, 4u) - 64);
        const u32 n0 = (u32)nlo, n1 = (u32)(nlo >> 32), n2 = (u32)nhi, n3 = (u32)(nhi >> 32);
        // a destination at an odd cell takes the eight cells a half dword up: one byte permute per dword written
        const u32 sel = (dofs & 1) ? 0x05040302u : 0x07060504u;
        u32 idle = 0, k = 0, spins = 0;       // idle: rounds in a row in which no lane of the wave moved (wave-uniform)
        const u64 slowm = ballot64(slow);
#if MTS_LZ2_STATS
        const u64 c2_ = LZ2_CLK();
        sg++; scp += __popcll(__ballot(cp)); sslow += __popcll(__ballot(slow && pv));
        bool first_ = true;
#endif
        u64 pm = ballot64(pv != 0);
        while (pm) {
#if MTS_LZ2_STATS
            sit++; spend += __popcll(pm);
#endif
            if (pv != 0 && !slow) {
                u32 x0, x1, x2, x3, x4;
#if MTS_LZ2_STATS
                const u64 cl0_ = LZ2_CLK();
#endif
                lz2_load_window(da, x0, x1, x2, x3, x4);
#if MTS_LZ2_STATS
                c_ld += LZ2_CLK() - cl0_;
#endif
                const u32 ssh = 16 * (sofs & 1);
                // the 8 cells from the source on, two per dword
                const u32 r0 = __builtin_amdgcn_alignbit(x1, x0, ssh), r1 = __builtin_amdgcn_alignbit(x2, x1, ssh);
                const u32 r2 = __builtin_amdgcn_alignbit(x3, x2, ssh), r3 = __builtin_amdgcn_alignbit(x4, x3, ssh);
                // all written: the smallest of the piece's cells is not zero (half by half over the four dwords)
                const u32 y0 = r0 | n0, y1 = r1 | n1, y2 = r2 | n2, y3 = r3 | n3;
                if (lz2_zero_half(lz2_pk_min(lz2_pk_min(y0, y1), lz2_pk_min(y2, y3))) == 0) {
                    const u32 w0 = y0 ^ n0, w1 = y1 ^ n1, w2 = y2 ^ n2, w3 = y3 ^ n3;      // (= r & ~n, without the ~n)
                    lz2_or_window(wa, __builtin_amdgcn_perm(w0, 0u, sel), __builtin_amdgcn_perm(w1, w0, sel), __builtin_amdgcn_perm(w2, w1, sel),
                                  __builtin_amdgcn_perm(w3, w2, sel), __builtin_amdgcn_perm(0u, w3, sel));
                    pv = 0;
                }
            } else if (pv != 0) {
                const u32 so = (src + k) & (LZ_RING - 1), dd = (dst + k) & (LZ_RING - 1);
                const u32 dv = lz2_load_cell(lds_data + 2 * so);
                if (dv) {
                    lz2_or_cell(lds_data + ((2 * dd) & ~3u), dv << (16 * (dd & 1)));
                    if (++k == len) pv = 0;
                    spins = 0;
                } else if ((++spins & 63u) == 0) {                    // (its own bound and its own look at the lost-chunk flag: a round
                    if (lds_ld(lds_bad)) pv = 0;                      //  with a cell-by-cell copy in it does not count as idle below)
                    else if (spins > (1u << 22)) { pv = 0; lds_st(lds_bad, 2); }
                }
            }
            const u64 pn = ballot64(pv != 0);
#if MTS_LZ2_STATS
            if (first_) { sfirst += __popcll(pm & ~pn); first_ = false; }
            if (pn == pm) snm++;
#endif
            // (what a copy waits for may never come -- damage: the wave gives up after 2^22 idle rounds, ~1 s.  The sleep is
            //  k_inf_lz's; 1, 2 or 8 here: the same 3.4 ms)
            u64 nx = pn;
            if (pn != pm || (pn & slowm)) idle = 0;                      // somebody is done (or goes cell by cell)
            else if (__builtin_amdgcn_readfirstlane((int)lds_ld(lds_bad))) nx = 0;            // (a scalar: the loop's condition stays one)
            else if (++idle > (1u << 22)) { nx = 0; lds_st(lds_bad, 2); }
            else __builtin_amdgcn_s_sleep(4);
            pm = nx;
        }
#if MTS_LZ2_STATS
        { const u64 c3_ = LZ2_CLK(); c_wait += c1_ - c0_; c_pre += c2_ - c1_; c_loop += c3_ - c2_; }
#endif
    }
#if MTS_LZ2_STATS
    if (lane == 0 && pass == 0) {
        const u64 v[13] = {sg, sit, snm, LZ2_CLK() - c_start, c_wait, c_pre, c_loop, spend, sslow, sfirst, scp, 1, c_ld};
        for (int k = 0; k < 13; k++) atomicAdd(&g_lz2_stats[k], v[k]);
    }
#endif
    if (lane == 0) lds_st(lds_prog + 4 * wave, 0xffffffffu);
    if (wave == 0) {
        for (u32 waits = 0; flushed() < Bend && waits < 4 * LZ_SPIN_MAX; waits++) __builtin_amdgcn_s_sleep(8);
        if (lane == 0 && (lds_ld(lds_bad) || flushed() < Bend)) res[ci].status = (pass || lds_ld(lds_bad) == 1) ? MTS_CHUNK_CORRUPT : LZ_STATUS_RETRY;
    }
}

// the real window of every segment, one after the other (one workgroup per chunk)
// after the retry pass: chunks still marked for a retry were resolved by it
__global__ __launch_bounds__(64) void k_inf_lz_settle(InfResult *__restrict__ res, int n_chunks)
{
    const int ci = blockIdx.x * 64 + threadIdx.x;
    if (ci < n_chunks && res[ci].status == LZ_STATUS_RETRY) res[ci].status = MTS_CHUNK_OK;
}

// The windows of a chunk's segments, one after the other: window k = the last 32 KiB of segment k - 1, its cells translated with
// window k - 1.  A step depends on the step before through those look-ups alone -- where its cells come from does not -- so the
// cells of step k + 1 are asked for before step k is translated, and the window a step translates with is the one the step
// before left in LDS.  A thread takes 32 CONSECUTIVE positions at a 32-byte boundary of the stream: 64 bytes of cells and 32
// bytes of the stream in six 16-byte loads, 32 bytes of window in two 16-byte stores; a window is therefore kept ROTATED, the
// byte of stream position q at q mod 32 KiB (k_inf_translate looks it up the same way), and the 32-byte block that holds the
// window's two ends is put together byte by byte, by thread 0 (its lower end) and the threads that take one extra position of
// the upper end each.
// (Until round 5 a thread took its 32 cells one by one, each three dependent trips to memory -- granule flag, cell, byte of the
// previous window, which lived in memory behind a __threadfence --: 24 us per step, i.e. 0.75 ms for the 32 segments a chunk is
// cut into when few chunks are decoded: a quarter of a cold read.)
__global__ __launch_bounds__(1024) void k_inf_windows(const InfChunk *__restrict__ chunks, const InfResult *__restrict__ res,
                                                      const LzPlan *__restrict__ plan, const u16 *__restrict__ sym, u8 *__restrict__ win,
                                                      const u8 *__restrict__ stream, const u8 *__restrict__ gflag)
{
    __shared__ __attribute__((aligned(16))) u8 wl[2][LZ_WIN];      // the window made by the step before, the one being made (both rotated)
    const int ci = blockIdx.x;
    if (res[ci].status != MTS_CHUNK_OK) return;
    const LzPlan *pl = plan + ci;
    const u32 nseg = pl->nseg;
    if (nseg < 2) return;
    const u16 *cells = sym + chunks[ci].stream_off;
    const u8 *bytes = stream + chunks[ci].stream_off;                 // (granules without a flag left the resolver as bytes)
    const u8 *gf = gflag + (chunks[ci].stream_off >> 12) + ci;
    u8 *W = win + (size_t)ci * LZ_MAXSEG * LZ_WIN;                  // W[k] = window of segment k, rotated
    const u32 t = threadIdx.x;
    struct Blk { uint4 c[4]; uint4 b[2]; u32 f; u32 xc; };             // a thread's 32 positions: cells, bytes, granule flag; its extra position's cell or byte
    // what step k translates.  Thread t: positions (p0 / 32 + t) * 32 ... + 31 -- thread 0's block starts p0 % 32 positions before
    // the window, and as many positions of the window's upper end lie in a block of their own: thread j < p0 % 32 takes one.
    auto fetch = [&](u32 k, Blk &x) {
        const u32 p0 = pl->b0[k] - LZ_WIN;                           // >= b0[k - 1]: segments are at least a window long
        const u32 q = ((p0 >> 5) + t) << 5;
        x.f = gf[q / LZ_FLUSH];                                      // (a block lies in one granule)
#pragma unroll
        for (int j = 0; j < 4; j++) x.c[j] = *(const uint4 *)(cells + q + 8 * j);
        x.b[0] = *(const uint4 *)(bytes + q); x.b[1] = *(const uint4 *)(bytes + q + 16);
        x.xc = 0;
        if (t < (p0 & 31)) { const u32 qx = (p0 & ~31u) + LZ_WIN + t; x.xc = gf[qx / LZ_FLUSH] ? cells[qx] : bytes[qx]; }
    };
    Blk cur, nxt;
    fetch(1, cur);
    for (u32 k = 1; k < nseg; k++) {
        if (k + 1 < nseg) fetch(k + 1, nxt);
        const u32 p0 = pl->b0[k] - LZ_WIN, rot = pl->b0[k - 1];     // window k - 1 holds position q at q % 32 Ki; its byte i is position b0[k - 1] - 32 Ki + i
        const u8 *prev = wl[(k - 1) & 1];                           // (step 1 never looks: segment 0 has no cells that point before it)
        u8 *mine = wl[k & 1];
        auto tr = [&](u32 c) -> u32 { return c < 256 ? c : (u32)prev[(rot + c - 256) & (LZ_WIN - 1)]; };
        const u32 cw[16] = {cur.c[0].x, cur.c[0].y, cur.c[0].z, cur.c[0].w, cur.c[1].x, cur.c[1].y, cur.c[1].z, cur.c[1].w,
                            cur.c[2].x, cur.c[2].y, cur.c[2].z, cur.c[2].w, cur.c[3].x, cur.c[3].y, cur.c[3].z, cur.c[3].w};
        const u32 bw[8] = {cur.b[0].x, cur.b[0].y, cur.b[0].z, cur.b[0].w, cur.b[1].x, cur.b[1].y, cur.b[1].z, cur.b[1].w};
        u32 ow[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u32 v = bw[j];
            if (cur.f) {
                const u32 a = cw[2 * j], b2 = cw[2 * j + 1];
                v = tr(a & 0xffff) | (tr(a >> 16) << 8) | (tr(b2 & 0xffff) << 16) | (tr(b2 >> 16) << 24);
            }
            ow[j] = v;
        }
        const u32 slot = (((p0 >> 5) + t) << 5) & (LZ_WIN - 1);
        const u32 lowcut = p0 & 31;                                  // positions of thread 0's block below the window
        if (t == 0 && lowcut) {
            for (u32 j = lowcut; j < 32; j++) { const u8 v = (u8)(ow[j >> 2] >> (8 * (j & 3))); mine[slot + j] = v; W[(size_t)k * LZ_WIN + slot + j] = v; }
        } else {
            *(uint4 *)(mine + slot) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            *(uint4 *)(mine + slot + 16) = make_uint4(ow[4], ow[5], ow[6], ow[7]);
            *(uint4 *)(W + (size_t)k * LZ_WIN + slot) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            *(uint4 *)(W + (size_t)k * LZ_WIN + slot + 16) = make_uint4(ow[4], ow[5], ow[6], ow[7]);
        }
        if (t < lowcut) {                                            // one position of the window's upper end: the same block as thread 0's, its lower bytes
            const u8 v = (u8)tr(cur.xc);
            const u32 sx = ((p0 & ~31u) & (LZ_WIN - 1)) + t;
            mine[sx] = v; W[(size_t)k * LZ_WIN + sx] = v;
        }
        __syncthreads();
        cur = nxt;
    }
}

// cells -> bytes: 16 cells per lane
// (also sums the bytes for the adler32 check -- A = sum b_i, B = sum (n - i) b_i, reduced per workgroup -- while they are in
// registers: the separate pass over the finished stream is skipped for the chunks that come through here)
__global__ __launch_bounds__(256) void k_inf_translate(const InfChunk *__restrict__ chunks, const InfResult *__restrict__ res,
                                                       const LzPlan *__restrict__ plan, const u16 *__restrict__ sym,
                                                       const u8 *__restrict__ win, u8 *__restrict__ stream, u64 *__restrict__ adler_acc,
                                                       const u8 *__restrict__ gflag)
{
    const int ci = blockIdx.y;
    const InfResult r = res[ci];
    if (r.status != MTS_CHUNK_OK) return;
    const LzPlan *pl = plan + ci;
    const u32 nseg = pl->nseg;
    if (nseg < 2) return;
    constexpr u32 TR_STEPS = 4;                                        // 4 x 4 KiB per workgroup: a quarter of the atomics on the chunk's two sums
    if (blockIdx.x * 256u * 16u * TR_STEPS >= r.n_out) return;
    __shared__ u64 red[8];
    u64 sa = 0, sb = 0;
    const u64 nn = chunks[ci].n_expect;                                // (= r.n_out for a chunk whose status is OK)
    auto finish = [&]() {
        for (int d = 32; d; d >>= 1) {                                  // (32-bit halves: the sums do not fit one shuffle)
            sa += ((u64)(u32)__shfl_xor((u32)(sa >> 32), d) << 32) | (u32)__shfl_xor((u32)sa, d);
            sb += ((u64)(u32)__shfl_xor((u32)(sb >> 32), d) << 32) | (u32)__shfl_xor((u32)sb, d);
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = sa; red[4 + (threadIdx.x >> 6)] = sb; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const u64 a = red[0] + red[1] + red[2] + red[3], b = red[4] + red[5] + red[6] + red[7];
            atomicAdd((unsigned long long *)&adler_acc[2 * ci], (unsigned long long)(a % 65521u));
            atomicAdd((unsigned long long *)&adler_acc[2 * ci + 1], (unsigned long long)(b % 65521u));
        }
    };
    // (every thread of the workgroup reaches finish() together: its shuffles and its barrier want whole waves)
    const u16 *cells = sym + chunks[ci].stream_off;
    u8 *out = stream + chunks[ci].stream_off;
    const u8 *W = win + (size_t)ci * LZ_MAXSEG * LZ_WIN;
    // (the segment starts come to LDS once: looked up in memory, every step of the searches below was a round trip)
    __shared__ u32 b0s[LZ_MAXSEG + 1];
    if (threadIdx.x <= nseg) b0s[threadIdx.x] = pl->b0[threadIdx.x];
    __syncthreads();
    auto seg_of = [&](u32 q) -> u32 {                               // last k with b0[k] <= q
        u32 lo = 0, hi = nseg - 1;
        while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (q >= b0s[mid]) lo = mid; else hi = mid - 1; }
        return lo;
    };
    // a step is one granule of the resolver's flushers (256 threads x 16 cells = LZ_FLUSH): only the flagged ones are cells
    static_assert(256 * 16 == LZ_FLUSH, "a translation step is a flush granule");
    const u8 *gf = gflag + (chunks[ci].stream_off >> 12) + ci;
    bool todo[TR_STEPS];
    // the cells of all steps are asked for together (the kernel waited for one pair of loads per step: 76 % of its wave cycles)
    uint4 ca[TR_STEPS], cb[TR_STEPS];
#pragma unroll
    for (u32 step = 0; step < TR_STEPS; step++) {
        const u32 p = ((blockIdx.x * TR_STEPS + step) * 256 + threadIdx.x) * 16;
        todo[step] = (blockIdx.x * TR_STEPS + step) * (u32)LZ_FLUSH < r.n_out && gf[blockIdx.x * TR_STEPS + step] != 0;
        if (todo[step] && p + 16 <= r.n_out) { ca[step] = *(const uint4 *)(cells + p); cb[step] = *(const uint4 *)(cells + p + 8); }
        else { ca[step] = make_uint4(0, 0, 0, 0); cb[step] = ca[step]; }
    }
    {
        bool any = false;
#pragma unroll
        for (u32 step = 0; step < TR_STEPS; step++) any |= todo[step];
        if (!any) return;                                               // (the same for every thread of the workgroup)
    }
#pragma unroll
    for (u32 step = 0; step < TR_STEPS; step++) {
    const u32 p = ((blockIdx.x * TR_STEPS + step) * 256 + threadIdx.x) * 16;
    if (!todo[step] || p >= r.n_out) {
    } else if (p + 16 <= r.n_out) {
        const uint4 a = ca[step], b = cb[step];
        const u32 cw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const u32 k_lo = seg_of(p), k_hi = seg_of(p + 15);
        u32 ow[4] = {0, 0, 0, 0}, s1 = 0, s2 = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const u32 c = (cw[j >> 1] >> (16 * (j & 1))) & 0xffff;
            u32 v = c;
            if (c >= 256) { const u32 k = (k_lo == k_hi) ? k_lo : seg_of(p + j); v = W[(size_t)k * LZ_WIN + ((b0s[k] + c - 256) & (LZ_WIN - 1))]; }      // (the windows are kept rotated: k_inf_windows)
            ow[j >> 2] |= (v & 0xff) << (8 * (j & 3));
            s1 += v & 0xff; s2 += (u32)j * (v & 0xff);
        }
        *(uint4 *)(out + p) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        sa += s1; sb += (nn - p) * (u64)s1 - s2;                        // sum (n - p - j) b_j with 32-bit sums inside the loop
    } else {
        for (u32 q = p; q < r.n_out; q++) {
            const u32 c = cells[q];
            const u32 kq = seg_of(q);
            const u8 v = (u8)(c < 256 ? c : W[(size_t)kq * LZ_WIN + ((b0s[kq] + c - 256) & (LZ_WIN - 1))]);
            out[q] = v;
            sa += v; sb += (nn - q) * (u64)v;
        }
    }
    }
    finish();
}

__global__ __launch_bounds__(64) void k_inf_finish(const InfChunk *__restrict__ chunks, InfResult *__restrict__ res,
                                                   const u64 *__restrict__ adler_acc, int n_chunks, int *__restrict__ status_out,
                                                   const u64 *__restrict__ seg_adler)
{
    const int ci = blockIdx.x * 64 + threadIdx.x;
    if (ci >= n_chunks) return;
    InfResult r = res[ci];
    if (r.status == MTS_CHUNK_OK && chunks[ci].n_need == 0) {                 // (a partial decode never reaches the check value)
        u64 fa = 0, fb = 0;                                                   // what the resolver's flushers wrote as bytes (zero for chunks that were not cut)
        if (seg_adler) for (int k = 0; k < LZ_MAXSEG * LZ_FLUSHERS; k++) { fa += seg_adler[((size_t)ci * LZ_MAXSEG * LZ_FLUSHERS + k) * 2]; fb += seg_adler[((size_t)ci * LZ_MAXSEG * LZ_FLUSHERS + k) * 2 + 1]; }
        const u32 a = (u32)((1 + adler_acc[2 * ci] + fa) % 65521u), b = (u32)((chunks[ci].n_expect + adler_acc[2 * ci + 1] + fb) % 65521u);
        if (((b << 16) | a) != r.adler_stored) r.status = MTS_CHUNK_CORRUPT;
    }
    res[ci].status = r.status;
    status_out[ci] = r.status;
}

// ------------------------------------------------------------------------------------------------
// host side of the inflate pipeline
// ------------------------------------------------------------------------------------------------
// which chunk a candidate slot / a true-block slot belongs to
__global__ __launch_bounds__(256) void k_inf_fill_slots(const InfFast *__restrict__ fast, u32 *__restrict__ slot_chunk, u32 *__restrict__ tslot_chunk)
{
    const InfFast f = fast[blockIdx.x];
    for (u32 k = threadIdx.x; k < f.cand_cap; k += 256) slot_chunk[f.cand_off + k] = blockIdx.x;
    for (u32 k = threadIdx.x; k < f.true_cap; k += 256) tslot_chunk[f.true_off + k] = blockIdx.x;
}

static inline u32 cand_cap_of(u64 c_len) { return (u32)(c_len / 4096 + 64); }

// scratch layout (all 256-B aligned): [so u64 n][nn u32 n][fast n][cand_cnt n][true_cnt n][seq_flag n]
//   [slot_chunk total_cand][tslot_chunk total_true][cand_pos][cand_tmp][cres][tblk][subs]
struct InfLayout {
    size_t so, nn, fast, cand_cnt, true_cnt, seq_flag, slot_chunk, tslot_chunk, cand_pos, cand_tmp, cres, tblk, subs, gb_off, gbase, tb_off, tile_base, surv, surv_cnt, plan, win, sym, rows,
        round_row, row_ctr, gflag, seg_adler, end;
    u32 total_cand, total_true, surv_cap, rounds_cap;
    int nseg;                // segments the resolver cuts every chunk into (1: none)
};
static InfLayout inf_layout(int n_chunks, const u64 *c_lens, const u32 *n_expect)
{
    InfLayout l;
    u64 tc = 0;
    for (int i = 0; i < n_chunks; i++) tc += cand_cap_of(c_lens[i]);
    l.total_cand = (u32)tc;
    l.total_true = (u32)(2 * tc);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t r = o; o += align_up(bytes, 256); return r; };
    // (what the host fills -- so, nn, fast, gb_off, tb_off -- lies in a row: one copy)
    l.so = take(8 * (size_t)n_chunks); l.nn = take(4 * (size_t)n_chunks); l.fast = take(sizeof(InfFast) * (size_t)n_chunks);
    l.gb_off = take(8 * (size_t)n_chunks); l.tb_off = take(8 * (size_t)n_chunks);
    l.cand_cnt = take(4 * (size_t)n_chunks); l.true_cnt = take(4 * (size_t)n_chunks); l.seq_flag = take(4 * (size_t)n_chunks);
    l.slot_chunk = take(4 * (size_t)l.total_cand); l.tslot_chunk = take(4 * (size_t)l.total_true);
    l.cand_pos = take(8 * (size_t)l.total_cand); l.cand_tmp = take(8 * (size_t)l.total_cand);
    l.cres = take(sizeof(CandRes) * (size_t)l.total_cand); l.tblk = take(sizeof(TrueBlk) * (size_t)l.total_true);
    l.subs = take(sizeof(uint2) * (size_t)SUBCAP * l.total_cand);
    size_t ng = 0;
    for (int i = 0; i < n_chunks; i++) ng += ((size_t)n_expect[i] + 2 + 63) / 64 + 2;       // tokens <= bytes + 1
    l.gbase = take(4 * ng);
    l.tile_base = take(4 * (ng / GS_TILE + 2 * (size_t)n_chunks + 16));
    u64 cbits = 0;
    for (int i = 0; i < n_chunks; i++) cbits += 8 * c_lens[i];
    l.surv_cap = (u32)(cbits / 256 + 65536);          // ~0.1 % of the bit offsets pass the cheap filters
    l.surv = take(8 * (size_t)l.surv_cap);
    l.surv_cnt = take(256);
    // LZ resolver: with fewer chunks than CUs a chunk is cut into segments (see LzPlan); cells and windows live here
    l.nseg = 1;
    if (n_chunks < 128) { l.nseg = 256 / n_chunks; if (l.nseg > LZ_MAXSEG) l.nseg = LZ_MAXSEG; }
    if (const char *e = getenv("MTS_LZ_SEGS")) { l.nseg = atoi(e); if (l.nseg < 1) l.nseg = 1; if (l.nseg > LZ_MAXSEG) l.nseg = LZ_MAXSEG; }
    l.plan = take(sizeof(LzPlan) * (size_t)n_chunks);
    l.win = take(l.nseg > 1 ? (size_t)n_chunks * LZ_MAXSEG * LZ_WIN : 0);
    size_t cells = 0;
    for (int i = 0; i < n_chunks; i++) cells += align_up((u64)n_expect[i] + STREAM_PAD, STREAM_ALIGN);
    // pass A's token rows are dead once pass B has run, the resolver's cells are written after that: one region for both
    u64 rounds = 0;
    for (int i = 0; i < n_chunks; i++) rounds += rows_rounds_of(c_lens[i]);
    if (const char *e = getenv("MTS_INF_ROW_ROUNDS")) rounds = (u64)atoll(e);      // (tests: a pool that runs out)
    if (rounds > 0x3ffffffull) rounds = 0x3ffffffull;                               // (row ids are 32 bits)
    l.rounds_cap = (u32)rounds;
    const size_t row_bytes = (size_t)rounds * 64 * ROWCAP * 4 + 256, sym_bytes = l.nseg > 1 ? 2 * cells + 4096 : 0;
    l.sym = take(row_bytes > sym_bytes ? row_bytes : sym_bytes);
    l.rows = l.sym;
    l.round_row = take(4 * (size_t)ROUNDS_MAX * l.total_cand);
    l.row_ctr = take(256);
    l.gflag = take(l.nseg > 1 ? cells / LZ_FLUSH + (size_t)n_chunks + 64 : 0);
    l.seg_adler = take(l.nseg > 1 ? (size_t)n_chunks * LZ_MAXSEG * LZ_FLUSHERS * 16 : 0);
    l.end = o;
    return l;
}

size_t inflate_scratch_bytes(int n_chunks, const u64 *c_lens, const u32 *n_expect) { return inf_layout(n_chunks, c_lens, n_expect).end + 256; }

static void dbg_status(hipStream_t st, const InfResult *d_res, int n_chunks, const char *where)
{
    if (!getenv("MTS_DEBUG_STATUS")) return;
    std::vector<InfResult> h(n_chunks);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), d_res, sizeof(InfResult) * n_chunks, hipMemcpyDeviceToHost);
    fprintf(stderr, "[status %s]", where);
    for (int i = 0; i < n_chunks; i++) fprintf(stderr, " %d:(st %d nout %u ntok %u)", i, h[i].status, h[i].n_out, h[i].ntok);
    fprintf(stderr, "\n");
}

int launch_inflate(hipStream_t st, const u8 *d_cdata, const InfChunk *d_chunks, const InfChunk *h_chunks, int n_chunks,
                   u8 *d_stream, u32 *d_tokens, InfResult *d_res, u64 *d_adler_acc, u32 max_n, int *d_status_out,
                   void *d_scratch, void *engine)
{
    if (n_chunks == 0) return MTS_OK;
    MTS_LDS_ATTR(k_inf_lz, LZ_LDS);
    MTS_LDS_ATTR(k_inf_lz_seg, LZ2_LDS);
    std::vector<u64> lens(n_chunks), so(n_chunks);
    std::vector<u32> nn(n_chunks);
    u64 max_clen = 0;
    for (int i = 0; i < n_chunks; i++) {
        lens[i] = h_chunks[i].c_len; so[i] = h_chunks[i].stream_off; nn[i] = h_chunks[i].n_expect;
        if (lens[i] > max_clen) max_clen = lens[i];
    }
    const InfLayout l = inf_layout(n_chunks, lens.data(), nn.data());
    if (getenv("MTS_DEBUG_STATUS"))
        for (int i = 0; i < n_chunks; i++) {
            std::vector<u8> hb(h_chunks[i].c_len);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hb.data(), d_cdata + h_chunks[i].c_off, hb.size(), hipMemcpyDeviceToHost);
            u32 a = 1, b2 = 0;
            for (u8 v : hb) { a = (a + v) % 65521u; b2 = (b2 + a) % 65521u; }
            fprintf(stderr, "[chunk %d] device copy of the compressed bytes: adler32 %08x\n", i, (b2 << 16) | a);
        }
    if (getenv("MTS_DEBUG_STATUS"))
        for (int i = 0; i < n_chunks; i++)
            fprintf(stderr, "[chunk %d] c_off %llu c_len %llu n_expect %u n_need %u stream_off %llu tok_off %llu\n", i, (unsigned long long)h_chunks[i].c_off,
                    (unsigned long long)h_chunks[i].c_len, h_chunks[i].n_expect, h_chunks[i].n_need, (unsigned long long)h_chunks[i].stream_off, (unsigned long long)h_chunks[i].tok_off);
    // the per-chunk index arrays go to the device in ONE copy (they lie in a row in the scratch area, see inf_layout); which chunk a
    // candidate / true-block slot belongs to -- ~6000 words per chunk -- is filled in on the device (the 60-chunk batch spent
    // 0.9 ms of host time and a 1.5 MB pageable copy on it before its first kernel)
    u8 *S = (u8 *)d_scratch;
    const size_t hst_bytes = l.cand_cnt - l.so;
    u8 *hst = inflate_host_stage(engine, hst_bytes);          // (kept by the engine until the next batch: the copy below is asynchronous)
    u64 *h_so = (u64 *)(hst + (l.so - l.so)), *h_gb = (u64 *)(hst + (l.gb_off - l.so)), *h_tb = (u64 *)(hst + (l.tb_off - l.so));
    u32 *h_nn = (u32 *)(hst + (l.nn - l.so));
    InfFast *h_fast = (InfFast *)(hst + (l.fast - l.so));
    {
        u64 a = 0, b2 = 0, co = 0;
        u32 to = 0;
        for (int i = 0; i < n_chunks; i++) {
            const u64 ng1 = ((u64)nn[i] + 2 + 63) / 64 + 2;
            h_gb[i] = a; a += ng1;
            h_tb[i] = b2; b2 += ng1 / GS_TILE + 2;
            h_so[i] = so[i]; h_nn[i] = nn[i];
            h_fast[i].cand_off = co; h_fast[i].cand_cap = cand_cap_of(lens[i]);
            h_fast[i].true_off = to; h_fast[i].true_cap = 2 * h_fast[i].cand_cap; h_fast[i].pad = 0;
            co += h_fast[i].cand_cap; to += h_fast[i].true_cap;
        }
    }
    MTS_HIP(hipMemcpyAsync(S + l.so, hst, hst_bytes, hipMemcpyHostToDevice, st));
    MTS_HIP(hipMemsetAsync(S + l.cand_cnt, 0, 4 * (size_t)n_chunks, st));
    hipLaunchKernelGGL(k_inf_fill_slots, dim3(n_chunks), dim3(256), 0, st, (const InfFast *)(S + l.fast), (u32 *)(S + l.slot_chunk), (u32 *)(S + l.tslot_chunk));
    const InfFast *d_fast = (const InfFast *)(S + l.fast);
    u64 *d_cand_pos = (u64 *)(S + l.cand_pos), *d_cand_tmp = (u64 *)(S + l.cand_tmp);
    u32 *d_cand_cnt = (u32 *)(S + l.cand_cnt), *d_true_cnt = (u32 *)(S + l.true_cnt);
    int *d_seq = (int *)(S + l.seq_flag);
    CandRes *d_cres = (CandRes *)(S + l.cres);
    TrueBlk *d_tblk = (TrueBlk *)(S + l.tblk);
    uint2 *d_subs = (uint2 *)(S + l.subs);
    const bool fast_path = getenv("MTS_INFLATE_SEQ") == nullptr;
    int lz_workers = LZ_WORKERS;
    if (const char *e = getenv("MTS_LZ_WORKERS")) { lz_workers = atoi(e); if (lz_workers < 1) lz_workers = 1; if (lz_workers > LZ_WORKERS) lz_workers = LZ_WORKERS; }
    if (fast_path) {
        const u64 max_bits = 8 * max_clen + 32;
        dim3 gscan((unsigned)((max_bits + (u64)SCAN_SPANS * SCAN_SPAN_BITS - 1) / ((u64)SCAN_SPANS * SCAN_SPAN_BITS)), n_chunks);
        u64 *d_surv = (u64 *)(S + l.surv);
        u32 *d_surv_cnt = (u32 *)(S + l.surv_cnt);
        MTS_HIP(hipMemsetAsync(d_surv_cnt, 0, 4, st));
        u64 final_zone_bits = SCAN_FINAL_ZONE_BITS;                  // (MTS_SCAN_FINAL_ZONE = bytes: the tests shrink it below a block's length)
        if (const char *e = getenv("MTS_SCAN_FINAL_ZONE")) final_zone_bits = 8ull * (u64)atoll(e);
        hipLaunchKernelGGL(k_inf_scan, gscan, dim3(SCAN_THREADS), 0, st, d_cdata, d_chunks, d_fast, d_cand_pos, d_cand_cnt, d_surv,
                           d_surv_cnt, l.surv_cap, final_zone_bits);
        hipLaunchKernelGGL(k_inf_validate, dim3(std::min((l.surv_cap + 255) / 256, 2048u)), dim3(256), 0, st, d_cdata, d_chunks, d_fast, d_surv,
                           d_surv_cnt, l.surv_cap, d_cand_pos, d_cand_cnt);
        hipLaunchKernelGGL(k_inf_sortc, dim3(n_chunks), dim3(256), 0, st, d_fast, d_cand_pos, d_cand_tmp, d_cand_cnt);
        inflate_mark(engine, st, "inflate_scan");
        const int use_rows = getenv("MTS_INF_NO_ROWS") ? 0 : 1;      // (A/B: pass B decodes every block again)
        MTS_HIP(hipMemsetAsync(S + l.row_ctr, 0, 4, st));
        hipLaunchKernelGGL(k_inf_passA, dim3(l.total_cand), dim3(64), 0, st, d_cdata, d_chunks, d_fast, (const u32 *)(S + l.slot_chunk),
                           d_cand_pos, d_cand_cnt, d_cres, d_subs, (u32 *)(S + l.rows), (u32 *)(S + l.row_ctr), use_rows ? l.rounds_cap : 0u,
                           (u32 *)(S + l.round_row));
        inflate_mark(engine, st, "inflate_passA");
        hipLaunchKernelGGL(k_inf_chain, dim3(n_chunks), dim3(64), 0, st, d_cdata, d_chunks, d_fast, d_cand_pos, d_cand_cnt, d_cres,
                           d_tblk, d_true_cnt, d_res, d_seq);
        inflate_mark(engine, st, "inflate_chain");
        hipLaunchKernelGGL(k_inf_passB, dim3(l.total_true), dim3(64 * PASSB_WAVES), 0, st, d_cdata, d_chunks, d_fast, (const u32 *)(S + l.tslot_chunk),
                           d_tblk, d_true_cnt, d_cres, d_subs, d_tokens, d_res, use_rows);
        if (use_rows)
            hipLaunchKernelGGL(k_inf_move_rows, dim3(l.total_true), dim3(MOVE_THREADS), 0, st, d_chunks, d_fast, (const u32 *)(S + l.tslot_chunk), d_tblk,
                               d_true_cnt, d_cres, d_subs, (const u32 *)(S + l.rows), (const u32 *)(S + l.round_row), d_tokens);
        inflate_mark(engine, st, "inflate_passB");
    }
    hipLaunchKernelGGL(k_inf_wave, dim3(n_chunks), dim3(WV_NT), 0, st, d_cdata, d_chunks, n_chunks, d_tokens, d_res, fast_path ? d_seq : nullptr);
    inflate_mark(engine, st, "inflate_wave_decoder");
    dbg_status(st, d_res, n_chunks, "after decode");
    {
        u32 max_groups = 1;
        for (int i = 0; i < n_chunks; i++) { const u32 gmax = (u32)(((u64)nn[i] + 2 + 63) / 64); if (gmax > max_groups) max_groups = gmax; }
        dim3 gg((max_groups + GS_TILE - 1) / GS_TILE, n_chunks);
        hipLaunchKernelGGL(k_inf_gsum, gg, dim3(GS_TILE), 0, st, d_tokens, d_chunks, d_res, (const u64 *)(S + l.gb_off),
                           (const u64 *)(S + l.tb_off), (u32 *)(S + l.gbase), (u32 *)(S + l.tile_base));
        hipLaunchKernelGGL(k_inf_gscan, dim3(n_chunks), dim3(64), 0, st, d_res, (const u64 *)(S + l.tb_off), (u32 *)(S + l.tile_base));
        hipLaunchKernelGGL(k_inf_plan, dim3((n_chunks + 63) / 64), dim3(64), 0, st, d_res, (const u64 *)(S + l.gb_off), (const u64 *)(S + l.tb_off),
                           (const u32 *)(S + l.gbase), (const u32 *)(S + l.tile_base), n_chunks, l.nseg, (LzPlan *)(S + l.plan));
    }
    inflate_mark(engine, st, "inflate_offsets");
    u64 *d_prof = nullptr;                   // MTS_LZ_PROF=1: per-wave cycle counters of the resolver, printed to stderr
    if (getenv("MTS_LZ_PROF")) { MTS_HIP(hipMalloc(&d_prof, (size_t)n_chunks * 16 * 8 * 8)); MTS_HIP(hipMemsetAsync(d_prof, 0, (size_t)n_chunks * 16 * 8 * 8, st)); }
    const int first_pass = getenv("MTS_LZ_FORCE_RETRY") ? 2 : 0;    // (tests: pretend every chunk's first run gave up, so that the retry pass does the work)
    hipLaunchKernelGGL(k_inf_lz, dim3(n_chunks), dim3(LZ_THREADS), LZ_LDS, st, d_tokens, d_chunks, d_res, (const u64 *)(S + l.gb_off),
                       (const u64 *)(S + l.tb_off), (const u32 *)(S + l.gbase), (const u32 *)(S + l.tile_base), d_stream, lz_workers,
                       d_prof, (const LzPlan *)(S + l.plan), first_pass);
    hipLaunchKernelGGL(k_inf_lz, dim3(n_chunks), dim3(LZ_THREADS), LZ_LDS, st, d_tokens, d_chunks, d_res, (const u64 *)(S + l.gb_off),
                       (const u64 *)(S + l.tb_off), (const u32 *)(S + l.gbase), (const u32 *)(S + l.tile_base), d_stream, 1,
                       nullptr, (const LzPlan *)(S + l.plan), 1);                                  // (retry pass: returns at once unless a wait expired)
    if (l.nseg > 1 && max_n > 0) {
        MTS_HIP(hipMemsetAsync(S + l.gflag, 0, l.seg_adler - l.gflag, st));
        MTS_HIP(hipMemsetAsync(S + l.seg_adler, 0, (size_t)n_chunks * LZ_MAXSEG * LZ_FLUSHERS * 16, st));
        for (int pass = 0; pass < 2; pass++)
            hipLaunchKernelGGL(k_inf_lz_seg, dim3(l.nseg, n_chunks), dim3(LZ_THREADS), LZ2_LDS, st, d_tokens, d_chunks, d_res,
                               (const u64 *)(S + l.gb_off), (const u64 *)(S + l.tb_off), (const u32 *)(S + l.gbase), (const u32 *)(S + l.tile_base),
                               (const LzPlan *)(S + l.plan), (u16 *)(S + l.sym), pass ? 1 : lz_workers, pass ? 1 : first_pass, d_stream,
                               (u8 *)(S + l.gflag), (u64 *)(S + l.seg_adler));
    }
    hipLaunchKernelGGL(k_inf_lz_settle, dim3((n_chunks + 63) / 64), dim3(64), 0, st, d_res, n_chunks);
    MTS_HIP(hipMemsetAsync(d_adler_acc, 0, sizeof(u64) * 2 * n_chunks, st));      // (the translation of segmented chunks adds their byte sums)
    if (l.nseg > 1 && max_n > 0) {
        hipLaunchKernelGGL(k_inf_windows, dim3(n_chunks), dim3(1024), 0, st, d_chunks, d_res, (const LzPlan *)(S + l.plan),
                           (const u16 *)(S + l.sym), (u8 *)(S + l.win), d_stream, (const u8 *)(S + l.gflag));
        hipLaunchKernelGGL(k_inf_translate, dim3((max_n + 16383) / 16384, n_chunks), dim3(256), 0, st, d_chunks, d_res,
                           (const LzPlan *)(S + l.plan), (const u16 *)(S + l.sym), (const u8 *)(S + l.win), d_stream, d_adler_acc, (const u8 *)(S + l.gflag));
    }
    MTS_HIP(hipGetLastError());
    if (d_prof) {
        std::vector<u64> hp((size_t)n_chunks * 16 * 8);
        MTS_HIP(hipStreamSynchronize(st));
        MTS_HIP(hipMemcpy(hp.data(), d_prof, hp.size() * 8, hipMemcpyDeviceToHost));
        MTS_HIP(hipFree(d_prof));
        u64 a[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int nw = 0;
        for (int c = 0; c < n_chunks; c++) for (int w = 0; w < 16; w++) { const u64 *q = &hp[((size_t)c * 16 + w) * 8]; if (!q[5]) continue; nw++; for (int k = 0; k < 8; k++) a[k] += q[k]; }
        if (nw) fprintf(stderr, "[lz prof] waves %d  per wave: total %.0f  wait %.0f  pre %.0f  loop %.0f cycles; groups %.0f  iters/group %.2f  cycles/group %.0f  vmwait %.0f  sleeps %.0f\n", nw,
                        (double)a[0] / nw, (double)a[1] / nw, (double)a[2] / nw, (double)a[3] / nw, (double)a[5] / nw, (double)a[4] / a[5], (double)a[0] / a[5], (double)a[6] / nw, (double)a[7] / nw);
    }
    inflate_mark(engine, st, "inflate_lz");
    dbg_status(st, d_res, n_chunks, "after lz");
    // byte sums of the chunks that did not go through the translation (resolved as bytes by k_inf_lz)
    int rc = launch_adler_stream(st, d_stream, (const u64 *)(S + l.so), (const u32 *)(S + l.nn), n_chunks, max_n, d_adler_acc,
                                 (const u32 *)(S + l.plan), (u32)(sizeof(LzPlan) / 4));
    if (rc) return rc;
    hipLaunchKernelGGL(k_inf_finish, dim3((n_chunks + 63) / 64), dim3(64), 0, st, d_chunks, d_res, d_adler_acc, n_chunks,
                       d_status_out, l.nseg > 1 && max_n > 0 ? (const u64 *)(S + l.seg_adler) : nullptr);
    MTS_HIP(hipGetLastError());
    inflate_mark(engine, st, "adler32");
    dbg_status(st, d_res, n_chunks, "after adler");
    return MTS_OK;
}

}  // namespace mts

#if MTS_LZ2_STATS
extern "C" int mts_debug_lz2_stats(unsigned long long *out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mts::g_lz2_stats), 128) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(mts::g_lz2_stats), z, 128) == hipSuccess ? 0 : -1;
}
#endif

