// Bit-exact zlib-1.2.11 DEFLATE on gfx950: levels 4..9 (deflate_slow) by the stages below, levels 1..3 (deflate_fast) by
// the sort + section F (its walks replace M and P).
//
// Replaces `zlib.compress(chunkd.tobytes(order))` (/root/reference/mtscomp.py:394).  The algorithm is
// the parallel formulation of SURVEY.md Appendix A.3, checked stage by stage against
// oracle/mtsc_oracle.c (orc_match_tables / orc_parse_tables / orc_deflate):
//
//   S  k_hash_sort     per tile (TILE owned positions + HALO history): stable 2-pass radix sort of the
//                      positions by their 15-bit zlib hash -> every hash chain becomes a contiguous,
//                      position-ordered run (parse independent because deflate_slow inserts every
//                      position).  Ranking by lane-ordered LDS atomics (probed, and checked where the order is used),
//                      ballots as fallback.
//   M  k_match5        (budgets <= 128) per tile: a wave walks contiguous 64-slot groups of the sorted order
//                      with a ring of per-slot entries in LDS; one lane per position scores, newest
//                      first, only the candidates its looked-up filter masks let through, and records
//                      the best match within the full budget and within budget>>2 (the
//                      prev_length >= good_match case).  The workgroups of one XCD share a tile (its window stays
//                      in that XCD's L2).  k_match6: budgets > 128, the same masks from a ring the workgroup shares.
//   F  k_fast_*        levels 1..3 (deflate_fast: the chains depend on the parse): per position the older members of its run
//                      with their match lengths, all at once; then one wave per chunk walks the chunk in order.
//   P  k_parse_*       lazy-evaluation state machine over the tables; one lane per SEG positions,
//                      speculative entry, iterated to a fixed point (walks re-converge after a few
//                      tokens); then count + emit tokens (LDS rows, written out coalesced).
//   T  k_block_trees   per 16383-token block: symbol histogram (LDS atomics) and zlib's exact
//                      build_tree / gen_bitlen / scan_tree by one lane; stored/fixed/dynamic choice.
//   L  k_block_layout  per chunk: bit offsets of the blocks, stream size, adler32.
//   B  k_block_pack    per block: codes OR-ed into an LDS image of the block at scanned bit positions,
//                      image written out as whole words.
#include <stdlib.h>

#include <mutex>

#include "deflate_dev.h"

namespace mts {

// ================================================================================================
// S: hash sort
// ================================================================================================
// The radix passes move 32-bit keys  (high 8 hash bits << 18) | window-relative position.  Everything else the
// match stage wants about a position (its bytes, its chain length) it derives from the window bytes.
#ifndef MTS_SORT_WAVES
#define MTS_SORT_WAVES 8
#endif
#ifndef MTS_SORT_KPL
#define MTS_SORT_KPL 12
#endif
constexpr int SORT_WAVES = MTS_SORT_WAVES; // waves per sort workgroup (one tile).  With the key loads a step ahead of their use (rank_pass) the best tiling moved from
                                           // 16 waves x 4 keys per lane and step (9.6 ms) to 8 x 12 (8.4): 8 x 4 10.1, 8 x 8 8.65, 8 x 10 8.6, 8 x 14 / 16 9.7, 4 x 8 8.9, 4 x 16 8.8,
                                           // 6 x 8 9.3, 12 x 8 9.9, 16 x 8 9.4 -- longer runs per store against workgroups per CU (69 KB of LDS: two)
constexpr int SORT_KPL = MTS_SORT_KPL;              // keys per lane and step: SORT_KPL * 64 keys are staged in destination order per wave
constexpr int SORT_NT = SORT_WAVES * 64;
#ifndef MTS_SORT_B1
#define MTS_SORT_B1 7
#endif
constexpr int SORT_B1 = MTS_SORT_B1;     // two passes: SORT_B1 + (15 - SORT_B1) hash bits (7 + 8: 7.96 ms, 8 + 7: 8.28 -- fewer, longer runs in the first pass's stores)
constexpr int SORT_B2 = 15 - SORT_B1;
// The match stage only wants equal hashes next to each other in position order, whatever the order of the runs, so the
// passes could sort on any bijective scramble of the hash.  zlib's hash of int16 deltas is lopsided (every other position
// is (high, low, high) bytes: ~120 values that differ in a few bits) and the lanes of a wave instruction that meet in one
// counter are served one after the other -- but the same lopsidedness keeps the keys of a step in few, long runs, and the
// stores matter more: MTS_SORT_SCRAMBLE=1 (xor-shift, odd multiplier, xor-shift) was measured at 15.0 ms against 11.2, and the
// match stage, whose window reads and table stores follow the order of the runs, at 26.8 against 25.9.
#ifndef MTS_SORT_SCRAMBLE
#define MTS_SORT_SCRAMBLE 0
#endif
__device__ __forceinline__ u32 sort_hash(u32 b012)
{
    u32 h = hash_of(b012);
#if MTS_SORT_SCRAMBLE
    h ^= h >> 8;
    h = (h * 0x5bd1u) & 0x7fffu;
    h ^= h >> 7;
#endif
    return h;
}
// Where the counter of a digit lives.  The digits that are common on recordings differ in their high bits (zlib's hash of int16
// deltas: the popular values of the low digit are b1[1:0] << 5 ^ b2 with b2 one of two bytes, those of the high digit multiples
// of 8 apart) -- and counters 32 apart share an LDS bank: the lanes of a rank instruction met in the same few banks
// (~21 cycles per instruction by the stream's statistics, the most popular digit alone accounts for ~9).  Folding the high
// bits into the low ones spreads them over the banks.
#ifndef MTS_SORT_SWZ
#define MTS_SORT_SWZ 1
#endif
__device__ __forceinline__ u32 cslot(u32 d) { return MTS_SORT_SWZ ? d ^ (d >> 5) : d; }

// per-(wave, digit) counts -> where each wave's keys of each digit start (digit-major, wave-minor: stable)
template <int NB>
__device__ __forceinline__ void bin_offsets(u32 (*cnt)[256], u32 *tot)
{
    constexpr int NBIN = 1 << NB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 cs = cslot(threadIdx.x);
    if (threadIdx.x < NBIN) {
        u32 run = 0;
        for (int w = 0; w < SORT_WAVES; w++) { const u32 c = cnt[w][cs]; cnt[w][cs] = run; run += c; }
        tot[threadIdx.x] = run;
    }
    __syncthreads();
    if (wave == 0) {
        u32 v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = (lane * 4 + k < NBIN) ? tot[lane * 4 + k] : 0; sum += v[k]; }
        u32 total;
        u32 ex = wave_excl_scan_u32(sum, total);
#pragma unroll
        for (int k = 0; k < 4; k++) { if (lane * 4 + k < NBIN) tot[lane * 4 + k] = ex; ex += v[k]; }
    }
    __syncthreads();
    if (threadIdx.x < NBIN) {
        const u32 base = tot[threadIdx.x];
        for (int w = 0; w < SORT_WAVES; w++) cnt[w][cs] += base;
    }
    __syncthreads();
}

// One stable counting-sort pass over the tile: wave w owns keys [w * per, (w + 1) * per) and ranks them 64 at a
// time with wave ballots.  The first pass makes the keys from the stream and, while it scatters them,
// counts the SECOND pass's digits per second-pass wave (a key's destination says which wave will own it),
// so the second pass needs no counting loop of its own.
// NB = bits of this pass's digit; FIRST: the digit is the low NB bits of the hash made from the stream bytes (and the key
// that is stored keeps the other hash bits above the position); otherwise the digit is NB bits at SHIFT above the position.
// NNB > 0: count the next pass's digit (the NNB bits above this pass's) for the wave that will own the key there.
template <int NB, bool FIRST, int SHIFT, int NNB>
__device__ __forceinline__ void rank_pass(const u8 *__restrict__ s, const u32 *__restrict__ src, u32 *__restrict__ dst, u32 wlen,
                                          u32 per, u32 (*cnt)[256], u32 (*cnt2)[1 << SORT_B2], u32 per_magic, int lane_ordered, u32 *stg, u32 *dlt)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 beg = min((u32)wave * per, wlen), end = min(beg + per, wlen);
    // first pass: the fetched word is the 15-bit hash (digit = its low 8 bits; what is stored and sorted on is
    // the key = high 7 hash bits : position); second pass: the key (digit = its 7 hash bits)
    // (unconditional: a load under a branch is waited for where the branch ends, and the four loads of a step went out one after
    //  the other, each behind the stores of the step before; a lane past the end reads the last key and drops it)
    auto fetch = [&](u32 i) -> u32 {
        const u32 ic = i < end ? i : end - 1;
        if (FIRST) return sort_hash(gld_u32_unaligned(s, ic));
        return src[ic];
    };
    auto digit_of = [&](u32 f) -> u32 { return FIRST ? f & ((1u << NB) - 1) : (f >> (REL_BITS + SHIFT)) & ((1u << NB) - 1); };
    auto key_of = [&](u32 f, u32 i) -> u32 { return FIRST ? ((f >> NB) << REL_BITS) | i : f; };
    auto next_digit_of = [&](u32 f) -> u32 { return (FIRST ? f >> NB : f >> (REL_BITS + SHIFT + NB)) & ((1u << (NNB > 0 ? NNB : 1)) - 1); };
    if (lane_ordered) {
        // The LDS retires the same-address atomics of one wave instruction in lane order (probed at start-up, see
        // k_probe_lds_order): the value returned by the add IS the stable destination -- no ballots, no barriers.
        // The kernel is bound by its scattered 4-byte stores (partial sectors), so the 256 keys of a step are first
        // put in destination order in LDS (keys of one digit sit together there: slot = destination - first destination
        // of the digit in this step + keys of smaller digits in this step) and then stored by consecutive lanes.
        constexpr int NBIN = 1 << NB;
        static_assert(NBIN % 64 == 0 && NBIN <= 256, "a lane owns the counters of digits lane, lane + 64, ...");
        constexpr int KPL = SORT_KPL, BK = 64 * KPL;
        u32 *K = stg + wave * (2 * BK), *A = K + BK, *D = dlt + wave * 256;
        // the keys of a step are asked for a step ahead, BEFORE the step's stores: what is waited for then is loads that went
        // out a whole step earlier, with that step's stores (a fixed number: lanes without a key repeat lane 0's store) behind them
        u32 key_n[KPL];
#pragma unroll
        for (int k = 0; k < KPL; k++) key_n[k] = fetch(beg + 64 * k + lane);
        // (the first keys are waited for HERE: entering the loop with them in flight, the wait at its top has to allow for this
        //  path -- no stores behind the loads -- and on every later step it then sat out the round trip of the step's own stores)
#pragma unroll
        for (int k = 0; k < KPL; k++) asm volatile("" :: "v"(key_n[k]));
        for (u32 base = beg; base < end; base += BK) {
            u32 key[KPL], at[KPL], bf[4];
#pragma unroll
            for (int k = 0; k < KPL; k++) { key[k] = key_n[k]; key_n[k] = fetch(base + BK + 64 * k + lane); }
            // (a lane looks after the counters of digits lane, lane + 64, ...: neighbouring lanes on neighbouring words.  With
            // digits 4 * lane + j, lanes 8 apart met in one LDS bank on each of these twelve accesses per step)
            constexpr int NJ = NBIN / 64;
#pragma unroll
            for (int j = 0; j < NJ; j++) bf[j] = cnt[wave][cslot(lane + 64 * j)];
#pragma unroll
            for (int k = 0; k < KPL; k++) at[k] = base + 64 * k + lane < end ? atomicAdd(&cnt[wave][cslot(digit_of(key[k]))], 1u) : 0u;
            u32 total = 0;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const u32 lcj = cnt[wave][cslot(lane + 64 * j)] - bf[j];
                const u32 inc = wave_incl_scan_dpp(lcj);
                D[cslot(lane + 64 * j)] = total + inc - lcj - bf[j];
                total += (u32)__builtin_amdgcn_readlane((int)inc, 63);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < KPL; k++) {
                if (base + 64 * k + lane < end) {
                    const u32 slot = at[k] + D[cslot(digit_of(key[k]))];
                    K[slot] = key_of(key[k], base + 64 * k + lane);
                    A[slot] = at[k];
                    if (NNB > 0) atomicAdd(&cnt2[per_magic ? __umulhi(at[k] >> 6, per_magic) : at[k] >> 6][cslot(next_digit_of(key[k]))], 1u);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < KPL; k++) { const u32 j = 64 * k + lane, jj = j < total ? j : 0u; dst[A[jj]] = K[jj]; }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        return;
    }
    // keys are fetched two steps ahead of their ranking
    u32 key_n = fetch(beg + lane), key_nn = fetch(beg + lane + 64);
    for (u32 base = beg; base < end; base += 64) {
        const u32 i = base + lane;
        const bool act = i < end;
        const u32 key = key_n;
        key_n = key_nn;
        key_nn = fetch(i + 128);
        const u32 d = digit_of(key);
        const u64 actm = __ballot(act);
        const u64 m = match_digit<NB>(d, actm);
        const u32 rank = __popcll(m & lanemask_lt()), count = __popcll(m);
        u32 off = 0;
        if (act) off = cnt[wave][cslot(d)];
        __builtin_amdgcn_wave_barrier();
        if (act) {
            dst[off + rank] = key_of(key, i);
            if (rank == count - 1) cnt[wave][cslot(d)] = off + count;
            if (NNB > 0) atomicAdd(&cnt2[per_magic ? __umulhi((off + rank) >> 6, per_magic) : (off + rank) >> 6][cslot(next_digit_of(key))], 1u);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
}

// Start-up probe for the property the fast ranking relies on: 64 lanes add 1 to counters picked at random (many
// lanes per counter); in lane order every lane must get back the number of lower lanes that picked its counter.
__global__ __launch_bounds__(1024) void k_probe_lds_order(u32 *bad, int iters)
{
    __shared__ u32 cnt[16][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32 x = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 99u, nbad = 0;
    for (int it = 0; it < iters; it++) {
        for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0;
        __builtin_amdgcn_wave_barrier();
        x = x * 1664525u + 1013904223u;
        const u32 d = (x >> 13) & (0xffu >> (it & 7));                  // 256, 128, ... 2 counters
        const u32 got = atomicAdd(&cnt[wave][d], 1u);
        const u64 m = match_digit<8>(d, ~0ull);
        if (got != (u32)__popcll(m & lanemask_lt())) nbad++;
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(bad, nbad);
}

struct OpTile;                                               // (the one-pass sort's per-tile record, below)
__device__ __forceinline__ bool op_two_pass(const OpTile *ot, u32 tile);
__global__ __launch_bounds__(SORT_NT) void k_hash_sort(const u8 *__restrict__ stream, const TileDesc *__restrict__ tiles,
                                                    u32 *__restrict__ tmp, u32 *__restrict__ sorted, int lane_ordered,
                                                    const OpTile *__restrict__ only /* null: every tile; else the tiles the one-pass sort left to this one */)
{
    if (only && !op_two_pass(only, blockIdx.x)) return;
    const TileDesc td = tiles[blockIdx.x];
    __shared__ u32 cnt[SORT_WAVES][256];
    __shared__ u32 cnt2[SORT_WAVES][1 << SORT_B2];
    __shared__ u32 tot[256];
    __shared__ u32 stg[SORT_WAVES * 2 * 64 * SORT_KPL];      // per wave: the keys of a step + their destinations, in destination order
    __shared__ u32 dlt[SORT_WAVES * 256];                     // per wave and digit: slot of the digit's first key of the step - its destination
    if (td.wlen == 0) return;
    const u8 *s = stream + td.stream_off + td.w;
    const u32 wlen = td.wlen;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 per = (((wlen + SORT_WAVES - 1) / SORT_WAVES) + 63) & ~63u;      // keys per wave (both passes)
    // __umulhi(x >> 6, magic) == x / per (per is a multiple of 64; x >> 6 < 2^12); magic 0 stands for per == 64 (x / per = x >> 6)
    const u32 per_magic = per > 64 ? 0xffffffffu / (per >> 6) + 1 : 0;
    for (int i = threadIdx.x; i < SORT_WAVES * 256; i += SORT_NT) (&cnt[0][0])[i] = 0;
    for (int i = threadIdx.x; i < SORT_WAVES * (1 << SORT_B2); i += SORT_NT) (&cnt2[0][0])[i] = 0;
    __syncthreads();
    {   // digits of the first pass: 4 positions per lane and step, so four loads are in flight
        const u32 beg = min((u32)wave * per, wlen), end = min(beg + per, wlen);
        constexpr int CK = 8;                                    // loads in flight per lane; the next step's go out before this step's atomics
        if (beg < end) {
            u32 vn[CK];
#pragma unroll
            for (int k = 0; k < CK; k++) vn[k] = gld_u32_unaligned(s, min(beg + lane + 64 * k, end - 1));
            for (u32 i0 = beg + lane; i0 < end; i0 += 64 * CK) {
                u32 v[CK];
#pragma unroll
                for (int k = 0; k < CK; k++) { v[k] = vn[k]; vn[k] = gld_u32_unaligned(s, min(i0 + 64 * CK + 64 * k, end - 1)); }
#pragma unroll
                for (int k = 0; k < CK; k++) if (i0 + 64 * k < end) atomicAdd(&cnt[wave][cslot(sort_hash(v[k]) & ((1u << SORT_B1) - 1))], 1u);
            }
        }
    }
    __syncthreads();
    u32 *tmp_t = tmp + td.sorted_off, *out_t = sorted + td.sorted_off;
    auto next_counts = [&]() {                       // the fused counts become the next pass's
        for (int i = threadIdx.x; i < SORT_WAVES * (1 << SORT_B2); i += SORT_NT) { cnt[i >> SORT_B2][i & ((1 << SORT_B2) - 1)] = cnt2[i >> SORT_B2][i & ((1 << SORT_B2) - 1)]; cnt2[i >> SORT_B2][i & ((1 << SORT_B2) - 1)] = 0; }
        __syncthreads();
    };
    bin_offsets<SORT_B1>(cnt, tot);
    rank_pass<SORT_B1, true, 0, SORT_B2>(s, nullptr, tmp_t, wlen, per, cnt, cnt2, per_magic, lane_ordered, stg, dlt);        // low hash bits
    next_counts();
    bin_offsets<SORT_B2>(cnt, tot);
    rank_pass<SORT_B2, false, 0, 0>(s, tmp_t, out_t, wlen, per, cnt, cnt2, per_magic, lane_ordered, stg, dlt);          // high bits
    // (three passes of 5 bits -- runs of ~8 keys per store -- were measured at 17.8 ms against 12.4: a pass costs what it costs
    // whatever its stores look like)
}


// ------------------------------------------------------------------------------------------------
// One-pass hash sort (round 5).  A recording's positions fall into few of the 32768 hash buckets (~970 per tile on the synthetic
// recordings: int16 deltas are small), so the 15-bit hash can be ranked in ONE scatter instead of two radix passes that write
// and read the 32-bit keys twice (28 GB per launch for 6.3 GB of sorted keys): a tile's live buckets get dense numbers in
// ascending hash order -- the same order the two passes produce, so the sorted keys are the same words -- and the keys go
// straight from the stream to their place.  Stability (positions ascending inside a bucket: the hash chains) by construction:
// a tile is cut into slices of OP_SLICE consecutive positions, a slice is a workgroup whose 8 waves take consecutive eighths,
// and a key's destination is  bucket start + keys of the bucket in earlier slices + in earlier waves of its slice + in earlier
// instructions / lower lanes of its wave (the lane-ordered LDS atomic the two-pass kernel ranks with, guarded the same way).
//   k_op_live     per tile (8 workgroups): bitmap of the hashes that occur                     [1 LDS read per key]
//   k_op_prefix   per tile: bucket numbers = prefix popcounts of the bitmap; tiles with more than OP_LMAX live buckets (uniform
//                 random data) or fewer than OP_MIN_KEYS keys are left to the two-pass kernel
//   k_op_count    per slice: keys per bucket                                                    [2 LDS reads + 1 atomic per key]
//   k_op_scan     per tile: bucket starts, and per slice and bucket the keys of earlier slices
//   k_op_scatter  per slice: per-wave counts, then rank and store                               [~8 LDS operations per key]
// The 64 workgroups of a tile's slices run on ONE XCD, like the match stage's (blockIdx -> tile as there): the tile's 1 MB of
// sorted keys, written four bytes at a time, and the counts between the kernels stay in that XCD's L2 until they are complete.
// Scratch: an OpTile per tile (8 KB) and, in the tile's part of the two-pass kernel's first-pass buffer, 6 KB per slice.
// ------------------------------------------------------------------------------------------------
#ifndef MTS_OP_SLICE
#define MTS_OP_SLICE 8192
#endif
constexpr int OP_SLICE = MTS_OP_SLICE;              // keys per slice = per workgroup of k_op_count / k_op_scatter
constexpr int OP_TILE_SLICES = WIN / OP_SLICE;       // slices of a full tile = workgroups of a tile side by side on its XCD
#ifndef MTS_OP_WAVES
#define MTS_OP_WAVES 16
#endif
constexpr int OP_WAVES = MTS_OP_WAVES;
constexpr int OP_NT = OP_WAVES * 64;
constexpr int OP_KPL = OP_SLICE / (OP_WAVES * 64);   // keys per lane
constexpr int OP_LMAX = 1024;                        // live buckets a tile may have
constexpr int OP_MIN_KEYS = 2 * OP_SLICE;            // smaller tiles: the two-pass kernel (their part of the scratch buffer is too small)
constexpr int OP_ROW_WORDS = 512 + 1024;             // a slice's row in the scratch: 1024 counts (u16), 1024 offsets (u32)
#ifndef MTS_OP_LIVE_PARTS
#define MTS_OP_LIVE_PARTS 8
#endif
constexpr int OP_LIVE_PARTS = MTS_OP_LIVE_PARTS;     // workgroups of k_op_live per tile
struct OpTile {
    uint2 bp[1024];                                  // .x: which of the hashes 32 w .. 32 w + 31 occur, .y: live buckets below hash 32 w (side by side: a key's bucket is ONE 8-byte LDS read)
    u32 live, two_pass;
    u32 pad[62];
};
__device__ __forceinline__ bool op_two_pass(const OpTile *ot, u32 tile) { return ot[tile].two_pass != 0; }
__device__ __forceinline__ u32 op_slices(u32 wlen) { return (wlen + OP_SLICE - 1) / OP_SLICE; }
// the workgroups of one tile side by side on one XCD (block b runs on XCD b % 8): tile and part of it for this block
__device__ __forceinline__ bool op_tile_part(int n_tiles, u32 parts, u32 &tile, u32 &part)
{
    const u32 xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    tile = (jb / parts) * 8 + xcd; part = jb % parts;
    return tile < (u32)n_tiles;
}
static unsigned op_grid(int n_tiles, unsigned parts) { return 8u * (((unsigned)n_tiles + 7) / 8) * parts; }

// four consecutive positions' hashes from the eight bytes at position p (a multiple of 4: one 8-byte load instead of the two
// 4-byte loads per position of gld_u32_unaligned)
typedef u32 u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ u32x2_a4 op_load8(const u8 *s, u32 p) { return *(const __attribute__((address_space(1))) u32x2_a4 *)(u64)(s + p); }
__device__ __forceinline__ u32 op_hash_at(u32x2_a4 x, int k /* 0 .. 3 */) { return sort_hash(k ? __builtin_amdgcn_alignbyte(x.y, x.x, (u32)k) : x.x); }

__global__ __launch_bounds__(512) void k_op_live(const u8 *__restrict__ stream, const TileDesc *__restrict__ tiles, int n_tiles, OpTile *__restrict__ ot)
{
    __shared__ u32 bm[1024];
    u32 tile, part;
    if (!op_tile_part(n_tiles, OP_LIVE_PARTS, tile, part)) return;
    const TileDesc td = tiles[tile];
    if (td.wlen < (u32)OP_MIN_KEYS) return;
    const u32 per = ((td.wlen + OP_LIVE_PARTS - 1) / OP_LIVE_PARTS + 3) & ~3u, beg = min(part * per, td.wlen), end = min(beg + per, td.wlen);
    if (beg >= end) return;
    const u8 *s = stream + td.stream_off + td.w;                   // (4-byte aligned: stream_off and w are multiples of 64)
    for (int i = threadIdx.x; i < 1024; i += 512) bm[i] = 0;
    __syncthreads();
    constexpr int U = 8;                                          // loads in flight per lane: 32 positions
    for (u32 i0 = beg + 4 * threadIdx.x; i0 < end; i0 += 4 * 512 * U) {
        u32x2_a4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = op_load8(s, min(i0 + 4u * 512 * k, (end - 1) & ~3u));
#pragma unroll
        for (int k = 0; k < U; k++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (i0 + 4u * 512 * k + q < end) {
                    const u32 h = op_hash_at(v[k], q), bit = 1u << (h & 31);
                    if (!(bm[h >> 5] & bit)) atomicOr(&bm[h >> 5], bit);   // (nearly always set already: a read, not an atomic)
                }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 512) if (bm[i]) atomicOr(&ot[tile].bp[i].x, bm[i]);
}

__global__ __launch_bounds__(1024) void k_op_prefix(const TileDesc *__restrict__ tiles, OpTile *__restrict__ ot)
{
    __shared__ u32 wsum[16];
    const u32 tile = blockIdx.x, w = threadIdx.x;
    const u32 c = (u32)__builtin_popcount(ot[tile].bp[w].x);
    u32 tot;
    const u32 ex = wave_excl_scan_u32(c, tot);
    if ((w & 63) == 0) wsum[w >> 6] = tot;
    __syncthreads();
    u32 before = 0, all = 0;
    for (u32 q = 0; q < 16; q++) { if (q < (w >> 6)) before += wsum[q]; all += wsum[q]; }
    ot[tile].bp[w].y = before + ex;
    if (w == 0) { ot[tile].live = all; ot[tile].two_pass = (tiles[tile].wlen < (u32)OP_MIN_KEYS || all > (u32)OP_LMAX) ? 1u : 0u; }
}

// a key's bucket number: live buckets below its hash
__device__ __forceinline__ u32 op_bucket(const uint2 *bp, u32 h)
{
    const uint2 e = bp[h >> 5];
    return e.y + (u32)__builtin_popcount(e.x & ((1u << (h & 31)) - 1u));
}

constexpr int OP_CNT_NT = 512;                      // threads of k_op_count (16 waves: 1.72 ms against 1.44)
__global__ __launch_bounds__(OP_CNT_NT) void k_op_count(const u8 *__restrict__ stream, const TileDesc *__restrict__ tiles, int n_tiles,
                                                        const OpTile *__restrict__ ot, u32 *__restrict__ scratch)
{
    __shared__ uint2 bp[1024];
    __shared__ u32 cnt[OP_LMAX];
    u32 tile, slice;
    if (!op_tile_part(n_tiles, OP_TILE_SLICES, tile, slice)) return;
    if (ot[tile].two_pass) return;
    const TileDesc td = tiles[tile];
    if (slice >= op_slices(td.wlen)) return;
    const u8 *s = stream + td.stream_off + td.w;
    for (int i = threadIdx.x; i < 1024; i += OP_CNT_NT) { bp[i] = ot[tile].bp[i]; cnt[i] = 0; }
    const u32 beg = slice * OP_SLICE, end = min(beg + (u32)OP_SLICE, td.wlen);
    constexpr int U = OP_SLICE / (4 * OP_CNT_NT);                 // 8-byte loads per lane: four consecutive positions each
    static_assert(OP_SLICE % (4 * OP_CNT_NT) == 0, "whole loads");
    u32x2_a4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = op_load8(s, min(beg + 4u * (OP_CNT_NT * k + threadIdx.x), (end - 1) & ~3u));
    __syncthreads();
#pragma unroll
    for (int k = 0; k < U; k++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (beg + 4u * (OP_CNT_NT * k + threadIdx.x) + q < end) atomicAdd(&cnt[op_bucket(bp, op_hash_at(v[k], q))], 1u);
    __syncthreads();
    u16 *row = (u16 *)(scratch + td.sorted_off + (size_t)slice * OP_ROW_WORDS);
    for (int i = threadIdx.x; i < 512; i += OP_CNT_NT) ((u32 *)row)[i] = cnt[2 * i] | (cnt[2 * i + 1] << 16);      // (a slice has at most OP_SLICE < 65536 keys)
}

__global__ __launch_bounds__(1024) void k_op_scan(const TileDesc *__restrict__ tiles, const OpTile *__restrict__ ot, u32 *__restrict__ scratch)
{
    __shared__ u32 wsum[16];
    const u32 tile = blockIdx.x, b = threadIdx.x;
    if (ot[tile].two_pass) return;
    const TileDesc td = tiles[tile];
    const u32 ns = op_slices(td.wlen);
    u32 *rows = scratch + td.sorted_off;
    // a bucket's counts of all slices at once (a loop over the slices, as it was, waited for every load before it asked for the
    // next: 56 round trips per workgroup, 0.37 ms of nothing but waiting)
    u32 c[OP_TILE_SLICES], total = 0;
#pragma unroll
    for (int sl = 0; sl < OP_TILE_SLICES; sl++) c[sl] = (u32)sl < ns ? ((const u16 *)(rows + (size_t)sl * OP_ROW_WORDS))[b] : 0u;
#pragma unroll
    for (int sl = 0; sl < OP_TILE_SLICES; sl++) total += c[sl];
    u32 tot;
    const u32 ex = wave_excl_scan_u32(total, tot);
    if ((b & 63) == 0) wsum[b >> 6] = tot;
    __syncthreads();
    u32 run = ex;
    for (u32 q = 0; q < (b >> 6); q++) run += wsum[q];
#pragma unroll
    for (int sl = 0; sl < OP_TILE_SLICES; sl++) {
        if ((u32)sl < ns) rows[(size_t)sl * OP_ROW_WORDS + 512 + b] = run;
        run += c[sl];
    }
}

__global__ __launch_bounds__(OP_NT) void k_op_scatter(const u8 *__restrict__ stream, const TileDesc *__restrict__ tiles, int n_tiles,
                                                    const OpTile *__restrict__ ot, const u32 *__restrict__ scratch, u32 *__restrict__ sorted)
{
    // The slice's 4096 keys are first put in destination order in LDS -- bucket after bucket, as they will lie in memory -- and
    // then stored by consecutive threads: stored straight from the lanes that ranked them (4 bytes each, every lane of an
    // instruction somewhere else) the kernel was bound by the number of write requests the L2s take (1.58e9 of them: 6.4 ms).
    // An entry is position | bucket << REL_BITS: the bucket says where the entry goes (D[bucket] + its place in the slice), and
    // what lies above the position in a sorted key is nobody's business (every reader masks it).
    __shared__ uint2 bp[1024];
    __shared__ u32 D[OP_LMAX];                                    // first: where the bucket's keys of this slice go; then: that minus the bucket's first place in the slice
    __shared__ u32 cw[OP_WAVES][OP_LMAX / 2];                     // per wave and bucket: two 16-bit counters a word
    __shared__ u32 S[OP_SLICE];
    __shared__ u32 wsum[OP_WAVES];
    static_assert(OP_LMAX <= (1 << (32 - REL_BITS)) && OP_SLICE < 65536 && OP_SLICE % (OP_WAVES * 64) == 0, "position and bucket share a word; places in a slice are 16-bit");
    u32 tile, slice;
    if (!op_tile_part(n_tiles, OP_TILE_SLICES, tile, slice)) return;
    if (ot[tile].two_pass) return;
    const TileDesc td = tiles[tile];
    if (slice >= op_slices(td.wlen)) return;
    const u8 *s = stream + td.stream_off + td.w;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 *row = scratch + td.sorted_off + (size_t)slice * OP_ROW_WORDS;
    for (int i = threadIdx.x; i < 1024; i += OP_NT) { bp[i] = ot[tile].bp[i]; D[i] = row[512 + i]; }
    for (int i = threadIdx.x; i < OP_WAVES * (OP_LMAX / 2); i += OP_NT) (&cw[0][0])[i] = 0;
    // wave w: keys beg + 64 k + lane, k = 0 .. 7 (ascending with k and with the lane: the order of the ranks below)
    const u32 s0 = slice * OP_SLICE, beg = s0 + (u32)wave * (OP_SLICE / OP_WAVES), end = min(s0 + (u32)OP_SLICE, td.wlen);
    u32 v[OP_KPL];
#pragma unroll
    for (int k = 0; k < OP_KPL; k++) v[k] = gld_u32_unaligned(s, min(beg + 64u * k + lane, td.wlen - 1));
    __syncthreads();
    u32 id[OP_KPL];
#pragma unroll
    for (int k = 0; k < OP_KPL; k++) {
        id[k] = op_bucket(bp, sort_hash(v[k]));
        if (beg + 64u * k + lane < end) atomicAdd(&cw[wave][id[k] >> 1], 1u << (16 * (id[k] & 1)));
    }
    __syncthreads();
    {   // thread j: buckets 2 j and 2 j + 1.  Their first places in the slice (an exclusive scan over the buckets), and per wave
        // where its keys of the bucket start (both halves of a word at once: no half exceeds 4096)
        const int j = threadIdx.x & (OP_LMAX / 2 - 1);              // (more threads than bucket pairs: the rest go along and write nothing)
        const bool mine = threadIdx.x < OP_LMAX / 2;
        u32 c[OP_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < OP_WAVES; w++) { c[w] = mine ? cw[w][j] : 0u; tot += c[w]; }
        const u32 c0 = tot & 0xffffu, c1 = tot >> 16;
        u32 wtot;
        const u32 ex = wave_excl_scan_u32(c0 + c1, wtot);
        if (lane == 0) wsum[wave] = wtot;
        __syncthreads();
        u32 st0 = ex;
        for (int w = 0; w < wave; w++) st0 += wsum[w];
        const u32 st1 = st0 + c0;
        u32 run = st0 | (st1 << 16);
        if (mine) {
#pragma unroll
            for (int w = 0; w < OP_WAVES; w++) { cw[w][j] = run; run += c[w]; }
            D[2 * j] -= st0; D[2 * j + 1] -= st1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < OP_KPL; k++) {
        const u32 i = beg + 64u * k + lane;
        if (i < end) {
            const u32 sh = 16 * (id[k] & 1);
            const u32 slot = (atomicAdd(&cw[wave][id[k] >> 1], 1u << sh) >> sh) & 0xffffu;      // lane-ordered: earlier lanes get lower places
            S[slot] = i | (id[k] << REL_BITS);
        }
    }
    __syncthreads();
    u32 *out = sorted + td.sorted_off;
    const u32 nk = end - s0;
#pragma unroll
    for (int k = 0; k < OP_KPL; k++) {
        const u32 j = threadIdx.x + (u32)OP_NT * k;
        if (j < nk) { const u32 e = S[j]; out[D[e >> REL_BITS] + j] = e; }
    }
}

// 1 if this device's LDS retires same-address atomics of a wave instruction in lane order (probed once per device)
static int lds_lane_ordered()
{
    static std::mutex mu;
    static int cached[64];
    static bool init = false;
    std::lock_guard<std::mutex> lk(mu);
    if (!init) { for (int i = 0; i < 64; i++) cached[i] = -1; init = true; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] >= 0) return cached[dev];
    u32 *d_bad = nullptr, h_bad = 1;
    if (hipMalloc(&d_bad, 4) != hipSuccess) return 0;
    if (hipMemset(d_bad, 0, 4) == hipSuccess) {
        hipLaunchKernelGGL(k_probe_lds_order, dim3(64), dim3(1024), 0, 0, d_bad, 512);
        if (hipMemcpy(&h_bad, d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) h_bad = 1;
    }
    (void)hipFree(d_bad);
    cached[dev] = h_bad == 0 ? 1 : 0;
    return cached[dev];
}

// test hook: swaps the first two neighbours of the first tile's sorted order that belong to one hash run, i.e. what a failure of the lane-ordered ranking would look like to the stages downstream
__global__ void k_inject_disorder(const u8 *__restrict__ stream, const TileDesc *__restrict__ tiles, u32 *__restrict__ sorted)
{
    const TileDesc td = tiles[0];
    const u8 *s = stream + td.stream_off + td.w;
    u32 *sk = sorted + td.sorted_off;
    for (u32 i = 0; i + 1 < td.wlen; i++) {
        const u32 a = sk[i], b = sk[i + 1];
        if (hash_of(gld_u32_unaligned(s, a & REL_MASK)) == hash_of(gld_u32_unaligned(s, b & REL_MASK)) && (b & REL_MASK) >= td.a - td.w) { sk[i] = b; sk[i + 1] = a; return; }
    }
}

size_t hash_sort_ws_bytes(int n_tiles) { return (size_t)(n_tiles + 1) * sizeof(OpTile); }

int launch_hash_sort(hipStream_t st, const u8 *d_stream, const TileDesc *d_tiles, int n_tiles, u32 *d_tmp, u32 *d_sorted, int force_ballot, void *d_ws)
{
    if (n_tiles == 0) return MTS_OK;
    const int ordered = (force_ballot == 1 || getenv("MTS_SORT_BALLOT")) ? 0 : lds_lane_ordered();      // MTS_SORT_BALLOT=1: force the ballot ranking (tests)
    // the one-pass sort ranks with lane-ordered LDS atomics: without them (or MTS_SORT_TWO_PASS=1: A/B runs, tests) everything is the two-pass kernel's
    static const bool two_pass_only = getenv("MTS_SORT_TWO_PASS") != nullptr;
    if (ordered && d_ws && !two_pass_only) {
        OpTile *ot = (OpTile *)d_ws;
        MTS_HIP(hipMemsetAsync(ot, 0, sizeof(OpTile) * (size_t)n_tiles, st));
        hipLaunchKernelGGL(k_op_live, dim3(op_grid(n_tiles, OP_LIVE_PARTS)), dim3(512), 0, st, d_stream, d_tiles, n_tiles, ot);
        hipLaunchKernelGGL(k_op_prefix, dim3(n_tiles), dim3(1024), 0, st, d_tiles, ot);
        hipLaunchKernelGGL(k_hash_sort, dim3(n_tiles), dim3(SORT_NT), 0, st, d_stream, d_tiles, d_tmp, d_sorted, ordered, (const OpTile *)ot);      // (the tiles left to it: usually none)
        hipLaunchKernelGGL(k_op_count, dim3(op_grid(n_tiles, OP_TILE_SLICES)), dim3(OP_CNT_NT), 0, st, d_stream, d_tiles, n_tiles, (const OpTile *)ot, d_tmp);
        hipLaunchKernelGGL(k_op_scan, dim3(n_tiles), dim3(1024), 0, st, d_tiles, (const OpTile *)ot, d_tmp);
        hipLaunchKernelGGL(k_op_scatter, dim3(op_grid(n_tiles, OP_TILE_SLICES)), dim3(OP_NT), 0, st, d_stream, d_tiles, n_tiles, (const OpTile *)ot, (const u32 *)d_tmp, d_sorted);
    } else
        hipLaunchKernelGGL(k_hash_sort, dim3(n_tiles), dim3(SORT_NT), 0, st, d_stream, d_tiles, d_tmp, d_sorted, ordered, (const OpTile *)nullptr);
    if (force_ballot == 2) hipLaunchKernelGGL(k_inject_disorder, dim3(1), dim3(1), 0, st, d_stream, d_tiles, d_sorted);      // (test hook: damage the order)
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}


// ================================================================================================
// P: lazy-evaluation parse over the tables -- orc_parse_tables() is the oracle
// ================================================================================================
// One step of the state machine from a base state (no pending match) at p0.  Emits the literals
// b[p0 .. mpos-1] followed by match (mlen, mdist) at mpos, or the single literal b[p0] when mlen == 0.
// Returns the next base position.
// the entries of positions p and p + 1 with one 16-byte load (8-byte aligned; the tables have slack past n): nearly
// every step looks at both, and for lanes that are each somewhere else in memory the number of load instructions
// is what the address unit charges for
typedef u32 u32x4_v __attribute__((ext_vector_type(4)));
typedef u32 u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef const __attribute__((address_space(1))) u32 *gptr_u32c;         // (global pointers spelled out: a generic pointer makes flat
typedef const __attribute__((address_space(1))) u32x4_v *gptr_uint4;    //  loads, which count as LDS operations as well)
typedef const __attribute__((address_space(1))) u32x4_a4 *gptr_uint4_a4;
// rd(q) = the table entry of position q;  rdq(q) = its entry in the side table (TE_QSIDE positions only).
// AHEAD: the entries of p0 + 1 .. p0 + 3 are read together with p0's (staged walkers: all four are in the LDS window, and
// one LDS latency per step instead of one per position the lazy evaluation moves on matters when two waves share a SIMD)
template <bool AHEAD, class RDU, class RD, class RDQ>
__device__ __forceinline__ u32 lazy_step(RDU &&rdu /* p0 .. p0 + 3: no bounds check */, RD &&rd, RDQ &&rdq, u32 p0, u32 n, const LevelCfg &cfg, u32 &mpos, u32 &mlen, u32 &mdist,
                                         u32 &mside /* 1: (mlen, mdist) is the side table's entry of mpos, 0: the table's */)
{
    u32 p = p0;
    mside = 0;
    const u32 e0 = rdu(p);
    u32 e1 = 0, e2 = 0, e3 = 0;
    if (AHEAD) { e1 = rdu(p + 1); e2 = rdu(p + 2); e3 = rdu(p + 3); }
    u32 len = te_len(e0), dist = te_dist(e0);
    if (len == MIN_MATCH && dist > (u32)TOO_FAR) len = 0;
    if (len < MIN_MATCH) { mpos = p0; mlen = 0; mdist = 0; return p0 + 1; }
    for (;;) {
        const u32 q = p + 1;
        if (q < n && len < (u32)cfg.lazy) {
            u32 d;
            if (AHEAD) { d = q == p0 + 1 ? e1 : q == p0 + 2 ? e2 : e3; if (q > p0 + 3) d = rd(q); }
            else d = rd(q);
            u32 side = 0;
            if (len >= (u32)cfg.good) {                          // the quarter-budget result
                if (d & TE_QNONE) d = 0;
                else if (d & TE_QSIDE) { d = rdq(q); side = 1; }
            }
            const u32 dl = te_len(d);
            if (dl > len) { p = q; len = dl; dist = te_dist(d); mside = side; continue; }
        }
        break;
    }
    mpos = p; mlen = len; mdist = dist;
    return p + len;
}
// the readers of a walker that goes straight to memory (one lane somewhere in its segment: the short re-walks)
#define MTS_PARSE_GLOBAL_READERS(T, TQ)                                 \
    auto rd = [&](u32 q) -> u32 { return (T)[q]; };                     \
    auto rdq = [&](u32 q) -> u32 { return (TQ)[q]; };

// ------------------------------------------------------------------------------------------------
// Staged walkers.  A wave walks 64 segments, one lane each, and every lane is somewhere else in memory: read entry by entry,
// each load instruction of the wave touched 64 different cache lines (the address unit takes a cycle per line: ~70 cycles per
// load instruction and CU, measured by doubling the loads) and every step of the walk waited for memory (75 % of the wave
// cycles).  So the table is brought in window by window: the wave loads, for every lane, the PARSE_WIN entries that start at
// the lane's current window base -- 16 bytes per lane and instruction with neighbouring lanes on neighbouring addresses,
// eight lanes' windows per instruction -- into LDS, and the lanes then walk PARSE_WIN positions there.
// Round 5: a window is exactly one 128-byte line of the table (PARSE_WIN = 32 entries at a 32-entry boundary).  The lazy
// evaluation looks up to four entries ahead; until round 4 a window therefore held 32 entries and advanced by 28, so that nearly
// every window straddled two lines and -- with 67 MB of windows in flight and the L2 keeping nothing of them -- every line of
// the table was fetched twice (FETCH_SIZE 5.75 GB raw = 2.07 x the table once doubled, §6 of DESIGN.md; 2.86 GB now).  The
// look-ahead of a window's last four positions is the head of the NEXT window, which is in flight while the window is walked:
// the walk of a window stops four positions short, the heads of the next window (the first 16-byte piece of every lane's, by
// then in the registers of the lanes that asked for them) go behind the window in LDS, and the last four positions follow.
// (A ring of two whole windows per lane was built first: 16.6 KB of LDS per wave instead of 9.5, 9 waves per CU instead of 16
// -- and 2.90 ms against 2.47 with half the bytes: what the kernel lives on is waves per CU, not bandwidth.)
// The 64 segments of a wave are consecutive segments of ONE chunk (the grid is laid out per chunk), so the window of lane l
// starts at  base + l * SEG + w * PARSE_WIN: plain arithmetic, no pointer per lane.
// ------------------------------------------------------------------------------------------------
constexpr int PARSE_WIN = 32;                       // entries per window: one 128-byte line per lane
constexpr int PARSE_AHEAD = 4;                      // entries of the next window kept behind it
constexpr int PARSE_ROW = PARSE_WIN + PARSE_AHEAD;  // entries a lane has in LDS
constexpr int PARSE_WIN_PITCH = PARSE_ROW + 1;      // LDS pitch in words: odd, so the lanes' reads fall into different banks
constexpr int PARSE_NWIN = SEG / PARSE_WIN;
constexpr int PARSE_PIECES = PARSE_WIN / 4;         // 16-byte pieces per window = lanes per window = load instructions per window step
static_assert(SEG % PARSE_WIN == 0 && 64 % PARSE_PIECES == 0, "windows tile a segment, eight lanes load one");

struct ParseStage {
    u32 *win;                   // LDS [64][PARSE_WIN_PITCH]
    const u32 *base;            // table entry of the first lane's segment start (a 128-byte boundary)
    int nlanes;                 // lanes that have a segment (the others' windows hold something harmless)
    long floor_off;             // lowest entry (relative to base) a window may start at: the chunk's first
    // ask for window w of every lane: lane l takes piece l % 8 of the windows of lanes it * 8 + l / 8.  head_only: only the first
    // piece (the four entries the last window of a segment looks ahead to)
    __device__ __forceinline__ void request(int w, u32x4_v (&pre)[PARSE_PIECES], bool head_only = false) const
    {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int it = 0; it < PARSE_PIECES; it++) {
            int wl = it * (64 / PARSE_PIECES) + lane / PARSE_PIECES;
            wl = wl < nlanes ? wl : nlanes - 1;
            long off = (long)wl * SEG + (long)w * PARSE_WIN;
            if (off < floor_off) off = floor_off;                  // (a warm-up window of the chunk's first segment: nobody walks it)
            const u32 *src = base + off + 4 * (lane % PARSE_PIECES);
            if (!head_only || lane % PARSE_PIECES == 0) pre[it] = *(gptr_uint4)(u64)src;
        }
    }
    __device__ __forceinline__ void land(const u32x4_v (&pre)[PARSE_PIECES]) const      // the requested window becomes the rows' window
    {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int it = 0; it < PARSE_PIECES; it++) {
            u32 *d = win + (it * (64 / PARSE_PIECES) + lane / PARSE_PIECES) * PARSE_WIN_PITCH + 4 * (lane % PARSE_PIECES);
            d[0] = pre[it].x; d[1] = pre[it].y; d[2] = pre[it].z; d[3] = pre[it].w;
        }
    }
    __device__ __forceinline__ void land_heads(const u32x4_v (&pre)[PARSE_PIECES]) const      // its first piece goes behind the rows' current window
    {
        const int lane = threadIdx.x & 63;
        if (lane % PARSE_PIECES == 0) {
#pragma unroll
            for (int it = 0; it < PARSE_PIECES; it++) {
                u32 *d = win + (it * (64 / PARSE_PIECES) + lane / PARSE_PIECES) * PARSE_WIN_PITCH + PARSE_WIN;
                d[0] = pre[it].x; d[1] = pre[it].y; d[2] = pre[it].z; d[3] = pre[it].w;
            }
        }
    }
};
// readers of a lane whose window starts at position wb (entries wb .. wb + rv - 1 are in the LDS row `row`: rv = PARSE_WIN while
// the heads of the next window are not there yet, PARSE_ROW after)
#define MTS_PARSE_STAGED_READERS(T, TQ, row, wb, rv)                                                              \
    auto rdu = [&](u32 q) -> u32 { return (row)[q - (wb)]; };     /* (q within the row: a step's first four entries are) */ \
    auto rd = [&](u32 q) -> u32 {                                                                                 \
        const u32 o = q - (wb);                                                                                   \
        if (o < (rv)) return (row)[o];                                                                            \
        return *(gptr_u32c)(u64)&(T)[q];                                                                          \
    };                                                                                                            \
    auto rdq = [&](u32 q) -> u32 { return *(gptr_u32c)(u64)&(TQ)[q]; };

__device__ __forceinline__ u64 readlane_u64(u64 v, int k)
{
    return (u64)(u32)__builtin_amdgcn_readlane((int)(u32)v, k) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), k) << 32);
}

// Speculative walk of one segment from `entry`.  Besides the exit (first base position at or beyond the
// segment end) and the token count it records checkpoints: the first base position at or beyond
// s + k * PARSE_CP (k = 1..7) and the tokens emitted before it.  A later walk from a corrected entry only
// has to run until it lands on a recorded checkpoint: from there on the two walks are the same walk.
#ifndef MTS_PARSE_NCP
#define MTS_PARSE_NCP 8
#endif
constexpr int PARSE_NCP = MTS_PARSE_NCP;           // checkpoints per segment + 1 (8: every 128 positions)
#ifndef MTS_PARSE_WARM
#define MTS_PARSE_WARM 2
#endif
constexpr int PARSE_WARM = MTS_PARSE_WARM;         // windows the speculative walk starts before its segment (see k_parse_spec)
constexpr int PARSE_CP = SEG / PARSE_NCP;

// The walks also leave MARKS: two bits per position of what the walk did there -- 0 nothing (inside a copy), 1 a literal, 2 a
// copy taken from the table's entry of the position, 3 a copy taken from the side table's -- in MARK_WORDS words per segment
// (16 positions a word, relative to the segment start; a walk's last token starts less than 256 positions past the segment:
// every lazy step that moves on has found a longer copy).  With the entries and exits settled, the tokens are then written by
// a kernel that does not walk: k_parse_emit_marks.  A re-walk that lands on a checkpoint keeps the marks from there on.
constexpr int MARK_WORDS = (SEG + 288) / 16;
// Word k of segment g lives at marks[((g / 64) * MARK_WORDS + k) * 64 + g % 64]: the 64 walkers of a wave are consecutive segments and write
// the same word at about the same time -- one 256-byte piece instead of 64 lines with 4 bytes each (as many write transactions as
// the speculative walk has reads; it was 0.7 ms slower for them)
__device__ __forceinline__ u32 *mark_base(u32 *marks, u32 g) { return marks + ((size_t)(g >> 6) * MARK_WORDS) * 64 + (g & 63); }
struct MarkW {
    u32 *base;                  // the segment's words: word k at base[64 k] (see mark_base)
    u32 a;                      // index of the lowest word held
    u64 lo, hi;                 // the marks of words a .. a + 3: stores happen where the caller wants them (a store in the walk's
                                // loop is waited for, with everything before it, by the next load that is waited for)
    bool on;
    __device__ __forceinline__ void start(u32 *b, u32 rel, bool on_) { base = b; a = rel >> 4; lo = 0; hi = 0; on = on_; }
    // word i (>= a) becomes the lowest one held; what lies below it goes to memory
    __device__ __forceinline__ void anchor(u32 i)
    {
        if (i <= a) return;
        const u32 d = i - a;
        if (on) {
            base[(size_t)(a) * 64] = (u32)lo;
            if (d > 1) base[(size_t)(a + 1) * 64] = (u32)(lo >> 32);
            if (d > 2) base[(size_t)(a + 2) * 64] = (u32)hi;
            if (d > 3) { base[(size_t)(a + 3) * 64] = (u32)(hi >> 32); for (u32 k = a + 4; k < i; k++) base[(size_t)(k) * 64] = 0; }
        }
        if (d == 1) { lo = (lo >> 32) | (hi << 32); hi >>= 32; }
        else if (d == 2) { lo = hi; hi = 0; }
        else if (d == 3) { lo = hi >> 32; hi = 0; }
        else { lo = 0; hi = 0; }
        a = i;
    }
    __device__ __forceinline__ void orbits(u64 pat, u32 sh)      // pat at bit sh of the 128 held (it fits: the callers see to that)
    {
        if (sh < 64) { lo |= pat << sh; if (sh) hi |= pat >> (64 - sh); }
        else hi |= pat << (sh - 64);
    }
    // a step of the walk: nl literals at rel0 .. rel0 + nl - 1, then a token of `type` at rel0 + nl.  The caller keeps
    // rel0 - 16 a below 48 (k_parse_spec anchors at its window; parse_rewalk when it gets there).
    __device__ __forceinline__ void step(u32 rel0, u32 nl, u32 type)
    {
        if (nl <= 12) {
            const u64 pat = (0x5555555555555555ull & ((1ull << (2 * nl)) - 1ull)) | ((u64)type << (2 * nl));
            orbits(pat, 2 * (rel0 - 16 * a));
        } else {                                                // (the lazy evaluation moved on more than 12 times: levels 8, 9)
            for (u32 q = 0; q <= nl; q++) {
                const u32 r = rel0 + q;
                if (r - 16 * a >= 48) anchor(r >> 4);
                orbits(q < nl ? 1u : type, 2 * (r - 16 * a));
            }
        }
    }
    // the walk ended at rel (its exit): no mark from there on
    __device__ __forceinline__ void finish(u32 rel)
    {
        const u32 last = min(rel ? (rel - 1) >> 4 : 0u, (u32)MARK_WORDS - 1u);       // (a last copy may reach past what the words cover: no token starts there)
        if (!on) return;
        base[(size_t)(a) * 64] = (u32)lo;
        if (last > a) base[(size_t)(a + 1) * 64] = (u32)(lo >> 32);
        if (last > a + 1) base[(size_t)(a + 2) * 64] = (u32)hi;
        if (last > a + 2) base[(size_t)(a + 3) * 64] = (u32)(hi >> 32);
        for (u32 k = a + 4; k <= last; k++) base[(size_t)(k) * 64] = 0;
    }
    // the walk met the one whose marks are there at rel: below rel the marks are this walk's, from rel on they stay
    __device__ __forceinline__ void merge(u32 rel)
    {
        const u32 i = rel >> 4;
        anchor(i);
        const u32 low = (1u << (2 * (rel & 15))) - 1u;
        if (on) base[(size_t)(i) * 64] = ((u32)lo & low) | (base[(size_t)(i) * 64] & ~low);
    }
};
// marks of the step that went from p0 to the token at mp (literals before it)
#define MTS_MARK_STEP(mk, s, p0, mp, ml, mside) (mk).step((p0) - (s), (mp) - (p0), (ml) ? 2u + (mside) : 1u)
#ifndef MTS_PARSE_SPEC_WAVES
#define MTS_PARSE_SPEC_WAVES 4          // waves per SIMD the speculative walk's registers leave room for
#endif

__global__ __launch_bounds__(64, MTS_PARSE_SPEC_WAVES) void k_parse_spec(const u32 *__restrict__ tables, const u32 *__restrict__ quarter, const ChunkDesc *__restrict__ chunks,
                                                   ParseBufs pb, int n_segs, LevelCfg cfg)
{
    __shared__ u32 win[64 * PARSE_WIN_PITCH];
    const int lane = threadIdx.x;
    const ChunkDesc ch = chunks[blockIdx.y];                    // grid: x = wave within the chunk, y = chunk
    const u32 k0seg = blockIdx.x * 64;                          // the wave's first segment within the chunk
    if (k0seg >= ch.nseg) return;
    const int nlanes = (int)min(64u, ch.nseg - k0seg);
    const bool valid = lane < nlanes;
    const int g = (int)(ch.seg0 + k0seg) + (valid ? lane : nlanes - 1);
    const u32 s = (k0seg + (u32)(valid ? lane : nlanes - 1)) * SEG, n = ch.n;
    const u32 segend = valid ? min(s + (u32)SEG, n) : 0;
    const u32 *T = tables + ch.stream_off, *TQ = quarter + ch.stream_off;
    const ParseStage st{win, T + (size_t)k0seg * SEG, nlanes, -(long)k0seg * SEG};
    const u32 *row = win + lane * PARSE_WIN_PITCH;
    u32 *cp = pb.cp + (u64)g * (2 * PARSE_NCP);
    // The walk starts PARSE_WARM windows BEFORE the segment, in a state that is as much a guess as "a token starts at s" was -- but
    // the lazy parse falls into step with the true one within a few tokens, whatever it started from: the first base position at
    // or behind s is the segment's true entry for 98.3 % of the segments after 64 positions (90 % after 32, 99.9 % after 96;
    // 28 % with no warm-up: tools/sim/match_walk_sim.c), and the fix round that follows re-walks the 1.7 % instead of the 72 %.
    // Nothing is recorded before s (no marks, no counts, no checkpoints: that stretch is the walk of the segment before).
    const bool warm = valid && s != 0;
    u32 pos = warm ? s - (u32)(PARSE_WARM * PARSE_WIN) : s, mp, ml, md, ms, cnt = 0, k = 1, entry = s;
    MarkW mk;
    mk.start(mark_base(pb.marks, (u32)g), 0, valid);
    u32x4_v pre[PARSE_PIECES];
    st.request(-PARSE_WARM, pre);
    for (int w = -PARSE_WARM; w < PARSE_NWIN; w++) {
        __syncthreads();                                         // (everybody is done with the window before)
        st.land(pre);
        __syncthreads();
        if (w == 0) entry = pos;                                 // (s, or where the warm-up came out)
        if (w > 0) mk.anchor(((u32)w * PARSE_WIN) >> 4);         // (the marks below this window: stored before the next loads are asked for)
        st.request(w + 1, pre, w + 1 == PARSE_NWIN);             // (of the window behind the segment only what the last one looks ahead to)
        const u32 wb = s + (u32)(w * PARSE_WIN);                 // (a warm-up window of a lane without one: wrapped around, and wend = 0 keeps the lane out)
        const u32 wend = w < 0 ? (warm ? wb + (u32)PARSE_WIN : 0u) : min(wb + (u32)PARSE_WIN, segend);
#pragma unroll 1
        for (int part = 0; part < 2; part++) {                   // all but the last four positions; then those, with the next window's heads behind the row
            const u32 rv = part ? (u32)PARSE_ROW : (u32)PARSE_WIN;
            const u32 lim = part ? wend : min(wend, wb + (u32)(PARSE_WIN - PARSE_AHEAD));
            if (part) {
                st.land_heads(pre);
                __syncthreads();
            }
            MTS_PARSE_STAGED_READERS(T, TQ, row, wb, rv)
            while (pos < lim) {
                while (k < (u32)PARSE_NCP && pos >= s + k * PARSE_CP) { cp[k - 1] = pos; cp[PARSE_NCP + k - 1] = cnt; k++; }
                const u32 p0 = pos;
                pos = lazy_step<true>(rdu, rd, rdq, pos, n, cfg, mp, ml, md, ms);
                if (w >= 0) {
                    cnt += mp - p0 + 1;
                    MTS_MARK_STEP(mk, s, p0, mp, ml, ms);
                }
            }
        }
        if (w >= 0 && !__any(pos < segend)) break;
    }
    if (!valid) return;
    mk.finish(pos - s);
    for (; k < (u32)PARSE_NCP; k++) { cp[k - 1] = pos; cp[PARSE_NCP + k - 1] = cnt; }      // checkpoints past the exit
    pb.entry[g] = entry;
    pb.exit_a[g] = pos;
    pb.cnt[g] = cnt;
}

// One segment walked again from the entry `ne` (the exit of the segment before it); returns its exit.  The walk stops early
// when it lands on a checkpoint of the previous walk (from there on the two are the same walk).
__device__ __forceinline__ u32 parse_rewalk(const u32 *__restrict__ T, const u32 *__restrict__ TQ, const ChunkDesc &ch, ParseBufs &pb, u32 g, u32 ne, u32 old_exit,
                                            LevelCfg cfg)
{
    pb.entry[g] = ne;
    const u32 s = pb.seg_start[g], n = ch.n;
    const u32 segend = min(s + (u32)SEG, n);
    MTS_PARSE_GLOBAL_READERS(T, TQ)
    u32 *cp = pb.cp + (u64)g * (2 * PARSE_NCP);
    const u32 old_cnt = pb.cnt[g];
    u32 pos = ne, mp, ml, md, ms, cnt = 0, k = 1;
    MarkW mk;
    mk.start(mark_base(pb.marks, g), ne - s, true);
    while (pos < segend) {
        bool merged = false;
        while (k < (u32)PARSE_NCP && pos >= s + k * PARSE_CP) {
            if (cp[k - 1] == pos) { merged = true; break; }
            cp[k - 1] = pos; cp[PARSE_NCP + k - 1] = cnt; k++;               // this walk's own checkpoint
        }
        if (merged) {
            // same walk from here on: keep the exit (and the marks), shift the counts of the remaining checkpoints
            const u32 at_old = cp[PARSE_NCP + k - 1];
            for (u32 q = k; q < (u32)PARSE_NCP; q++) cp[PARSE_NCP + q - 1] = cp[PARSE_NCP + q - 1] - at_old + cnt;
            pb.cnt[g] = cnt + (old_cnt - at_old);
            mk.merge(pos - s);
            return old_exit;
        }
        const u32 p0 = pos;
        pos = lazy_step<false>(rd, rd, rdq, pos, n, cfg, mp, ml, md, ms);
        cnt += mp - p0 + 1;
        if (p0 - s - 16 * mk.a >= 48) mk.anchor((p0 - s) >> 4);
        MTS_MARK_STEP(mk, s, p0, mp, ml, ms);
    }
    mk.finish(pos - s);
    for (; k < (u32)PARSE_NCP; k++) { cp[k - 1] = pos; cp[PARSE_NCP + k - 1] = cnt; }
    pb.cnt[g] = cnt;
    return pos;
}

__global__ __launch_bounds__(64) void k_parse_fix(const u32 *__restrict__ tables, const u32 *__restrict__ quarter, const ChunkDesc *__restrict__ chunks,
                                                  ParseBufs pb, int n_segs, LevelCfg cfg, int round)
{
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= n_segs) return;
    const u32 *exit_in = (round & 1) ? pb.exit_b : pb.exit_a;
    u32 *exit_out = (round & 1) ? pb.exit_a : pb.exit_b;
    const u32 ci = pb.seg_chunk[g];
    const ChunkDesc ch = chunks[ci];
    const u32 old_exit = exit_in[g];
    const bool first = (u32)g == ch.seg0;
    const u32 ne = first ? 0 : exit_in[g - 1];
    if (ne == pb.entry[g]) { exit_out[g] = old_exit; return; }
    const u32 e = parse_rewalk(tables + ch.stream_off, quarter + ch.stream_off, ch, pb, (u32)g, ne, old_exit, cfg);
    exit_out[g] = e;
    if (e != old_exit) *pb.changed = 1;
}

// Data whose parse never re-synchronises (a run of zeros is one 258-byte match after the other, in whatever phase the walk
// starts) would need one parallel round per segment.  After PARSE_PARALLEL_ROUNDS rounds the rest is done in order
// instead, each segment from the exit of the one before; a single pass is exact.
__global__ __launch_bounds__(64) void k_parse_fix_serial(const u32 *__restrict__ tables, const u32 *__restrict__ quarter, const ChunkDesc *__restrict__ chunks,
                                                         ParseBufs pb, int n_chunks, LevelCfg cfg, u32 *__restrict__ exits)
{
    // one wave per chunk: 64 segments at a time are checked for an entry that is not the exit before it; the first such
    // segment is walked again, and so is what follows it for as long as the exits keep changing
    const int ci = blockIdx.x, lane = threadIdx.x;
    if (ci >= n_chunks) return;
    const ChunkDesc ch = chunks[ci];
    const u32 *T = tables + ch.stream_off, *TQ = quarter + ch.stream_off;
    const u32 g_end = ch.seg0 + ch.nseg;
    u32 g0 = ch.seg0;
    while (g0 < g_end) {
        const u32 g = g0 + lane;
        bool wrong = false;
        if (g < g_end) wrong = ((g == ch.seg0) ? 0u : exits[g - 1]) != pb.entry[g];
        const u64 m = __ballot(wrong);
        if (m == 0) { g0 += 64; continue; }
        u32 gg = g0 + (u32)__ffsll((unsigned long long)m) - 1;
        if (lane == 0) {
            for (; gg < g_end; gg++) {
                const u32 ne = gg == ch.seg0 ? 0u : exits[gg - 1];
                if (ne == pb.entry[gg]) break;                   // consistent again: back to scanning
                const u32 old_exit = exits[gg];
                exits[gg] = parse_rewalk(T, TQ, ch, pb, gg, ne, old_exit, cfg);
            }
        }
        gg = (u32)__shfl((int)gg, 0);
        __threadfence_block();
        g0 = gg;                                               // (gg is consistent, or the end)
    }
}

// exclusive scan of the per-segment token counts of each chunk (one workgroup per chunk)
constexpr int SEG_SCAN_NT = 1024, SEG_SCAN_PER = 8;   // a thread takes SEG_SCAN_PER consecutive segments a trip (256 threads, one each: 88 trips of three barriers per chunk, 83 us)
__global__ __launch_bounds__(SEG_SCAN_NT) void k_seg_scan(const ChunkDesc *__restrict__ chunks, ParseBufs pb, ChunkOut *cout)
{
    const ChunkDesc ch = chunks[blockIdx.x];
    __shared__ u32 wsum[SEG_SCAN_NT / 64];
    __shared__ u32 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (u32 base = 0; base < ch.nseg; base += SEG_SCAN_NT * SEG_SCAN_PER) {
        const u32 k0 = base + threadIdx.x * SEG_SCAN_PER;
        u32 v[SEG_SCAN_PER], mine = 0;
#pragma unroll
        for (int j = 0; j < SEG_SCAN_PER; j++) v[j] = k0 + j < ch.nseg ? pb.cnt[ch.seg0 + k0 + j] : 0;
#pragma unroll
        for (int j = 0; j < SEG_SCAN_PER; j++) mine += v[j];
        u32 tot;
        u32 ex = wave_excl_scan_u32(mine, tot);
        if (lane == 0) wsum[wave] = tot;
        __syncthreads();
        u32 add = carry_s, all = 0;
        for (int w = 0; w < SEG_SCAN_NT / 64; w++) { if (w < wave) add += wsum[w]; all += wsum[w]; }
        u32 run = ex + add;
#pragma unroll
        for (int j = 0; j < SEG_SCAN_PER; j++) { if (k0 + j < ch.nseg) pb.tokbase[ch.seg0 + k0 + j] = run; run += v[j]; }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cout[blockIdx.x].ntok = carry_s;
        if (ch.n == 0) cout[blockIdx.x].trailing = 0;
    }
}

// The tokens from the marks (see MarkW): a wave per segment, a lane per word of 16 positions.  The segment's first token index
// is known (k_seg_scan), a wave scan of the words' token counts gives every lane its own, and a token is the stream's byte or
// the table's entry at the marked position: nothing is walked.  (The walking version above read the table a second time entry
// by entry through LDS windows, 10 waves per CU: 3.5 ms.)
__global__ __launch_bounds__(256) void k_parse_emit_marks(const u8 *__restrict__ stream, const u32 *__restrict__ tables, const u32 *__restrict__ quarter,
                                                          const ChunkDesc *__restrict__ chunks, ParseBufs pb, const u32 *__restrict__ exits,
                                                          u32 *__restrict__ tokens, u32 *__restrict__ blk_in_start, ChunkOut *__restrict__ cout)
{
    __shared__ u32 tokst[4][64 * 16];                           // a wave's tokens of one pass, in order: they leave by consecutive lanes
    const u32 ci = blockIdx.y;                                  // grid: x = four segments of the chunk (see below), y = chunk
    const ChunkDesc ch = chunks[ci];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // The marks of 64 consecutive segments are interleaved (mark_base): a 128-byte line of them belongs to 32 segments, i.e. to
    // eight of these workgroups -- which, numbered in a row, are one on each XCD: every XCD's L2 fetched every line of marks for
    // itself (FETCH_SIZE 2 x 4.94 GB where table, stream and marks are 7.4: round 5).  So the 16 workgroups of a group of 64
    // segments are put on ONE XCD, one after the other: block 8 j + c (XCD c: the grid's width is a multiple of 128) takes
    // quad j % 16 of segment group (j / 16) * 8 + c.  FETCH_SIZE 2 x 3.65 GB = what the kernel needs -- and the same 1.55-1.65 ms:
    // the lines the other XCDs had fetched came out of the Infinity Cache, not out of HBM; the kernel moves its 8.5 GB at 5.5 TB/s
    // either way.  (All of a segment's loads asked for before any is waited for, independent of its marks: 96 registers, 20 waves
    // per CU instead of 32, 1.83 ms -- built, measured, not kept.)
    const u32 xj = blockIdx.x >> 3, xc = blockIdx.x & 7;
    const u32 seg = ((((xj >> 4) * 8 + xc) * 16 + (xj & 15)) * 4 + (u32)wave) - (ch.seg0 & 63);      // (the chunk's first segment need not start a group: segment numbers count from the group's start)
    if (seg >= ch.nseg) return;                                 // (also what lies before the chunk's first segment: a huge number)
    const u32 g = ch.seg0 + seg, s = seg * SEG;
    const u32 entry = pb.entry[g], ex = exits[g], k0 = pb.tokbase[g], ntok = cout[ci].ntok;
    const u32 *T = tables + ch.stream_off, *TQ = quarter + ch.stream_off;
    const u8 *b = stream + ch.stream_off;
    u32 *tk = tokens + ch.tok_off;
    u32 *bis = blk_in_start + ch.blk0;
    const u32 *mw = mark_base(pb.marks, g);
    u32 *mine = tokst[wave];
    const u32 bnd = (k0 + (u32)BLOCK_TOKENS - 1) / (u32)BLOCK_TOKENS * (u32)BLOCK_TOKENS;      // the block start at or after k0 (a segment meets one at most)
    u32 kbase = k0;
    for (u32 w0 = 0; w0 < (u32)MARK_WORDS; w0 += 64) {
        if (s + 16 * w0 >= ex) break;
        const u32 wi = w0 + lane, lo = s + 16 * wi;
        u32 w = 0;
        if (wi < (u32)MARK_WORDS && lo < ex && lo + 16 > entry) {
            w = mw[(size_t)wi * 64];
            if (entry > lo) w &= ~0u << (2 * (entry - lo));                 // only [entry, exit) is this segment's
            if (ex < lo + 16) w &= (1u << (2 * (ex - lo))) - 1u;
        }
        // the word's 16 table entries and 16 stream bytes, asked for together (the tokens are made from registers below)
        u32 e[16], bb[4];
#pragma unroll
        for (int q = 0; q < 16; q++) e[q] = 0;
        bb[0] = bb[1] = bb[2] = bb[3] = 0;
        if (w) {
            const u32x4_v t0 = *(gptr_uint4)(u64)(T + lo), t1 = *(gptr_uint4)(u64)(T + lo + 4), t2 = *(gptr_uint4)(u64)(T + lo + 8),
                          t3 = *(gptr_uint4)(u64)(T + lo + 12), sb = *(gptr_uint4)(u64)(b + lo);
            e[0] = t0.x; e[1] = t0.y; e[2] = t0.z; e[3] = t0.w; e[4] = t1.x; e[5] = t1.y; e[6] = t1.z; e[7] = t1.w;
            e[8] = t2.x; e[9] = t2.y; e[10] = t2.z; e[11] = t2.w; e[12] = t3.x; e[13] = t3.y; e[14] = t3.z; e[15] = t3.w;
            bb[0] = sb.x; bb[1] = sb.y; bb[2] = sb.z; bb[3] = sb.w;
        }
        const u32 st = (w | (w >> 1)) & 0x55555555u;
        const u32 c = (u32)__builtin_popcount(st);
        const u32 incl = wave_incl_scan_dpp(c);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        u32 idx = incl - c;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const u32 m = (w >> (2 * q)) & 3;
            if (m) {
                u32 v;
                if (m == 1) v = ((bb[q >> 2] >> (8 * (q & 3))) & 0xffu) << 16;
                else { const u32 en = m == 3 ? TQ[lo + q] : e[q]; v = (((en >> 15) & 0xffu) << 16) | (en & TE_DIST); }
                mine[idx] = v;
                const u32 kk = kbase + idx;
                if (kk == bnd) bis[kk / (u32)BLOCK_TOKENS] = lo + q;            // first token of a block: where its input starts
                if (kk + 1 == ntok) cout[ci].trailing = m == 1 ? 1u : 0u;       // last token of the chunk
                idx++;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (u32 t = lane; t < total; t += 64) tk[kbase + t] = mine[t];
        __builtin_amdgcn_wave_barrier();
        kbase += total;
    }
}

size_t parse_cp_words() { return 2 * (size_t)PARSE_NCP; }
size_t parse_marks_words(size_t n_segs) { return (n_segs + 64) / 64 * 64 * MARK_WORDS; }

int launch_parse_emit_marks(hipStream_t st, const u8 *d_stream, const u32 *d_tables, const u32 *d_quarter, const ChunkDesc *d_chunks,
                            ParseBufs pb, int rounds_done, u32 *d_tokens, u32 *d_blk_in_start, ChunkOut *d_cout, int n_chunks, u32 max_nseg)
{
    if (max_nseg == 0) return MTS_OK;
    const u32 *exits = ((rounds_done - 1) & 1) ? pb.exit_a : pb.exit_b;      // where the last fix round left the exits
    // (width: quads of 4 segments, 63 segments of slack for a chunk whose first segment is not the first of its group of 64, rounded up to 16 quads on each of 8 XCDs)
    hipLaunchKernelGGL(k_parse_emit_marks, dim3((max_nseg + 63 + 3) / 4 / 128 * 128 + 128, n_chunks), dim3(256), 0, st, d_stream, d_tables, d_quarter, d_chunks, pb, exits,
                       d_tokens, d_blk_in_start, d_cout);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

int launch_parse_spec(hipStream_t st, const u32 *d_tables, const u32 *d_quarter, const ChunkDesc *d_chunks, ParseBufs pb, int n_segs,
                      LevelCfg cfg, int n_chunks, u32 max_nseg)
{
    if (n_segs == 0) return MTS_OK;
    hipLaunchKernelGGL(k_parse_spec, dim3((max_nseg + 63) / 64, n_chunks), dim3(64), 0, st, d_tables, d_quarter, d_chunks, pb, n_segs, cfg);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}
int launch_parse_fix(hipStream_t st, const u32 *d_tables, const u32 *d_quarter, const ChunkDesc *d_chunks, ParseBufs pb, int n_segs,
                     LevelCfg cfg, int round)
{
    if (n_segs == 0) return MTS_OK;
    hipLaunchKernelGGL(k_parse_fix, dim3((n_segs + 63) / 64), dim3(64), 0, st, d_tables, d_quarter, d_chunks, pb, n_segs, cfg, round);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}
int launch_parse_fix_serial(hipStream_t st, const u32 *d_tables, const u32 *d_quarter, const ChunkDesc *d_chunks, ParseBufs pb, int n_chunks, LevelCfg cfg,
                            int rounds_done)
{
    if (n_chunks == 0) return MTS_OK;
    u32 *exits = ((rounds_done - 1) & 1) ? pb.exit_a : pb.exit_b;        // where the last parallel round left the exits
    hipLaunchKernelGGL(k_parse_fix_serial, dim3(n_chunks), dim3(64), 0, st, d_tables, d_quarter, d_chunks, pb, n_chunks, cfg, exits);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}
int launch_parse_count(hipStream_t st, const u32 *d_tables, const ChunkDesc *d_chunks, ParseBufs pb, int n_segs,
                       int n_chunks, LevelCfg cfg, ChunkOut *d_cout)
{
    (void)d_tables; (void)n_segs; (void)cfg;      // the counts come out of the spec/fix walks
    hipLaunchKernelGGL(k_seg_scan, dim3(n_chunks), dim3(SEG_SCAN_NT), 0, st, d_chunks, pb, d_cout);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}
// ================================================================================================
// F: levels 1..3 (zlib's deflate_fast) -- orc_deflate()'s fast branch is the oracle
// ================================================================================================
// deflate_fast is greedy (no lazy evaluation) and it does NOT enter the inside of a match longer than max_insert_length
// (= the level's `lazy` field) into the hash chains, so the chains -- unlike deflate_slow's -- depend on the parse.  What
// stays parse independent is the ORDER: the same hash sort as for the other levels lists every position of a hash run
// newest-last, a position finds its own slot through an inverse map (k_inverse_map), and the older members of its run with
// the common prefix each would give can be listed for every position at once (k_fast_cands).  Which of them the parse
// entered into the chain is one bit per position that only the parse so far can tell: the state of deflate_fast is the exact
// insertion pattern of the last 32 KiB, and a chunk is a sequential computation.  k_fast_seq walks every chunk in order
// with one wave, speculating only inside the 64 positions the wave looks at together.
// (Round 2 first had speculative walks of 1024-position segments iterated to the fixed point: wrong guesses about the
// insertion pattern healed in ~450 rounds at levels 1 and 3 and not at all at level 2 on the 385-channel workload.)
// (also checks what the walks rely on -- positions ascending inside a hash run -- like k_match5 does for the other levels)
__global__ __launch_bounds__(256) void k_inverse_map(const u8 *__restrict__ stream, const TileDesc *__restrict__ tiles, int n_tiles,
                                                     const u32 *__restrict__ sorted, u32 *__restrict__ inv, u32 *__restrict__ flags)
{
    for (int ti = blockIdx.y; ti < n_tiles; ti += gridDim.y) {
        const TileDesc td = tiles[ti];
        const u32 halo = td.a - td.w;
        const u8 *s = stream + td.stream_off + td.w;
        for (u32 i = blockIdx.x * 256 + threadIdx.x; i < td.wlen; i += gridDim.x * 256) {
            const u32 rel = sorted[td.sorted_off + i] & REL_MASK;
            if (rel >= halo) inv[td.stream_off + td.w + rel] = i;
            if (i > 0) {
                const u32 prev = sorted[td.sorted_off + i - 1] & REL_MASK;
                if (prev >= rel && hash_of(gld_u32_unaligned(s, prev)) == hash_of(gld_u32_unaligned(s, rel))) atomicOr(flags, 1u);
            }
        }
    }
}

// common prefix of the strings at a and b (a > b), at most maxlen; the streams are padded, so 4-byte reads past n are fine
__device__ __forceinline__ u32 common_len(const u8 *__restrict__ s, u32 a, u32 b, u32 maxlen)
{
    u32 len = 0;
    while (len < maxlen) {
        const u32 x = gld_u32_unaligned(s, (u64)a + len) ^ gld_u32_unaligned(s, (u64)b + len);
        if (x) { len += (u32)__builtin_ctz(x) >> 3; break; }
        len += 4;
    }
    return len < maxlen ? len : maxlen;
}

// One wave per chunk, 64 positions (a window) at a time:
//   * every lane decides for its own position, as if it were a decision point, from its list: members before the window are
//     looked up in a bitmap of the last 32 KiB kept in LDS (final), members inside the window count as inserted unless the
//     wave already knows better (`nonins`); the lane remembers which in-window members it relied on (`dep`).  The decision
//     is mask arithmetic: E = members that count, C = the first `chain` of them, X = C up to the first nice one, the result
//     the maximum of the keys in X;
//   * a scalar loop hops from decision point to decision point with v_readlane (about ten scalar instructions per plain
//     token).  A match longer than max_insert_length leaves its inside out of the chains: lanes behind it that relied on
//     one of those positions are marked dirty and decide again when the loop reaches one of them (then exactly:
//     everything before it is final);
//   * a list that ends before the decision does (long runs of skipped members, more than FQ_KW members inside the window,
//     a member at exactly MAX_DIST) is finished by all 64 lanes scanning the sorted run itself (fast_exact);
//   * the window's tokens are written side by side, its insertion bits go into the LDS bitmap.
// The lists are made in phases of W positions per chunk so that they fit a fixed workspace whatever the batch; the walk's
// state (position, token count, bitmap) is parked in memory between phases.
constexpr u32 FQ_NEED = 1u << 10, FQ_LONG = 1u << 9;
constexpr int FQ_KW = 8;                        // list members inside the window a lane follows itself (more: fast_exact)

struct FastSeqState { u32 pos, k, carry_long, pad; u32 ring[1024]; };

template <int K> struct FqMask { typedef u32 type; };
template <> struct FqMask<48> { typedef u64 type; };
template <int K> constexpr int fq_rows() { return K + (K > 32 ? 3 : 2); }      // K members + N mask word(s) + header

// List member k of position p: len << 22 | (63 - k) << 16 | dist -- the maximum over a set of members is the longest, among
// equals the first, and carries its distance.  Behind the members: the mask of those with len >= nice_match and a header
// (members | members inside the 64-aligned window << 8 | "the list is not the whole story" << 16).
template <int K>
__global__ __launch_bounds__(256) void k_fast_cands(const u8 *__restrict__ stream, const ChunkDesc *__restrict__ chunks, const TileDesc *__restrict__ tiles,
                                                    const u32 *__restrict__ sorted, const u32 *__restrict__ inv, u32 *__restrict__ tab, u32 W, u32 phase,
                                                    LevelCfg cfg)
{
    constexpr int R = fq_rows<K>();
    const u32 bpc = W / 256;                                      // blocks per chunk (a 1-D grid: any number of chunks)
    const u32 ci = blockIdx.x / bpc;
    const ChunkDesc ch = chunks[ci];
    const u32 pl = (blockIdx.x - ci * bpc) * 256 + threadIdx.x;
    const u64 p64 = (u64)phase * W + pl;
    if (p64 >= ch.n) return;
    const u32 p = (u32)p64, n = ch.n, look = n - p, lane = pl & 63;
    u32 *row = tab + (((u64)ci * (W / 64) + pl / 64) * R) * 64 + lane;
    if (look < (u32)MIN_MATCH) { row[(R - 1) * 64] = 0; return; }
    const u8 *s = stream + ch.stream_off;
    const TileDesc &td = tiles[ch.tile0 + p / TILE];
    const u32 *sk = sorted + td.sorted_off;
    const u32 j = inv[ch.stream_off + p];
    const u32 own4 = gld_u32_unaligned(s, p), myhash = hash_of(own4);
    const u32 maxlen = look < (u32)MAX_MATCH ? look : (u32)MAX_MATCH;
    const u32 nice = (u32)cfg.nice < look ? (u32)cfg.nice : look;
    u32 q[K], c4[K];
#pragma unroll
    for (int k = 0; k < K; k++) q[k] = (u32)k < j ? td.w + (sk[j - 1 - k] & REL_MASK) : 0xffffffffu;
#pragma unroll
    for (int k = 0; k < K; k++) c4[k] = q[k] != 0xffffffffu ? gld_u32_unaligned(s, q[k]) : 0;
    bool run = true;
    u32 nlist = 0, kin = 0, inc = 0;
    u64 nm = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (run) {
            const u32 dist = p - q[k];
            // the run ends, the window ends, or position 0 (zlib's NIL): nothing behind it counts
            if (q[k] == 0xffffffffu || hash_of(c4[k]) != myhash || dist > (u32)MAX_DIST || dist == p) run = false;
            else if (dist == (u32)MAX_DIST) { run = false; inc = 1; }      // (allowed as the chain's head only: fast_exact knows)
            else {
                const u32 x0 = c4[k] ^ own4;
                u32 len = x0 ? (u32)__builtin_ctz(x0) >> 3 : 4 + common_len(s, p + 4, q[k] + 4, maxlen > 4 ? maxlen - 4 : 0);
                len = len < maxlen ? len : maxlen;
                row[k * 64] = (len << 22) | ((u32)(63 - k) << 16) | dist;
                nlist = k + 1;
                if (dist <= lane) kin = k + 1;
                if (len >= nice) nm |= 1ull << k;
            }
        }
    }
    if (run) inc = 1;                                             // K members and the run goes on
    row[K * 64] = (u32)nm;
    if (K > 32) row[(K + 1) * 64] = (u32)(nm >> 32);
    row[(R - 1) * 64] = nlist | (kin << 8) | (inc << 16);
}

// One position's decision from its list, given which members count as inserted (E).  X: the members the chain walk looks at.
template <int K>
__device__ __forceinline__ void fast_decide(const u32 (&c)[K], typename FqMask<K>::type oldE, typename FqMask<K>::type listm, typename FqMask<K>::type nicem,
                                            u32 kin, u32 inc, const u32 (&bk)[FQ_KW], u64 nonins, u32 look, const LevelCfg &cfg,
                                            u32 &vbest, u32 &vdist, u64 &dep, u32 &vinfo)
{
    typedef typename FqMask<K>::type M;
    // members inside the window: inserted unless the wave knows better
    u32 notw = 0;
#pragma unroll
    for (int j = 0; j < FQ_KW; j++) notw |= ((u32)(nonins >> bk[j]) & 1u) << j;
    const u32 kw = kin < (u32)FQ_KW ? kin : (u32)FQ_KW;
    const M E = oldE | ((M)(~notw & ((1u << kw) - 1)) & listm);
    // the first `chain` of them ...
    M C;
    if (K > 32) {
        C = E;
        while (__popcll((u64)C) > cfg.chain) C &= ~((M)1 << (63 - __builtin_clzll((u64)C)));
    } else {
        M r = E;
        for (int i = 0; i < cfg.chain; i++) r &= r - 1;
        C = E ^ r;
    }
    // ... up to and including the first that is nice enough
    const M En = C & nicem;
    const M X = En ? C & (En ^ (En - 1)) : C;
    const bool ended = En != 0 || (K > 32 ? __popcll((u64)X) : __popc((u32)X)) == cfg.chain;
    u32 key = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const u32 m = (u32)((X >> k) & 1);
        key = max(key, c[k] & (0u - m));
    }
    u64 d = 0;
    const u32 xw = (u32)X & ((1u << kw) - 1);
#pragma unroll
    for (int j = 0; j < FQ_KW; j++) d |= (u64)((xw >> j) & 1u) << bk[j];
    const u32 best = key >> 22;
    const bool is_match = best >= (u32)MIN_MATCH;
    vbest = is_match ? best : (u32)MIN_MATCH - 1; vdist = key & 0xffff; dep = d;
    const bool lng = is_match && !(best <= (u32)cfg.lazy && look - best >= (u32)MIN_MATCH);
    const bool need = kin > (u32)FQ_KW || (!ended && inc);
    vinfo = (is_match ? best : 1u) | (lng ? FQ_LONG : 0u) | (need ? FQ_NEED : 0u);
}

// the decision at pq by the whole wave from the sorted run itself (everything before pq is final)
__device__ __forceinline__ void fast_exact(const u8 *__restrict__ s, const ChunkDesc &ch, const TileDesc *__restrict__ tiles, const u32 *__restrict__ sorted,
                                           const u32 *__restrict__ inv, const u32 *ring, u32 ws, u64 nonins, u32 pq, const LevelCfg &cfg, u32 lane,
                                           u32 &obest, u32 &odist)
{
    const u32 n = ch.n, look = n - pq;
    const u32 maxlen = look < (u32)MAX_MATCH ? look : (u32)MAX_MATCH;
    const u32 nice = (u32)cfg.nice < look ? (u32)cfg.nice : look;
    const TileDesc &td = tiles[ch.tile0 + pq / TILE];
    const u32 *sk = sorted + td.sorted_off;
    const u32 tw = td.w;
    u32 j = (u32)__builtin_amdgcn_readfirstlane((int)inv[ch.stream_off + pq]);      // (uniform: the scalar loop of the caller stays scalar)
    const u32 own4 = (u32)__builtin_amdgcn_readfirstlane((int)gld_u32_unaligned(s, pq)), myhash = hash_of(own4);
    const bool farp = pq > (u32)MAX_DIST;
    u32 best = MIN_MATCH - 1, bdist = 0, examined = 0;
    bool done = false;
    while (!done && j > 0) {
        const bool have = lane < j;
        u32 qq = 0, c4 = 0;
        if (have) { qq = tw + (sk[j - 1 - lane] & REL_MASK); c4 = gld_u32_unaligned(s, qq); }
        const u32 dist = pq - qq;
        const bool ends = !have || hash_of(c4) != myhash || dist > (u32)MAX_DIST;
        const u64 em = __ballot(ends);
        const u32 cut = em ? (u32)__ffsll((long long)em) - 1 : 64u;
        bool ins = false;
        if (lane < cut) ins = qq >= ws ? !((nonins >> (qq - ws)) & 1) : (ring[(qq >> 5) & 1023] >> (qq & 31)) & 1;
        u32 len = 0;
        if (ins) {
            const u32 x0 = c4 ^ own4;
            len = x0 ? (u32)__builtin_ctz(x0) >> 3 : 4 + common_len(s, pq + 4, qq + 4, maxlen > 4 ? maxlen - 4 : 0);
            len = len < maxlen ? len : maxlen;
        }
        u64 im = __ballot(ins);
        while (im && !done) {
            const int i = __ffsll((long long)im) - 1;
            im &= im - 1;
            const u32 d = (u32)__builtin_amdgcn_readlane((int)dist, i), l = (u32)__builtin_amdgcn_readlane((int)len, i);
            const bool stop = (examined == 0 || !farp) ? d == pq : d >= (u32)MAX_DIST;
            if (stop) { done = true; break; }
            examined++;
            if (l > best) { best = l; bdist = d; if (l >= nice) done = true; }
            if (examined == (u32)cfg.chain) done = true;
        }
        if (cut < 64) done = true;
        j = j > 64 ? j - 64 : 0;
    }
    obest = (u32)__builtin_amdgcn_readfirstlane((int)best); odist = (u32)__builtin_amdgcn_readfirstlane((int)bdist);
}

template <int K>
__global__ __launch_bounds__(64) void k_fast_seq(const u8 *__restrict__ stream, const ChunkDesc *__restrict__ chunks, const TileDesc *__restrict__ tiles,
                                                 const u32 *__restrict__ sorted, const u32 *__restrict__ inv, const u32 *__restrict__ tab, u32 W, u32 phase,
                                                 FastSeqState *__restrict__ state, LevelCfg cfg, u32 *__restrict__ tokens, u32 *__restrict__ blk_in_start,
                                                 ChunkOut *__restrict__ cout)
{
    __shared__ u32 ring[1024];
    const u32 ci = blockIdx.x, lane = threadIdx.x;
    const ChunkDesc ch = chunks[ci];
    const u32 n = ch.n;
    const u64 pb64 = (u64)phase * W;
    if (pb64 >= n) return;
    const u32 p_begin = (u32)pb64, p_end = (u64)p_begin + W < n ? p_begin + W : n;
    const u8 *s = stream + ch.stream_off;
    FastSeqState *S = state + ci;
    u32 pos = 0, k = 0, carry_long = 0;
    if (phase == 0) { for (int i = lane; i < 1024; i += 64) ring[i] = 0; }
    else {
        pos = S->pos; k = S->k; carry_long = S->carry_long;
        for (int i = lane; i < 1024; i += 64) ring[i] = S->ring[i];
    }
    __syncthreads();
    pos = (u32)__builtin_amdgcn_readfirstlane((int)pos); k = (u32)__builtin_amdgcn_readfirstlane((int)k);
    carry_long = (u32)__builtin_amdgcn_readfirstlane((int)carry_long);
    u32 *tk = tokens + ch.tok_off;
    u32 *bis = blk_in_start + ch.blk0;
    constexpr int R = fq_rows<K>();
    typedef typename FqMask<K>::type M;
    const u8 *ringb = (const u8 *)ring;
    const u32 *trow = tab + ((u64)ci * (W / 64)) * R * 64 + lane;
    // the lists of G windows at a time: fetched into registers a group ahead (one wave alone hides no latency), parked in
    // LDS while the group is worked on
    constexpr int G = K <= 12 ? 4 : 2;
    __shared__ u32 stage[G][R + 1][64];
    u32 pre[G][R + 1];
    auto fetch = [&](u32 wsg) {
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int i = 0; i < R; i++) pre[g][i] = trow[(g * R + i) * 64];
            pre[g][R] = wsg + g * 64 + lane < n ? s[wsg + g * 64 + lane] : 0;
        }
        trow += G * R * 64;
    };
    fetch(p_begin);
    for (u32 wsg = p_begin; wsg < p_end; wsg += 64 * G) {
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int i = 0; i <= R; i++) stage[g][i][lane] = pre[g][i];
    if (wsg + 64 * G < p_end) fetch(wsg + 64 * G);
#pragma unroll 1
    for (u32 g = 0; g < (u32)G; g++) {
        const u32 ws = wsg + g * 64;
        if (ws >= p_end) break;
        const u32 p = ws + lane;
        const u32 look = p < n ? n - p : 0;
        const bool valid = look >= (u32)MIN_MATCH;
        const u32 byte = stage[g][R][lane];
        const u32 lim = n - ws < 64 ? n - ws : 64;
        const u32 e0 = (u32)__builtin_amdgcn_readfirstlane((int)(pos - ws));
        const u64 vm = __ballot(valid);
        u64 nonins = ~vm;
        if (carry_long) nonins |= e0 >= 64 ? ~0ull : ((1ull << e0) - 1);
        u64 B = 0;
        u32 vbest = MIN_MATCH - 1, vdist = 0;
        if (e0 < lim) {
            u32 c[K];
#pragma unroll
            for (int i = 0; i < K; i++) c[i] = stage[g][i][lane];
            const u32 hdr = valid ? stage[g][R - 1][lane] : 0;
            const u32 nlist = hdr & 0xff, kin = (hdr >> 8) & 0xff, inc = (hdr >> 16) & 1;
            const M listm = nlist >= (u32)(8 * sizeof(M)) ? ~(M)0 : ((M)1 << nlist) - 1;
            M nicem = stage[g][K][lane];
            if (K > 32) nicem |= (M)((u64)stage[g][K + 1][lane] << 32);
            nicem &= listm;
            u32 bk[FQ_KW];
#pragma unroll
            for (int j = 0; j < FQ_KW; j++) bk[j] = (lane - (c[j] & 0xffff)) & 63;
            // members before the window: the bitmap of the last 32 KiB has them
            M oldE = 0;
#pragma unroll
            for (int i = 0; i < K; i++) {
                const u32 q = p - (c[i] & 0xffff);
                oldE |= (M)((ringb[(q >> 3) & 4095] >> (q & 7)) & 1) << i;
            }
            const u32 kw = kin < (u32)FQ_KW ? kin : (u32)FQ_KW;
            oldE &= listm & ~(M)((1u << kw) - 1);
            u64 dep; u32 vinfo;
            fast_decide<K>(c, oldE, listm, nicem, kin, inc, bk, nonins, look, cfg, vbest, vdist, dep, vinfo);
            u32 q = (u32)__builtin_amdgcn_readfirstlane((int)e0), lastlong = 0;
            u64 slow = __ballot(vinfo > 0x1ffu);                // lanes whose token is not a plain one: a long match, fast_exact, dirty
            u64 dirty = 0;                                      // lanes that relied on a position a later long match left out
            for (;;) {
                // literals and short matches: a handful of scalar instructions each
                // (written out: the compiler's version of this loop is 13 instructions and two taken branches per token, and one wave
                // alone on its SIMD pays for every one of them)
                {
                    u32 xp;
                    asm volatile("s_bitcmp1_b64 %[slow], %[q]\n\t"
                                 "s_cbranch_scc1 2f\n"
                                 "1:\n\t"
                                 "v_readlane_b32 %[x], %[vinfo], %[q]\n\t"
                                 "s_bitset1_b64 %[B], %[q]\n\t"
                                 "s_mov_b32 %[ll], 0\n\t"
                                 "s_add_u32 %[q], %[q], %[x]\n\t"
                                 "s_cmp_ge_u32 %[q], %[lim]\n\t"
                                 "s_cbranch_scc1 2f\n\t"
                                 "s_bitcmp1_b64 %[slow], %[q]\n\t"
                                 "s_cbranch_scc0 1b\n"
                                 "2:\n"
                                 : [q] "+s"(q), [B] "+s"(B), [x] "=&s"(xp), [ll] "+s"(lastlong)
                                 : [vinfo] "v"(vinfo), [slow] "s"(slow), [lim] "s"(lim)
                                 : "scc");
                }
                if (q >= lim) break;
                if ((dirty >> q) & 1) {
                    // decide again, now that everything before q is final (the other dirty lanes too: most will not change again)
                    if ((dirty >> lane) & 1) fast_decide<K>(c, oldE, listm, nicem, kin, inc, bk, nonins, look, cfg, vbest, vdist, dep, vinfo);
                    dirty = 0;
                    slow = __ballot(vinfo > 0x1ffu);
                } else {
                    u32 x = (u32)__builtin_amdgcn_readlane((int)vinfo, (int)q);
                    if (x & FQ_NEED) {
                        u32 ob, od;
                        fast_exact(s, ch, tiles, sorted, inv, ring, ws, nonins, ws + q, cfg, lane, ob, od);
                        const u32 lk = n - (ws + q);
                        const bool im = ob >= (u32)MIN_MATCH;
                        const bool lg = im && !(ob <= (u32)cfg.lazy && lk - ob >= (u32)MIN_MATCH);
                        x = (im ? ob : 1u) | (lg ? FQ_LONG : 0u);
                        if (lane == q) { vbest = ob; vdist = od; vinfo = x; }
                        x = (u32)__builtin_amdgcn_readfirstlane((int)x);
                    }
                    B |= 1ull << q;
                    const u32 adv = x & 0x1ff;
                    lastlong = (x >> 9) & 1;
                    if (lastlong) {
                        const u32 hi = q + adv < 64 ? q + adv : 64;
                        const u64 below = hi >= 64 ? ~0ull : (1ull << hi) - 1;
                        const u64 nb = below & ~((2ull << q) - 1);            // positions (q, q + adv)
                        nonins |= nb;
                        const u64 dn = __ballot((dep & nb) != 0) & ~below;
                        dirty |= dn; slow |= dn;
                    }
                    q = (u32)__builtin_amdgcn_readfirstlane((int)(q + adv));
                    if (q >= lim) break;
                }
            }
            pos = (u32)__builtin_amdgcn_readfirstlane((int)(ws + q));
            carry_long = (u32)__builtin_amdgcn_readfirstlane((int)(q > 64 ? lastlong : 0));
        }
        // the window's insertion bits
        const u64 ins = ~nonins;
        if (lane == 0) { ring[(ws >> 5) & 1023] = (u32)ins; ring[((ws >> 5) + 1) & 1023] = (u32)(ins >> 32); }
        __builtin_amdgcn_wave_barrier();                        // (one wave: its LDS accesses are in order; no s_barrier, no wait for the stores in flight)
        // its tokens, side by side
        if (B) {
            const bool base = (B >> lane) & 1;
            const u32 rank = (u32)__popcll(B & ((1ull << lane) - 1));
            if (base) {
                const u32 kk = k + rank;
                if (kk % BLOCK_TOKENS == 0) bis[kk / BLOCK_TOKENS] = p;
                tk[kk] = vbest >= (u32)MIN_MATCH ? ((vbest - MIN_MATCH) << 16) | vdist : byte << 16;
            }
            k += (u32)__popcll(B);
        }
    }
    }
    if (lane == 0) {
        S->pos = pos; S->k = k; S->carry_long = carry_long;
        if (p_end == n) cout[ci].ntok = k;
    }
    for (int i = lane; i < 1024; i += 64) S->ring[i] = ring[i];
}

size_t fast_seq_state_bytes(int n_chunks) { return sizeof(FastSeqState) * (size_t)n_chunks; }
int fast_list_len(int level) { return level <= 1 ? 12 : level == 2 ? 24 : 48; }
int fast_list_rows(int level) { const int K = fast_list_len(level); return K + (K > 32 ? 3 : 2); }

int launch_inverse_map(hipStream_t st, const u8 *d_stream, const TileDesc *d_tiles, int n_tiles, const u32 *d_sorted, u32 *d_inv, u32 *d_flags)
{
    if (n_tiles == 0) return MTS_OK;
    hipLaunchKernelGGL(k_inverse_map, dim3(64, n_tiles < 65535 ? n_tiles : 65535), dim3(256), 0, st, d_stream, d_tiles, n_tiles, d_sorted, d_inv, d_flags);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

int launch_fast_cands(hipStream_t st, const u8 *d_stream, const ChunkDesc *d_chunks, const TileDesc *d_tiles, const u32 *d_sorted, const u32 *d_inv,
                      u32 *d_lists, u32 W, u32 phase, int n_chunks, int level, LevelCfg cfg)
{
    if (n_chunks == 0) return MTS_OK;
    const dim3 gc((W / 256) * (u32)n_chunks);                    // (bounded by the list workspace: W * n_chunks * rows * 4 bytes)
    const int K = fast_list_len(level);
#define MTS_FAST_CANDS(K) hipLaunchKernelGGL(k_fast_cands<K>, gc, dim3(256), 0, st, d_stream, d_chunks, d_tiles, d_sorted, d_inv, d_lists, W, phase, cfg)
    if (K == 12) { MTS_FAST_CANDS(12); } else if (K == 24) { MTS_FAST_CANDS(24); } else { MTS_FAST_CANDS(48); }
#undef MTS_FAST_CANDS
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

int launch_fast_seq(hipStream_t st, const u8 *d_stream, const ChunkDesc *d_chunks, const TileDesc *d_tiles, const u32 *d_sorted, const u32 *d_inv,
                    const u32 *d_lists, u32 W, u32 phase, void *d_state, int n_chunks, int level, LevelCfg cfg, u32 *d_tokens, u32 *d_blk_in_start,
                    ChunkOut *d_cout)
{
    if (n_chunks == 0) return MTS_OK;
    FastSeqState *S = (FastSeqState *)d_state;
    const int K = fast_list_len(level);
#define MTS_FAST_SEQ(K)                                                                                                                        \
    hipLaunchKernelGGL(k_fast_seq<K>, dim3(n_chunks), dim3(64), 0, st, d_stream, d_chunks, d_tiles, d_sorted, d_inv, d_lists, W, phase, S, cfg, \
                       d_tokens, d_blk_in_start, d_cout)
    if (K == 12) { MTS_FAST_SEQ(12); } else if (K == 24) { MTS_FAST_SEQ(24); } else { MTS_FAST_SEQ(48); }
#undef MTS_FAST_SEQ
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

// ================================================================================================
// T: per-block Huffman trees (zlib trees.c, exact) -- flush_block() in the oracle
// ================================================================================================
__constant__ u8 c_extra_lbits[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__constant__ u8 c_extra_dbits[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
__constant__ u8 c_extra_blbits[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
__constant__ u8 c_bl_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

// length code (0..28) of lc = len - 3, distance code (0..29) of d = dist - 1 (trees.c _length_code/_dist_code)
__device__ __forceinline__ u32 len_code(u32 lc, u32 &extra)
{
    if (lc < 8) { extra = 0; return lc; }
    if (lc == 255) { extra = 0; return 28; }
    const u32 k = 31 - __builtin_clz(lc);
    extra = k - 2;
    return 4 * extra + 4 + ((lc >> extra) & 3);
}
__device__ __forceinline__ u32 dist_code(u32 d, u32 &extra)
{
    if (d < 4) { extra = 0; return d; }
    const u32 k = 31 - __builtin_clz(d);
    extra = k - 1;
    return 2 * k + ((d >> (k - 1)) & 1);
}
__device__ __forceinline__ u32 static_llen(u32 n) { return n <= 143 ? 8 : n <= 255 ? 9 : n <= 279 ? 7 : 8; }
__device__ __forceinline__ u32 bit_reverse(u32 code, int len) { return __brev(code) >> (32 - len); }

constexpr int HEAP_SIZE = 2 * L_CODES + 1;       // 573

// Scratch of one tree build (LDS).  The workgroup is one wave: the heap (zlib's exact order, ties and all) is one lane's
// serial work, everything around it (heap fill, leaf depths, bl_count / opt_len sums, code assignment) uses all 64 lanes.
struct TreeWS {
    u16 freq[L_CODES + 2];                       // leaf frequencies; internal nodes carry theirs in the heap entries
    u16 len[HEAP_SIZE + 1];
    u16 dad[HEAP_SIZE];
    u32 bl_count[16];
    u32 next_code[16];
    int heap_len, heap_max, max_code, n_nodes, height, serial_lengths;
    long long opt_acc, static_acc;               // lane 0's contributions (forced nodes, serial gen_bitlen)
    // zlib's heap[] with the sort key carried in the entry:  freq << 15 | depth << 10 | node.  A block holds at most 16384
    // symbols, so freq < 2^15 + 1 and (Fibonacci bound on Huffman heights) depth < 32; smaller() is "key <= key" on
    // entry >> 10, and the two children of a slot are one aligned 8-byte LDS read.
    __attribute__((aligned(8))) u32 hp[HEAP_SIZE + 1];
};

typedef u32 tw_pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 tw_entry(u32 freq, u32 depth, u32 node) { return freq << 15 | depth << 10 | node; }
// (the heap's length comes by value: read from the structure it was loaded again at every level -- the stores into the heap
//  might have changed it, for all the compiler knows)
__device__ __forceinline__ void tw_downheap(TreeWS &t, int k, const int heap_len)
{
    const u32 v = t.hp[k];
    int j = k << 1;
    while (j <= heap_len) {
        const tw_pair pr = *(const tw_pair *)&t.hp[j];                // children j, j + 1 (j is even)
        u32 c = pr.x;
        if (j < heap_len && (pr.y >> 10) <= (c >> 10)) { j++; c = pr.y; }
        if ((v >> 10) <= (c >> 10)) break;
        t.hp[k] = c; k = j; j <<= 1;
    }
    t.hp[k] = v;
}

__device__ __forceinline__ int tw_xbits(int kind, int n, int base)
{
    if (n < base) return 0;
    return kind == 0 ? c_extra_lbits[n - base] : kind == 1 ? c_extra_dbits[n - base] : c_extra_blbits[n - base];
}

// zlib's gen_bitlen as written (one lane): only needed when the tree is higher than max_length, where the length
// repair walks the heap order.  Accumulates into t.opt_acc / t.static_acc.
__device__ void tw_gen_bitlen_serial(TreeWS &t, int kind, int max_length, int base)
{
    const int max_code = t.max_code;
    int h, n, m, bits, overflow = 0;
    long long opt_len = 0, static_len = 0;
    for (bits = 0; bits <= 15; bits++) t.bl_count[bits] = 0;
    t.len[t.hp[t.heap_max] & 1023] = 0;
    for (h = t.heap_max + 1; h < HEAP_SIZE; h++) {
        n = (int)(t.hp[h] & 1023);
        bits = t.len[t.dad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        t.len[n] = (u16)bits;
        if (n > max_code) continue;
        t.bl_count[bits]++;
        const int xbits = tw_xbits(kind, n, base);
        const long long f = t.freq[n];
        opt_len += f * (bits + xbits);
        if (kind == 0) static_len += f * ((long long)static_llen(n) + xbits);
        else if (kind == 1) static_len += f * (5 + xbits);
    }
    if (overflow > 0) {
        do {
            bits = max_length - 1;
            while (t.bl_count[bits] == 0) bits--;
            t.bl_count[bits]--; t.bl_count[bits + 1] += 2; t.bl_count[max_length]--;
            overflow -= 2;
        } while (overflow > 0);
        for (bits = max_length; bits != 0; bits--) {
            n = (int)t.bl_count[bits];
            while (n != 0) {
                m = (int)(t.hp[--h] & 1023);
                if (m > max_code) continue;
                if (t.len[m] != (u16)bits) {
                    opt_len += ((long long)bits - (long long)t.len[m]) * (long long)t.freq[m];
                    t.len[m] = (u16)bits;
                }
                n--;
            }
        }
    }
    t.opt_acc += opt_len; t.static_acc += static_len;
}

__device__ __forceinline__ u32 wave_sum_u32(u32 v)
{
    for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// kind 0 literal/length, 1 distance, 2 bit-length.  t.freq[0 .. elems) is the input; t.len[], t.bl_count[], t.max_code the
// output.  Called by all 64 lanes; opt_len / static_len stay uniform.
__device__ void tw_build(TreeWS &t, int kind, int lane, long &opt_len, long &static_len)
{
    const int elems = kind == 0 ? L_CODES : kind == 1 ? D_CODES : BL_CODES;
    const int max_length = kind == 2 ? 7 : 15;
    const int base = kind == 0 ? 257 : 0;
    const u64 lt = ((u64)1 << lane) - 1;
    // heap fill in symbol order
    int heap_len = 0, max_code = -1;
    for (int n0 = 0; n0 < elems; n0 += 64) {
        const int n = n0 + lane;
        const u32 f = n < elems ? t.freq[n] : 0;
        const u64 m = __ballot(f != 0);
        if (f) t.hp[heap_len + 1 + __popcll(m & lt)] = tw_entry(f, 0, (u32)n);
        else if (n < elems) t.len[n] = 0;
        if (m) max_code = n0 + 63 - __clzll(m);
        heap_len += __popcll(m);
    }
    __syncthreads();
    if (lane == 0) {
        int node;
        long long oa = 0, sa = 0;
        while (heap_len < 2) {
            node = (max_code < 2 ? ++max_code : 0);
            t.hp[++heap_len] = tw_entry(1, 0, (u32)node);
            t.freq[node] = 1; oa--;
            if (kind == 0) sa -= static_llen(node); else if (kind == 1) sa -= 5;
        }
        t.opt_acc = oa; t.static_acc = sa;
        t.heap_len = heap_len; t.max_code = max_code;
        int hl = heap_len, hmax = HEAP_SIZE;                          // (heap_len / heap_max of zlib, in registers while the heap is worked)
        for (int n = hl / 2; n >= 1; n--) tw_downheap(t, n, hl);
        node = elems;
        do {
            const u32 en = t.hp[1];
            t.hp[1] = t.hp[hl--]; tw_downheap(t, 1, hl);
            const u32 em = t.hp[1];
            t.hp[--hmax] = en; t.hp[--hmax] = em;
            const u32 dn = (en >> 10) & 31, dm = (em >> 10) & 31;
            t.dad[en & 1023] = t.dad[em & 1023] = (u16)node;
            t.hp[1] = tw_entry((en >> 15) + (em >> 15), (dn >= dm ? dn : dm) + 1, (u32)node);
            node++;
            tw_downheap(t, 1, hl);
        } while (hl >= 2);
        hmax--; t.hp[hmax] = t.hp[1];
        t.heap_len = hl; t.heap_max = hmax;
        t.n_nodes = node;
        t.height = (int)((t.hp[1] >> 10) & 31);
        t.serial_lengths = t.height > max_length;
        if (t.serial_lengths) tw_gen_bitlen_serial(t, kind, max_length, base);
    }
    __syncthreads();
    max_code = t.max_code;
    u32 o = 0, sl = 0;
    if (!t.serial_lengths) {
        // no length exceeds max_length: a leaf's length is its depth.  Pointer jumping over dad[] (in place, read phase
        // then write phase), len[] holds the distance covered so far.
        const int n_nodes = t.n_nodes, root = n_nodes - 1, height = t.height;
        constexpr int PER = (HEAP_SIZE + 63) / 64;
        for (int k = 0; k < PER; k++) {
            const int i = lane + 64 * k;
            if (i < n_nodes && (i >= elems || t.freq[i] != 0)) { t.len[i] = i == root ? 0 : 1; if (i == root) t.dad[i] = (u16)root; }
        }
        __syncthreads();
        for (int span = 1; span < height; span <<= 1) {
            u16 da[PER], aa[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int i = lane + 64 * k;
                da[k] = 0; aa[k] = 0;
                if (i < n_nodes && (i >= elems || t.freq[i] != 0)) { const int a = t.dad[i]; da[k] = t.len[a]; aa[k] = t.dad[a]; }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int i = lane + 64 * k;
                if (i < n_nodes && (i >= elems || t.freq[i] != 0)) { t.len[i] += da[k]; t.dad[i] = aa[k]; }
            }
            __syncthreads();
        }
        if (lane < 16) t.bl_count[lane] = 0;
        __syncthreads();
        for (int n = lane; n <= max_code; n += 64) {
            const u32 f = t.freq[n];
            if (f) {
                const u32 bits = t.len[n], xbits = (u32)tw_xbits(kind, n, base);
                atomicAdd(&t.bl_count[bits], 1u);
                o += f * (bits + xbits);
                if (kind == 0) sl += f * (static_llen(n) + xbits); else if (kind == 1) sl += f * (5 + xbits);
            }
        }
        o = wave_sum_u32(o); sl = wave_sum_u32(sl);
        __syncthreads();
    }
    opt_len += (long)t.opt_acc + (long)o;
    static_len += (long)t.static_acc + (long)sl;
}

// canonical codes (bit reversed) from t.len[0..max_code] and t.bl_count; out[n] = code | len << 16.  All 64 lanes: a symbol's
// code is next_code[len] plus the number of earlier symbols of the same length (ballot ranks, 64 symbols a round).
__device__ void tw_gen_codes(const TreeWS &t, int lane, u32 *out, int elems)
{
    u32 mine = 0, present = 0;                   // lane L (1..15) owns next_code[L]
    {
        u32 code = 0;
        for (int bits = 1; bits <= 15; bits++) {
            code = (code + t.bl_count[bits - 1]) << 1;
            if (bits == lane) mine = code;
            present |= (t.bl_count[bits] ? 1u : 0u) << bits;
        }
    }
    const u64 lt = ((u64)1 << lane) - 1;
    const int max_code = t.max_code;
    for (int n0 = 0; n0 < elems; n0 += 64) {
        const int n = n0 + lane;
        const u32 l = (n < elems && n <= max_code) ? t.len[n] : 0;
        u32 code = 0;
        for (u32 pm = present; pm; pm &= pm - 1) {
            const int bits = __ffs(pm) - 1;
            const u64 m = __ballot(l == (u32)bits);
            const u32 nc = __shfl(mine, bits);
            if (l == (u32)bits) code = nc + (u32)__popcll(m & lt);
            if (lane == bits) mine += (u32)__popcll(m);
        }
        if (n < elems) out[n] = l ? (bit_reverse(code, (int)l) | (l << 16)) : 0;
    }
}

struct BitW { u32 *w; u64 acc; int nb; u32 pos; };   // sequential LSB-first writer into zeroed words
__device__ __forceinline__ void bw_put(BitW &b, u32 v, int n)
{
    b.acc |= (u64)v << b.nb; b.nb += n;
    if (b.nb >= 32) { b.w[b.pos++] = (u32)b.acc; b.acc >>= 32; b.nb -= 32; }
}
// ------------------------------------------------------------------------------------------------
// scan_tree / send_tree by the whole wave.  zlib walks the code lengths symbol by symbol and flushes a count when it reaches
// max_count or the value changes; what it emits for a MAXIMAL run of R equal lengths v depends on v and R alone:
//   v = 0 :  R div 138 times code 18 (138 zeros), then for the rest r: nothing (0), r zeros (r < 3), code 17 (r <= 10), code 18;
//   v > 0 :  the run starts with max_count 7 / min_count 4 (the length before it differs): c = min(R, 7) of it are v x c (c < 4)
//            or v followed by code 16 for c - 1; after a full 7, groups of 6 are code 16 each (max_count 6 / min_count 3, the
//            length before is v now), and the rest r is v x r (r < 3) or one more code 16.
// A lane per symbol finds the runs (ballots of "differs from the symbol before"), the lane of a run's first symbol counts
// (scan) or writes (send: bit offsets by a wave scan of the runs' sizes, codes OR-ed into an LDS image of the header) what the
// run emits.  Lane 0 doing both walks alone was a quarter of the kernel's scalar instructions (k_block_trees is bound by the
// CU's scalar unit: the compiler runs one-lane code there).
// ------------------------------------------------------------------------------------------------
struct TreeRun { u32 v, nlit, n16, x16_first, q6, r16, n17, x17, n18_full, n18, x18; };    // what a run emits (see above)
__device__ __forceinline__ TreeRun tree_run(u32 v, u32 R)
{
    TreeRun t = {v, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (v == 0) {
        t.n18_full = R / 138;
        const u32 r = R - t.n18_full * 138;
        if (r < 3) t.nlit = r;
        else if (r <= 10) { t.n17 = 1; t.x17 = r - 3; }
        else { t.n18 = 1; t.x18 = r - 11; }
    } else {
        const u32 c = R < 7 ? R : 7;
        if (c < 4) t.nlit = c;
        else { t.nlit = 1; t.n16 = 1; t.x16_first = c - 4; }
        const u32 rest = R - c;                                   // (only behind a full 7)
        t.q6 = rest / 6;
        const u32 r = rest - t.q6 * 6;
        if (r < 3) t.nlit += r; else t.r16 = r;                  // r16: a last code 16 for r (3..5) lengths
    }
    return t;
}
// the runs of len[0 .. max_code]: calls f(k, is_start, v, R) for every symbol 64 k + lane, all lanes together, k ascending
template <class F>
__device__ __forceinline__ void tree_runs(const u16 *len, int max_code, int lane, F &&f)
{
    constexpr int RK = (L_CODES + 63) / 64;
    u64 M[RK];
    u32 v[RK];
#pragma unroll
    for (int k = 0; k < RK; k++) {
        const int n = 64 * k + lane;
        v[k] = n <= max_code ? (u32)len[n] : 0xffffu;
        const u32 pv = (n > 0 && n <= max_code) ? (u32)len[n - 1] : 0xfffeu;
        M[k] = __ballot(n <= max_code && v[k] != pv);
    }
#pragma unroll
    for (int k = 0; k < RK; k++) {
        const int n = 64 * k + lane;
        const bool st = (M[k] >> lane) & 1;
        // the next run start behind this symbol (or the end)
        u32 next = (u32)max_code + 1;
        const u64 above = lane == 63 ? 0 : M[k] & (~0ull << (lane + 1));
        if (above) next = 64 * k + (u32)__ffsll((long long)above) - 1;
        else {
            bool found = false;
#pragma unroll
            for (int kk = k + 1; kk < RK; kk++)
                if (!found && M[kk]) { next = 64 * kk + (u32)__ffsll((long long)M[kk]) - 1; found = true; }
        }
        f(k, st, v[k], st ? next - (u32)n : 0u);
    }
}
// scan_tree: blf[0 .. 18] += the codes the tree's runs emit (blf in LDS, zeroed by the caller)
__device__ void tw_scan_tree_wave(const u16 *len, int max_code, u32 *blf, int lane)
{
    tree_runs(len, max_code, lane, [&](int, bool st, u32 v, u32 R) {
        if (!st) return;
        const TreeRun t = tree_run(v, R);
        if (t.nlit) atomicAdd(&blf[v], t.nlit);
        const u32 n16 = t.n16 + t.q6 + (t.r16 ? 1u : 0u);
        if (n16) atomicAdd(&blf[16], n16);
        if (t.n17) atomicAdd(&blf[17], 1u);
        if (t.n18_full + t.n18) atomicAdd(&blf[18], t.n18_full + t.n18);
    });
}
// send_tree: the runs' codes into the header image `img` (LDS words, zero where nothing has been written) from bit `bit0` on;
// returns the bit behind them.  blc[c] = code | length << 16.
__device__ u32 tw_send_tree_wave(const u16 *len, int max_code, const u32 *blc, u32 *img, u32 bit0, int lane)
{
    u32 carry = bit0;
    auto put = [&](u32 &pos, u32 val, u32 nb) {                   // nb <= 14 bits at `pos`
        const u32 w = pos >> 5, sh = pos & 31;
        atomicOr(&img[w], val << sh);
        if (sh + nb > 32) atomicOr(&img[w + 1], val >> (32 - sh));
        pos += nb;
    };
    tree_runs(len, max_code, lane, [&](int, bool st, u32 v, u32 R) {
        TreeRun t = tree_run(st ? v : 0u, st ? R : 0u);
        const u32 lv = st ? blc[v] >> 16 : 0u, l16 = blc[16] >> 16, l17 = blc[17] >> 16, l18 = blc[18] >> 16;
        const u32 n16 = t.n16 + t.q6 + (t.r16 ? 1u : 0u);
        const u32 bits = st ? t.nlit * lv + n16 * (l16 + 2) + t.n17 * (l17 + 3) + (t.n18_full + t.n18) * (l18 + 7) : 0u;
        u32 tot;
        u32 pos = carry + wave_excl_scan_u32(bits, tot);
        carry += tot;
        if (!st) return;
        const u32 cv = blc[v] & 0xffff, c16 = blc[16] & 0xffff, c17 = blc[17] & 0xffff, c18 = blc[18] & 0xffff;
        if (v == 0) {
            for (u32 k = 0; k < t.n18_full; k++) put(pos, c18 | (127u << l18), l18 + 7);
            for (u32 k = 0; k < t.nlit; k++) put(pos, cv, lv);
            if (t.n17) put(pos, c17 | (t.x17 << l17), l17 + 3);
            if (t.n18) put(pos, c18 | (t.x18 << l18), l18 + 7);
        } else {
            const u32 c = R < 7 ? R : 7;
            if (c < 4) { for (u32 k = 0; k < c; k++) put(pos, cv, lv); }
            else { put(pos, cv, lv); put(pos, c16 | (t.x16_first << l16), l16 + 2); }
            for (u32 k = 0; k < t.q6; k++) put(pos, c16 | (3u << l16), l16 + 2);
            const u32 r = R - c - t.q6 * 6;
            if (r >= 3) put(pos, c16 | ((r - 3) << l16), l16 + 2);
            else for (u32 k = 0; k < r; k++) put(pos, cv, lv);
        }
    });
    return carry;
}

__device__ __forceinline__ u32 chunk_nblk(u32 ntok, u32 trailing)
{
    if (ntok == 0) return 1;
    return (ntok + BLOCK_TOKENS - 1) / BLOCK_TOKENS + ((ntok % BLOCK_TOKENS == 0 && !trailing) ? 1 : 0);
}

// number of fill_window slides once the loop top at absolute position q has run (oracle: orc_slides_at)
__device__ __forceinline__ u32 slides_at(u32 q, u32 n)
{
    u32 k = 0;
    for (;;) {
        const u64 edge = (u64)(k + 2) * WSIZE;
        const u64 theta = edge - 261 - (n < edge ? 1 : 0);
        if (q >= theta) k++; else break;
    }
    return k;
}

__global__ __launch_bounds__(64) void k_block_trees(const ChunkDesc *__restrict__ chunks, const u32 *__restrict__ blk_chunk,
                                                    int total_blk_cap, const u32 *__restrict__ tokens,
                                                    const u32 *__restrict__ blk_in_start, const ChunkOut *__restrict__ cout,
                                                    BlockRec *__restrict__ blocks, u32 *__restrict__ blk_codes,
                                                    u32 *__restrict__ blk_hdr, int fast)
{
    const int b = blockIdx.x;
    if (b >= total_blk_cap) return;
    const u32 ci = blk_chunk[b];
    const ChunkDesc ch = chunks[ci];
    const u32 bi = b - ch.blk0;
    const u32 ntok_c = cout[ci].ntok;
    const u32 nblk = chunk_nblk(ntok_c, cout[ci].trailing);
    if (bi >= nblk) return;
    const u32 tok0 = bi * BLOCK_TOKENS;
    const u32 nt = ntok_c > tok0 ? min((u32)BLOCK_TOKENS, ntok_c - tok0) : 0;
    const u32 last = bi == nblk - 1;
    const u32 in_start = nt ? blk_in_start[b] : ch.n;
    const u32 in_end = tok0 + nt < ntok_c ? blk_in_start[b + 1] : ch.n;
    __shared__ u32 dfreq[D_CODES + 2];
    __shared__ TreeWS tw;
    __shared__ u16 llen_s[L_CODES + 2], dlen_s[D_CODES + 2];
    u32 *lfreq = tw.hp;                              // the literal/length histogram lives in the heap's storage until it is copied out
    const int lane = threadIdx.x;
    for (int i = lane; i < L_CODES + 2; i += 64) lfreq[i] = 0;
    if (lane < D_CODES + 2) dfreq[lane] = 0;
    __syncthreads();
    const u32 *tk = tokens + ch.tok_off + tok0;
    {
        typedef u32 tok4 __attribute__((ext_vector_type(4), aligned(4)));
        auto count = [&](u32 t) {
            const u32 dist = t & 0xffff, lc = t >> 16;
            u32 ex;
            if (dist == 0) atomicAdd(&lfreq[lc], 1u);
            else { atomicAdd(&lfreq[257 + len_code(lc, ex)], 1u); atomicAdd(&dfreq[dist_code(dist - 1, ex)], 1u); }
        };
        const u32 nt4 = nt & ~3u;
        constexpr int TA = 4;                                   // 16-byte loads in flight per lane
        for (u32 i0 = (u32)lane * 4; i0 < nt4; i0 += 256 * TA) {
            tok4 q[TA];
#pragma unroll
            for (int a = 0; a < TA; a++) { const u32 i = i0 + 256 * a; q[a] = i < nt4 ? *(const tok4 *)(tk + i) : tok4{0, 0, 0, 0}; }
#pragma unroll
            for (int a = 0; a < TA; a++) if (i0 + 256 * a < nt4) { count(q[a].x); count(q[a].y); count(q[a].z); count(q[a].w); }
        }
        if ((u32)lane < nt - nt4) count(tk[nt4 + lane]);
    }
    __syncthreads();
    long opt_len = 0, static_len = 0;
    // literal/length tree
    for (int i = lane; i < L_CODES; i += 64) tw.freq[i] = i == 256 ? (u16)1 : (u16)lfreq[i];
    __syncthreads();
    tw_build(tw, 0, lane, opt_len, static_len);
    const int l_max = tw.max_code;
    for (int i = lane; i <= l_max; i += 64) llen_s[i] = tw.len[i];
    u32 *codes = blk_codes + (u64)b * BLK_CODE_WORDS;
    tw_gen_codes(tw, lane, codes, L_CODES);
    __syncthreads();
    // distance tree
    if (lane < D_CODES) tw.freq[lane] = (u16)dfreq[lane];
    __syncthreads();
    tw_build(tw, 1, lane, opt_len, static_len);
    const int d_max = tw.max_code;
    if (lane <= d_max) dlen_s[lane] = tw.len[lane];
    tw_gen_codes(tw, lane, codes + L_CODES, D_CODES);
    __syncthreads();
    // bit-length tree
    __shared__ u32 blf[BL_CODES + 1];
    __shared__ u32 hdr_img[BLK_HDR_WORDS];
    if (lane < BL_CODES) blf[lane] = 0;
    __syncthreads();
    tw_scan_tree_wave(llen_s, l_max, blf, lane);
    tw_scan_tree_wave(dlen_s, d_max, blf, lane);
    __syncthreads();
    if (lane < BL_CODES) tw.freq[lane] = (u16)blf[lane];
    __syncthreads();
    tw_build(tw, 2, lane, opt_len, static_len);
    int max_blindex;
    for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--)
        if (tw.len[c_bl_order[max_blindex]] != 0) break;
    opt_len += 3 * (max_blindex + 1) + 5 + 5 + 4;
    long opt_lenb = (opt_len + 3 + 7) >> 3;
    const long static_lenb = (static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
    const u32 in_len = in_end - in_start;
    // _tr_flush_block's `buf != NULL`: the block start must still be inside the sliding window.
    // The flush runs in the loop iteration whose top is at q: the literal branch flushes before
    // strstart++ (q = block end), the match branch after the skip (q = position after the match's
    // first byte + 1 = start of the last token + 1); the final flush follows the loop top at n.
    u32 q_top;
    if (last) q_top = ch.n;
    else {
        const u32 t_last = tk[nt - 1];
        // (deflate_fast tallies a token in the iteration whose top is at the token's own start, and flushes there)
        if (fast) q_top = (t_last & 0xffff) ? in_end - ((t_last >> 16) + MIN_MATCH) : in_end - 1;
        else q_top = (t_last & 0xffff) ? in_end - ((t_last >> 16) + MIN_MATCH) + 1 : in_end;
    }
    const bool buf_ok = (u64)in_start >= (u64)slides_at(q_top, ch.n) * WSIZE;
    BlockRec r;
    r.tok0 = tok0; r.ntok = nt; r.in_start = in_start; r.in_len = in_len; r.last = last; r.hdr_bits = 0; r.bit_start = 0;
    if ((long)in_len + 4 <= opt_lenb && buf_ok) {
        r.btype = 0; r.nbits = 0;
    } else if (static_lenb == opt_lenb) {
        r.btype = 1; r.nbits = (u32)(3 + static_len);
        // static codes: 0..143 -> 8 bits from 0x30, 144..255 -> 9 bits from 0x190, 256..279 -> 7 bits from 0, 280..287 -> 8 bits from 0xc0
        for (u32 n = lane; n < L_CODES; n += 64) {
            const u32 l = static_llen(n);
            const u32 code = n <= 143 ? 0x30 + n : n <= 255 ? 0x190 + (n - 144) : n <= 279 ? n - 256 : 0xc0 + (n - 280);
            codes[n] = bit_reverse(code, (int)l) | (l << 16);
        }
        if (lane < D_CODES) codes[L_CODES + lane] = bit_reverse((u32)lane, 5) | (5u << 16);
    } else {
        r.btype = 2; r.nbits = (u32)(3 + opt_len);
        u32 *hw = blk_hdr + (u64)b * BLK_HDR_WORDS;
        __shared__ u32 blc[BL_CODES + 1];
        for (int i = lane; i < BLK_HDR_WORDS; i += 64) hdr_img[i] = 0;
        if (lane == 0) {
            u32 next_code[16], code = 0;
            for (int bits = 1; bits <= 7; bits++) { code = (code + tw.bl_count[bits - 1]) << 1; next_code[bits] = code; }
            for (int n = 0; n < BL_CODES; n++) {
                const int l = n <= tw.max_code ? tw.len[n] : 0;
                blc[n] = l ? (bit_reverse(next_code[l]++, l) | ((u32)l << 16)) : 0;
            }
        }
        __syncthreads();
        if (lane == 0) {
            BitW bw = {hdr_img, 0, 0, 0};
            bw_put(bw, (u32)(l_max + 1 - 257), 5);
            bw_put(bw, (u32)(d_max + 1 - 1), 5);
            bw_put(bw, (u32)(max_blindex + 1 - 4), 4);
            for (int rank = 0; rank < max_blindex + 1; rank++) {
                const int sym = c_bl_order[rank];
                bw_put(bw, sym <= tw.max_code ? tw.len[sym] : 0, 3);
            }
            if (bw.nb) hdr_img[bw.pos] = (u32)bw.acc;              // (at most 71 bits: the runs' codes are OR-ed in behind them)
        }
        __syncthreads();
        u32 hb = 14u + 3u * (u32)(max_blindex + 1);
        hb = tw_send_tree_wave(llen_s, l_max, blc, hdr_img, hb, lane);
        hb = tw_send_tree_wave(dlen_s, d_max, blc, hdr_img, hb, lane);
        __syncthreads();
        for (int i = lane; i < BLK_HDR_WORDS; i += 64) hw[i] = hdr_img[i];
        r.hdr_bits = hb;
    }
    if (lane != 0) return;
    blocks[b] = r;
}

// ================================================================================================
// L: block layout per chunk
// ================================================================================================
// One wave per chunk: the blocks 64 at a time, bit offsets by a wave scan of the block sizes (one lane walking a chunk's ~300
// block records one dependent load after the other took 0.2 ms).  A stored block starts its bytes on a byte boundary, so its
// size depends on where it starts: groups with a stored block are walked in order by lane 0.
__global__ __launch_bounds__(64) void k_block_layout(const ChunkDesc *__restrict__ chunks, int n_chunks,
                                                     BlockRec *__restrict__ blocks, ChunkOut *__restrict__ cout,
                                                     const u64 *__restrict__ adler_acc)
{
    const int ci = blockIdx.x, lane = threadIdx.x;
    if (ci >= n_chunks) return;
    const ChunkDesc ch = chunks[ci];
    ChunkOut co = cout[ci];
    const u32 nblk = chunk_nblk(co.ntok, co.trailing);
    u64 bit = 16;
    for (u32 b0 = 0; b0 < nblk; b0 += 64) {
        const u32 bi = b0 + lane;
        const bool have = bi < nblk;
        BlockRec *r = blocks + ch.blk0 + (have ? bi : b0);
        const u32 btype = have ? r->btype : 1u;
        const u64 nbits = have ? (u64)r->nbits : 0ull;
        if (__any(btype == 0)) {
            if (lane == 0) {
                for (u32 k = b0; k < min(b0 + 64u, nblk); k++) {
                    BlockRec &q = blocks[ch.blk0 + k];
                    q.bit_start = bit;
                    if (q.btype == 0) { bit += 3; bit = (bit + 7) & ~7ull; bit += 32 + 8ull * q.in_len; }
                    else bit += q.nbits;
                }
            }
            bit = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)bit, 0)) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(bit >> 32), 0) << 32);
            continue;
        }
        u64 x = nbits;                                          // inclusive scan (64-bit: a chunk's stream may pass 2^32 bits)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 lo = __shfl_up((u32)x, o, 64), hi = __shfl_up((u32)(x >> 32), o, 64);
            if (lane >= o) x += ((u64)hi << 32) | lo;
        }
        if (have) r->bit_start = bit + x - nbits;
        const u64 tot = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)x, 63)) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(x >> 32), 63) << 32);
        bit += tot;
    }
    if (lane != 0) return;
    bit = (bit + 7) & ~7ull;
    co.nblk = nblk;
    co.nbytes = bit / 8 + 4;
    const u32 a = (u32)((1 + adler_acc[2 * ci]) % 65521u), bb = (u32)((ch.n + adler_acc[2 * ci + 1]) % 65521u);
    co.adler = (bb << 16) | a;
    cout[ci] = co;
}

// ================================================================================================
// B: bit packing
// ================================================================================================
// A lane's bit sink: it owns the bit range it writes.  Words it fully covers are stored, the two
// edge words are OR-ed atomically (the output is zero-filled beforehand).
struct Sink {
    u32 *out; u64 acc; int nb; u64 word; bool first;
    __device__ __forceinline__ void init(u32 *o, u64 bitpos) { out = o; word = bitpos >> 5; nb = (int)(bitpos & 31); acc = 0; first = true; }
    __device__ __forceinline__ void put(u32 v, int n)
    {
        acc |= (u64)v << nb; nb += n;
        if (nb >= 32) {
            if (first) { atomicOr(&out[word], (u32)acc); first = false; } else out[word] = (u32)acc;
            word++; acc >>= 32; nb -= 32;
        }
    }
    __device__ __forceinline__ void flush() { if (nb > 0 && (u32)acc) atomicOr(&out[word], (u32)acc); }
};

__device__ __forceinline__ u32 token_bits(u32 t, const u32 *lc_tab, const u32 *dc_tab)
{
    const u32 dist = t & 0xffff, lc = t >> 16;
    if (dist == 0) return lc_tab[lc] >> 16;
    u32 el, ed;
    const u32 lcode = len_code(lc, el), dcode = dist_code(dist - 1, ed);
    return (lc_tab[257 + lcode] >> 16) + el + (dc_tab[dcode] >> 16) + ed;
}
__device__ __forceinline__ void token_put(Sink &s, u32 t, const u32 *lc_tab, const u32 *dc_tab)
{
    const u32 dist = t & 0xffff, lc = t >> 16;
    if (dist == 0) { const u32 c = lc_tab[lc]; s.put(c & 0xffff, (int)(c >> 16)); return; }
    u32 el, ed;
    const u32 lcode = len_code(lc, el), dcode = dist_code(dist - 1, ed);
    u32 c = lc_tab[257 + lcode];
    // length code + extra bits (<= 15 + 5), distance code + extra bits (<= 15 + 13)
    s.put((c & 0xffff) | ((lc & ((1u << el) - 1)) << (c >> 16)), (int)((c >> 16) + el));
    c = dc_tab[dcode];
    s.put((c & 0xffff) | (((dist - 1) & ((1u << ed) - 1)) << (c >> 16)), (int)((c >> 16) + ed));
}

// code bits of one token, LSB first: <= 48 bits (15 + 5 length, 15 + 13 distance)
__device__ __forceinline__ u64 token_code(u32 t, const u32 *lc_tab, const u32 *dc_tab, u32 &nb)
{
    const u32 dist = t & 0xffff, lc = t >> 16;
    if (dist == 0) { const u32 c = lc_tab[lc]; nb = c >> 16; return c & 0xffff; }
    u32 el, ed;
    const u32 lcode = len_code(lc, el), dcode = dist_code(dist - 1, ed);
    const u32 c = lc_tab[257 + lcode], d = dc_tab[dcode];
    const u32 n1 = (c >> 16) + el, n2 = (d >> 16) + ed;
    const u64 v1 = (c & 0xffff) | ((lc & ((1u << el) - 1)) << (c >> 16));
    const u64 v2 = (d & 0xffff) | (((dist - 1) & ((1u << ed) - 1)) << (d >> 16));
    nb = n1 + n2;
    return v1 | (v2 << n1);
}

// One workgroup per block.  The bits of the block are assembled in LDS and leave as whole words, coalesced:
// a wave owns 2048 consecutive tokens and takes them 64 at a time (one coalesced load), a wave scan of
// the code lengths gives every token its bit position, and the code is OR-ed into the (zeroed) LDS image.
// Only the first and the last word of the image can be shared with the neighbouring blocks: those two are
// OR-ed into the (zeroed) output, the rest is stored.  A block too large for the image (it holds 16 bits
// per token; zlib needs 9-10 on this kind of data) ORs the overhang straight into the output.
#ifndef MTS_PACK_THREADS
#define MTS_PACK_THREADS 512
#endif
constexpr int PACK_THREADS = MTS_PACK_THREADS;              // 256: 1.19 ms, 512: 1.01, 1024: 1.60 (the image's LDS lets four workgroups share a CU)
constexpr int PACK_WTOK = 16384 / (PACK_THREADS / 64);      // token indices per wave (together >= 16383 tokens + end-of-block)
constexpr int PACK_IMG_WORDS = 8192;

__global__ __launch_bounds__(PACK_THREADS) void k_block_pack(const u8 *__restrict__ stream, const ChunkDesc *__restrict__ chunks,
                                                             const u32 *__restrict__ blk_chunk, int total_blk_cap,
                                                             const u32 *__restrict__ tokens, const BlockRec *__restrict__ blocks,
                                                             const u32 *__restrict__ blk_codes, const u32 *__restrict__ blk_hdr,
                                                             const ChunkOut *__restrict__ cout, u8 *__restrict__ outb, u32 zhdr)
{
    const int b = blockIdx.x;
    if (b >= total_blk_cap) return;
    const u32 ci = blk_chunk[b];
    const ChunkDesc ch = chunks[ci];
    const u32 bi = b - ch.blk0;
    const ChunkOut co = cout[ci];
    if (bi >= co.nblk) return;
    const BlockRec r = blocks[b];
    u32 *out = (u32 *)(outb + ch.out_off);
    __shared__ u32 codes[BLK_CODE_WORDS];
    __shared__ u32 wsum[PACK_THREADS / 64];
    __shared__ u32 img[PACK_IMG_WORDS];
    const int tid = threadIdx.x;
    if (r.btype == 0) {
        // stored: [3 bits][pad][LEN][NLEN][bytes]
        Sink sk;
        const u64 hdr_bit = r.bit_start;
        const u64 pay_bit = ((hdr_bit + 3 + 7) & ~7ull) + 32;
        if (tid == 0) {
            sk.init(out, bi == 0 ? 0 : hdr_bit);
            if (bi == 0) sk.put(zhdr, 16);
            sk.put(r.last, 3);
            sk.flush();
            sk.init(out, pay_bit - 32);
            sk.put(r.in_len & 0xffff, 16);
            sk.put((~r.in_len) & 0xffff, 16);
            sk.flush();
        }
        const u32 per = (r.in_len + PACK_THREADS - 1) / PACK_THREADS;
        const u32 b0 = min((u32)tid * per, r.in_len), b1 = min(b0 + per, r.in_len);
        if (b1 > b0) {
            const u8 *src = stream + ch.stream_off + r.in_start;
            sk.init(out, pay_bit + 8ull * b0);
            for (u32 i = b0; i < b1; i++) sk.put(src[i], 8);
            sk.flush();
        }
        if (r.last && tid == 0) {
            sk.init(out, pay_bit + 8ull * r.in_len);
            sk.put(__builtin_bswap32(co.adler), 32);
            sk.flush();
        }
        return;
    }
    // the image starts at the output word holding the first bit of the block (block 0: the zlib header)
    const u64 first_bit = bi == 0 ? 0 : r.bit_start;
    const u64 word0 = first_bit >> 5;
    const u64 end_bit = r.last ? ((r.bit_start + r.nbits + 7) & ~7ull) + 32 : r.bit_start + r.nbits;
    const u32 nwords = (u32)(((end_bit + 31) >> 5) - word0);
    const u32 nimg = nwords < (u32)PACK_IMG_WORDS ? nwords : (u32)PACK_IMG_WORDS;
    for (int i = tid; i < BLK_CODE_WORDS; i += PACK_THREADS) codes[i] = i < L_CODES + D_CODES ? blk_codes[(u64)b * BLK_CODE_WORDS + i] : 0;
    for (u32 i = tid; i < nimg; i += PACK_THREADS) img[i] = 0;
    __syncthreads();
    // nb bits of v at absolute bit position `at`
    auto or_bits = [&](u64 at, u64 v, u32 nb) {
        if (nb == 0) return;
        const u32 rel = (u32)(at - (word0 << 5)), w = rel >> 5, sh = rel & 31;
        const u64 lo = v << sh;
        const u32 x[3] = {(u32)lo, (u32)(lo >> 32), sh ? (u32)(v >> (64 - sh)) : 0u};
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (x[j]) { if (w + j < (u32)PACK_IMG_WORDS) atomicOr(&img[w + j], x[j]); else atomicOr(&out[word0 + w + j], x[j]); }
    };
    const u32 *lc_tab = codes, *dc_tab = codes + L_CODES;
    const u32 *tk = tokens + ch.tok_off + r.tok0;
    const int wave = tid >> 6, lane = tid & 63;
    // token index ntok is the end-of-block symbol
    const u32 i_beg = (u32)wave * PACK_WTOK, i_end = min(i_beg + (u32)PACK_WTOK, r.ntok + 1);
    // (the kernel was bound by the latency of its token loads, one per lane and step with the scan and the LDS work of the step
    // behind it: the tokens of PACK_AHEAD steps are asked for together)
    constexpr int PACK_AHEAD = 8;
    auto code_of = [&](u32 i, u32 t, u32 &nb) -> u64 {
        if (i < r.ntok) return token_code(t, lc_tab, dc_tab, nb);
        if (i == r.ntok) { const u32 c = lc_tab[256]; nb = c >> 16; return c & 0xffff; }
        nb = 0;
        return 0;
    };
    u32 mybits = 0;
    for (u32 i0 = i_beg; i0 < i_end; i0 += 64 * PACK_AHEAD) {
        u32 t[PACK_AHEAD];
#pragma unroll
        for (int a = 0; a < PACK_AHEAD; a++) { const u32 i = i0 + 64 * a + lane; t[a] = i < r.ntok ? tk[i] : 0u; }
#pragma unroll
        for (int a = 0; a < PACK_AHEAD; a++) { const u32 i = i0 + 64 * a + lane; u32 nb; if (i < i_end) { code_of(i, t[a], nb); mybits += nb; } }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) mybits += __shfl_xor(mybits, o, 64);
    if (lane == 0) wsum[wave] = mybits;
    // zlib header, block header, code-length header
    if (tid == 0) {
        if (bi == 0) or_bits(0, zhdr, 16);
        or_bits(r.bit_start, r.last | (r.btype << 1), 3);
        if (r.last) or_bits(end_bit - 32, __builtin_bswap32(co.adler), 32);           // adler32, big-endian
    }
    if (r.btype == 2) {
        const u32 *hw = blk_hdr + (u64)b * BLK_HDR_WORDS;
        for (u32 k = tid; 32 * k < r.hdr_bits; k += PACK_THREADS) {
            const u32 left = r.hdr_bits - 32 * k;
            or_bits(r.bit_start + 3 + 32 * k, left >= 32 ? hw[k] : hw[k] & ((1u << left) - 1), left >= 32 ? 32 : left);
        }
    }
    __syncthreads();
    u64 pos = r.bit_start + 3 + r.hdr_bits;
    for (int w = 0; w < wave; w++) pos += wsum[w];
    for (u32 i0 = i_beg; i0 < i_end; i0 += 64 * PACK_AHEAD) {
        u32 t[PACK_AHEAD];
#pragma unroll
        for (int a = 0; a < PACK_AHEAD; a++) { const u32 i = i0 + 64 * a + lane; t[a] = i < r.ntok ? tk[i] : 0u; }
#pragma unroll
        for (int a = 0; a < PACK_AHEAD; a++) {
            if (i0 + 64 * a >= i_end) break;                     // (wave uniform)
            const u32 i = i0 + 64 * a + lane;
            u32 nb = 0;
            const u64 v = i < i_end ? code_of(i, t[a], nb) : 0ull;
            const u32 x = wave_incl_scan_dpp(nb);                // (six shuffles through the LDS crossbar were the step's longest chain)
            or_bits(pos + x - nb, v, nb);
            pos += (u32)__builtin_amdgcn_readlane((int)x, 63);
        }
    }
    __syncthreads();
    for (u32 j = tid; j < nimg; j += PACK_THREADS) {
        const u32 v = img[j];
        if (j == 0 || j == nwords - 1) { if (v) atomicOr(&out[word0 + j], v); } else out[word0 + j] = v;
    }
}

int launch_block_trees(hipStream_t st, const ChunkDesc *d_chunks, const u32 *d_blk_chunk, int total_blk_cap,
                       const u32 *d_tokens, const u32 *d_blk_in_start, const ChunkOut *d_cout, BlockRec *d_blocks,
                       u32 *d_blk_codes, u32 *d_blk_hdr, int fast)
{
    if (total_blk_cap == 0) return MTS_OK;
    hipLaunchKernelGGL(k_block_trees, dim3(total_blk_cap), dim3(64), 0, st, d_chunks, d_blk_chunk, total_blk_cap, d_tokens,
                       d_blk_in_start, d_cout, d_blocks, d_blk_codes, d_blk_hdr, fast);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}
int launch_block_layout(hipStream_t st, const ChunkDesc *d_chunks, int n_chunks, BlockRec *d_blocks, ChunkOut *d_cout,
                        const u64 *d_adler_acc)
{
    if (n_chunks == 0) return MTS_OK;
    hipLaunchKernelGGL(k_block_layout, dim3(n_chunks), dim3(64), 0, st, d_chunks, n_chunks, d_blocks, d_cout, d_adler_acc);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}
int launch_block_pack(hipStream_t st, const u8 *d_stream, const ChunkDesc *d_chunks, const u32 *d_blk_chunk,
                      int total_blk_cap, const u32 *d_tokens, const BlockRec *d_blocks, const u32 *d_blk_codes,
                      const u32 *d_blk_hdr, const ChunkOut *d_cout, u8 *d_out, int level)
{
    if (total_blk_cap == 0) return MTS_OK;
    // zlib header: CMF 0x78, FLG = level flags << 6 made a multiple of 31 (deflate.c)
    const u32 lf = level < 2 ? 0 : level < 6 ? 1 : level == 6 ? 2 : 3;
    u32 header = (0x78u << 8) | (lf << 6);
    header += 31 - (header % 31);
    const u32 zhdr = (header >> 8) | ((header & 0xff) << 8);      // first byte in the low bits
    hipLaunchKernelGGL(k_block_pack, dim3(total_blk_cap), dim3(PACK_THREADS), 0, st, d_stream, d_chunks, d_blk_chunk,
                       total_blk_cap, d_tokens, d_blocks, d_blk_codes, d_blk_hdr, d_cout, d_out, zhdr);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

// The packer stores the words a block (or, in a stored block, a thread) covers alone and ORs the ones it shares -- so those must
// be zero beforehand: the first and the last word of every block, what a block has beyond its LDS image, stored blocks as a
// whole, the check value.  (Rounds 1-2 zeroed every slot over its whole compress_bound: 1.39 GB of writes per 60 chunks.)
// One workgroup per block, after the layout.
__global__ __launch_bounds__(256) void k_zero_edges(const ChunkDesc *__restrict__ chunks, const u32 *__restrict__ blk_chunk, int total_blk_cap,
                                                    const BlockRec *__restrict__ blocks, const ChunkOut *__restrict__ cout, u8 *__restrict__ outb)
{
    const int b = blockIdx.x;
    if (b >= total_blk_cap) return;
    const u32 ci = blk_chunk[b];
    const ChunkDesc ch = chunks[ci];
    const u32 bi = b - ch.blk0;
    const ChunkOut co = cout[ci];
    if (bi >= co.nblk) return;
    const BlockRec r = blocks[b];
    u32 *out = (u32 *)(outb + ch.out_off);
    const u64 first_bit = bi == 0 ? 0 : r.bit_start;
    const u64 word0 = first_bit >> 5;
    u64 end_bit;
    if (r.btype == 0) end_bit = ((r.bit_start + 3 + 7) & ~7ull) + 32 + 8ull * r.in_len;
    else end_bit = r.bit_start + r.nbits;
    if (r.last) end_bit = ((end_bit + 7) & ~7ull) + 32;
    const u64 wend = (end_bit + 31) >> 5;                       // one past the last word
    if (r.btype == 0) {
        for (u64 w = word0 + threadIdx.x; w < wend; w += 256) out[w] = 0;
        return;
    }
    if (threadIdx.x == 0) { out[word0] = 0; out[wend - 1] = 0; if (r.last && wend >= 2) out[wend - 2] = 0; }
    for (u64 w = word0 + PACK_IMG_WORDS + threadIdx.x; w < wend; w += 256) out[w] = 0;      // (a block too large for the packer's image)
}
int launch_zero_edges(hipStream_t st, const ChunkDesc *d_chunks, const u32 *d_blk_chunk, int total_blk_cap, const BlockRec *d_blocks,
                      const ChunkOut *d_cout, u8 *d_out)
{
    if (total_blk_cap == 0) return MTS_OK;
    hipLaunchKernelGGL(k_zero_edges, dim3(total_blk_cap), dim3(256), 0, st, d_chunks, d_blk_chunk, total_blk_cap, d_blocks, d_cout, d_out);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

}  // namespace mts
