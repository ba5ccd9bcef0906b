// M: per-position best matches of the bit-exact zlib-1.2.11 DEFLATE (levels 4..9) on gfx950 -- k_match5 / k_match6.
// A translation unit of its own: the candidate loops run under wave-uniform branches, and the structurizer's copies of the loop
// state at every join go away with  -mllvm -structurizecfg-skip-uniform-regions  (Makefile: this file only; the same flag
// miscompiles the level 1..3 kernels of deflate.hip).  Replaces the longest_match() calls inside `zlib.compress`
// (/root/reference/mtscomp.py:394); orc_match_tables() of oracle/mtsc_oracle.c is the oracle.
#include <stdlib.h>

#include <type_traits>

#include "deflate_dev.h"

namespace mts {

// ================================================================================================
// M: per-position best matches (t_full, t_quarter) -- orc_match_tables() is the oracle
// ================================================================================================
// The kernel scores a candidate from a 64-bit "entry" built (from the LDS window) for every staged slot:
//   w0 [17:0]  rel   window-relative position
//      [26:18] d     9 bits that, TOGETHER WITH AN EQUAL 15-BIT HASH, prove bytes 0..2 equal (any order of them does: b1[2:0] : b0[7:5] : b0[2:0] is stored):
//                    h = (b0<<10 ^ b1<<5 ^ b2) & 0x7fff exposes b0[4:3], b1[4:3], b2[4:0] directly and
//                    b0[2:0]^b1[7:5], b1[2:0]^b2[7:5]; b0[7:5] not at all.  d = b0[7:5] : b0[2:0] : b1[2:0].
//      [31:27] low 5 bits of byte 7
//   w1         bytes 3..6
// so the common prefix of two same-hash positions is known exactly up to 7 bytes from the entries alone;
// only longer matches go back to the window bytes.
// A table entry is ONE word per position:  [14:0] dist (0: no match), [22:15] len - 3  = the full-budget result, and two flags
// for the quarter-budget result (what the walk looks at when it already holds a match of >= good_match bytes):
//   neither   the same as the full-budget result
//   TE_QNONE  another one that cannot matter: it is not longer than good_match, and the walk only takes what is LONGER than the
//             match it holds
//   TE_QSIDE  another one that can: it is in the side table quarter[p] (same packing), which is written for these positions
//             only (0.4 % of them on the synthetic recordings; the debug tap of the tests has it written everywhere)
// (Rounds 1-2 kept both results and the position's byte in 8 bytes per position: twice the table traffic in the match store
// and in both parse walks, and half as many walkers per CU, whose windows of the table live in LDS.)
// the value of the lane before (wave_shr:1); lane 0 gets `first`
__device__ __forceinline__ u32 prev_lane(u32 v, u32 first) { return (u32)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false); }
// lane-wise select by a lane mask held in scalar registers: mask bit set -> a, else b
__device__ __forceinline__ u32 sel64(u64 mask, u32 a, u32 b)
{
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
// a & (b ^ c) in one instruction (v_bitop3_b32; truth table index = a << 2 | b << 1 | c)
__device__ __forceinline__ u32 and_xor(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x60); }
__device__ __forceinline__ u64 make_entry(u32 rel, u32 lo, u32 hi)       // lo = bytes 0..3, hi = bytes 4..7
{
    // (the nine bits in the order they are cheapest to take: b1[2:0] : b0[7:5] is one field of `lo`, b0[2:0] another)
    const u32 d = (((lo >> 5) & 0x3f) << 3) | (lo & 7);
    const u32 w0 = rel | (d << REL_BITS) | (((hi >> 24) & ((1u << (32 - REL_BITS - 9)) - 1)) << (REL_BITS + 9));
    const u32 w1 = (lo >> 24) | (hi << 8);
    return (u64)w0 | ((u64)w1 << 32);
}
__device__ __forceinline__ u32 lds_u32(const u32 *win, u32 addr)
{
    const u32 a = addr >> 2;
    return alignbyte(win[a + 1], win[a], addr & 3);
}

// ------------------------------------------------------------------------------------------------
// The walk is filtered: it produces exactly what the plain newest-first walk produces, scoring far fewer candidates:
// a candidate can only replace the current best if it can be LONGER than it, so once the best length is L only candidates
// whose first L + 1 bytes may equal the position's own are looked at; all others are skipped -- which changes nothing,
// because zlib's walk would have compared and rejected them (they still count against the chain budget: budgets are
// positions in the run, not candidates scored).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// k_match5 (chain budget <= 128): the filter masks are looked up.  Every wave keeps, next to its 256-slot ring of
// entries, four tables of 32 rows x 256 bits (the first one 64 rows): row k of table d (d = 4..7) has bit r set iff the slot at
// ring position r has key_d = k, where key_d is a 5-bit hash of bytes 3 .. d-1 of the slot's string (6 bits of byte 3 for the
// first table, which the walk consults most: with 32 rows a lane met ~4 candidates per group that only shared the row).  A
// slot entering the ring clears the bits of the slot it replaces and sets its own (8 LDS atomics per 64
// slots).  A lane reads the rows of its OWN keys and funnel-shifts out the 128 bits of the slots before
// it: M_d = candidates whose first d bytes may equal its own (hash + bytes 3..d-1; a superset, which is
// all the filter needs: a candidate that passes is scored exactly, one that fails cannot be longer than
// d-1).  With best length L the walk only pops candidates of M_(L+1) (M_7 from 6 on), so the number of
// scored candidates is about the number of times the best length improves.
// ------------------------------------------------------------------------------------------------
#ifndef MTS_M5_NT_KEYS
#define MTS_M5_NT_KEYS 0     // 1: the sorted keys, read once, by non-temporal loads (round 3: they keep table lines in L2, 17.0 -> 13.4 GB written, same time).
                             // Round 6: with plain loads the kernel takes 18.2-18.3 ms in every process; with the non-temporal ones it took one of three
                             // times from process to process (18.4 / 18.8 / 19.0: the physical placement of the workspace, tools/m5_addr_times.py) -- six
                             // alternating runs each: 18.21-18.32 against 18.55-18.96.  The bytes written are not what the kernel waits for.
#endif
#ifndef MTS_M5_STATS
#define MTS_M5_STATS 0
#endif
#ifndef MTS_M5_BIT_AND
#define MTS_M5_BIT_AND 0
#endif
#if MTS_M5_STATS
__device__ unsigned long long g_m5_stats[16];          // groups walked, rounds of the newest word, rounds of the other 96, scorings of each
#endif
constexpr int M5_WAVES = 8;                         // k_match6's workgroup
#ifndef MTS_M5_SX
#define MTS_M5_SX 1                                  // 1: bytes 7..12 of every slot in LDS beside its entry (2 KB per wave more: 8 waves per workgroup); 0: read from the window when wanted (10 waves)
#endif
#ifndef MTS_M5_WAVES
#define MTS_M5_WAVES (MTS_M5_SX == 1 ? 8 : MTS_M5_SX == 2 ? 9 : 10)
#endif
constexpr int M5W = MTS_M5_WAVES;                    // k_match5's waves per workgroup: two workgroups per CU either way
constexpr int M5_SLICES = 64;                        // workgroups per tile (MTS_MATCH_SLICES overrides: experiments) = the workgroups an XCD holds: ONE tile per XCD at a
                                                     // time.  Round 4, time / bytes written per launch (60 chunks; the table is 5.5 GB): 32 slices 24.7 ms / 38-44 GB (two tiles
                                                     // share an L2: lines leave half filled again and again), 64: 24.9 ms / 13.4 GB, 96: 26.2 ms / 9.7 GB, 128: 7.8 GB (a tile is
                                                     // finished sooner, but a wave's two history groups and empty tables are spread over fewer groups)
constexpr int M6_SLICES = 64;
constexpr int M5_RING = 256;
constexpr int M5_ROWS = 32;
constexpr int M5_LEVELS = 4;                         // tables for prefix lengths 4, 5, 6, 7
constexpr int M5_ROW_WORDS = M5_RING / 32 + 1;       // a row is 8 words of bits + 1 of padding, so that rows start in different LDS banks (32-byte rows whose word
                                                     // addresses are (offset & 28) | row save ten instructions per group and cost 1.8 ms in bank conflicts; whole rows read
                                                     // from one address -- four ds_read2_b32 -- and the five words picked in registers save fourteen and cost 0.5 ms: round 4)
constexpr int M5_TABLE = M5_ROWS * M5_ROW_WORDS * 4; // bytes per table
constexpr int M5_SLOTS = M5_LEVELS + 1;                // the first table has 64 rows (a 6-bit key of byte 3): two table slots
constexpr int M5_SX_BYTES = MTS_M5_SX == 1 ? 8 : MTS_M5_SX == 2 ? 4 : 0;      // per slot, of its bytes 7..
constexpr int M5_WAVE_LDS = M5_RING * (8 + M5_SX_BYTES) + M5_SLOTS * M5_TABLE;      // 9856: entries, bytes 7..12, tables (7808 without the bytes)
__device__ __forceinline__ constexpr int m5_slot(int d) { return d ? d + 1 : 0; }
// requested LDS is padded so that TWO workgroups share a CU, not three (16 waves per CU keep the vector units busy)
#ifndef MTS_M5_LDS_PAD
#define MTS_M5_LDS_PAD 0
#endif
constexpr int M5_VLUT = 129 * 16;                    // the budget masks (one table per workgroup, behind the waves' areas)
constexpr int MATCH5_LDS = MTS_M5_LDS_PAD ? MTS_M5_LDS_PAD : (M5W * M5_WAVE_LDS + M5_VLUT > 56 * 1024 ? M5W * M5_WAVE_LDS + M5_VLUT : 56 * 1024);      // (MTS_M5_LDS_PAD: occupancy experiments)

// the equal bytes among bytes 3..6 (x1 = their xor), 0..4: v_ffbl_b32 gives -1 for 0, and 0xffffffff >> 3 is still more than 4.
// (k_match5 keeps match lengths as length - 3 -- what the table word stores --: one addition less per candidate)
__device__ __forceinline__ int m5_len04(u32 x1, bool &all4)
{
    u32 f;
    asm("v_ffbl_b32 %0, %1" : "=v"(f) : "v"(x1));
    all4 = f > 31;                                       // (no bit set: bytes 3..6 are equal; the caller sets 4 itself, so no v_min here)
    return (int)(f >> 3);
}
// 5-bit keys of the prefixes (b3), (b3,b4), (b3..b5), (b3..b6) of e1 = bytes 3..6
// (v_mul_u32_u24 by name: the compiler sees that the bits taken do not depend on the operand's top byte, drops the mask in front
//  of the multiplication and is left with a 32-bit multiply -- a quarter-rate instruction, four issue slots instead of one)
__device__ __forceinline__ u32 m5_mul24(u32 x, u32 c)
{
    u32 r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
    return r;
}
__device__ __forceinline__ u32 m5_hash24(u32 x) { return (m5_mul24(x, 0x9E3779u) >> 19) & 31; }
// leading zeros, -1 for 0 (v_ffbh_u32 by name: __builtin_clz(0) is undefined)
__device__ __forceinline__ u32 m5_ffbh(u32 x)
{
    u32 r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// k - 8 z in one instruction (the ring offset of the candidate at bit 31 - z of a mask word whose bit 31 is at offset k)
__device__ __forceinline__ u32 m5_off8(u32 z, u32 k)
{
    u32 r;
    asm("v_mad_i32_i24 %0, %1, -8, %2" : "=v"(r) : "v"(z), "v"(k));
    return r;
}
__device__ __forceinline__ void m5_keys(u32 e1, u32 (&k)[M5_LEVELS])
{
    // ONE multiplication for the first three: bit i of a product depends on the bits 0..i of the factor alone, so bits 2..7 of
    // (bytes 3..5) x C are a key of byte 3, bits 11..15 one of bytes 3..4 and bits 19..23 one of bytes 3..5 (round 5: each had a
    // mask and a multiplication of its own, four instructions more per slot)
    const u32 p = m5_mul24(e1, 0x9E3779u);                    // (the instruction takes the low 24 bits)
    k[0] = (p >> 2) & 63;
    k[1] = (p >> 11) & 31;
    k[2] = (p >> 19) & 31;
    k[3] = m5_hash24(e1 ^ (e1 >> 11));
}

// Workgroups share tiles: `nsl` consecutive workgroups OF ONE XCD (block b runs on XCD b % 8) take the `nsl` slices of one
// tile's sorted order, so an XCD has 64 / nsl tiles in flight and their windows (read at random) and table regions (written
// at random) stay in its L2 until they are complete (L2 hit rate 12 % -> 92 %, 10x fewer misses: tools/pmc_cache.sh).
// flags[0] |= 1 when the sorted order is found NOT to be position-ordered inside a hash run (the sort's ranking relies on
// a hardware property, see rank_pass): the caller then sorts again with the ballot ranking and repeats the stage.
__global__ __launch_bounds__(M5W * 64) void k_match5(const u8 *__restrict__ stream, const TileDesc *__restrict__ tiles, int n_tiles, int nsl,
                                                          const u32 *__restrict__ sorted, u32 *__restrict__ tables, u32 *__restrict__ quarter, LevelCfg cfg,
                                                          u32 *__restrict__ flags, int all_quarters)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const u32 tile_id = (jb / (u32)nsl) * 8 + xcd, slice = jb % (u32)nsl;
    if (tile_id >= (u32)n_tiles) return;
    const TileDesc td = tiles[tile_id];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // LDS: the waves' entry rings (2 KB each, 2 KB aligned: a slot's address is (offset & 2040) | base), their rings of bytes 7..12
    // (the same offset + 16 KB), their tables, the table of budget masks
    const u8 *gwin = stream + td.stream_off + td.w;
    auto wread = [&](u32 addr) -> u32 { return gld_u32_unaligned(gwin, addr); };
    u64 *SE = (u64 *)(smem + wave * (M5_RING * 8));
    u64 *SX = (u64 *)(smem + M5W * M5_RING * 8 + wave * (M5_RING * 8));      // bytes 7..12 of every slot: matches up to 13 never leave the LDS
    u32 *TB = (u32 *)(smem + M5W * M5_RING * (8 + M5_SX_BYTES) + M5_VLUT + wave * (M5_SLOTS * M5_TABLE));      // [level][row][8 words + 1]
    const uint4 *VLUT = (const uint4 *)(smem + M5W * M5_RING * (8 + M5_SX_BYTES));      // [0 .. 128]: the newest n of 128 bits (one table per workgroup; behind the rings, where its offset fits a ds instruction's)
    typedef __attribute__((address_space(3))) const u64 *lds_u64p;
    const u32 se_base = (u32)(size_t)(__attribute__((address_space(3))) u8 *)(u8 *)SE;      // (byte address in LDS)
    if (se_base & (M5_RING * 8 - 1)) __builtin_trap();            // (the kernel has no static LDS: the dynamic area starts at 0)
    // entry / bytes 7..12 of the ring slot at byte offset o8 (any multiple of 8; only its low 11 bits count)
    auto ring_e = [&](u32 o8) -> u64 { return *(lds_u64p)(size_t)((o8 & (M5_RING * 8 - 8)) | se_base); };
#if MTS_M5_SX == 2
    typedef __attribute__((address_space(3))) const u32 *lds_u32p;
    auto ring_x = [&](u32 o8) -> u32 { return *(lds_u32p)(size_t)((((o8 & (M5_RING * 8 - 8)) | se_base) >> 1) + M5W * M5_RING * 8); };      // bytes 7..10
#else
    auto ring_x = [&](u32 o8) -> u64 { return *(lds_u64p)(size_t)(((o8 & (M5_RING * 8 - 8)) | se_base) + M5W * M5_RING * 8); };
#endif
    u32 *T = tables + td.stream_off, *TQ = quarter + td.stream_off;
    u32 *sink = flags + 63 + (blockIdx.x & (MATCH_SINK_BYTES / 256 - 1)) * 64 + (threadIdx.x & 63);      // (this lane's word of a line per workgroup)
    if (threadIdx.x < 2 && slice == 0) {
        const u32 hashed_end = td.w + td.wlen;
        const u32 p = hashed_end + threadIdx.x;
        if (p >= td.a && p < td.own_end) { T[p] = 0; if (all_quarters) TQ[p] = 0; }
    }
    if (td.wlen == 0) return;
    const u32 *sk = sorted + td.sorted_off;
    const u32 wlen = td.wlen, n = td.n;
    const u32 ngroups = (wlen + 63) / 64;
    const u32 halo = td.a - td.w;
    const u32 chain = (u32)cfg.chain, qchain = (u32)cfg.chain >> 2;
    const u32 nwv = (u32)nsl * M5W;
    const u32 gpw = (ngroups + nwv - 1) / nwv;
    const u32 g_begin = (slice * M5W + wave) * gpw, g_end = min(ngroups, g_begin + gpw);
#if MTS_M5_NT_KEYS
    auto slot_rel = [&](int idx) -> u32 { return __builtin_nontemporal_load(&sk[idx < 0 ? 0 : (u32)idx < wlen ? (u32)idx : wlen - 1]) & REL_MASK; };      // (read once: keep them out of the way of the window and the table lines in L2)
#else
    auto slot_rel = [&](int idx) -> u32 { return sk[idx < 0 ? 0 : (u32)idx < wlen ? (u32)idx : wlen - 1] & REL_MASK; };
#endif
#if MTS_M5_NT_KEYS
    auto slot_fwd = [&](u32 idx) -> u32 { return __builtin_nontemporal_load(&sk[min(idx, wlen - 1)]) & REL_MASK; };      // (an index that cannot be negative: one v_min)
#else
    auto slot_fwd = [&](u32 idx) -> u32 { return sk[min(idx, wlen - 1)] & REL_MASK; };
#endif
    // A wave walks ~8 groups; its start was four dependent round trips to memory (keys of the history, its bytes, keys of the first
    // group, its bytes) during which its share of the CU did nothing.  The four key loads go out before anything else (the budget
    // masks, the barrier and the empty tables are made while they fly), the three loads of bytes together behind them.
    const int i_first = (int)g_begin * 64 + lane;
    const u32 r_ha = slot_rel(i_first - 128), r_hb = slot_rel(i_first - 64), r_0 = slot_rel(i_first), r_1 = slot_rel(i_first + 64);
    for (int k = threadIdx.x; k <= 128; k += M5W * 64) {
        uint4 v;
        v.w = k >= 32 ? 0xffffffffu : k ? 0xffffffffu << (32 - k) : 0u;
        v.z = k >= 64 ? 0xffffffffu : k > 32 ? 0xffffffffu << (64 - k) : 0u;
        v.y = k >= 96 ? 0xffffffffu : k > 64 ? 0xffffffffu << (96 - k) : 0u;
        v.x = k >= 128 ? 0xffffffffu : k > 96 ? 0xffffffffu << (128 - k) : 0u;
        ((uint4 *)(smem + M5W * M5_RING * (8 + M5_SX_BYTES)))[k] = v;
    }
    __syncthreads();                                               // (the only barrier: every wave is still here)
    if (g_begin >= g_end) return;
    // the tables start empty (the entry ring may hold anything: it is only read where table bits point)
    for (int k = lane; k < M5_SLOTS * M5_TABLE / 16; k += 64) ((uint4 *)TB)[k] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
    u32 h_carry = 0xffffffffu;                                     // hash of the slot before the one lane 0 commits next
#if MTS_M5_STATS
    u32 st_groups = 0, st_r1 = 0, st_r2 = 0, st_s1 = 0, st_s2 = 0;
    u64 st_t[6] = {0, 0, 0, 0, 0, 0};                           // (MTS_M5_STATS=2) wave clocks: commit, masks, newest word, other 96, store, groups without an owned slot
    u64 st_mark = 0;
#define M5_MARK(k) do { if (MTS_M5_STATS == 2) { const u64 t_ = (u64)clock64(); st_t[k] += t_ - st_mark; st_mark = t_; } } while (0)
#endif
    u32 rc_carry = 0;                                              // its position
    u32 run_carry = 0;                                             // slots between the newest run start and lane 0 of the group being committed (capped)
    const u32 le_lo = lane >= 31 ? 0xffffffffu : (2u << lane) - 1, le_hi = lane < 32 ? 0u : lane == 63 ? 0xffffffffu : (2u << (lane - 32)) - 1;      // lanes <= this one
    u32 inv[6];                                                    // lane r builds row r of the first table: bit j of r clear -> all ones
#pragma unroll
    for (int j = 0; j < 6; j++) inv[j] = ((lane >> j) & 1) ? 0u : 0xffffffffu;
    // slot idx -> position -> its 13 bytes -> entry; enters the ring at idx & 255, replacing slot idx - 256.  The ring
    // positions of a 64-slot group are two whole words of every table row: they are cleared and set again (the first table,
    // whose keys repeat most -- 32 lanes adding the same bit to the same word would be serialised by the LDS -- is rebuilt
    // from five ballots by the lane that owns the row; the others take one atomic OR per slot).  `nbv` = the chain behind
    // the slot = the slots back to the start of its hash run (at most 128 matter), from the ballot of the run starts.
    auto commit = [&](int idx, u32 rc, u32 lo, u32 hi, u64 x, u32 (&key)[M5_LEVELS], u32 &nbv, u64 &vm) -> u64 {
        const u32 rp = (u32)idx & (M5_RING - 1), word = rp >> 5, bit = 1u << (rp & 31);
        // (lane masks are made by single comparisons and combined as scalars: the ballot of a compound condition goes through a
        //  0 / 1 register and a second comparison)
        const bool valid = (u32)idx < wlen;                        // (a negative index is a huge one)
        vm = ballot64(valid);
        const u64 ce = valid ? make_entry(rc, lo, hi) : ~0ull;
        // a slot starts a run when its hash differs from its predecessor's (slots are committed in order)
        const u32 h = valid ? hash_of(lo) : 0xfffffffeu;
        const u32 hp = prev_lane(h, h_carry), rcp = prev_lane(rc, rc_carry);
        const u64 sr = ballot64(h != hp);
        if (vm & ~sr & ballot64(rc <= rcp)) { if (lane == 0) atomicOr(flags, 1u); }      // positions must increase inside a run
        h_carry = (u32)__builtin_amdgcn_readlane((int)h, 63);
        rc_carry = (u32)__builtin_amdgcn_readlane((int)rc, 63);
        // slots back to the start of the lane's run: lane + what the groups before carried, or lane - (the last run start at or
        // before the lane).  Runs are long (a few hundred slots on the recordings): the starts of a group are few, and each is two
        // instructions (lanes before a start wrap around to huge values and keep what they have)
        if (__builtin_popcountll(sr) <= 6) {
            nbv = (u32)lane + run_carry;
            for (u64 m = sr; m; m &= m - 1) { const u32 d = (u32)lane - (u32)__builtin_ctzll(m); nbv = d < nbv ? d : nbv; }
        } else {
            const u32 mlo = (u32)sr & le_lo, mhi = (u32)(sr >> 32) & le_hi;
            const u32 top = mhi ? 63u - (u32)__builtin_clz(mhi) : 31u - (u32)__builtin_clz(mlo | 1u);
            nbv = (mlo | mhi) ? (u32)lane - top : (u32)lane + run_carry;
        }
        nbv = nbv < 128u ? nbv : 128u;
        run_carry = sr ? (u32)__builtin_clzll(sr) + 1u : (run_carry + 64u < 128u ? run_carry + 64u : 128u);
        m5_keys((u32)(ce >> 32), key);
        const u32 wp = word & ~1u;                                 // the group's word pair
        {
            u32 *t1 = TB + 2 * (M5_TABLE / 4) + lane * M5_ROW_WORDS + wp;   // rows of tables 1..3 are contiguous: 96 rows
            t1[0] = 0; t1[1] = 0;
            if (lane < 32) { t1[64 * M5_ROW_WORDS] = 0; t1[64 * M5_ROW_WORDS + 1] = 0; }
            // (atomic ORs for this table as well take 42 vector instructions off a group -- 8 % of the kernel's -- and give them back
            //  as LDS waits: byte 3 of an even position is the high byte of a small delta, two values, 32 lanes per word;
            //  SQ_WAIT_INST_LDS x5, the same 25.2 ms)
            u32 m0 = (u32)vm, m1 = (u32)(vm >> 32);
#pragma unroll
            for (int j = 0; j < 6; j++) {
#if MTS_M5_BIT_AND
                // (the bit by v_and_b32, an opcode of the fast class -- profiles/r6_valu_issue_table.txt --, instead of the v_bfe_u32 the
                //  compiler makes of a shift and a mask)
                u32 t;
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(t) : "n"(1 << j), "v"(key[0]));
                const u64 B = ballot64(t != 0);
#else
                const u64 B = ballot64((key[0] >> j) & 1);
#endif
                m0 = and_xor(m0, (u32)B, inv[j]); m1 = and_xor(m1, (u32)(B >> 32), inv[j]);
            }
            { u32 *t0 = TB + lane * M5_ROW_WORDS + wp; t0[0] = m0; t0[1] = m1; }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int d = 1; d < M5_LEVELS; d++)
            if (valid) atomicOr(&TB[m5_slot(d) * (M5_TABLE / 4) + key[d] * M5_ROW_WORDS + word], bit);
        SE[rp] = ce;
#if MTS_M5_SX == 1
        SX[rp] = x;
#elif MTS_M5_SX == 2
        ((u32 *)(smem + M5W * M5_RING * 8))[wave * M5_RING + rp] = (u32)x;
#endif
        return ce;
    };
    // bytes 0..12 of the string at window offset r with ONE 16-byte load (the four dwords around it): the lanes are each
    // somewhere else in the window, and the address unit charges by the instruction
    typedef u32 u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    auto load16 = [&](u32 r, u32 &lo, u32 &hi, u64 &x) {
        const u8 *q = gwin + (r & ~3u);
        const u32x4_a4 w = *(const u32x4_a4 *)q;
        lo = alignbyte(w.y, w.x, r);
        hi = alignbyte(w.z, w.y, r);
        const u32 b8 = alignbyte(w.w, w.z, r), b12 = alignbyte(0u, w.w, r);
        x = (u64)__builtin_amdgcn_alignbit(b8, hi, 24) | ((u64)(__builtin_amdgcn_alignbit(b12, b8, 24) & 0xffffu) << 32);      // bytes 7..10, 11..12
    };
    // pipeline: (rc, lo, hi) of the group about to be walked, rc of the one after
    u32 rc_c = r_0, rc_n = r_1;
    u32 lo_c, hi_c;
    u64 x_c;
    {
        u32 kk[M5_LEVELS], nb;
        u64 vmm;
        const u32 ra = r_ha, rb = r_hb;
        u32 la, ha, lb, hb;
        u64 xa, xb;
        load16(ra, la, ha, xa);
        load16(rb, lb, hb, xb);
        load16(rc_c, lo_c, hi_c, x_c);
        *sink = 0;                                                 // (a store behind the loads, as at the end of every iteration)
        commit(i_first - 128, ra, la, ha, xa, kk, nb, vmm);
        __builtin_amdgcn_wave_barrier();
        commit(i_first - 64, rb, lb, hb, xb, kk, nb, vmm);
    }
    for (u32 g = g_begin; g < g_end; g++) {
        const u32 i0 = g * 64, i = i0 + lane;
        u32 key[M5_LEVELS], nbv;
        __builtin_amdgcn_wave_barrier();
#if MTS_M5_STATS == 2
        st_mark = (u64)clock64();
#endif
        u64 vm;
        const u64 e = commit((int)i, rc_c, lo_c, hi_c, x_c, key, nbv, vm);
        const u64 ex = x_c;
        __builtin_amdgcn_wave_barrier();
        // next group's words, and the position of the one after
        rc_c = rc_n;
        load16(rc_c, lo_c, hi_c, x_c);
        rc_n = slot_fwd(i + 128);
        const u32 e0 = (u32)e, e1 = (u32)(e >> 32);
        const u32 rel_p = e0 & REL_MASK;
        const bool own = i < wlen && rel_p >= halo;
        const u64 ownm = vm & ballot64(rel_p >= halo);             // (the same as a lane mask, without the ballot of a compound condition)
#if MTS_M5_STATS == 2
        if (!ownm) { *sink = 0; M5_MARK(5); continue; }
        M5_MARK(0);
#else
        if (!ownm) { *sink = 0; continue; }                             // (the store every path has: see the end of the loop)
#endif
        const u32 p_abs = td.w + rel_p;
        const u32 look = n - p_abs;
        const u32 maxlen = look < (u32)MAX_MATCH ? look : (u32)MAX_MATCH;
        const u32 nice = (u32)cfg.nice < look ? (u32)cfg.nice : look;
        // window offsets a candidate must lie above: the head of the chain, the others
        const int lim1 = (int)(p_abs > (u32)MAX_DIST + 1 ? p_abs - MAX_DIST - 1 : 0) - (int)td.w;
        const int limn = (int)(p_abs > (u32)MAX_DIST ? p_abs - MAX_DIST : 0) - (int)td.w;
        // the 128 slots before this lane's own are ring positions lo .. lo + 127 (mod 256); candidate j
        // (1 = newest) is bit 128 - j of the masks.  V = the candidates inside this lane's chain budget.
        const u32 lo = (i + 128) & (M5_RING - 1), w0 = lo >> 5, sh = lo & 31, lo8 = lo << 3;
        nbv = own ? (nbv < chain ? nbv : chain) : 0;
        u32 V[4], A4[4], A5[4], A6[4], A7[4];                     // V = inside the budget; A_d = V & "first d bytes may match"
        {   // the newest nbv of the 128 bits: one 16-byte read of a table of the 129 masks (24 instructions of shifts and selects otherwise)
            const uint4 v = VLUT[nbv];
            V[0] = v.x; V[1] = v.y; V[2] = v.z; V[3] = v.w;
        }
        auto rowmask = [&](const int d, const u32 (&in)[4], u32 (&out)[4]) __attribute__((always_inline)) {
            const u8 *row = (const u8 *)(TB + m5_slot(d) * (M5_TABLE / 4) + key[d] * M5_ROW_WORDS);
            // (byte offsets ((w0 + k) << 2) & 28: an add-shift and an AND per word; as (w0 + k) & 7 first it was an add, an AND and a shift)
            auto word = [&](const u32 k) -> u32 { return *(const u32 *)(row + (((w0 + k) << 2) & 28u)); };
            const u32 W0 = word(0), W1 = word(1), W2 = word(2), W3 = word(3), W4 = word(4);
            out[0] = in[0] & __builtin_amdgcn_alignbit(W1, W0, sh);
            out[1] = in[1] & __builtin_amdgcn_alignbit(W2, W1, sh);
            out[2] = in[2] & __builtin_amdgcn_alignbit(W3, W2, sh);
            out[3] = in[3] & __builtin_amdgcn_alignbit(W4, W3, sh);
        };
        rowmask(0, V, A4);
        rowmask(1, A4, A5);
        rowmask(2, A5, A6);
        rowmask(3, A6, A7);
#if MTS_M5_STATS == 2
        asm volatile("" :: "v"(A7[0]), "v"(A7[1]), "v"(A7[2]), "v"(A7[3]));
        M5_MARK(1);
#endif
        // The walks.  NEAR_END = some lane of the group is within 258 bytes of the end of its chunk (one group in a thousand of a
        // chunk's last tile): match lengths are capped by what is left and nice_match shrinks with it.  Everywhere else both are
        // constants, the cap never binds below the long compare, and nice_match (>= 16 at every level) can only be reached there:
        // the common path of a round carries neither the cap nor the test.
        int best = -1, qbest = -1;                                // match lengths - 3 (-1: none)
        u32 bdist = 0, qdist = 0;
        asm volatile("" : "+v"(best), "+v"(bdist));          // (two registers from here on: as constants they are made again on every path that does not change them)
        auto walks = [&](auto near_end) __attribute__((always_inline)) {
        constexpr bool NEAR_END = decltype(near_end)::value;
        // a lane whose walk has ended (nice_match reached) holds a match of >= 3 bytes: every later pick of it is one of the A masks,
        // and those are emptied where the walk ends -- nothing asks "has it ended?" afterwards
        auto set_stop = [&]() {
#pragma unroll
            for (int w = 0; w < 4; w++) { A4[w] = 0; A5[w] = 0; A6[w] = 0; A7[w] = 0; }
        };
        // one candidate: scored against this lane's string; a longer match than the one held is taken and `narrow()` leaves the
        // candidates that can still beat it; `finish()` ends the lane's walk (nice_match reached).  Everything a candidate can do
        // to the walk's state happens in the branch where it is found out: nothing is handed to the code behind through a value
        // that the other paths would have to make up (each of those was a v_mov per round and path).
        auto cand = [&](const u32 o8 /* byte offset of the candidate's ring slot */, const u32 c0, const u32 c1, const u32 rel_c, auto &&narrow, auto &&finish) __attribute__((always_inline)) {
            const u32 x0 = (c0 ^ e0) >> REL_BITS, x1 = c1 ^ e1;
            if ((x0 & 0x1ff) == 0) {
                bool all4;
                int len = m5_len04(x1, all4);                         // (length - 3)
                if (all4) len = 4;
                if (all4 && (x0 >> 9) == 0) {
#if MTS_M5_SX == 1
                    const u64 y = ring_x(o8) ^ ex;
                    const u32 len0 = 13;
#elif MTS_M5_SX == 2
                    const u32 y = ring_x(o8) ^ (u32)ex;
                    const u32 len0 = 11;
#else
                    const u32 y = wread(rel_c + 7) ^ (u32)ex;       // bytes 7..10 (this lane's own are in registers)
                    const u32 len0 = 11;
#endif
                    if (y) len = 4 + (int)((u32)__builtin_ctzll((u64)y) >> 3);
                    else {
                        const u32 cap = NEAR_END ? maxlen : (u32)MAX_MATCH;
                        u32 full = len0;                                // (bytes, in this branch)
                        while (full < cap) {
                            const u32 x = wread(rel_c + full) ^ wread(rel_p + full);
                            if (x) { full += (u32)__builtin_ctz(x) >> 3; break; }
                            full += 4;
                        }
                        full = full < cap ? full : cap;
                        len = (int)full - 3;
                        // (a candidate that reaches nice_match is an improvement: a match that long already held would have ended the walk)
                        if (!NEAR_END && full >= (u32)cfg.nice) { best = len; bdist = rel_p - rel_c; set_stop(); finish(); }
                    }
                }
                if (NEAR_END) len = len < (int)maxlen - 3 ? len : (int)maxlen - 3;
                if (len > best) {
                    best = len; bdist = rel_p - rel_c;
                    narrow();                                           // fewer candidates can still win now
                    if (NEAR_END && len >= (int)nice - 3) { set_stop(); finish(); }
                }
            }
        };
        // candidates of word w restricted to `part`, newest first
        auto walk = [&](const u32 tb, const u32 &m0, const u32 &m1, const u32 &m2, const u32 &m3, const u32 &m4, const u32 part, const bool head) __attribute__((always_inline)) {      // (the masks by reference: a walk that ends empties them)
            // (the empty asm statements keep the compiler from turning the select chain into a table in scratch memory)
            auto pick = [&]() -> u32 {
                u32 r = best >= 0 ? m1 : m0;
                asm volatile("" : "+v"(r));
                r = best >= 1 ? m2 : r;
                asm volatile("" : "+v"(r));
                r = best >= 2 ? m3 : r;
                asm volatile("" : "+v"(r));
                return best >= 3 ? m4 : r;
            };
            auto pick_longer = [&]() -> u32 {                      // the same once a match is held (best >= 3)
                u32 r = best >= 1 ? m2 : m1;
                asm volatile("" : "+v"(r));
                r = best >= 2 ? m3 : r;
                asm volatile("" : "+v"(r));
                return best >= 3 ? m4 : r;
            };
            u32 el = head ? m0 & part : pick() & part;             // (the head's walk is the first: nothing is held yet)
            if (head) {
                // The head of the chain (the slot before this lane's own: bit 31 of the newest word) is every lane's first candidate,
                // and the only one that may be MAX_DIST away (zlib checks the head against MAX_DIST, the others against the limit
                // one nearer): scored here, by all lanes at once, so that the rounds below know one limit and one kind of candidate.
#if MTS_M5_STATS
                st_r1++; st_s1 += (u32)__popcll(ballot64(el != 0));
#endif
                const u32 o8 = lo8 + 127 * 8;
                const u64 c = ring_e(o8);
                if ((int)el < 0) {                                  // (bit 31: the lane has a chain)
                    el &= 0x7fffffffu;
                    const u32 c0 = (u32)c, c1 = (u32)(c >> 32);
                    const u32 rel_c = c0 & REL_MASK;
                    if ((int)rel_c > lim1) cand(o8, c0, c1, rel_c, [&]() { el &= pick_longer(); }, [&]() { el = 0; });
                    else el = 0;
                }
            }
            while (any64(el != 0)) {
#if MTS_M5_STATS
                st_r1++; st_s1 += (u32)__popcll(ballot64(el != 0));
#endif
                {
                    // The newest candidate left: bit 31 - z; out of the mask by a shifted constant, its ring offset by one multiply-add.
                    // EVERY lane goes through this part, also one without candidates (z = -1: it reads some slot of the ring and is
                    // masked out with the lanes whose candidate is out of range): under `if (el)` the lanes that skip it would need
                    // their zero made and copied at the join, two moves per round, for instructions the wave issues anyway.
                    const bool had = el != 0;
                    const u32 z = m5_ffbh(el);
                    const u32 t = tb + 31 - z;
                    const u32 o8 = m5_off8(z, lo8 + ((tb + 31) << 3));
                    const u64 c = ring_e(o8);
                    const u32 c0 = (u32)c, c1 = (u32)(c >> 32);
                    const u32 rel_c = c0 & REL_MASK;
                    // out of range: so is everything older -- the lane's mask goes by an AND, not in an else branch whose zero every path would
                    // have to carry (a later walk of the lane finds the same out with its first candidate: rare)
                    const bool in_range = had && (int)rel_c > ((!head && t == 127) ? lim1 : limn);
                    u32 rm = in_range ? 0xffffffffu : 0u;
                    asm volatile("" : "+v"(rm));                      // (a value: the compiler would make the branch of it again)
                    el = el & ~(0x80000000u >> (z & 31)) & rm;
                    if (in_range) cand(o8, c0, c1, rel_c, [&]() { el &= pick_longer(); }, [&]() { el = 0; });
                }
            }
        };
        // candidates 1 .. qchain first: what the walk holds then is the quarter-budget result
        const u32 qpart = qchain >= 32 ? 0xffffffffu : ~(0xffffffffu >> qchain);        // qchain <= 32 (chain <= 128)
#define MTS_WALK(w, part, head) walk(32 * (w), V[w], A4[w], A5[w], A6[w], A7[w], part, head)
        MTS_WALK(3, qpart, true);
        qbest = best; qdist = bdist;
        if (qpart != 0xffffffffu) MTS_WALK(3, ~qpart, false);
#undef MTS_WALK
#if MTS_M5_STATS == 2
        M5_MARK(2);
#endif
        {
            // The other 96 candidates in ONE loop (a lane takes its own next candidate, whichever of the three words it is in):
            // word by word the wave ran as many rounds as the busiest lane of EACH word needed; together it is the busiest lane
            // over all three.
            auto pickw = [&](const int w) __attribute__((always_inline)) -> u32 {
                u32 r = best >= 1 ? A5[w] : A4[w];                   // (after an improvement: a match is held)
                asm volatile("" : "+v"(r));
                r = best >= 2 ? A6[w] : r;
                asm volatile("" : "+v"(r));
                return best >= 3 ? A7[w] : r;
            };
            // (the three words' first picks share their four comparisons: the lane masks are kept and the selects take them as they are)
            const u64 c3 = ballot64(best >= 0), c4 = ballot64(best >= 1), c5 = ballot64(best >= 2), c6 = ballot64(best >= 3);
            auto pick0 = [&](const int w) __attribute__((always_inline)) -> u32 {
                u32 r = sel64(c3, A4[w], V[w]);
                r = sel64(c4, A5[w], r);
                r = sel64(c5, A6[w], r);
                return sel64(c6, A7[w], r);
            };
            u32 f2 = pick0(2), f1 = pick0(1), f0 = pick0(0);
            u32 ko2 = lo8 + ((64 + 31) << 3), ko1 = lo8 + ((32 + 31) << 3), ko0 = lo8 + (31 << 3);
            asm volatile("" : "+v"(ko2), "+v"(ko1), "+v"(ko0));      // (three registers for the loop: taken apart again they are an addition per round)
            while (any64((f2 | f1 | f0) != 0)) {
#if MTS_M5_STATS
                st_r2++; st_s2 += (u32)__popcll(ballot64((f2 | f1 | f0) != 0));
#endif
                if (f2 | f1 | f0) {                                  // (here the branch is cheaper than masking the lanes without candidates: measured in instructions)
                    const bool t2 = f2 != 0, t1 = f1 != 0;
                    const u32 cur = t2 ? f2 : t1 ? f1 : f0;
                    const u32 tbo = t2 ? ko2 : t1 ? ko1 : ko0;                   // ring offset of bit 31 of the word taken
                    const u32 z = (u32)__builtin_clz(cur);
                    const u32 o8 = m5_off8(z, tbo);
                    const u64 c = ring_e(o8);
                    const u32 c0 = (u32)c, c1 = (u32)(c >> 32);
                    const u32 rel_c = c0 & REL_MASK;
                    // the word the candidate came from goes back without it -- or, the candidate out of range (so is everything
                    // older), all three as nothing: by a mask, not in an else branch whose zeros every path would have to carry
                    const bool in_range = (int)rel_c > limn;
                    u32 rm = in_range ? 0xffffffffu : 0u;
                    asm volatile("" : "+v"(rm));                      // (a value: the compiler would make the branch of it again)
                    const u32 ncur = cur & ~(0x80000000u >> z);
                    f2 = (t2 ? ncur : f2) & rm;
                    f1 = ((!t2 && t1) ? ncur : f1) & rm;
                    f0 = ((!t2 && !t1) ? ncur : f0) & rm;
                    if (in_range) cand(o8, c0, c1, rel_c, [&]() { f2 &= pickw(2); f1 &= pickw(1); f0 &= pickw(0); }, [&]() { f2 = 0; f1 = 0; f0 = 0; });
                }
            }
        }
        };
        if (ownm & ballot64(look < (u32)MAX_MATCH)) walks(std::true_type{}); else walks(std::false_type{});
#if MTS_M5_STATS == 2
        M5_MARK(3);
#endif
        {   // te_store(), with the table word stored by EVERY lane, on every path (lanes that own no position: into the sink).  The
            // loads of the next group's bytes and of the keys behind it went out at the top of this iteration and are waited for at
            // the top of the next; the counter they are waited on counts stores as well, in order of issue, and the compiler must
            // allow for the path with the fewest operations behind a load: with the store under a branch that is none, the wait
            // became vmcnt(0), and every wave sat out the round trip of its own 64 scattered stores before it entered the next
            // group (a quarter of its time: tools/m5_stats.py with -DMTS_M5_STATS=2).  One store on every path: vmcnt(1).
            const u32 f = bdist | ((u32)(best > 0 ? best : 0) << 15), q = qdist | ((u32)(qbest > 0 ? qbest : 0) << 15);      // te_pack() of lengths - 3
            u32 e = f;
            const bool differ = q != f, side = differ && qbest > (int)cfg.good - 3;      // (the side table's word is wanted: asked of the comparisons, not of the flag in e)
            if (differ) e |= side ? TE_QSIDE : TE_QNONE;
            if (own && (side || all_quarters)) TQ[p_abs] = q;      // (rare; before the table word: the waits allow ONE younger operation)
            u32 *dst = own ? T + p_abs : sink;
            *dst = e;
        }
#if MTS_M5_STATS
        st_groups++;
#if MTS_M5_STATS == 2
        M5_MARK(4);
#endif
#endif
    }
#if MTS_M5_STATS
    if (lane == 0) { atomicAdd(&g_m5_stats[0], st_groups); atomicAdd(&g_m5_stats[1], st_r1); atomicAdd(&g_m5_stats[2], st_r2); atomicAdd(&g_m5_stats[3], st_s1); atomicAdd(&g_m5_stats[4], st_s2); for (int k = 0; k < 6; k++) atomicAdd(&g_m5_stats[5 + k], st_t[k]); }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_match6 (chain budget > 128: levels 7..9): k_match5's looked-up filter masks and walk, block after block of 128
// candidates.  The eight waves of a workgroup take eight consecutive groups
// of 64 slots and go through the blocks together: what their lanes' candidates of block b lie in is 640 consecutive slots of
// the sorted order, 128 slots older with every block -- ONE ring of 1024 slots per workgroup (entries, bytes 7..12, the bits
// of the four key tables) into which two waves enter the 128 new slots of the next block while the current block is walked;
// nothing is ever staged twice.  (Round 1's kernel for these levels restaged 192 slots per wave and block and compared two byte
// keys per candidate with SWAR arithmetic: ~840 instructions per block before the first candidate was looked at.)
// The quarter-budget result is what the walk holds when it has seen chain/4 candidates: at a block boundary for chain
// 1024 and 4096, between the two halves of the first block for chain 256.
// ------------------------------------------------------------------------------------------------
constexpr int M6_RING = 1024;
constexpr int M6_ROW_WORDS = M6_RING / 32 + 1;
constexpr int M6_TABLE = M5_ROWS * M6_ROW_WORDS * 4;
constexpr int MATCH6_LDS = 2 * M6_RING * 8 + M5_SLOTS * M6_TABLE + M5_VLUT;       // 39568 per workgroup (rings, tables, the table of budget masks)

__global__ __launch_bounds__(M5_WAVES * 64) void k_match6(const u8 *__restrict__ stream, const TileDesc *__restrict__ tiles, int n_tiles, int nsl,
                                                          const u32 *__restrict__ sorted, u32 *__restrict__ tables, u32 *__restrict__ quarter, LevelCfg cfg,
                                                          u32 *__restrict__ flags, int all_quarters)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    __shared__ u32 wg_h[M5_WAVES], wg_rc[M5_WAVES], wg_tail[M5_WAVES], wg_min;      // per group of the set: hash and position of its last slot, slots since its last run start (0: none starts in it)
    const u32 xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const u32 tile_id = (jb / (u32)nsl) * 8 + xcd, slice = jb % (u32)nsl;
    if (tile_id >= (u32)n_tiles) return;
    const TileDesc td = tiles[tile_id];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u8 *gwin = stream + td.stream_off + td.w;
    auto wread = [&](u32 addr) -> u32 { return gld_u32_unaligned(gwin, addr); };
    u64 *SE = (u64 *)smem;
    u64 *SX = SE + M6_RING;
    u32 *TB = (u32 *)(smem + 2 * M6_RING * 8);                     // [level][row][32 words + 1]
    u32 *T = tables + td.stream_off, *TQ = quarter + td.stream_off;
    if (threadIdx.x < 2 && slice == 0) {
        const u32 hashed_end = td.w + td.wlen;
        const u32 p = hashed_end + threadIdx.x;
        if (p >= td.a && p < td.own_end) { T[p] = 0; if (all_quarters) TQ[p] = 0; }
    }
    if (td.wlen == 0) return;
    const u32 *sk = sorted + td.sorted_off;
    const u32 wlen = td.wlen, n = td.n;
    const u32 ngroups = (wlen + 63) / 64;
    const u32 halo = td.a - td.w;
    const u32 chain = (u32)cfg.chain, qchain = (u32)cfg.chain >> 2;
    const u32 gpb = ((ngroups + (u32)nsl - 1) / (u32)nsl + M5_WAVES - 1) / M5_WAVES * M5_WAVES;      // groups per workgroup: whole sets of eight
    const u32 gb_begin = slice * gpb, gb_end = min(ngroups, gb_begin + gpb);
    if (gb_begin >= gb_end) return;                                // (the whole workgroup)
    for (int k = threadIdx.x; k < M5_SLOTS * M6_TABLE / 4; k += M5_WAVES * 64) TB[k] = 0;
    const uint4 *VLUT = (const uint4 *)(smem + 2 * M6_RING * 8 + M5_SLOTS * M6_TABLE);      // [0 .. 128]: the newest n of 128 bits (as in k_match5)
    for (int k = threadIdx.x; k <= 128; k += M5_WAVES * 64) {
        uint4 v;
        v.w = k >= 32 ? 0xffffffffu : k ? 0xffffffffu << (32 - k) : 0u;
        v.z = k >= 64 ? 0xffffffffu : k > 32 ? 0xffffffffu << (64 - k) : 0u;
        v.y = k >= 96 ? 0xffffffffu : k > 64 ? 0xffffffffu << (96 - k) : 0u;
        v.x = k >= 128 ? 0xffffffffu : k > 96 ? 0xffffffffu << (128 - k) : 0u;
        ((uint4 *)(smem + 2 * M6_RING * 8 + M5_SLOTS * M6_TABLE))[k] = v;
    }
    u32 inv[6];                                                    // lane r builds row r of the first table: bit j of r clear -> all ones
#pragma unroll
    for (int j = 0; j < 6; j++) inv[j] = ((lane >> j) & 1) ? 0u : 0xffffffffu;
    typedef u32 u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    auto load16 = [&](u32 r, u32 &lo, u32 &hi, u64 &x) {
        const u8 *q = gwin + (r & ~3u);
        const u32x4_a4 w = *(const u32x4_a4 *)q;
        lo = alignbyte(w.y, w.x, r);
        hi = alignbyte(w.z, w.y, r);
        const u32 b8 = alignbyte(w.w, w.z, r), b12 = alignbyte(0u, w.w, r) & 0xffu;
        x = (u64)(hi >> 24) | ((u64)b8 << 8) | ((u64)b12 << 40);                    // bytes 7..12
    };
    // a batch = 64 consecutive slots from idx0 (a multiple of 64, possibly negative or beyond the window: no bits then)
    struct Batch { bool valid; u32 rc, lo, hi; u64 x; };
    auto fetch = [&](int idx0, Batch &bt) {
        const int idx = idx0 + lane;
        bt.valid = idx >= 0 && (u32)idx < wlen;
        bt.rc = bt.valid ? sk[idx] & REL_MASK : 0;
        load16(bt.rc, bt.lo, bt.hi, bt.x);
    };
    // it takes word pair (idx0 / 32) mod 32 of every table row: cleared and set again (first table from five ballots by the
    // lane that owns the row, the others by atomic OR, as in k_match5)
    auto commit = [&](int idx0, const Batch &bt) {
        const u32 rp = (u32)(idx0 + lane) & (M6_RING - 1), word = rp >> 5, bit = 1u << (rp & 31);
        const u64 ce = bt.valid ? make_entry(bt.rc, bt.lo, bt.hi) : ~0ull;
        u32 key[M5_LEVELS];
        m5_keys((u32)(ce >> 32), key);
        const u32 wp = word & ~1u;
        u32 *t1 = TB + 2 * (M6_TABLE / 4) + lane * M6_ROW_WORDS + wp;       // rows of tables 1..3 are contiguous: 96 rows
        t1[0] = 0; t1[1] = 0;
        if (lane < 32) { t1[64 * M6_ROW_WORDS] = 0; t1[64 * M6_ROW_WORDS + 1] = 0; }
        const u64 vm = ballot64(bt.valid);
        u32 m0 = (u32)vm, m1 = (u32)(vm >> 32);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const u64 B = ballot64((key[0] >> j) & 1);
            m0 = and_xor(m0, (u32)B, inv[j]); m1 = and_xor(m1, (u32)(B >> 32), inv[j]);
        }
        { u32 *t0 = TB + lane * M6_ROW_WORDS + wp; t0[0] = m0; t0[1] = m1; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int d = 1; d < M5_LEVELS; d++)
            if (bt.valid) atomicOr(&TB[m5_slot(d) * (M6_TABLE / 4) + key[d] * M6_ROW_WORDS + word], bit);
        SE[rp] = ce;
        SX[rp] = bt.x;
    };
    // The chain behind a slot = the slots back to the start of its hash run (the sorted order lists a run's positions in a row):
    // inside a group from the ballot of the run starts, across groups carried along -- c_tail = slots since the last run start
    // before the set's first group (capped), c_h / c_rc = hash and position of the slot before it; the same in every wave.
    // Before the workgroup's first group the whole workgroup looks back for the start of that run.
    u32 c_h = 0xfffffffdu, c_rc = 0, c_tail = 0;
    if (gb_begin > 0) {
        const u32 i1 = gb_begin * 64 - 1;
        c_rc = sk[i1] & REL_MASK;
        c_h = hash_of(wread(c_rc));
        if (threadIdx.x == 0) wg_min = 0xffffffffu;
        __syncthreads();
        for (u32 r = 0; r * (M5_WAVES * 64) < chain; r++) {
            const u32 t = 1 + threadIdx.x + r * (M5_WAVES * 64);            // is slot i1 - t still in the run of slot i1?
            const bool mism = t > i1 || hash_of(wread(sk[i1 - (t > i1 ? 0 : t)] & REL_MASK)) != c_h;
            if (mism) atomicMin(&wg_min, t);
            __syncthreads();
            const u32 found = wg_min;
            // Everybody has read it before anybody goes round again: a wave that read "nothing yet" and went on to the next round's
            // atomicMin let a slower wave read THAT and leave the loop a round early -- the two then met at different barriers, and
            // everything behind (ring, run carries) was out of step.  Only budgets > 512 make a second round (level 9, chains that
            // reach back further than that: zeros), where the run-order guard then fired at random: found by the fuzzer in round 4.
            __syncthreads();
            if (found != 0xffffffffu) break;
        }
        c_tail = wg_min < chain ? wg_min : chain;
        __syncthreads();
    }
    for (u32 gs = gb_begin; gs < gb_end; gs += M5_WAVES) {
        const int G0 = (int)gs * 64;
        const u32 g = gs + (u32)wave;
        const bool active = g < gb_end;
        const u32 i = g * 64 + lane;
        const u32 rp0 = (active && i < wlen) ? sk[i] & REL_MASK : 0;
        u32 own_lo, own_hi;
        u64 ex;
        load16(rp0, own_lo, own_hi, ex);
        // block 0 needs slots G0 - 128 .. G0 + 511: ten batches, wave v enters batches v and v + 8
        Batch ba, bb;
        fetch(G0 - 128 + 64 * wave, ba);
        if (wave < 2) fetch(G0 - 128 + 64 * (wave + 8), bb);
        const u64 e = make_entry(rp0, own_lo, own_hi);
        const u32 e0 = (u32)e, e1 = (u32)(e >> 32);
        const u32 rel_p = e0 & REL_MASK;
        const bool own = active && i < wlen && rel_p >= halo;
        u32 key[M5_LEVELS];
        m5_keys(e1, key);
        const u32 p_abs = td.w + rel_p;
        const u32 look = n - p_abs;
        const u32 maxlen = look < (u32)MAX_MATCH ? look : (u32)MAX_MATCH;
        const u32 nice = (u32)cfg.nice < look ? (u32)cfg.nice : look;
        const int lim1 = (int)(p_abs > (u32)MAX_DIST + 1 ? p_abs - MAX_DIST - 1 : 0) - (int)td.w;
        const int limn = (int)(p_abs > (u32)MAX_DIST ? p_abs - MAX_DIST : 0) - (int)td.w;
        u32 best = 2, bdist = 0, qbest = 2, qdist = 0;
        bool stop = false, qtaken = false;
        const bool have = active && i < wlen;
        const u32 h_own = have ? hash_of(own_lo) : 0xfffffffeu;
        if (lane == 63) { wg_h[wave] = h_own; wg_rc[wave] = rel_p; }
        __syncthreads();                                           // the ring is free (first set: the tables are zero)
        commit(G0 - 128 + 64 * wave, ba);
        if (wave < 2) commit(G0 - 128 + 64 * (wave + 8), bb);
        u32 nbv;
        {
            u32 hp = __shfl_up(h_own, 1, 64), rcp = __shfl_up(rel_p, 1, 64);
            if (lane == 0) { hp = wave ? wg_h[wave - 1] : c_h; rcp = wave ? wg_rc[wave - 1] : c_rc; }
            const bool starts_run = h_own != hp;
            if (any64(have && !starts_run && rel_p <= rcp)) { if (lane == 0) atomicOr(flags, 1u); }      // positions must increase inside a run (see k_match5)
            const u64 sr = ballot64(starts_run);
            if (lane == 0) wg_tail[wave] = sr ? (u32)__builtin_clzll(sr) + 1u : 0u;
            __syncthreads();
            u32 t = c_tail;                                         // slots since the last run start before this wave's group
            for (int w = 0; w < wave; w++) { const u32 x = wg_tail[w]; t = x ? x : (t + 64u < chain ? t + 64u : chain); }
            const u32 le_lo = lane >= 31 ? 0xffffffffu : (2u << lane) - 1, le_hi = lane < 32 ? 0u : lane == 63 ? 0xffffffffu : (2u << (lane - 32)) - 1;
            const u32 mlo = (u32)sr & le_lo, mhi = (u32)(sr >> 32) & le_hi;
            const u32 top = mhi ? 63u - (u32)__builtin_clz(mhi) : 31u - (u32)__builtin_clz(mlo | 1u);
            nbv = (mlo | mhi) ? (u32)lane - top : (u32)lane + t;
            nbv = own ? (nbv < chain ? nbv : chain) : 0;
            // what the next set starts from (every wave works it out for itself)
            u32 tt = c_tail;
            for (int w = 0; w < M5_WAVES; w++) { const u32 x = wg_tail[w]; tt = x ? x : (tt + 64u < chain ? tt + 64u : chain); }
            c_tail = tt; c_h = wg_h[M5_WAVES - 1]; c_rc = wg_rc[M5_WAVES - 1];
        }
        // The 128 slots block b + 1 adds lie just below what block b reads (other ring positions, other table words, and what
        // they replace in the ring is 1024 slots newer: beyond anything still read), so they are entered DURING block b, by the
        // wave pair (b mod 4), which fetched them during block b - 1: one barrier per block, and entering overlaps walking.
        if (wave < 2) fetch(G0 - 256 + 64 * wave, ba);
        for (u32 jbase = 0, blk = 0;; jbase += 128, blk++) {
            if ((u32)(wave >> 1) == (blk & 3)) commit(G0 - (int)jbase - 256 + 64 * (wave & 1), ba);
            if ((u32)(wave >> 1) == ((blk + 1) & 3)) fetch(G0 - (int)jbase - 384 + 64 * (wave & 1), ba);
            if (jbase == qchain && !qtaken) { qbest = best; qdist = bdist; qtaken = true; }      // (chain / 4 a multiple of 128)
            const u32 nbl = nbv > jbase ? (nbv - jbase < 128 ? nbv - jbase : 128) : 0;          // candidates of this lane in the block
            if (any64(nbl != 0 && !stop)) {
                // candidate jj (1 = newest) of this lane is slot i - jbase - jj: bit 128 - jj of the 128 ring positions from lo
                const u32 lo = (u32)((int)i - (int)jbase - 128) & (M6_RING - 1), w0 = lo >> 5, sh = lo & 31;
                u32 V[4], A4[4], A5[4], A6[4], A7[4];                 // V = inside the budget; A_d = V & "first d bytes may match"
                {   // the newest nbl of the 128 bits: one 16-byte read of the table of the 129 masks
                    const uint4 v = VLUT[nbl];
                    V[0] = v.x; V[1] = v.y; V[2] = v.z; V[3] = v.w;
                }
                auto rowmask = [&](const int d, const u32 (&in)[4], u32 (&out)[4]) __attribute__((always_inline)) {
                    const u8 *row = (const u8 *)(TB + m5_slot(d) * (M6_TABLE / 4) + key[d] * M6_ROW_WORDS);
                    auto word = [&](const u32 k) -> u32 { return *(const u32 *)(row + (((w0 + k) << 2) & 124u)); };      // (byte offsets: see k_match5)
                    const u32 W0 = word(0), W1 = word(1), W2 = word(2), W3 = word(3), W4 = word(4);
                    out[0] = in[0] & __builtin_amdgcn_alignbit(W1, W0, sh);
                    out[1] = in[1] & __builtin_amdgcn_alignbit(W2, W1, sh);
                    out[2] = in[2] & __builtin_amdgcn_alignbit(W3, W2, sh);
                    out[3] = in[3] & __builtin_amdgcn_alignbit(W4, W3, sh);
                };
                rowmask(0, V, A4);
                rowmask(1, A4, A5);
                rowmask(2, A5, A6);
                rowmask(3, A6, A7);
                auto pickw = [&](const int w) __attribute__((always_inline)) -> u32 {
                    u32 r = best >= 3 ? A4[w] : V[w];
                    asm volatile("" : "+v"(r));
                    r = best >= 4 ? A5[w] : r;
                    asm volatile("" : "+v"(r));
                    r = best >= 5 ? A6[w] : r;
                    asm volatile("" : "+v"(r));
                    return best >= 6 ? A7[w] : r;
                };
                const bool first = jbase == 0;
                // a lane takes its own next candidate, newest first, whichever of the enabled words it is in
                auto walk = [&](const bool u3, const bool u2, const bool u1, const bool u0) __attribute__((always_inline)) {
                    u32 f3 = (stop || !u3) ? 0u : pickw(3), f2 = (stop || !u2) ? 0u : pickw(2);
                    u32 f1 = (stop || !u1) ? 0u : pickw(1), f0 = (stop || !u0) ? 0u : pickw(0);
                    while (any64((f3 | f2 | f1 | f0) != 0)) {
                        if (f3 | f2 | f1 | f0) {
                            const bool t3 = f3 != 0, t2 = f2 != 0, t1 = f1 != 0;
                            const u32 cur = t3 ? f3 : t2 ? f2 : t1 ? f1 : f0;
                            const u32 tb = t3 ? 96u : t2 ? 64u : t1 ? 32u : 0u;
                            const u32 z = (u32)__builtin_clz(cur);
                            const u32 ncur = cur & ~(0x80000000u >> z);        // the word the candidate came from, without it
                            f3 = t3 ? ncur : f3;
                            f2 = (!t3 && t2) ? ncur : f2;
                            f1 = (!t3 && !t2 && t1) ? ncur : f1;
                            f0 = (!t3 && !t2 && !t1) ? ncur : f0;
                            const u32 t = tb + 31 - z;
                            const u32 slot = (lo + t) & (M6_RING - 1);
                            const u64 c = SE[slot];
                            const u32 c0 = (u32)c, c1 = (u32)(c >> 32);
                            const u32 rel_c = c0 & REL_MASK;
                            if ((int)rel_c > ((first && t == 127) ? lim1 : limn)) {
                                const u32 x0 = (c0 ^ e0) >> REL_BITS, x1 = c1 ^ e1;
                                if ((x0 & 0x1ff) == 0) {
                                    u32 len = x1 ? 3 + ((u32)__builtin_ctz(x1) >> 3) : 7;
                                    if (x1 == 0 && (x0 >> 9) == 0) {
                                        const u64 y = SX[slot] ^ ex;
                                        if (y) len = 7 + ((u32)__builtin_ctzll(y) >> 3);
                                        else {
                                            len = 13;
                                            while (len < maxlen) {
                                                const u32 x = wread(rel_c + len) ^ wread(rel_p + len);
                                                if (x) { len += (u32)__builtin_ctz(x) >> 3; break; }
                                                len += 4;
                                            }
                                        }
                                    }
                                    len = len < maxlen ? len : maxlen;
                                    if (len > best) {
                                        best = len; bdist = rel_p - rel_c;
                                        if (len >= nice) stop = true;
                                        f3 &= pickw(3); f2 &= pickw(2); f1 &= pickw(1); f0 &= pickw(0);      // fewer candidates can still win now
                                    }
                                }
                            } else stop = true;                 // out of range: so is everything older
                            if (stop) { f3 = 0; f2 = 0; f1 = 0; f0 = 0; }
                        }
                    }
                };
                if (first && qchain == 64) {
                    walk(true, true, false, false);
                    qbest = best; qdist = bdist; qtaken = true;
                    walk(false, false, true, true);
                } else walk(true, true, true, true);
            } else if (jbase == 0 && qchain == 64) { qbest = best; qdist = bdist; qtaken = true; }
            // another block while any lane of the workgroup has candidates left (this is also where everybody is done reading)
            if (!__syncthreads_or(!stop && nbv > jbase + 128)) break;
        }
        if (!qtaken) { qbest = best; qdist = bdist; }
        if (own) te_store(T, TQ, p_abs, best, bdist, qbest, qdist, (u32)cfg.good, all_quarters);
    }
}

// ------------------------------------------------------------------------------------------------
// Built, measured and taken out again in round 4: k_match7, a POOLED walk (the commit before this one has it).  One ring per
// workgroup like k_match6's, every slot entered once; a lane scored two candidates of its newest word in place (64 and 48 lanes of
// 64 busy; the later rounds run with 5, the rounds of the other 96 candidates with 23, 4, 1 -- tools/sim/match_walk_sim.c,
// confirmed by the instrumented kernel, tools/m5_stats.py) and left the rest as an 80-byte work item in LDS (what it holds, its
// remaining candidates, the masks a longer match narrows them with), which the workgroup's waves worked off after a barrier,
// refilling lanes from the queue.  Bit-exact on the first run, and slower: 8 waves per workgroup 42.4 ms (every wave takes 64
// items and finds the queue empty: no refills, SQ_INSTS_VALU -5 %) / 55 ms (one wave works the queue off: -18 % instructions, seven
// waves wait), 4 waves 38-42 ms, 2 waves 34.4 ms (-12 % instructions) against k_match5's 25.2 -- three barriers per step leave
// the vector units idle, and parking, fetching and the barriers' bookkeeping (SALU x2.3) eat most of what the idle lanes cost.
// ------------------------------------------------------------------------------------------------
int launch_match(hipStream_t st, const u8 *d_stream, const TileDesc *d_tiles, int n_tiles, const u32 *d_sorted, u32 *d_tables, u32 *d_quarter,
                 LevelCfg cfg, u32 *d_flags, int all_quarters)
{
    if (n_tiles == 0) return MTS_OK;
    if (cfg.chain <= 128) {
        if (MATCH5_LDS > 65536) MTS_LDS_ATTR(k_match5, MATCH5_LDS);
        int nsl = M5_SLICES;
        if (const char *e = getenv("MTS_MATCH_SLICES")) nsl = atoi(e) > 0 ? atoi(e) : nsl;
        const int grid = (n_tiles + 7) / 8 * 8 * nsl;
        hipLaunchKernelGGL(k_match5, dim3(grid), dim3(M5W * 64), MATCH5_LDS, st, d_stream, d_tiles, n_tiles, nsl, d_sorted, d_tables, d_quarter, cfg, d_flags, all_quarters);
    } else {
        if (cfg.chain != 256 && (cfg.chain >> 2) % 128 != 0) { set_error("match: chain budget %d unsupported", cfg.chain); return MTS_E_INTERNAL; }
        const int nsl = M6_SLICES;
        const int grid = (n_tiles + 7) / 8 * 8 * nsl;
        hipLaunchKernelGGL(k_match6, dim3(grid), dim3(M5_WAVES * 64), MATCH6_LDS, st, d_stream, d_tiles, n_tiles, nsl, d_sorted, d_tables, d_quarter, cfg, d_flags, all_quarters);
    }
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

}  // namespace mts

#if MTS_M5_STATS
extern "C" int mts_debug_m5_stats(unsigned long long *out)      // (instrumented builds only: tools/m5_stats.py)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mts::g_m5_stats), 128) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(mts::g_m5_stats), z, 128) == hipSuccess ? 0 : -1;
}
#endif
