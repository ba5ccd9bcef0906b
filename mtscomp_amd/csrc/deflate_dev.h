// Device helpers shared by deflate.hip and match.hip (gfx950 only).
#pragma once
#include "common.h"

namespace mts {

// ================================================================================================
// small device helpers
// ================================================================================================
__device__ __forceinline__ u32 alignbyte(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }

__device__ __forceinline__ u32 gld_u32_unaligned(const u8 *s, u64 p)
{
    const u32 *q = (const u32 *)(s + (p & ~(u64)3));
    return alignbyte(q[1], q[0], (u32)p & 3);
}
__device__ __forceinline__ u32 hash_of(u32 b012) { return (((b012 & 0xff) << 10) ^ (((b012 >> 8) & 0xff) << 5) ^ ((b012 >> 16) & 0xff)) & 0x7fff; }

__device__ __forceinline__ u64 lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1; }

// lanes (among `active`) holding the same NB-bit digit as this lane
template <int NB>
__device__ __forceinline__ u64 match_digit(u32 d, u64 active)
{
    u64 m = active;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const bool bit = (d >> b) & 1;
        const u64 bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

__device__ __forceinline__ u32 wave_excl_scan_u32(u32 v, u32 &total)
{
    const int lane = threadIdx.x & 63;
    u32 x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}

// inclusive wave scan on the vector ALU alone (row shifts + row broadcasts: no LDS crossbar)
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 x)
{
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);     // row_bcast:15
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);     // row_bcast:31
    return x;
}

__device__ __forceinline__ u32 __reduce_max_sync_u32(u32 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (u32)__shfl_xor(v, off, 64));
    return v;
}


// A match-table entry is ONE word per position (see match.hip):  [14:0] dist (0: no match), [22:15] len - 3 = the full-budget result,
// TE_QNONE / TE_QSIDE say what the quarter-budget result is.
constexpr u32 TE_DIST = 0x7fffu, TE_QNONE = 1u << 23, TE_QSIDE = 1u << 24;
// (dist is 0 whenever len < MIN_MATCH: the walks only record a distance with a length of 3 and more)
__device__ __forceinline__ u32 te_pack(u32 len, u32 dist) { return dist | (((len > (u32)MIN_MATCH ? len : (u32)MIN_MATCH) - MIN_MATCH) << 15); }
__device__ __forceinline__ u32 te_dist(u32 e) { return e & TE_DIST; }
__device__ __forceinline__ u32 te_len(u32 e) { return (e & TE_DIST) ? ((e >> 15) & 0xffu) + MIN_MATCH : 0u; }
__device__ __forceinline__ void te_store(u32 *__restrict__ T, u32 *__restrict__ TQ, u32 p, u32 best, u32 bdist, u32 qbest, u32 qdist, u32 good, int all_quarters)
{
    const u32 f = te_pack(best, bdist), q = te_pack(qbest, qdist);
    u32 e = f;
    if (q != f) {
        if (qbest > good) { e |= TE_QSIDE; TQ[p] = q; }
        else e |= TE_QNONE;
    }
    T[p] = e;
    if (all_quarters) TQ[p] = q;
}

}  // namespace mts
