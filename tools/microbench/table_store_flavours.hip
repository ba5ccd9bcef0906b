#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
// How do k_match5's table stores leave the chip?  Each "tile" is a region of TILE entries that the 64 workgroups of ONE XCD
// (block b runs on XCD b % 8) fill completely within a few hundred microseconds, every lane storing ONE entry at a position
// that is scattered over the whole region (what sorting by hash does to positions).  Variants of the store:
//   0 plain 8-byte store            1 non-temporal 8-byte store       2 64-bit atomic exchange (relaxed, agent scope)
//   3 64-bit atomic OR              4 plain 4-byte store              5 two adjacent entries per lane: 16-byte store
//   6 four adjacent entries per lane as two 16-byte stores (32 B)     7 plain 8-byte store after the workgroup has READ the region
// Run under rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes) and --kernel-trace --stats.
typedef unsigned long long u64;
typedef unsigned u32;
constexpr u32 TILE = 229376, SLICES = 64;
template <int MODE>
__global__ __launch_bounds__(512) void k_store(u64 *__restrict__ tab, u32 n_tiles, u32 mult)
{
    const u32 xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const u32 tile = (jb / SLICES) * 8 + xcd, slice = jb % SLICES;
    if (tile >= n_tiles) return;
    u64 *T = tab + (size_t)tile * TILE;
    constexpr u32 PER = MODE == 5 ? 2 : MODE == 6 ? 4 : 1;          // adjacent entries per lane
    const u32 units = TILE / PER, per_slice = units / SLICES;
    if (MODE == 7) {
        u64 acc = 0;
        for (u32 i = threadIdx.x; i < TILE / 8; i += 512) acc += T[(size_t)(slice * (TILE / 64)) / 8 * 8 + i % (TILE / 64)];   // touch this slice's share
        if (acc == 0x1234567) T[0] = acc;
    }
    for (u32 k = threadIdx.x; k < per_slice; k += 512) {
        const u32 idx = slice * per_slice + k;
        const u32 pos = (u32)(((u64)idx * mult) % units) * PER;       // a permutation of the units (mult odd, coprime with units)
        const u64 v = ((u64)idx << 32) | pos | 1;
        if (MODE == 0 || MODE == 7) T[pos] = v;
        if (MODE == 1) __builtin_nontemporal_store(v, &T[pos]);
        if (MODE == 2) __hip_atomic_exchange(&T[pos], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) __hip_atomic_fetch_or(&T[pos], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 4) ((u32 *)T)[pos] = (u32)v;
        if (MODE == 5) *(ulonglong2 *)&T[pos] = make_ulonglong2(v, v + 1);
        if (MODE == 6) { *(ulonglong2 *)&T[pos] = make_ulonglong2(v, v + 1); *(ulonglong2 *)&T[pos + 2] = make_ulonglong2(v + 2, v + 3); }
    }
}
template <int MODE> void run(const char *name, u64 *d, u32 n_tiles)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const u32 mult = 48271 * 2 + 1;           // odd; TILE = 2^15 * 7: coprime needs mult not divisible by 7
    const u32 grid = (n_tiles + 7) / 8 * 8 * SLICES;
    k_store<MODE><<<grid, 512>>>(d, n_tiles, mult);
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_store<MODE><<<grid, 512>>>(d, n_tiles, mult);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n_tiles * TILE * (MODE == 4 ? 4 : 8);
    printf("%-28s %8.3f ms per launch  payload %.2f GB  -> %.0f GB/s\n", name, ms / 3, bytes / 1e9, bytes / (ms / 3 * 1e-3) / 1e9);
}
int main(int argc, char **argv)
{
    const u32 n_tiles = argc > 1 ? atoi(argv[1]) : 1024;      // 1024 tiles * 229376 * 8 B = 1.88 GB (beyond the 256 MB Infinity Cache)
    u64 *d; hipMalloc(&d, (size_t)n_tiles * TILE * 8 + 64); hipMemset(d, 0, (size_t)n_tiles * TILE * 8);
    if ((48271 * 2 + 1) % 7 == 0) { printf("bad multiplier\n"); return 1; }
    run<0>("plain 8 B", d, n_tiles);
    run<1>("nontemporal 8 B", d, n_tiles);
    run<2>("atomic exchange 8 B", d, n_tiles);
    run<3>("atomic or 8 B", d, n_tiles);
    run<4>("plain 4 B", d, n_tiles);
    run<5>("16 B (2 adjacent entries)", d, n_tiles);
    run<6>("32 B (4 adjacent entries)", d, n_tiles);
    run<7>("plain 8 B, region read first", d, n_tiles);
    hipFree(d);
    return 0;
}
