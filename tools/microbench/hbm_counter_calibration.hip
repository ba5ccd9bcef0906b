#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
// What do rocprofv3's FETCH_SIZE / WRITE_SIZE report for a KNOWN number of bytes, per access pattern?  (MI355X_MICROARCH.md: on
// gfx950 FETCH_SIZE is half the bytes of a wide coalesced streaming read; other widths and WRITE_SIZE are uncalibrated.)
// Every kernel touches `bytes` of a 4 GiB buffer exactly once (beyond L2 and the 256 MiB Infinity Cache), so the bytes that
// must cross the memory interface are known:
//   k_read16_stream    16 B per lane, consecutive lanes consecutive addresses
//   k_read4_stream     4 B per lane, consecutive
//   k_read16_gather    16 B per lane at scattered 16-B-aligned addresses, every 128-B line touched by exactly one lane
//                      (k_match5's window loads look like this on a miss): 128 B must be fetched per 16 B used if lines are the unit,
//                      32 B if sectors are
//   k_write16_stream   16 B per lane, consecutive;   k_write4_stream   4 B per lane, consecutive
//   k_write4_scatter   4 B per lane, one lane per 128-B line (the worst case of the table store)
//   k_read16_stride112 (round 5) the speculative parse walk's pattern before its windows were line aligned: eight lanes read
//                      128 consecutive bytes of a 4 KiB row at offset 112 w, w = 0 .. 36 (37 windows per row, nearly every one
//                      across two lines), window w of ALL rows before window w + 1 of any: nothing of a line is still cached when
//                      its other part is asked for.  4736 bytes are asked for per 4096-byte row, 37 x 2 - 5 = 69 lines touched.
//   k_read16_sum       the pure-read ceiling: 16 B per lane, consecutive, 8 loads in flight per lane, every byte summed
// The program also times every kernel with HIP events (GB/s of the known bytes): the read ceiling, next to the 6.29 TB/s copy rate
// the guide quotes, is what says whether "7 TB/s" can be a real figure for a kernel that only reads.
// Run under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes): tools/mb_calib.sh prints counter / known bytes.
typedef unsigned long long u64;
typedef unsigned u32;
__global__ void k_read16_stream(const uint4 *p, u64 n16, u32 *sink)
{
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read4_stream(const u32 *p, u64 n4, u32 *sink)
{
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (u64)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read16_gather(const uint4 *p, u64 n_lines, u32 *sink)      // line j -> one 16-B piece at a varying offset inside it
{
    u32 acc = 0;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_lines; j += (u64)gridDim.x * blockDim.x) {
        const u64 line = (j * 0x9E3779B97F4A7C15ull) % n_lines;             // scattered: neighbouring lanes far apart
        const uint4 v = p[line * 8 + (j & 7)];
        acc ^= v.x ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_write16_stream(uint4 *p, u64 n16)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) p[i] = make_uint4((u32)i, 1, 2, 3);
}
__global__ void k_write4_stream(u32 *p, u64 n4)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (u64)gridDim.x * blockDim.x) p[i] = (u32)i;
}
__global__ void k_write4_scatter(u32 *p, u64 n_lines)
{
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_lines; j += (u64)gridDim.x * blockDim.x) {
        const u64 line = (j * 0x9E3779B97F4A7C15ull) % n_lines;
        p[line * 32 + (j & 31)] = (u32)j;
    }
}
__global__ void k_read16_stride112(const uint4 *p, u64 n_rows, u32 *sink)
{
    u32 acc = 0;
    const u64 groups = (u64)gridDim.x * blockDim.x / 8, g0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / 8;
    const u32 piece = threadIdx.x & 7;
    for (u32 w = 0; w < 37; w++)
        for (u64 r = g0; r < n_rows; r += groups) {
            const uint4 v = *(const uint4 *)((const char *)p + r * 4096 + (u64)w * 112 + piece * 16);      // (the last windows read a few bytes into the next row)
            acc ^= v.x ^ v.w;
        }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read16_sum(const uint4 *p, u64 n16, u32 *sink)
{
    u32 acc = 0;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    for (; i < n16; i += stride) { const uint4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_copy16(const uint4 *src, uint4 *dst, u64 n16)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main()
{
    const u64 bytes = 4ull << 30;
    void *d; u32 *sink;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    const u64 n_lines = bytes / 128;            // (odd multiplier, power-of-two modulus: a permutation of the lines)
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](const char *name, double known_bytes, auto &&launch) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("time %-20s %8.3f ms  %7.1f GB/s of %.3f GB\n", name, ms, known_bytes / ms / 1e6, known_bytes / 1e9);
    };
    const u64 n_rows = bytes / 4096 - 1;
    timed("k_read16_stream", (double)bytes, [&] { k_read16_stream<<<8192, 256>>>((const uint4 *)d, bytes / 16, sink); });
    timed("k_read4_stream", (double)bytes, [&] { k_read4_stream<<<8192, 256>>>((const u32 *)d, bytes / 4, sink); });
    timed("k_read16_gather", (double)n_lines * 128, [&] { k_read16_gather<<<8192, 256>>>((const uint4 *)d, n_lines, sink); });
    timed("k_write16_stream", (double)bytes, [&] { k_write16_stream<<<8192, 256>>>((uint4 *)d, bytes / 16); });
    timed("k_write4_stream", (double)bytes, [&] { k_write4_stream<<<8192, 256>>>((u32 *)d, bytes / 4); });
    timed("k_write4_scatter", (double)n_lines * 128, [&] { k_write4_scatter<<<8192, 256>>>((u32 *)d, n_lines); });
    timed("k_read16_stride112", (double)n_rows * 69 * 128, [&] { k_read16_stride112<<<8192, 256>>>((const uint4 *)d, n_rows, sink); });
    timed("k_read16_sum", (double)bytes, [&] { k_read16_sum<<<4096, 256>>>((const uint4 *)d, bytes / 16, sink); });
    timed("k_copy16 (r + w)", (double)bytes, [&] { k_copy16<<<8192, 256>>>((const uint4 *)d, (uint4 *)((char *)d + bytes / 2), bytes / 32); });
    hipDeviceSynchronize();
    printf("known bytes: streams %llu; gather %llu used (%llu by 128-B lines, %llu by 32-B sectors); scatter %llu stored (%llu by lines, %llu by sectors)\n",
           bytes, n_lines * 16, n_lines * 128, n_lines * 32, n_lines * 4, n_lines * 128, n_lines * 32);
    return 0;
}
