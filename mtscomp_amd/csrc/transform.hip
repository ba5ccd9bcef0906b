// K1 / K2: the np.diff + tobytes(order) transform and its cumsum inverse, adler32, synthetic data.
//   K1 replaces diff_along_axis + ndarray.tobytes       (/root/reference/mtscomp.py:143-159, :381-394)
//   K2 replaces reshape(order) + cumsum_along_axis + ascontiguousarray       (mtscomp.py:622-635)
// Integer items of 1/2/4/8 bytes; arithmetic wraps in the item width like numpy's.
#include <stdlib.h>

#include "common.h"

namespace mts {


// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 wave_sum_u64(u64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        u32 lo = __shfl_down((u32)v, off, 64), hi = __shfl_down((u32)(v >> 32), off, 64);
        v += ((u64)hi << 32) | lo;
    }
    return v;
}

// block-wide sum of two u64 values (256 threads); result valid in thread 0
__device__ __forceinline__ void block_sum2(u64 &a, u64 &b, u64 *sh /* 8 entries */)
{
    a = wave_sum_u64(a);
    b = wave_sum_u64(b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (lane == 0) { sh[wave] = a; sh[4 + wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0; b = 0;
        for (int w = 0; w < nw; w++) { a += sh[w]; b += sh[4 + w]; }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: delta + transpose (+ adler32 partial sums of the produced stream)
//   grid (time tiles of 64, channel tiles of 64, chunks), 256 threads.
//   adler: A = 1 + sum b_i, B = n + sum (n - i) b_i  (mod 65521); partial sums are atomically
//   accumulated per chunk (already reduced mod 65521 per workgroup, so the u64 cannot overflow).
// ------------------------------------------------------------------------------------------------
// bit pattern of an item (the adler32 sums run over the bytes of the stream)
__device__ __forceinline__ u64 item_bits(u8 v) { return v; }
__device__ __forceinline__ u64 item_bits(u16 v) { return v; }
__device__ __forceinline__ u64 item_bits(u32 v) { return v; }
__device__ __forceinline__ u64 item_bits(u64 v) { return v; }
__device__ __forceinline__ u64 item_bits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ u64 item_bits(double v) { return (u64)__double_as_longlong(v); }

// T = unsigned integer of the item size (wrap-around arithmetic, like numpy's), or float / double (MTS_FLAG_FLOAT: every
// difference rounded in T, the time difference first, like np.diff applied twice)
template <typename T>
__global__ __launch_bounds__(256) void k_delta_transpose(const u8 *__restrict__ raw, u8 *__restrict__ stream,
                                                         const ChunkDesc *__restrict__ chunks, int nc,
                                                         int flags, u64 *adler_acc)
{
    const ChunkDesc ch = chunks[blockIdx.z];
    const long nt = ch.n_rows;
    const long t0 = (long)blockIdx.x * 64;
    if (t0 >= nt) return;
    const int c0 = blockIdx.y * 64;
    const T *x = (const T *)(raw + ch.raw_off);
    T *out = (T *)(stream + ch.stream_off);
    __shared__ T tile[65][66];          // row 0 = t0 - 1, col 0 = c0 - 1
    __shared__ u64 red[8];
    for (int idx = threadIdx.x; idx < 65 * 65; idx += 256) {
        const int r = idx / 65, k = idx - r * 65;
        const long t = t0 - 1 + r;
        const int c = c0 - 1 + k;
        T v = 0;
        if (t >= 0 && t < nt && c >= 0 && c < nc) v = x[t * nc + c];
        tile[r][k] = v;
    }
    __syncthreads();
    const bool td = flags & MTS_FLAG_TIME_DIFF, sd = flags & MTS_FLAG_SPATIAL_DIFF, of = flags & MTS_FLAG_ORDER_F;
    const u64 nbytes = (u64)nt * nc * sizeof(T);
    u64 sa = 0, sb = 0;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        int tt, cc;
        if (of) { cc = idx >> 6; tt = idx & 63; } else { tt = idx >> 6; cc = idx & 63; }
        const long t = t0 + tt;
        const int c = c0 + cc;
        if (t >= nt || c >= nc) continue;
        T d = tile[tt + 1][cc + 1];
        if (td && t > 0) d -= tile[tt][cc + 1];
        if (sd && c > 0) {
            T dl = tile[tt + 1][cc];
            if (td && t > 0) dl -= tile[tt][cc];
            d -= dl;
        }
        const u64 I = of ? (u64)c * nt + t : (u64)t * nc + c;
        out[I] = d;
        const u64 v = item_bits(d);
#pragma unroll
        for (int k = 0; k < (int)sizeof(T); k++) {
            const u64 b = (v >> (8 * k)) & 0xff;
            sa += b;
            sb += (nbytes - (I * sizeof(T) + k)) * b;
        }
    }
    block_sum2(sa, sb, red);
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long *)&adler_acc[2 * blockIdx.z], (unsigned long long)(sa % 65521u));
        atomicAdd((unsigned long long *)&adler_acc[2 * blockIdx.z + 1], (unsigned long long)(sb % 65521u));
    }
}

// ------------------------------------------------------------------------------------------------
// Row-tile variants for the layout the reference actually uses (time difference, channel-major stream,
// no spatial difference; items of 1, 2 or 4 bytes).  A workgroup owns TT consecutive rows of ALL channels:
// on the C-order side that is one contiguous piece of memory (read or written as whole dwords, fully
// coalesced), on the stream side TT consecutive items of every channel.  The transpose goes through LDS
// with a row pitch of an odd number of dwords.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 wave_incl_scan_dpp32(u32 x)
{
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);     // row_bcast:15
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);     // row_bcast:31
    return x;
}
// LDS row pitch in items: at least nc, an odd number of dwords
__host__ __device__ inline int rows_pitch(int nc, int itemsize)
{
    const int per = 4 / itemsize;                       // items per dword (itemsize 1, 2, 4)
    int dw = (nc + per - 1) / per;
    if (!(dw & 1)) dw++;
    return dw * per;
}
// largest tile height whose LDS image ((TT + 1) rows) stays within 64 KiB
static int rows_tile(int nc, int itemsize)
{
    const long pitch_b = (long)rows_pitch(nc, itemsize) * itemsize;
    for (int tt = 64; tt >= 16; tt >>= 1)
        if ((tt + 1) * pitch_b <= 64 * 1024) return tt;
    return 0;
}

// Row tiles that follow each other in time touch the same cache lines on the stream side (a channel's 64 items of a tile are
// 128 bytes at a 32-byte phase): consecutive tiles of a chunk are given to ONE XCD (workgroup b runs on XCD b % 8), so that its
// L2 merges the halves; with tile = blockIdx.x neighbours sat on different XCDs and every line crossed the fabric twice.
// Linear workgroup b -> (chunk, tile); false when the tile does not exist.
__device__ __forceinline__ bool xcd_row_tile(int ntile, int &tile, int &chunk)
{
    const u32 per = ((u32)ntile + 7) / 8;                      // tiles of a chunk per XCD
    const u32 xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    chunk = (int)(j / per);
    tile = (int)(xcd * per + j % per);
    return tile < ntile;
}
static unsigned xcd_row_grid(int ntile, int n_chunks) { return 8u * (((unsigned)ntile + 7) / 8) * (unsigned)n_chunks; }

template <typename T>
__global__ __launch_bounds__(256) void k_delta_rows(const u8 *__restrict__ raw, u8 *__restrict__ stream,
                                                    const ChunkDesc *__restrict__ chunks, int nc, int tt_rows, int pitch,
                                                    u32 nc_magic, u64 *adler_acc, int ntile_max)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem_rows[];
    T *tile = (T *)smem_rows;                                   // [tt_rows + 1][pitch]; row 0 = t0 - 1
    __shared__ u64 red[8];
    int tile_x, chunk_y;
    if (!xcd_row_tile(ntile_max, tile_x, chunk_y)) return;
    const ChunkDesc ch = chunks[chunk_y];
    const long nt = ch.n_rows;
    const long t0 = (long)tile_x * tt_rows;
    if (t0 >= nt) return;
    constexpr int EPD = 4 / (int)sizeof(T);
    const T *x = (const T *)(raw + ch.raw_off);
    const int nrow = (int)min((long)tt_rows, nt - t0);
    // tile row 0 = row t0 - 1 (zeros before the chunk's first row), item by item
    for (int c = threadIdx.x; c < nc; c += 256) tile[c] = t0 > 0 ? x[(t0 - 1) * nc + c] : (T)0;
    // rows t0 .. : one contiguous piece of memory, copied a dword (EPD items) per lane
    const T *xs = x + t0 * nc;
    const long e_n = (long)nrow * nc;
    if ((((u64)xs) & 15) == 0) {
        // 16 bytes (VW items) per lane and step, LOADS_IN_FLIGHT steps at a time: what bounds this kernel is the memory
        // latency, i.e. how many bytes a CU has on their way (one load per lane and loop turn kept 12 KB per CU in flight and
        // the kernel at 2.8 TB/s; the chip needs ~40 KB per CU to stream at its rate)
        constexpr int VW = 16 / (int)sizeof(T);
        constexpr int LOADS_IN_FLIGHT = 7;
        const uint4 *xq = (const uint4 *)xs;
        const long n_full = e_n / VW;                              // whole 16-byte pieces
        for (long base = 0; base < n_full; base += 256 * LOADS_IN_FLIGHT) {
            uint4 q[LOADS_IN_FLIGHT];
#pragma unroll
            for (int u = 0; u < LOADS_IN_FLIGHT; u++) {
                const long i = base + u * 256 + threadIdx.x;
                q[u] = i < n_full ? xq[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < LOADS_IN_FLIGHT; u++) {
                const long i = base + u * 256 + threadIdx.x;
                if (i < n_full) {
                    const u32 w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                    u32 r = __umulhi((u32)i * VW, nc_magic), c = (u32)i * VW - r * nc;
#pragma unroll
                    for (int j = 0; j < VW; j++) {
                        tile[(r + 1) * pitch + c] = (T)(w[j / EPD] >> (8 * sizeof(T) * (j % EPD)));
                        if (++c == (u32)nc) { c = 0; r++; }
                    }
                }
            }
        }
        for (long e = n_full * VW + threadIdx.x; e < e_n; e += 256) {     // ragged end: never read past the rows
            const u32 r = __umulhi((u32)e, nc_magic), c = (u32)e - r * nc;
            tile[(r + 1) * pitch + c] = xs[e];
        }
    } else {
        for (long i = threadIdx.x; i < e_n; i += 256) {
            const u32 r = __umulhi((u32)i, nc_magic), c = (u32)i - r * nc;
            tile[(r + 1) * pitch + c] = xs[i];
        }
    }
    __syncthreads();
    T *out = (T *)(stream + ch.stream_off);
    const u64 nbytes = (u64)nt * nc * sizeof(T);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u64 sa = 0, sb = 0;
    auto adler_item = [&](T d, u64 I) {
        u32 bs = 0, bw = 0;                                      // sum of the item's bytes, sum of k * byte_k
#pragma unroll
        for (int k = 0; k < (int)sizeof(T); k++) { const u32 b = ((u32)d >> (8 * k)) & 0xff; bs += b; bw += k * b; }
        sa += bs;
        sb += (nbytes - I * sizeof(T)) * bs - bw;
    };
    constexpr int IPL = 8 / (int)sizeof(T);                      // items per lane on the stream side (8 bytes)
    if (tt_rows % IPL == 0 && ((u64)nt * sizeof(T)) % 8 == 0 && (((u64)out) & 7) == 0 && ((u64)t0 * sizeof(T)) % 8 == 0) {
        // a lane owns IPL consecutive items of one channel; 64 / (tt_rows / IPL) channels per wave and step
        const int lpc = tt_rows / IPL, cpw = 64 / lpc;           // lanes per channel (8 or 16 or ...), channels per wave step
        const int q = lane % lpc, cc = lane / lpc;
        for (int c0 = wave * cpw; c0 < nc; c0 += 4 * cpw) {
            const int c = c0 + cc;
            const int tt = q * IPL;
            if (cc < cpw && c < nc && t0 + tt < nt) {
                u32 w[2] = {0, 0};
                T prev = tile[tt * pitch + c];
#pragma unroll
                for (int j = 0; j < IPL; j++) {
                    const T cur = tile[(tt + j + 1) * pitch + c];
                    const T d = (T)(cur - prev);
                    prev = cur;
                    if (t0 + tt + j < nt) w[j / EPD] |= (u32)d << (8 * sizeof(T) * (j % EPD));      // (items past the chunk stay zero: no weight in the sums)
                }
                const u64 I = (u64)c * nt + t0 + tt;
                {   // adler32 of the lane's 8 bytes: byte k of them is byte I * sizeof(T) + k of the stream
                    // (sum of the 8 bytes and sum of k * byte_k: two byte dot products each instead of eight extracts, adds and multiplies)
                    const u32 bs = __builtin_amdgcn_udot4(w[1], 0x01010101u, __builtin_amdgcn_udot4(w[0], 0x01010101u, 0u, false), false);
                    const u32 bw = __builtin_amdgcn_udot4(w[1], 0x07060504u, __builtin_amdgcn_udot4(w[0], 0x03020100u, 0u, false), false);
                    sa += bs;
                    sb += (nbytes - I * sizeof(T)) * bs - bw;
                }
                if (t0 + tt + IPL <= nt) *(uint2 *)&out[I] = make_uint2(w[0], w[1]);
                else for (int j = 0; t0 + tt + j < nt; j++) out[I + j] = (T)(w[j / EPD] >> (8 * sizeof(T) * (j % EPD)));
            }
        }
    } else {
        for (int tb = 0; tb < tt_rows; tb += 64) {
            const int tt = tb + lane;
            const long t = t0 + tt;
            const bool ok = tt < tt_rows && t < nt;
            for (int c = wave; c < nc; c += 4) {
                if (ok) {
                    const T d = (T)(tile[(tt + 1) * pitch + c] - tile[tt * pitch + c]);
                    const u64 I = (u64)c * nt + t;
                    out[I] = d;
                    adler_item(d, I);
                }
            }
        }
    }
    block_sum2(sa, sb, red);
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long *)&adler_acc[2 * chunk_y], (unsigned long long)(sa % 65521u));
        atomicAdd((unsigned long long *)&adler_acc[2 * chunk_y + 1], (unsigned long long)(sb % 65521u));
    }
}

template <typename T>
static void run_delta_rows(hipStream_t st, const u8 *raw, u8 *stream, const ChunkDesc *d_chunks, int n_chunks, u32 max_rows,
                           int nc, u64 *d_adler_acc)
{
    const int tt = rows_tile(nc, (int)sizeof(T)), pitch = rows_pitch(nc, (int)sizeof(T));
    const size_t lds = (size_t)(tt + 1) * pitch * sizeof(T);
    (void)ensure_dynamic_lds((const void *)k_delta_rows<T>, 64 * 1024);
    const int ntile = (int)((max_rows + tt - 1) / tt);
    hipLaunchKernelGGL(k_delta_rows<T>, dim3(xcd_row_grid(ntile, n_chunks)), dim3(256), lds, st, raw, stream, d_chunks, nc, tt, pitch, 0xffffffffu / (u32)nc + 1,
                       d_adler_acc, ntile);
}

int launch_delta_transpose(hipStream_t st, const void *d_raw, void *d_stream, const ChunkDesc *d_chunks,
                           int n_chunks, u32 max_rows, int n_channels, int itemsize, int flags,
                           u64 *d_adler_acc)
{
    if (n_chunks == 0 || max_rows == 0) return MTS_OK;
    MTS_HIP(hipMemsetAsync(d_adler_acc, 0, sizeof(u64) * 2 * n_chunks, st));
    dim3 grid((max_rows + 63) / 64, (n_channels + 63) / 64, n_chunks), block(256);
    const u8 *raw = (const u8 *)d_raw;
    u8 *stream = (u8 *)d_stream;
    if (flags == (MTS_FLAG_TIME_DIFF | MTS_FLAG_ORDER_F) && itemsize <= 4 && n_channels >= 2 && rows_tile(n_channels, itemsize) && !getenv("MTS_K12_GENERIC")) {
        switch (itemsize) {
        case 1: run_delta_rows<u8>(st, raw, stream, d_chunks, n_chunks, max_rows, n_channels, d_adler_acc); break;
        case 2: run_delta_rows<u16>(st, raw, stream, d_chunks, n_chunks, max_rows, n_channels, d_adler_acc); break;
        default: run_delta_rows<u32>(st, raw, stream, d_chunks, n_chunks, max_rows, n_channels, d_adler_acc); break;
        }
        MTS_HIP(hipGetLastError());
        return MTS_OK;
    }
    if (flags & MTS_FLAG_FLOAT) {
        if (itemsize == 4) hipLaunchKernelGGL(k_delta_transpose<float>, grid, block, 0, st, raw, stream, d_chunks, n_channels, flags, d_adler_acc);
        else if (itemsize == 8) hipLaunchKernelGGL(k_delta_transpose<double>, grid, block, 0, st, raw, stream, d_chunks, n_channels, flags, d_adler_acc);
        else return MTS_E_ARG;
        MTS_HIP(hipGetLastError());
        return MTS_OK;
    }
    switch (itemsize) {
    case 1: hipLaunchKernelGGL(k_delta_transpose<u8>, grid, block, 0, st, raw, stream, d_chunks, n_channels, flags, d_adler_acc); break;
    case 2: hipLaunchKernelGGL(k_delta_transpose<u16>, grid, block, 0, st, raw, stream, d_chunks, n_channels, flags, d_adler_acc); break;
    case 4: hipLaunchKernelGGL(k_delta_transpose<u32>, grid, block, 0, st, raw, stream, d_chunks, n_channels, flags, d_adler_acc); break;
    case 8: hipLaunchKernelGGL(k_delta_transpose<u64>, grid, block, 0, st, raw, stream, d_chunks, n_channels, flags, d_adler_acc); break;
    default: return MTS_E_ARG;
    }
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

// ------------------------------------------------------------------------------------------------
// adler32 partial sums of plain streams (inflate side: verify the trailer)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adler_stream(const u8 *__restrict__ stream, const u64 *__restrict__ stream_off,
                                                      const u32 *__restrict__ n_arr, u64 *adler_acc, const u32 *__restrict__ skip, u32 skip_stride)
{
    const int chunk = blockIdx.y;
    if (skip && skip[(size_t)chunk * skip_stride] >= 2) return;         // (summed where its bytes were made: k_inf_translate)
    const u64 n = n_arr[chunk];
    const u64 base = (u64)blockIdx.x * (256 * 64);
    if (base >= n) return;
    const u8 *s = stream + stream_off[chunk];
    __shared__ u64 red[8];
    u64 sa = 0, sb = 0;
    // 16 bytes per lane per step, 4 steps
    for (int k = 0; k < 4; k++) {
        const u64 i0 = base + ((u64)k * 256 + threadIdx.x) * 16;
        if (i0 >= n) break;
        if (i0 + 16 <= n) {
            const uint4 v = *(const uint4 *)(s + i0);
            const u32 w[4] = {v.x, v.y, v.z, v.w};
            u32 bs = 0, bw = 0;                                  // sum of the 16 bytes, sum of j * byte_j (byte dot products)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                bs = __builtin_amdgcn_udot4(w[q], 0x01010101u, bs, false);
                bw = __builtin_amdgcn_udot4(w[q], 0x03020100u + 0x04040404u * (u32)q, bw, false);
            }
            sa += bs; sb += (n - i0) * (u64)bs - bw;
        } else {
            for (u64 i = i0; i < n; i++) { const u64 b = s[i]; sa += b; sb += (n - i) * b; }
        }
    }
    block_sum2(sa, sb, red);
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long *)&adler_acc[2 * chunk], (unsigned long long)(sa % 65521u));
        atomicAdd((unsigned long long *)&adler_acc[2 * chunk + 1], (unsigned long long)(sb % 65521u));
    }
}

int launch_adler_stream(hipStream_t st, const u8 *d_stream, const u64 *d_stream_off, const u32 *d_n,
                        int n_chunks, u32 max_n, u64 *d_adler_acc, const u32 *d_skip, u32 skip_stride)
{
    if (n_chunks == 0) return MTS_OK;
    if (!d_skip) MTS_HIP(hipMemsetAsync(d_adler_acc, 0, sizeof(u64) * 2 * n_chunks, st));      // (with d_skip the caller zeroed the sums before the chunks that are skipped here added theirs)
    if (max_n == 0) return MTS_OK;
    dim3 grid((max_n + 256 * 64 - 1) / (256 * 64), n_chunks), block(256);
    hipLaunchKernelGGL(k_adler_stream, grid, block, 0, st, d_stream, d_stream_off, d_n, d_adler_acc, d_skip, skip_stride);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

// ------------------------------------------------------------------------------------------------
// K2: (spatial cumsum) + time cumsum + transpose back to C order
//   pass S (only with do_spatial_diff): in-place prefix sum over channels, one thread per row
//   pass A: per (segment of SEGR rows, channel) sums
//   pass B: scan inside the segment with the carry of the previous segments, write C-order rows
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_spatial_cumsum(u8 *stream, const u64 *__restrict__ stream_off,
                                                        const u32 *__restrict__ rows, const int *__restrict__ status,
                                                        int nc, int order_f)
{
    const int chunk = blockIdx.y;
    if (status && status[chunk] != 0) return;
    const long nt = rows[chunk];
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nt) return;
    T *d = (T *)(stream + stream_off[chunk]);
    T acc = 0;
    for (int c = 0; c < nc; c++) {
        const u64 I = order_f ? (u64)c * nt + t : (u64)t * nc + c;
        acc = c ? acc + d[I] : d[I];             // (not 0 + d: -0.0 stays -0.0 for float items)
        d[I] = acc;
    }
}

template <typename T, int SEGR>
__global__ __launch_bounds__(256) void k_seg_sums(const u8 *__restrict__ stream, const u64 *__restrict__ stream_off,
                                                  const u32 *__restrict__ rows, const int *__restrict__ status,
                                                  int nc, int order_f, int nseg_max, u64 *__restrict__ segsums)
{
    const int chunk = blockIdx.z;
    if (status && status[chunk] != 0) return;
    const long nt = rows[chunk];
    const long t0 = (long)blockIdx.x * SEGR;
    if (t0 >= nt) return;
    const int c0 = blockIdx.y * 64;
    const T *d = (const T *)(stream + stream_off[chunk]);
    __shared__ u64 part[4][64];
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (order_f) {
        // wave q sums channels q*16 .. q*16+15; lanes run along time
        for (int k = 0; k < 16; k++) {
            const int cc = q * 16 + k, c = c0 + cc;
            u64 s = 0;
            if (c < nc)
                for (int tt = lane; tt < SEGR; tt += 64)
                    if (t0 + tt < nt) s += (u64)d[(u64)c * nt + t0 + tt];
            s = wave_sum_u64(s);
            if (lane == 0) part[0][cc] = s;
        }
        __syncthreads();
        if (threadIdx.x < 64 && c0 + threadIdx.x < nc)
            segsums[((u64)chunk * nseg_max + blockIdx.x) * nc + c0 + threadIdx.x] = part[0][threadIdx.x];
    } else {
        const int c = c0 + lane;
        u64 s = 0;
        if (c < nc)
            for (int tt = q; tt < SEGR; tt += 4)
                if (t0 + tt < nt) s += (u64)d[(u64)(t0 + tt) * nc + c];
        part[q][lane] = s;
        __syncthreads();
        if (threadIdx.x < 64 && c < nc)
            segsums[((u64)chunk * nseg_max + blockIdx.x) * nc + c] =
                part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    }
}

template <typename T, int SEGR>
__global__ __launch_bounds__(256) void k_cumsum_transpose(const u8 *__restrict__ stream, u8 *__restrict__ outb,
                                                          const u64 *__restrict__ stream_off,
                                                          const u64 *__restrict__ out_off, const u32 *__restrict__ rows,
                                                          const int *__restrict__ status, int nc, int flags,
                                                          int nseg_max, const u64 *__restrict__ segsums)
{
    const int chunk = blockIdx.z;
    if (status && status[chunk] != 0) return;
    const long nt = rows[chunk];
    const long t0 = (long)blockIdx.x * SEGR;
    if (t0 >= nt) return;
    const int c0 = blockIdx.y * 64;
    const bool td = flags & MTS_FLAG_TIME_DIFF, of = flags & MTS_FLAG_ORDER_F;
    const T *d = (const T *)(stream + stream_off[chunk]);
    T *out = (T *)(outb + out_off[chunk]);
    constexpr int PITCH = SEGR + (sizeof(T) >= 4 ? 1 : (int)(4 / sizeof(T)));   // odd number of dwords for 2-byte items
    __shared__ T tile[64][PITCH];
    __shared__ T carry[64];
    __shared__ T qsum[4][64];
    if (threadIdx.x < 64) {
        T cy = 0;
        const int c = c0 + threadIdx.x;
        if (td && c < nc)
            for (unsigned s = 0; s < blockIdx.x; s++) cy += (T)segsums[((u64)chunk * nseg_max + s) * nc + c];
        carry[threadIdx.x] = cy;
    }
    for (int idx = threadIdx.x; idx < 64 * SEGR; idx += 256) {
        int cc, tt;
        if (of) { cc = idx / SEGR; tt = idx - cc * SEGR; } else { tt = idx >> 6; cc = idx & 63; }
        const long t = t0 + tt;
        const int c = c0 + cc;
        T v = 0;
        if (t < nt && c < nc) v = d[of ? (u64)c * nt + t : (u64)t * nc + c];
        tile[cc][tt] = v;
    }
    __syncthreads();
    if (td) {
        // thread (cc, q) scans a quarter of the segment sequentially, then the quarters are chained
        const int cc = threadIdx.x & 63, q = threadIdx.x >> 6;
        constexpr int QR = SEGR / 4;
        T acc = 0;
        for (int k = 0; k < QR; k++) { acc += tile[cc][q * QR + k]; tile[cc][q * QR + k] = acc; }
        qsum[q][cc] = acc;
        __syncthreads();
        T add = carry[cc];
        for (int j = 0; j < q; j++) add += qsum[j][cc];
        for (int k = 0; k < QR; k++) tile[cc][q * QR + k] += add;
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < 64 * SEGR; idx += 256) {
        const int tt = idx >> 6, cc = idx & 63;
        const long t = t0 + tt;
        const int c = c0 + cc;
        if (t < nt && c < nc) out[(u64)t * nc + c] = tile[cc][tt];
    }
}

// Float items: np.cumsum adds one after the other (r[t] = r[t-1] + d[t], rounded in T each time), and so does this -- one
// lane per channel, 30000 dependent adds -- because any tree-shaped scan rounds differently.  (Loads are issued eight
// ahead; the C-order stores of a wave are consecutive channels.)
template <typename T>
__global__ __launch_bounds__(256) void k_cumsum_seq(const u8 *__restrict__ stream, u8 *__restrict__ outb,
                                                    const u64 *__restrict__ stream_off, const u64 *__restrict__ out_off,
                                                    const u32 *__restrict__ rows, const int *__restrict__ status, int nc, int flags)
{
    const int chunk = blockIdx.y;
    if (status && status[chunk] != 0) return;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    const long nt = rows[chunk];
    const T *d = (const T *)(stream + stream_off[chunk]);
    T *o = (T *)(outb + out_off[chunk]);
    const bool td = flags & MTS_FLAG_TIME_DIFF, of = flags & MTS_FLAG_ORDER_F;
    const long step = of ? 1 : nc;
    const T *p = d + (of ? (long)c * nt : (long)c);
    T acc = 0;
    long t = 0;
    for (; t + 8 <= nt; t += 8) {
        T v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = p[(t + k) * step];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            acc = (td && t + k > 0) ? acc + v[k] : v[k];
            o[(t + k) * nc + c] = acc;
        }
    }
    for (; t < nt; t++) {
        const T v = p[t * step];
        acc = (td && t > 0) ? acc + v : v;
        o[t * nc + c] = acc;
    }
}

template <typename T>
static int run_cumsum_float(hipStream_t st, const u8 *stream, u8 *out, const u64 *d_stream_off, const u64 *d_out_off,
                            const u32 *d_rows, const int *d_status, int n_chunks, u32 max_rows, int nc, int flags)
{
    const int of = (flags & MTS_FLAG_ORDER_F) ? 1 : 0;
    if (flags & MTS_FLAG_SPATIAL_DIFF) {
        dim3 g((max_rows + 255) / 256, n_chunks);
        hipLaunchKernelGGL(k_spatial_cumsum<T>, g, dim3(256), 0, st, (u8 *)stream, d_stream_off, d_rows, d_status, nc, of);
    }
    hipLaunchKernelGGL(k_cumsum_seq<T>, dim3((nc + 255) / 256, n_chunks), dim3(256), 0, st, stream, out, d_stream_off, d_out_off, d_rows,
                       d_status, nc, flags);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

template <typename T, int SEGR>
static int run_cumsum(hipStream_t st, const u8 *stream, u8 *out, const u64 *d_stream_off, const u64 *d_out_off,
                      const u32 *d_rows, const int *d_status, int n_chunks, u32 max_rows, int nc, int flags,
                      u64 *segsums)
{
    const int nseg = (max_rows + SEGR - 1) / SEGR;
    const int of = (flags & MTS_FLAG_ORDER_F) ? 1 : 0;
    if (flags & MTS_FLAG_SPATIAL_DIFF) {
        dim3 g((max_rows + 255) / 256, n_chunks);
        hipLaunchKernelGGL(k_spatial_cumsum<T>, g, dim3(256), 0, st, (u8 *)stream, d_stream_off, d_rows, d_status, nc, of);
    }
    dim3 grid(nseg, (nc + 63) / 64, n_chunks), block(256);
    if (flags & MTS_FLAG_TIME_DIFF)
        hipLaunchKernelGGL((k_seg_sums<T, SEGR>), grid, block, 0, st, stream, d_stream_off, d_rows, d_status, nc, of, nseg, segsums);
    hipLaunchKernelGGL((k_cumsum_transpose<T, SEGR>), grid, block, 0, st, stream, out, d_stream_off, d_out_off, d_rows,
                       d_status, nc, flags, nseg, segsums);
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

#ifndef MTS_K2_KU
#define MTS_K2_KU 13     // (5: 1.06 ms, 9: 1.02, 13: 1.00, 25: 1.02 for the 60-chunk workload)
#endif
// K2, row-tile variant: (1) per (tile, channel) sums, (2) exclusive scan of the sums over the tiles of a chunk,
// (3) scan inside the tile with the carry and write whole C-order rows.
template <typename T>
__global__ __launch_bounds__(256) void k_rows_sums(const u8 *__restrict__ stream, const u64 *__restrict__ stream_off,
                                                   const u32 *__restrict__ rows, const int *__restrict__ status,
                                                   int nc, int tt_rows, int ntile_max, u32 *__restrict__ sums)
{
    int tile_x, chunk;
    if (!xcd_row_tile(ntile_max, tile_x, chunk)) return;
    if (status && status[chunk] != 0) return;
    const long nt = rows[chunk];
    const long t0 = (long)tile_x * tt_rows;
    if (t0 >= nt) return;
    const T *d = (const T *)(stream + stream_off[chunk]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32 *o = sums + ((u64)chunk * ntile_max + tile_x) * nc;
    constexpr int EPD = 4 / (int)sizeof(T), IPL = 8 / (int)sizeof(T);
    if (tt_rows % IPL == 0 && ((u64)nt * sizeof(T)) % 8 == 0 && (((u64)d) & 7) == 0 && ((u64)t0 * sizeof(T)) % 8 == 0 && t0 + tt_rows <= nt) {
        // a lane reads IPL consecutive items (8 bytes) of one channel; the lanes of a channel are neighbours
        const int lpc = tt_rows / IPL, cpw = 64 / lpc;
        const int q = lane % lpc, cc = lane / lpc;
        constexpr int KU = MTS_K2_KU;                             // wave steps whose loads are in flight together
        for (int c00 = wave * cpw; c00 < nc; c00 += 4 * cpw * KU) {
            uint2 wq[KU];
#pragma unroll
            for (int k = 0; k < KU; k++) wq[k] = *(const uint2 *)&d[(u64)min(c00 + k * 4 * cpw + cc, nc - 1) * nt + t0 + q * IPL];
#pragma unroll
            for (int k = 0; k < KU; k++) {
                const int c0 = c00 + k * 4 * cpw;
                const int c = min(c0 + cc, nc - 1);
                const u32 ww[2] = {wq[k].x, wq[k].y};
                u32 v = 0;
#pragma unroll
                for (int j = 0; j < IPL; j++) v += (u32)(T)(ww[j / EPD] >> (8 * sizeof(T) * (j % EPD)));
                v = wave_incl_scan_dpp32(v);
                const u32 before = (u32)__shfl((int)v, (lane - q - 1) & 63, 64);   // total of the lanes before this channel's
                if (q == lpc - 1 && c0 + cc < nc) o[c] = v - (lane - q ? before : 0u);
            }
        }
    } else {
        for (int c = wave; c < nc; c += 4) {
            u32 v = 0;
            for (int tt = lane; tt < tt_rows; tt += 64)
                if (t0 + tt < nt) v += (u32)d[(u64)c * nt + t0 + tt];
            v = wave_incl_scan_dpp32(v);
            if (lane == 63) o[c] = v;
        }
    }
}
// (a column's tiles in ROWS_SCAN_PARTS consecutive parts, a wave per part and 64 columns: the parts' totals meet in LDS.  One
//  thread per column walking all ~470 tiles, eight dependent loads at a time, was 70 us of a 0.98 ms stage)
constexpr int ROWS_SCAN_PARTS = 8;
__global__ __launch_bounds__(64 * ROWS_SCAN_PARTS) void k_rows_scan(const u32 *__restrict__ rows, const int *__restrict__ status, int nc,
                                                   int tt_rows, int ntile_max, u32 *__restrict__ sums)
{
    __shared__ u32 part_sum[ROWS_SCAN_PARTS][64];
    const int chunk = blockIdx.y;
    if (status && status[chunk] != 0) return;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int ntile = (int)((rows[chunk] + tt_rows - 1) / tt_rows);
    const int per = (ntile + ROWS_SCAN_PARTS - 1) / ROWS_SCAN_PARTS, k_beg = min(part * per, ntile), k_end = min(k_beg + per, ntile);
    u32 *p = sums + (u64)chunk * ntile_max * nc + (c < nc ? c : 0);
    u32 total = 0;
    if (c < nc)
        for (int k0 = k_beg; k0 < k_end; k0 += 8) {
            u32 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = k0 + j < k_end ? p[(u64)(k0 + j) * nc] : 0;
#pragma unroll
            for (int j = 0; j < 8; j++) total += v[j];
        }
    part_sum[part][lane] = total;
    __syncthreads();
    u32 run = 0;
    for (int q = 0; q < part; q++) run += part_sum[q][lane];
    if (c >= nc) return;
    for (int k0 = k_beg; k0 < k_end; k0 += 8) {
        u32 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = k0 + j < k_end ? p[(u64)(k0 + j) * nc] : 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { if (k0 + j < k_end) p[(u64)(k0 + j) * nc] = run; run += v[j]; }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_cumsum_rows(const u8 *__restrict__ stream, u8 *__restrict__ outb,
                                                     const u64 *__restrict__ stream_off, const u64 *__restrict__ out_off,
                                                     const u32 *__restrict__ rows, const int *__restrict__ status,
                                                     int nc, int tt_rows, int pitch, u32 nc_magic, int ntile_max,
                                                     const u32 *__restrict__ sums)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem_rows[];
    T *tile = (T *)smem_rows;                                   // [tt_rows][pitch]
    int tile_x, chunk;
    if (!xcd_row_tile(ntile_max, tile_x, chunk)) return;
    if (status && status[chunk] != 0) return;
    const long nt = rows[chunk];
    const long t0 = (long)tile_x * tt_rows;
    if (t0 >= nt) return;
    constexpr int EPD = 4 / (int)sizeof(T);
    const T *d = (const T *)(stream + stream_off[chunk]);
    const u32 *carry = sums + ((u64)chunk * ntile_max + tile_x) * nc;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int IPL = 8 / (int)sizeof(T);
    if (tt_rows % IPL == 0 && ((u64)nt * sizeof(T)) % 8 == 0 && (((u64)d) & 7) == 0 && ((u64)t0 * sizeof(T)) % 8 == 0 && t0 + tt_rows <= nt) {
        const int lpc = tt_rows / IPL, cpw = 64 / lpc;
        const int q = lane % lpc, cc = lane / lpc;
        constexpr int KU = MTS_K2_KU;                             // wave steps whose loads are in flight together (see k_delta_rows)
        for (int c00 = wave * cpw; c00 < nc; c00 += 4 * cpw * KU) {
            uint2 wq[KU];
            u32 cy[KU];
#pragma unroll
            for (int k = 0; k < KU; k++) {
                const int c = min(c00 + k * 4 * cpw + cc, nc - 1);
                wq[k] = *(const uint2 *)&d[(u64)c * nt + t0 + q * IPL];
                cy[k] = carry[c];
            }
#pragma unroll
            for (int k = 0; k < KU; k++) {
                const int c0 = c00 + k * 4 * cpw;
                const int c = min(c0 + cc, nc - 1);
                const u32 ww[2] = {wq[k].x, wq[k].y};
                u32 x[IPL], tot = 0;
#pragma unroll
                for (int j = 0; j < IPL; j++) { tot += (u32)(T)(ww[j / EPD] >> (8 * sizeof(T) * (j % EPD))); x[j] = tot; }
                const u32 v = wave_incl_scan_dpp32(tot);
                const u32 prev = (u32)__shfl((int)v, (lane - q - 1) & 63, 64);      // (unconditional: a shuffle cannot read a lane that is masked off)
                const u32 before = (lane - q) ? prev : 0u;                          // total of the lanes of the channels before this one
                const u32 add = cy[k] + v - tot - before;
                if (c0 + cc < nc) {
#pragma unroll
                    for (int j = 0; j < IPL; j++) tile[(q * IPL + j) * pitch + c] = (T)(x[j] + add);
                }
            }
        }
    } else {
        for (int c = wave; c < nc; c += 4) {
            u32 run = carry[c];
            for (int tb = 0; tb < tt_rows; tb += 64) {
                const int tt = tb + lane;
                const u32 v = (tt < tt_rows && t0 + tt < nt) ? (u32)d[(u64)c * nt + t0 + tt] : 0u;
                const u32 x = wave_incl_scan_dpp32(v) + run;
                if (tt < tt_rows) tile[tt * pitch + c] = (T)x;
                run = (u32)__builtin_amdgcn_readlane((int)x, 63);
            }
        }
    }
    __syncthreads();
    const int nrow = (int)min((long)tt_rows, nt - t0);
    T *out = (T *)(outb + out_off[chunk]) + t0 * nc;
    const long e_n = (long)nrow * nc;
    if ((((u64)out) & 15) == 0 && sizeof(T) <= 4) {
        // the rows of the tile are one contiguous piece of memory: 16 bytes (VW items) per lane and store, except a ragged end
        constexpr int VW = 16 / (int)sizeof(T);
        const long n_full = e_n / VW;
        uint4 *oq = (uint4 *)out;
        for (long i = threadIdx.x; i < n_full; i += 256) {
            u32 r = __umulhi((u32)i * VW, nc_magic), c = (u32)i * VW - r * nc;
            u32 w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < VW; j++) {
                w[j / EPD] |= (u32)tile[r * pitch + c] << (8 * sizeof(T) * (j % EPD));
                if (++c == (u32)nc) { c = 0; r++; }
            }
            oq[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        for (long e = n_full * VW + threadIdx.x; e < e_n; e += 256) {
            const u32 r = __umulhi((u32)e, nc_magic), c = (u32)e - r * nc;
            out[e] = tile[r * pitch + c];
        }
    } else if ((((u64)out) & 3) == 0) {
        // ... whole dwords (EPD items)
        u32 *od = (u32 *)out;
        for (long i = threadIdx.x; i * EPD < e_n; i += 256) {
            u32 v = 0;
            bool full = true;
#pragma unroll
            for (int j = 0; j < EPD; j++) {
                const u32 e = (u32)i * EPD + j;
                if (e < e_n) {
                    const u32 r = __umulhi(e, nc_magic), c = e - r * nc;
                    v |= (u32)tile[r * pitch + c] << (8 * sizeof(T) * j);
                } else full = false;
            }
            if (full) od[i] = v;
            else for (int j = 0; j < EPD; j++) if ((long)i * EPD + j < e_n) out[i * EPD + j] = (T)(v >> (8 * sizeof(T) * j));
        }
    } else {
        for (long i = threadIdx.x; i < e_n; i += 256) {
            const u32 r = __umulhi((u32)i, nc_magic), c = (u32)i - r * nc;
            out[i] = tile[r * pitch + c];
        }
    }
}

template <typename T>
static void run_cumsum_rows(hipStream_t st, const u8 *stream, u8 *out, const u64 *d_stream_off, const u64 *d_out_off,
                            const u32 *d_rows, const int *d_status, int n_chunks, u32 max_rows, int nc, u32 *sums)
{
    const int tt = rows_tile(nc, (int)sizeof(T)), pitch = rows_pitch(nc, (int)sizeof(T));
    const size_t lds = (size_t)tt * pitch * sizeof(T);
    (void)ensure_dynamic_lds((const void *)k_cumsum_rows<T>, 64 * 1024);
    const int ntile = (max_rows + tt - 1) / tt;
    dim3 grid(xcd_row_grid(ntile, n_chunks));
    hipLaunchKernelGGL(k_rows_sums<T>, grid, dim3(256), 0, st, stream, d_stream_off, d_rows, d_status, nc, tt, ntile, sums);
    hipLaunchKernelGGL(k_rows_scan, dim3((nc + 63) / 64, n_chunks), dim3(64 * ROWS_SCAN_PARTS), 0, st, d_rows, d_status, nc, tt, ntile, sums);
    hipLaunchKernelGGL(k_cumsum_rows<T>, grid, dim3(256), lds, st, stream, out, d_stream_off, d_out_off, d_rows, d_status, nc, tt, pitch,
                       0xffffffffu / (u32)nc + 1, ntile, sums);
}

static int segr_for(int itemsize) { return itemsize <= 2 ? 256 : itemsize == 4 ? 128 : 64; }

size_t cumsum_scratch_bytes(int n_chunks, u32 max_rows, int n_channels)
{
    // generic kernels: u64 sums per (segment of >= 64 rows, channel); row-tile kernels: u32 sums per (tile of >= 16 rows,
    // channel) -- the wide-channel case (tiles of 16 rows from 1024 int16 channels on) needs twice the first bound
    const size_t seg = (size_t)n_chunks * ((max_rows + 63) / 64 + 1) * n_channels * sizeof(u64);
    const size_t til = (size_t)n_chunks * ((max_rows + 15) / 16 + 1) * n_channels * sizeof(u32);
    return (seg > til ? seg : til) + 256;
}

int launch_cumsum_transpose(hipStream_t st, const void *d_stream, void *d_out, const u64 *d_stream_off,
                            const u64 *d_out_off, const u32 *d_rows, const int *d_status, int n_chunks,
                            u32 max_rows, int n_channels, int itemsize, int flags, void *d_segsums)
{
    if (n_chunks == 0 || max_rows == 0) return MTS_OK;
    const u8 *s = (const u8 *)d_stream;
    u8 *o = (u8 *)d_out;
    u64 *ss = (u64 *)d_segsums;
    (void)segr_for;
    if (flags & MTS_FLAG_FLOAT) {
        if (itemsize == 4) return run_cumsum_float<float>(st, s, o, d_stream_off, d_out_off, d_rows, d_status, n_chunks, max_rows, n_channels, flags);
        if (itemsize == 8) return run_cumsum_float<double>(st, s, o, d_stream_off, d_out_off, d_rows, d_status, n_chunks, max_rows, n_channels, flags);
        return MTS_E_ARG;
    }
    if (flags == (MTS_FLAG_TIME_DIFF | MTS_FLAG_ORDER_F) && itemsize <= 4 && n_channels >= 2 && rows_tile(n_channels, itemsize) && !getenv("MTS_K12_GENERIC")) {
        switch (itemsize) {
        case 1: run_cumsum_rows<u8>(st, s, o, d_stream_off, d_out_off, d_rows, d_status, n_chunks, max_rows, n_channels, (u32 *)ss); break;
        case 2: run_cumsum_rows<u16>(st, s, o, d_stream_off, d_out_off, d_rows, d_status, n_chunks, max_rows, n_channels, (u32 *)ss); break;
        default: run_cumsum_rows<u32>(st, s, o, d_stream_off, d_out_off, d_rows, d_status, n_chunks, max_rows, n_channels, (u32 *)ss); break;
        }
        MTS_HIP(hipGetLastError());
        return MTS_OK;
    }
    switch (itemsize) {
    case 1: return run_cumsum<u8, 256>(st, s, o, d_stream_off, d_out_off, d_rows, d_status, n_chunks, max_rows, n_channels, flags, ss);
    case 2: return run_cumsum<u16, 256>(st, s, o, d_stream_off, d_out_off, d_rows, d_status, n_chunks, max_rows, n_channels, flags, ss);
    case 4: return run_cumsum<u32, 128>(st, s, o, d_stream_off, d_out_off, d_rows, d_status, n_chunks, max_rows, n_channels, flags, ss);
    case 8: return run_cumsum<u64, 64>(st, s, o, d_stream_off, d_out_off, d_rows, d_status, n_chunks, max_rows, n_channels, flags, ss);
    default: return MTS_E_ARG;
    }
}

// ------------------------------------------------------------------------------------------------
// synthetic int16 recording: integer-exact, counter based
// ------------------------------------------------------------------------------------------------
__constant__ int c_taps[64] = {256, 230, 207, 187, 168, 151, 136, 122, 110, 99, 89, 80, 72, 65, 59, 53,
                               47, 43, 38, 35, 31, 28, 25, 23, 20, 18, 17, 15, 13, 12, 11, 10,
                               9, 8, 7, 6, 6, 5, 5, 4, 4, 3, 3, 3, 2, 2, 2, 2,
                               2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0};

__device__ __forceinline__ int synth_noise(long t, int c, int nc, u64 seedx)
{
    u64 idx = (u64)(t * (long)nc + c) ^ seedx;
    u64 z = idx + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (int)((z & 0xffff) + ((z >> 16) & 0xffff) + ((z >> 32) & 0xffff) + (z >> 48)) - 131070;
}

// each thread produces RUN consecutive time samples of one channel, reusing the noise window
constexpr int SYNTH_RUN = 32;
__global__ __launch_bounds__(256) void k_synth(int16_t *out, long t0, long t1, int nc, u64 seedx)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    const long ta = t0 + (long)blockIdx.y * SYNTH_RUN;
    if (ta >= t1) return;
    int s[64 + SYNTH_RUN - 1];
#pragma unroll
    for (int k = 0; k < 64 + SYNTH_RUN - 1; k++) s[k] = synth_noise(ta - 63 + k, c, nc, seedx);
#pragma unroll
    for (int r = 0; r < SYNTH_RUN; r++) {
        const long t = ta + r;
        if (t >= t1) break;
        int y = 0;
#pragma unroll
        for (int k = 0; k < 60; k++) y += c_taps[k] * s[63 + r - k];
        out[(t - t0) * nc + c] = (int16_t)((4 * y) >> 23);
    }
}

int launch_synth_int16(hipStream_t st, int16_t *d_out, long t0, long t1, int n_channels, long seed)
{
    if (t1 <= t0) return MTS_OK;
    const u64 seedx = (u64)seed * 0xD1B54A32D192ED03ull;
    const long nrun = (t1 - t0 + SYNTH_RUN - 1) / SYNTH_RUN;
    // grid.y is limited to 65535: loop in slabs
    for (long y0 = 0; y0 < nrun; y0 += 65535) {
        const long ny = nrun - y0 < 65535 ? nrun - y0 : 65535;
        dim3 grid((n_channels + 255) / 256, (unsigned)ny);
        hipLaunchKernelGGL(k_synth, grid, dim3(256), 0, st, d_out + (y0 * SYNTH_RUN) * n_channels, t0 + y0 * SYNTH_RUN, t1, n_channels, seedx);
    }
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

// ================================================================================================
// rectangular pieces of decoded chunks (Reader[rows, columns]; several requests per launch)
// ================================================================================================
// One thread per requested item; consecutive threads take consecutive columns of one row, so with a column step of 1 the
// reads and the writes are both contiguous.  The chunks of the call are few: the owning chunk of a row is found by bisection.
template <typename T>
__global__ __launch_bounds__(256) void k_gather_slices(const GatherChunk *__restrict__ chunks, int n_chunks, const GatherReq *__restrict__ req,
                                                       int n_channels, u8 *__restrict__ out)
{
    const GatherReq q = req[blockIdx.y];
    const u64 items = (u64)q.nr * (u64)q.ncol;
    for (u64 e = (u64)blockIdx.x * 256 + threadIdx.x; e < items; e += (u64)gridDim.x * 256) {
        const long i = (long)(e / (u64)q.ncol), j = (long)(e % (u64)q.ncol);
        const long row = q.rb + i * q.rs, col = q.cb + j * q.cs;
        int lo = 0, hi = n_chunks - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (chunks[mid].row0 <= row) lo = mid; else hi = mid - 1; }
        const GatherChunk c = chunks[lo];
        if (!c.base) continue;                                         // (a chunk that failed to decode: its rows are not written)
        ((T *)(out + q.out_off))[e] = ((const T *)c.base)[(u64)(row - c.row0) * (u64)c.pitch + (u64)col];
    }
}

int launch_gather_slices(hipStream_t st, const GatherChunk *d_chunks, int n_chunks, const GatherReq *d_req, int n_req, u64 max_items,
                         int n_channels, int itemsize, u8 *d_out)
{
    if (n_req == 0 || max_items == 0) return MTS_OK;
    const u64 nb = (max_items + 255) / 256;
    if (itemsize != 1 && itemsize != 2 && itemsize != 4 && itemsize != 8) return MTS_E_ARG;
    // grid.y (one request per row of workgroups) is limited to 65535: slabs of requests
    for (int r0 = 0; r0 < n_req; r0 += 65535) {
        const int nr = n_req - r0 < 65535 ? n_req - r0 : 65535;
        dim3 grid((unsigned)(nb < 4096 ? nb : 4096), (unsigned)nr);
        switch (itemsize) {
        case 1: hipLaunchKernelGGL(k_gather_slices<u8>, grid, dim3(256), 0, st, d_chunks, n_chunks, d_req + r0, n_channels, d_out); break;
        case 2: hipLaunchKernelGGL(k_gather_slices<u16>, grid, dim3(256), 0, st, d_chunks, n_chunks, d_req + r0, n_channels, d_out); break;
        case 4: hipLaunchKernelGGL(k_gather_slices<u32>, grid, dim3(256), 0, st, d_chunks, n_chunks, d_req + r0, n_channels, d_out); break;
        default: hipLaunchKernelGGL(k_gather_slices<u64>, grid, dim3(256), 0, st, d_chunks, n_chunks, d_req + r0, n_channels, d_out); break;
        }
    }
    MTS_HIP(hipGetLastError());
    return MTS_OK;
}

}  // namespace mts
