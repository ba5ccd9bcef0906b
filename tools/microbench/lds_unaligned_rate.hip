#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// mode bits: 1 = unaligned addresses, 2 = include 4 b32 writes, 4 = sequential-ish addresses (lane*8 + small)
__global__ __launch_bounds__(1024) void k(uint64_t *out, int mode, int iters, int nw)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s[];
    const uint32_t base = (uint32_t)(uintptr_t)s;
    for (int i = threadIdx.x; i < 32768; i += 1024) ((volatile uint32_t *)s)[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= nw) return;
    uint32_t x = threadIdx.x * 2654435761u + 12345u;
    uint64_t acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        uint32_t a = (x >> 8) & 0xfff8;
        if (mode & 4) a = ((it * 512 + lane * 8) & 0xfff8);
        if (mode & 1) a = (a + ((x >> 28) & 7)) & 0xffff;
        if (a > 65528) a = 65528;
        uint64_t t, d;
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t), "=&v"(d) : "v"(base + a), "v"(base + 65536 + a) : "memory");
        acc += t ^ d;
        if (mode & 2) {
            uint32_t b = (a * 7 + 13) & 0xffff; if (!(mode & 1)) b &= 0xfffc; if (b > 65520) b = 65520;
            asm volatile("ds_write_b32 %0, %2\n\tds_write_b32 %1, %3\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %1, %3 offset:4"
                         :: "v"(base + b), "v"(base + 65536 + b), "v"((uint32_t)d), "v"((uint32_t)t) : "memory");
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = acc; }
}
int main()
{
    uint64_t *d, h[32];
    hipMalloc(&d, sizeof h);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 20000;
    for (int nw : {1, 4, 8, 14}) for (int mode = 0; mode < 8; mode++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(1024), 131072, 0, d, mode, iters, nw);
        if (hipDeviceSynchronize() != hipSuccess) { printf("fail\n"); return 1; }
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double avg = 0; for (int w = 0; w < nw; w++) avg += (double)h[w * 2] / iters; avg /= nw;
        printf("waves %2d mode %d (%s%s%s): %.0f cycles/iter\n", nw, mode, mode & 1 ? "unaligned " : "aligned ", mode & 2 ? "+writes " : "", mode & 4 ? "seq" : "random", avg);
    }
    return 0;
}
