#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// mode 0: 3 reads (read2,read2,read) ; 1: reads + 5 ds_or ; 2: reads + 5 ds_write_b32 ; 3: only 5 ds_or ; 4: only 5 writes; 5: reads + 2 or
__global__ __launch_bounds__(1024) void k(uint64_t *out, int mode, int iters, int nw, int active)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s[];
    const uint32_t base = (uint32_t)(uintptr_t)s;
    for (int i = threadIdx.x; i < 20000; i += 1024) ((volatile uint32_t *)s)[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= nw) return;
    uint32_t x = threadIdx.x * 2654435761u + 12345u;
    uint64_t acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t a = (x >> 8) & 0xfff0, b = (x >> 12) & 0x1ffc, c = ((x >> 5) * 7) & 0xfff0, d = (x >> 3) & 0x1ffc;
        if (lane < active) {
        if (mode != 3 && mode != 4) {
            u32x2 t, dd; uint32_t d2;
            asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %4 offset1:1\n\tds_read_b32 %2, %4 offset:8\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(t), "=&v"(dd), "=&v"(d2) : "v"(base + 65536 + b), "v"(base + a) : "memory");
            acc += t.x ^ dd.y ^ d2;
        }
        if (mode == 1 || mode == 3)
            asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %0, %2 offset:4\n\tds_or_b32 %0, %3 offset:8\n\tds_or_b32 %4, %5\n\tds_or_b32 %4, %6 offset:4"
                         :: "v"(base + c), "v"(x), "v"(x >> 3), "v"(x >> 7), "v"(base + 65536 + d), "v"(x >> 9), "v"(x >> 11) : "memory");
        if (mode == 2 || mode == 4)
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %0, %3 offset:8\n\tds_write_b32 %4, %5\n\tds_write_b32 %4, %6 offset:4"
                         :: "v"(base + c), "v"(x), "v"(x >> 3), "v"(x >> 7), "v"(base + 65536 + d), "v"(x >> 9), "v"(x >> 11) : "memory");
        if (mode == 5)
            asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %2, %3" :: "v"(base + c), "v"(x), "v"(base + 65536 + d), "v"(x >> 9) : "memory");
        if (mode == 3 || mode == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = acc; }
}
int main()
{
    uint64_t *d, h[32];
    hipMalloc(&d, sizeof h);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    const int iters = 20000;
    const char *names[] = {"3 reads", "3 reads + 5 or", "3 reads + 5 write", "5 or", "5 write", "3 reads + 2 or"};
    for (int active : {64, 8}) for (int nw : {1, 4, 14}) for (int mode = 0; mode < 6; mode++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(1024), 81920, 0, d, mode, iters, nw, active);
        if (hipDeviceSynchronize() != hipSuccess) { printf("fail\n"); return 1; }
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double avg = 0; for (int w = 0; w < nw; w++) avg += (double)h[w * 2] / iters; avg /= nw;
        printf("lanes %2d waves %2d %-18s: %.0f cycles/iter\n", active, nw, names[mode], avg);
    }
    return 0;
}
