#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t *out, int off)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (uint8_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    uint32_t a = (uint32_t)(uintptr_t)(s) + off + lane;     // arbitrary alignment
    uint32_t v, w0, w1; uint64_t v64;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(a) : "memory");
    out[lane] = v; out[64 + lane] = (uint32_t)v64; out[128 + lane] = (uint32_t)(v64 >> 32);
    __syncthreads();
    // unaligned stores: lane 0..7 store 0xAABBCCDD at byte offset 100 + 5*lane
    if (lane < 8) { uint32_t b = (uint32_t)(uintptr_t)(s) + 100 + 5 * lane; uint32_t d = 0xA0B0C0D0u + lane;
        asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(b), "v"(d) : "memory"); }
    if (lane == 8) { uint32_t b = (uint32_t)(uintptr_t)(s) + 151; uint32_t d = 0x1234;
        asm volatile("ds_write_b16 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(b), "v"(d) : "memory"); }
    __syncthreads();
    for (int i = threadIdx.x; i < 64; i += 64) out[192 + i] = ((uint32_t *)s)[i];
}
int main()
{
    uint32_t *d, h[256];
    hipMalloc(&d, sizeof h);
    for (int off = 0; off < 4; off++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, off);
        if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 1; }
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) {
            uint32_t e = 0; uint64_t e64 = 0;
            for (int b = 0; b < 4; b++) e |= (uint32_t)((off + l + b) & 255) << (8 * b);
            for (int b = 0; b < 8; b++) e64 |= (uint64_t)((off + l + b) & 255) << (8 * b);
            if (h[l] != e) bad++;
            if (h[64 + l] != (uint32_t)e64 || h[128 + l] != (uint32_t)(e64 >> 32)) bad += 100;
        }
        printf("off %d read mismatches %d\n", off, bad);
    }
    uint8_t *sb = (uint8_t *)&h[192];
    for (int i = 96; i < 160; i++) printf("%02x ", sb[i]);
    printf("\n");
    return 0;
}
