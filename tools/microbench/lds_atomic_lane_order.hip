#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// Do same-address LDS atomics of ONE wave instruction retire in lane order?  (old values increase with the lane id)
__global__ __launch_bounds__(1024) void k(uint32_t *bad, int iters, int nbins_mask)
{
    __shared__ uint32_t cnt[16][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t x = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 99u;
    uint32_t nbad = 0;
    for (int it = 0; it < iters; it++) {
        for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0;
        __builtin_amdgcn_wave_barrier();
        x = x * 1664525u + 1013904223u;
        const uint32_t d = (x >> 13) & nbins_mask;
        const uint32_t old = atomicAdd(&cnt[wave][d], 1u);
        // expected: number of lower lanes with the same digit
        uint64_t m = ~0ull;
        for (int b = 0; b < 8; b++) { const bool bit = (d >> b) & 1; const uint64_t bal = __ballot(bit); m &= bit ? bal : ~bal; }
        const uint32_t rank = __popcll(m & ((1ull << lane) - 1));
        if (old != rank) nbad++;
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main()
{
    uint32_t *d, h;
    hipMalloc(&d, 4);
    for (int mask : {0, 1, 3, 15, 63, 255}) {
        hipMemset(d, 0, 4);
        hipLaunchKernelGGL(k, dim3(512), dim3(1024), 0, 0, d, 2000, mask);
        hipDeviceSynchronize();
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("bins %3d: %u out-of-order of %llu\n", mask + 1, h, 512ull * 1024 * 2000);
    }
    return 0;
}
