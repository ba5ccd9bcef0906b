#include <hip/hip_runtime.h>
#include <stdio.h>
// integer VALU throughput per SIMD: N independent chains of xor/add/and per wave, W waves per CU
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters)
{
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 77, d = a + 5, e = a * 7, f = a ^ 3, g = a + 11, h = a * 13;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (KIND == 0) { a ^= b + k; b += c; c ^= d; d += e ^ k; e ^= f; f += g; g ^= h; h += a; }
            if (KIND == 1) { a = __builtin_amdgcn_alignbit(a, b, c); b = __builtin_amdgcn_alignbit(b, c, d); c = __builtin_amdgcn_alignbit(c, d, e); d = __builtin_amdgcn_alignbit(d, e, f); e = __builtin_amdgcn_alignbit(e, f, g); f = __builtin_amdgcn_alignbit(f, g, h); g = __builtin_amdgcn_alignbit(g, h, a); h = __builtin_amdgcn_alignbit(h, a, b); }
            if (KIND == 2) { a = a > b ? c : a; b = b > c ? d : b; c = c > d ? e : c; d = d > e ? f : d; e = e > f ? g : e; f = f > g ? h : f; g = g > h ? a : g; h = h > a ? b : h; }
            if (KIND == 3) { a = __builtin_clz(a | 1) + b; b = __builtin_clz(b | 1) + c; c = __builtin_clz(c | 1) + d; d = __builtin_clz(d | 1) + e; e = __builtin_clz(e | 1) + f; f = __builtin_clz(f | 1) + g; g = __builtin_clz(g | 1) + h; h = __builtin_clz(h | 1) + a; }
            if (KIND == 4) { a = __umul24(a, b) + 1; b = __umul24(b, c) + 1; c = __umul24(c, d) + 1; d = __umul24(d, e) + 1; e = __umul24(e, f) + 1; f = __umul24(f, g) + 1; g = __umul24(g, h) + 1; h = __umul24(h, a) + 1; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
template <int KIND> void run(const char *name, int opsPerInner, int blocksPerCU)
{
    unsigned *d; hipMalloc(&d, 256 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    k<KIND><<<256 * blocksPerCU, 256>>>(d, 10);
    hipEventRecord(e0);
    k<KIND><<<256 * blocksPerCU, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waveinst = (double)256 * blocksPerCU * 4 * iters * 16 * opsPerInner;   // wave-instructions
    printf("%-10s blocks/CU %d: %.3f ms  %.3e wave-inst/s  -> cycles per wave-inst per SIMD @2.4GHz: %.2f\n", name, blocksPerCU, ms, waveinst / (ms * 1e-3), 1024 * 2.4e9 / (waveinst / (ms * 1e-3)));
    hipFree(d);
}
int main()
{
    for (int b : {1, 2, 4}) {
        run<0>("xor/add", 10, b);     // 8 ops + 2 (b+k, e^k)
        run<1>("alignbit", 8, b);
        run<2>("cmp+cndmask", 16, b);
        run<3>("clz+or+add", 24, b);
        run<4>("mul24+add", 16, b);
    }
    return 0;
}
