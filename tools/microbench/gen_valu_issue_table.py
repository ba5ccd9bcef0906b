#!/usr/bin/env python3
"""Writes valu_issue_table.hip: one kernel per opcode, each a straight line of 64 independent instances of ONE instruction
(inline asm, destinations rotating over 8 registers so that no instruction waits for its predecessor), looped; every wave
times itself with s_memtime and the host times the launch.  Output of the binary: cycles per wave64 instruction and SIMD at
1 / 2 / 4 / 8 waves per SIMD, per opcode -- the table behind DESIGN.md's statement about what bounds k_match5.

    python3 tools/microbench/gen_valu_issue_table.py > tools/microbench/valu_issue_table.hip
    hipcc --offload-arch=gfx950 -O2 -o tools/microbench/build/valu_issue_table tools/microbench/valu_issue_table.hip
"""
# (name, instruction template, extra clobbers).  {d} = destination / accumulator register k, {a} {b} = registers k+1, k+2 (mod 8)
V = [
    # --- what k_match5's hot loop is made of
    ('v_mov_b32', 'v_mov_b32 {d}, {a}', ''),
    ('v_add_u32', 'v_add_u32 {d}, {d}, {a}', ''),
    ('v_sub_u32', 'v_sub_u32 {d}, {d}, {a}', ''),
    ('v_and_b32', 'v_and_b32 {d}, {d}, {a}', ''),
    ('v_or_b32', 'v_or_b32 {d}, {d}, {a}', ''),
    ('v_xor_b32', 'v_xor_b32 {d}, {d}, {a}', ''),
    ('v_lshlrev_b32', 'v_lshlrev_b32 {d}, 3, {a}', ''),
    ('v_lshrrev_b32', 'v_lshrrev_b32 {d}, 3, {a}', ''),
    ('v_lshlrev_b32(v)', 'v_lshlrev_b32 {d}, {a}, {b}', ''),
    ('v_bfe_u32', 'v_bfe_u32 {d}, {a}, 3, 5', ''),
    ('v_bfi_b32', 'v_bfi_b32 {d}, {a}, {b}, {d}', ''),
    ('v_alignbit_b32', 'v_alignbit_b32 {d}, {a}, {b}, {d}', ''),
    ('v_alignbyte_b32', 'v_alignbyte_b32 {d}, {a}, {b}, {d}', ''),
    ('v_perm_b32', 'v_perm_b32 {d}, {a}, {b}, {d}', ''),
    ('v_bitop3_b32', 'v_bitop3_b32 {d}, {d}, {a}, {b} bitop3:0x60', ''),
    ('v_and_or_b32', 'v_and_or_b32 {d}, {d}, {a}, {b}', ''),
    ('v_or3_b32', 'v_or3_b32 {d}, {d}, {a}, {b}', ''),
    ('v_add3_u32', 'v_add3_u32 {d}, {d}, {a}, {b}', ''),
    ('v_lshl_add_u32', 'v_lshl_add_u32 {d}, {a}, 3, {b}', ''),
    ('v_lshl_or_b32', 'v_lshl_or_b32 {d}, {a}, 3, {b}', ''),
    ('v_xad_u32', 'v_xad_u32 {d}, {d}, {a}, {b}', ''),
    ('v_lshl_add_u64', 'v_lshl_add_u64 {D}, {A}, 2, {D}', ''),
    ('v_min_u32', 'v_min_u32 {d}, {d}, {a}', ''),
    ('v_max_u32', 'v_max_u32 {d}, {d}, {a}', ''),
    ('v_min3_u32', 'v_min3_u32 {d}, {d}, {a}, {b}', ''),
    ('v_med3_u32', 'v_med3_u32 {d}, {d}, {a}, {b}', ''),
    ('v_ffbl_b32', 'v_ffbl_b32 {d}, {a}', ''),
    ('v_ffbh_u32', 'v_ffbh_u32 {d}, {a}', ''),
    ('v_bcnt_u32_b32', 'v_bcnt_u32_b32 {d}, {a}, {d}', ''),
    ('v_mbcnt_lo_u32_b32', 'v_mbcnt_lo_u32_b32 {d}, {a}, {d}', ''),
    ('v_bfrev_b32', 'v_bfrev_b32 {d}, {a}', ''),
    ('v_not_b32', 'v_not_b32 {d}, {a}', ''),
    ('v_cndmask_b32(vcc)', 'v_cndmask_b32 {d}, {d}, {a}, vcc', ''),
    ('v_cndmask_b32(sgpr)', 'v_cndmask_b32_e64 {d}, {d}, {a}, s[20:21]', ''),
    ('v_cmp_ne_u32->vcc', 'v_cmp_ne_u32 vcc, {d}, {a}', 'vcc'),
    ('v_cmp_lt_u32->sgpr', 'v_cmp_lt_u32_e64 s[22:23], {d}, {a}', 's22,s23'),
    ('v_cmp_ne_u64->vcc', 'v_cmp_ne_u64 vcc, {D}, {A}', 'vcc'),
    ('v_cmpx_ge_u32', 'v_cmpx_ge_u32_e32 vcc, {d}, {d}', 'vcc'),
    ('v_mul_u32_u24', 'v_mul_u32_u24 {d}, {d}, {a}', ''),
    ('v_mad_u32_u24', 'v_mad_u32_u24 {d}, {d}, {a}, {b}', ''),
    ('v_mad_i32_i24', 'v_mad_i32_i24 {d}, {a}, -8, {b}', ''),
    ('v_mul_lo_u32', 'v_mul_lo_u32 {d}, {d}, {a}', ''),
    ('v_mul_hi_u32', 'v_mul_hi_u32 {d}, {d}, {a}', ''),
    ('v_mad_u64_u32', 'v_mad_u64_u32 {D}, s[22:23], {a}, {b}, {D}', 's22,s23'),
    ('v_add_co_u32', 'v_add_co_u32 {d}, vcc, {d}, {a}', 'vcc'),
    ('v_addc_co_u32', 'v_addc_co_u32 {d}, vcc, {d}, {a}, vcc', 'vcc'),
    ('v_lshlrev_b64', 'v_lshlrev_b64 {D}, 3, {A}', ''),
    ('v_lshrrev_b64', 'v_lshrrev_b64 {D}, 3, {A}', ''),
    ('v_mov_b64', 'v_mov_b64 {D}, {A}', ''),
    # --- packed 16-bit, bytes
    ('v_pk_add_u16', 'v_pk_add_u16 {d}, {d}, {a}', ''),
    ('v_pk_sub_u16', 'v_pk_sub_u16 {d}, {d}, {a}', ''),
    ('v_pk_min_u16', 'v_pk_min_u16 {d}, {d}, {a}', ''),
    ('v_pk_max_u16', 'v_pk_max_u16 {d}, {d}, {a}', ''),
    ('v_pk_lshlrev_b16', 'v_pk_lshlrev_b16 {d}, {a}, {d}', ''),
    ('v_pk_lshrrev_b16', 'v_pk_lshrrev_b16 {d}, {a}, {d}', ''),
    ('v_pk_mul_lo_u16', 'v_pk_mul_lo_u16 {d}, {d}, {a}', ''),
    ('v_pk_mad_u16', 'v_pk_mad_u16 {d}, {d}, {a}, {b}', ''),
    ('v_pk_add_f32', 'v_pk_add_f32 {D}, {D}, {A}', ''),
    ('v_pk_fma_f32', 'v_pk_fma_f32 {D}, {D}, {A}, {D}', ''),
    ('v_fma_f32', 'v_fma_f32 {d}, {d}, {a}, {b}', ''),
    ('v_add_f32', 'v_add_f32 {d}, {d}, {a}', ''),
    ('v_fma_f64', 'v_fma_f64 {D}, {D}, {A}, {D}', ''),
    ('v_sad_u8', 'v_sad_u8 {d}, {d}, {a}, {b}', ''),
    ('v_sad_u32', 'v_sad_u32 {d}, {d}, {a}, {b}', ''),
    ('v_msad_u8', 'v_msad_u8 {d}, {d}, {a}, {b}', ''),
    ('v_qsad_pk_u16_u8', 'v_qsad_pk_u16_u8 {D}, {D}, {a}, {D}', ''),
    ('v_mqsad_pk_u16_u8', 'v_mqsad_pk_u16_u8 {D}, {D}, {a}, {D}', ''),
    ('v_dot4_u32_u8', 'v_dot4_u32_u8 {d}, {d}, {a}, {b}', ''),
    ('v_dot4_i32_i8', 'v_dot4_i32_i8 {d}, {d}, {a}, {b}', ''),
    ('v_dot2_u32_u16', 'v_dot2_u32_u16 {d}, {d}, {a}, {b}', ''),
    ('v_cvt_pk_u8_f32', 'v_cvt_pk_u8_f32 {d}, {d}, {a}, {b}', ''),
    ('v_lerp_u8', 'v_lerp_u8 {d}, {d}, {a}, {b}', ''),
    # --- SDWA / DPP / cross-lane
    ('v_lshlrev_b32_sdwa', 'v_lshlrev_b32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1', ''),
    ('v_xor_b32_sdwa', 'v_xor_b32_sdwa {d}, {d}, {a} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2', ''),
    ('v_mov_b32_dpp(wave_shr)', 'v_mov_b32_dpp {d}, {a} wave_shr:1 row_mask:0xf bank_mask:0xf', ''),
    ('v_mov_b32_dpp(row_shr)', 'v_mov_b32_dpp {d}, {a} row_shr:1 row_mask:0xf bank_mask:0xf', ''),
    ('v_mov_b32_dpp(row_bcast15)', 'v_mov_b32_dpp {d}, {a} row_bcast:15 row_mask:0xa bank_mask:0xf', ''),
    ('v_mov_b32_dpp(quad_perm)', 'v_mov_b32_dpp {d}, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf', ''),
    ('v_add_u32_dpp(row_shr)', 'v_add_u32_dpp {d}, {a}, {d} row_shr:1 row_mask:0xf bank_mask:0xf', ''),
    ('v_or_b32_dpp(row_shr)', 'v_or_b32_dpp {d}, {a}, {d} row_shr:2 row_mask:0xf bank_mask:0xf', ''),
    ('v_readlane_b32', 'v_readlane_b32 s22, {a}, 63', 's22'),
    ('v_readfirstlane_b32', 'v_readfirstlane_b32 s22, {a}', 's22'),
    ('v_writelane_b32', 'v_writelane_b32 {d}, s20, 5', ''),
    ('v_permlane32_swap', 'v_permlane32_swap_b32 {d}, {a}', ''),
    ('v_permlane16_swap', 'v_permlane16_swap_b32 {d}, {a}', ''),
    ('v_swap_b32', 'v_swap_b32 {d}, {a}', ''),
    ('v_accvgpr_write', 'v_accvgpr_write_b32 a{k}, {a}', ''),
    ('v_accvgpr_read', 'v_accvgpr_read_b32 {d}, a{k}', ''),
    # --- the other pipes, alone
    ('s_add_u32', 's_add_u32 s2{k8}, s2{k8}, s20', 's20,s21,s22,s23,s24,s25,s26,s27,scc'),
    ('s_and_b64', 's_and_b64 s[22:23], s[22:23], s[20:21]', 's22,s23,scc'),
    ('s_bcnt1_i32_b64', 's_bcnt1_i32_b64 s22, s[20:21]', 's22,scc'),
    ('s_ff1_i32_b64', 's_ff1_i32_b64 s22, s[20:21]', 's22'),
    ('s_lshl_b64', 's_lshl_b64 s[22:23], s[20:21], 3', 's22,s23,scc'),
    ('s_nop', 's_nop 0', ''),
    ('ds_read_b32', 'ds_read_b32 {d}, {l}', ''),
    ('ds_read_b64', 'ds_read_b64 {D}, {l8}', ''),
    ('ds_read_b128', 'ds_read_b128 {Q}, {l16}', ''),
    ('ds_read2_b32', 'ds_read2_b32 {D}, {l} offset1:1', ''),
    ('ds_write_b32', 'ds_write_b32 {l}, {a}', ''),
    ('ds_write_b64', 'ds_write_b64 {l8}, {A}', ''),
    ('ds_or_b32', 'ds_or_b32 {l}, {a}', ''),
    ('ds_bpermute_b32', 'ds_bpermute_b32 {d}, {l}, {a}', ''),
    ('ds_swizzle_b32', 'ds_swizzle_b32 {d}, {a} offset:swizzle(SWAP,1)', ''),
]
# mixes: (name, [templates]) -- does a second pipe's instruction cost issue time next to VALU work?
MIX = [
    ('v_add_u32 + s_add_u32 (1:1)', ['v_add_u32 {d}, {d}, {a}', 's_add_u32 s22, s22, s20'], 's22,scc'),
    ('v_add_u32 + s_and_b64 (1:1)', ['v_add_u32 {d}, {d}, {a}', 's_and_b64 s[22:23], s[22:23], s[20:21]'], 's22,s23,scc'),
    ('v_add_u32 + ds_read_b32 (1:1)', ['v_add_u32 {d}, {d}, {a}', 'ds_read_b32 {e}, {l}'], ''),
    ('v_add_u32 + ds_read_b32 (3:1)', ['v_add_u32 {d}, {d}, {a}', 'v_add_u32 {d}, {d}, {a}', 'v_add_u32 {d}, {d}, {a}', 'ds_read_b32 {e}, {l}'], ''),
    ('v_add_u32 + ds_bpermute_b32 (3:1)', ['v_add_u32 {d}, {d}, {a}', 'v_add_u32 {d}, {d}, {a}', 'v_add_u32 {d}, {d}, {a}', 'ds_bpermute_b32 {e}, {l}, {a}'], ''),
    ('v_add_u32 + v_cmp->vcc + v_cndmask(vcc) dependent', ['v_cmp_ne_u32 vcc, {d}, {a}', 'v_cndmask_b32 {d}, {d}, {a}, vcc'], 'vcc'),
    ('v_cmp->sgpr + v_cndmask(sgpr) dependent', ['v_cmp_ne_u32_e64 s[22:23], {d}, {a}', 'v_cndmask_b32_e64 {d}, {d}, {a}, s[22:23]'], 's22,s23'),
    ('v_add_u32 dependent chain', ['v_add_u32 {z}, {z}, {z}'], ''),
    ('v_xor_b32 + v_ffbl_b32 dependent pair', ['v_xor_b32 {d}, {d}, {a}', 'v_ffbl_b32 {d}, {d}'], ''),
    ('v_cmp->vcc + v_bitop3(vcc_lo as data)', ['v_cmp_ne_u32 vcc, {d}, {a}', 'v_bitop3_b32 {d}, {d}, vcc_lo, {b} bitop3:0x60'], 'vcc'),
    ('v_readlane + s_add (dependent)', ['v_readlane_b32 s22, {a}, 3', 's_add_u32 s23, s22, s23'], 's22,s23,scc'),
    ('s_cbranch_scc0 not taken + v_add', ['s_cmp_eq_u32 s20, s21', 'v_add_u32 {d}, {d}, {a}'], 'scc'),
    # a lane mask combined on the scalar unit, then used by a select: through VCC / through another SGPR pair
    ('v_cmp->vcc; s_and vcc; v_cndmask(vcc)', ['v_cmp_ne_u32 vcc, {d}, {a}', 's_and_b64 vcc, vcc, s[20:21]', 'v_cndmask_b32 {d}, {d}, {a}, vcc'], 'vcc,scc'),
    ('v_cmp->sgpr; s_and sgpr; v_cndmask(sgpr)', ['v_cmp_ne_u32_e64 s[22:23], {d}, {a}', 's_and_b64 s[22:23], s[22:23], s[20:21]', 'v_cndmask_b32_e64 {d}, {d}, {a}, s[22:23]'], 's22,s23,scc'),
    ('v_cmp->vcc; s_and sgpr<-vcc; v_cndmask(sgpr)', ['v_cmp_ne_u32 vcc, {d}, {a}', 's_and_b64 s[22:23], vcc, s[20:21]', 'v_cndmask_b32_e64 {d}, {d}, {a}, s[22:23]'], 'vcc,s22,s23,scc'),
    ('s_mov vcc; v_cndmask(vcc)', ['s_mov_b64 vcc, s[20:21]', 'v_cndmask_b32 {d}, {d}, {a}, vcc'], 'vcc'),
    ('s_mov sgpr; v_cndmask(sgpr)', ['s_mov_b64 s[22:23], s[20:21]', 'v_cndmask_b32_e64 {d}, {d}, {a}, s[22:23]'], 's22,s23'),
]
N = 64            # instructions per asm block


def body(tpls):
    out = []
    k = 0
    while len(out) < N:
        for t in tpls:
            r = k % 8
            d = 'v%d' % (8 + r)
            a = 'v%d' % (8 + (r + 1) % 8)
            b = 'v%d' % (8 + (r + 2) % 8)
            p = 2 * (k % 4)
            D = 'v[%d:%d]' % (16 + p, 17 + p)
            A = 'v[%d:%d]' % (16 + (p + 2) % 8, 17 + (p + 2) % 8)
            Q = 'v[%d:%d]' % (24 + 4 * (k % 2), 27 + 4 * (k % 2))
            out.append(t.format(d=d, a=a, b=b, D=D, A=A, Q=Q, l='v32', l8='v37', l16='v38', e='v%d' % (33 + k % 4), z='v8', k=k % 8, k8=2 + k % 6))
            k += 1
    return out[:N]


def kernel(idx, tpls, clob):
    lines = body(tpls)
    asm = '\\n\\t'.join(lines)
    allv = ['v%d' % i for i in range(8, 39)] + ['a%d' % i for i in range(8)]
    cl = ', '.join('"%s"' % c for c in allv + [c for c in clob.split(',') if c] + ['memory'])
    return '''
__global__ __launch_bounds__(1024) void k%d(unsigned long long *out, int iters)
{
    __shared__ unsigned lds[4096];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 1; lds[threadIdx.x + 2048] = 2; lds[threadIdx.x + 3072] = 3;
    __syncthreads();
    const unsigned t = threadIdx.x;
    asm volatile("v_mov_b32 v8, %%0\\n\\tv_add_u32 v9, 1, %%0\\n\\tv_add_u32 v10, 2, %%0\\n\\tv_add_u32 v11, 3, %%0\\n\\t"
                 "v_add_u32 v12, 5, %%0\\n\\tv_add_u32 v13, 7, %%0\\n\\tv_add_u32 v14, 11, %%0\\n\\tv_add_u32 v15, 13, %%0\\n\\t"
                 "v_mov_b32 v16, %%0\\n\\tv_mov_b32 v17, 0\\n\\tv_mov_b32 v18, %%0\\n\\tv_mov_b32 v19, 0\\n\\tv_mov_b32 v20, %%0\\n\\tv_mov_b32 v21, 0\\n\\tv_mov_b32 v22, %%0\\n\\tv_mov_b32 v23, 0\\n\\t"
                 "v_mov_b32 v24, 0\\n\\tv_mov_b32 v25, 0\\n\\tv_mov_b32 v26, 0\\n\\tv_mov_b32 v27, 0\\n\\tv_mov_b32 v28, 0\\n\\tv_mov_b32 v29, 0\\n\\tv_mov_b32 v30, 0\\n\\tv_mov_b32 v31, 0\\n\\t"
                 "v_lshlrev_b32 v32, 2, %%0\\n\\tv_lshlrev_b32 v37, 3, %%0\\n\\tv_lshlrev_b32 v38, 4, %%0\\n\\tv_mov_b32 v33, 0\\n\\tv_mov_b32 v34, 0\\n\\tv_mov_b32 v35, 0\\n\\tv_mov_b32 v36, 0\\n\\t"
                 "s_mov_b64 s[20:21], 0x55\\n\\ts_mov_b64 s[22:23], 0\\n\\ts_mov_b64 s[24:25], 0\\n\\ts_mov_b64 s[26:27], 0\\n\\ts_mov_b64 vcc, 0x33"
                 :: "v"(t) : %s, "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc");
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %%0\\n\\ts_memtime %%1\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0) :: "memory");
    for (int i = 0; i < iters; i++) {
        asm volatile("%s" ::: %s);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_memtime %%0\\n\\ts_memrealtime %%1\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    unsigned v;
    asm volatile("v_xor_b32 %%0, v8, v9\\n\\tv_xor_b32 %%0, %%0, v16\\n\\tv_xor_b32 %%0, %%0, v33" : "=v"(v) :: %s);
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[3 * w] = t1 - t0; out[3 * w + 1] = r1 - r0; out[3 * w + 2] = v + lds[threadIdx.x] + lds[threadIdx.x + 3072];
    }
}
''' % (idx, cl, asm, cl, cl)


def main():
    allk = [(n, [t], c) for n, t, c in V] + MIX
    print('// GENERATED by tools/microbench/gen_valu_issue_table.py -- do not edit.  gfx950 only.')
    print('#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <string.h>\n#include <vector>\n#include <algorithm>')
    for i, (n, t, c) in enumerate(allk):
        print(kernel(i, t, c))
    print('typedef void (*kfn)(unsigned long long *, int);')
    print('struct K { const char *name; kfn f; };')
    print('static K ks[] = {')
    for i, (n, t, c) in enumerate(allk):
        print('    {"%s", k%d},' % (n, i))
    print('};')
    print(r'''
int main(int argc, char **argv)
{
    const char *only = argc > 1 ? argv[1] : nullptr;
    int ncu = 256;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); ncu = pr.multiProcessorCount;
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("# device %s, %d CUs, clockRate attribute %d kHz; N = %d instructions per block x iters\n", pr.name, ncu, clk_khz, ''' + str(N) + r''');
    printf("# columns: waves per SIMD 1 / 2 / 4 / 8 -> s_memtime ticks per instruction and SIMD | the same from the launch's wall time at the clockRate attribute | memtime ticks per us\n");
    unsigned long long *d; hipMalloc(&d, 3 * 8 * ncu * 16 * 4);
    std::vector<unsigned long long> h(3 * ncu * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    for (auto &k : ks) {
        if (only && !strstr(k.name, only)) continue;
        printf("%-44s", k.name);
        double wallc[4], tick[4], tpu = 0;
        int col = 0;
        for (int wps : {1, 2, 4}) {                       // waves per SIMD: one workgroup of 256 * wps threads per CU
            const int threads = 256 * wps, blocks = ncu;
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, 20);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d, 3 * 8 * blocks * (threads / 64), hipMemcpyDeviceToHost);
            const int nw = blocks * threads / 64;
            std::vector<double> dt(nw);
            double rt = 0, st = 0;
            for (int w = 0; w < nw; w++) { dt[w] = (double)h[3 * w]; st += (double)h[3 * w]; rt += (double)h[3 * w + 1]; }
            std::sort(dt.begin(), dt.end());
            const double ninst = (double)iters * ''' + str(N) + r''';
            tick[col] = dt[nw / 2] / (ninst * wps);
            wallc[col] = ms * 1e-3 * (clk_khz * 1e3) / (ninst * wps);
            tpu = st / (rt / 100.0);                       // s_memrealtime: 100 MHz
            col++;
        }
        {   // 8 waves per SIMD: two workgroups of 1024 threads per CU (both resident: no LDS to speak of, few registers)
            const int threads = 1024, blocks = 2 * ncu;
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, 20);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d, 3 * 8 * blocks * (threads / 64), hipMemcpyDeviceToHost);
            const int nw = blocks * threads / 64;
            std::vector<double> dt(nw);
            for (int w = 0; w < nw; w++) dt[w] = (double)h[3 * w];
            std::sort(dt.begin(), dt.end());
            const double ninst = (double)iters * ''' + str(N) + r''';
            tick[3] = dt[nw / 2] / (ninst * 8);
            wallc[3] = ms * 1e-3 * (clk_khz * 1e3) / (ninst * 8);
        }
        printf(" %6.2f %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f | %8.1f\n", tick[0], tick[1], tick[2], tick[3], wallc[0], wallc[1], wallc[2], wallc[3], tpu);
        fflush(stdout);
    }
    return 0;
}''')


if __name__ == '__main__':
    main()
