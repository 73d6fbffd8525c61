"""Development aid (round 3): issue cost of single VALU instruction kinds on gfx950 by encoding / datapath.
Generates issue_bench.hip; build: hipcc --offload-arch=gfx950 -O3 -o issue_bench issue_bench.hip"""
KINDS = [
    ("v_add_u32_e32", "v_add_u32_e32 {d}, {d}, {s}"),
    ("v_add_u32_e64", "v_add_u32_e64 {d}, {d}, {s}"),
    ("v_max_u32_e32", "v_max_u32_e32 {d}, {d}, {s}"),
    ("v_and_b32_e32", "v_and_b32_e32 {d}, {d}, {s}"),
    ("v_mov_b32_e32", "v_mov_b32_e32 {d}, {s}"),
    ("v_add_u32_lit", "v_add_u32_e32 {d}, 0x12345, {d}"),
    ("v_add3_u32", "v_add3_u32 {d}, {d}, {s}, {s}"),
    ("v_max3_u32", "v_max3_u32 {d}, {d}, {s}, {s}"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {d}, 1, {s}"),
    ("v_perm_b32", "v_perm_b32 {d}, {d}, {s}, {s}"),
    ("v_bfe_u32", "v_bfe_u32 {d}, {d}, 1, 31"),
    ("v_alignbit_b32", "v_alignbit_b32 {d}, {d}, {s}, 3"),
    ("v_pk_add_u16", "v_pk_add_u16 {d}, {d}, {s}"),
    ("v_pk_max_u16", "v_pk_max_u16 {d}, {d}, {s}"),
    ("v_pk_sub_u16_clamp", "v_pk_sub_u16 {d}, {d}, {s} clamp"),
    ("v_pk_min_u16", "v_pk_min_u16 {d}, {d}, {s}"),
    ("v_add_u16_e32", "v_add_u16_e32 {d}, {d}, {s}"),
    ("v_max_u16_e32", "v_max_u16_e32 {d}, {d}, {s}"),
    ("v_add_u32_sdwa", "v_add_u32_sdwa {d}, {d}, {s} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0"),
    ("v_max_u16_sdwa_hi", "v_max_u16_sdwa {d}, {d}, {s} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"),
    ("v_mov_dpp_row_shr1", "v_mov_b32_dpp {d}, {d} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"),
    ("v_mov_dpp_wave_shr1", "v_mov_b32_dpp {d}, {d} wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"),
    ("v_max_u32_dpp_row_shr1", "v_max_u32_dpp {d}, {d}, {d} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"),
    ("v_max_u32_dpp_row_shr8", "v_max_u32_dpp {d}, {d}, {d} row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"),
    ("v_max_u32_dpp_bcast15", "v_max_u32_dpp {d}, {d}, {d} row_bcast:15 row_mask:0xa bank_mask:0xf"),
    ("v_max_u32_dpp_bcast31", "v_max_u32_dpp {d}, {d}, {d} row_bcast:31 row_mask:0xc bank_mask:0xf"),
    ("v_max_u32_dpp_quad", "v_max_u32_dpp {d}, {d}, {d} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("v_cndmask_e32", "v_cndmask_b32_e32 {d}, {d}, {s}, vcc"),
    ("v_cndmask_e64", "v_cndmask_b32_e64 {d}, {d}, {s}, s[20:21]"),
    ("v_cmp_eq_e32", "v_cmp_eq_u32_e32 vcc, {d}, {s}"),
    ("v_cmp_eq_e64", "v_cmp_eq_u32_e64 s[20:21], {d}, {s}"),
    ("v_readlane", "v_readlane_b32 s22, {d}, 5"),
    ("v_readfirstlane", "v_readfirstlane_b32 s22, {d}"),
    ("v_mbcnt_lo", "v_mbcnt_lo_u32_b32 {d}, s20, {d}"),
    ("v_mul_u32_u24", "v_mul_u32_u24_e32 {d}, {d}, {s}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {d}, {d}, {s}, {s}"),
    ("v_lshlrev_e32", "v_lshlrev_b32_e32 {d}, 1, {d}"),
    ("v_sub_u32_e32", "v_sub_u32_e32 {d}, {d}, {s}"),
    ("v_xad_u32", "v_xad_u32 {d}, {d}, {s}, {s}"),
    ("v_and_or_b32", "v_and_or_b32 {d}, {d}, {s}, {s}"),
    ("v_or3_b32", "v_or3_b32 {d}, {d}, {s}, {s}"),
    ("v_pk_lshlrev_b16", "v_pk_lshlrev_b16 {d}, 1, {d}"),
    ("v_pk_mad_u16", "v_pk_mad_u16 {d}, {d}, {s}, {s}"),
    ("s_nop", "s_nop 0"),
    ("s_add_u32", "s_add_u32 s22, s22, 1"),
    ("mix_add_pk", "v_add_u32_e32 {d}, {d}, {s}\n v_pk_add_u16 {d}, {d}, {s}"),
    ("mix_add_dpp", "v_add_u32_e32 {d}, {d}, {s}\n v_max_u32_dpp {d}, {d}, {d} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"),
    ("dep_add_u32", None),        # one dependent chain
    ("dep_pk_add", None),
    ("dep_dpp", None),
]
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <cstring>', '#include <cstdlib>',
       'template <int KIND> __global__ __launch_bounds__(64) void k(unsigned *out, int iters)', '{',
       '    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = a0 * 3 + 1;',
       '    asm volatile("s_mov_b64 s[20:21], 0x5555\\n s_mov_b32 s22, 0\\n s_mov_b64 vcc, 0x3333" ::: "s20", "s21", "s22", "vcc");',
       '    unsigned long long t0 = __builtin_amdgcn_s_memtime();',
       '    for (int i = 0; i < iters; i++) {', '#pragma unroll', '        for (int u = 0; u < 8; u++) {']
for idx, (name, fmt) in enumerate(KINDS):
    if fmt is None:
        base = {"dep_add_u32": "v_add_u32_e32 %0, %0, %8", "dep_pk_add": "v_pk_add_u16 %0, %0, %8",
                "dep_dpp": "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"}[name]
        body = "\\n ".join([base] * 8)
    else:
        body = "\\n ".join(fmt.replace("\n", "\\n").format(d="%%%d" % r, s="%8") for r in range(8))
    src.append('            if (KIND == %d) asm volatile("%s" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "vcc");' % (idx, body))
src += ['        }', '    }', '    unsigned long long t1 = __builtin_amdgcn_s_memtime();',
        '    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;',
        '    if (threadIdx.x == 0) ((unsigned long long *)(out + 2097152))[blockIdx.x] = t1 - t0;', '}',
        'template <int KIND> void run(const char *name, unsigned *d, int per)', '{',
        '    const int iters = 1000;',
        '    for (int wps : {1, 2, 4, 8}) {',
        '        int blocks = 256 * 4 * wps;',
        '        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);',
        '        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 10); hipDeviceSynchronize();',
        '        hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1);',
        '        hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);',
        '        std::vector<unsigned long long> t(blocks);',
        '        hipMemcpy(t.data(), d + 2097152, blocks * 8, hipMemcpyDeviceToHost);',
        '        double avg = 0; for (auto x : t) avg += (double)x; avg /= blocks;',
        '        double instr = (double)iters * 64.0 * per;',
        '        printf("%-24s w/SIMD %d: memtime %.2f ticks/instr/wave = %.2f per SIMD issue; wall %.3f ns per instr per SIMD\\n", name, wps, avg / instr, avg / instr / wps, ms * 1e6 / (instr * wps));',
        '    }', '}', 'int main(int argc, char **argv)', '{',
        '    unsigned *d; hipMalloc(&d, 2097152 * 4 + 8192 * 8 * 2);']
for idx, (name, fmt) in enumerate(KINDS):
    per = 2 if name.startswith("mix_") else 1
    src.append('    if (argc < 2 || strstr("%s", argv[1])) run<%d>("%s", d, %d);' % (name, idx, name, per))
src += ['    return 0;', '}']
open(__file__.replace("gen_issue_bench.py", "issue_bench.hip"), "w").write("\n".join(src) + "\n")
