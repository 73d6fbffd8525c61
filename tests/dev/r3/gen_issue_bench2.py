"""Development aid (round 3): issue cost of more VALU instruction kinds and of fast/slow instruction PATTERNS on gfx950.
Generates issue_bench2.hip; build: hipcc --offload-arch=gfx950 -O3 -o issue_bench2 issue_bench2.hip.  8 waves per SIMD only."""
A = "v_add_u32_e32 {d}, {d}, {s}"
P = "v_pk_add_u16 {d}, {d}, {s}"
M16 = "v_max_u16_e32 {d}, {d}, {s}"
KINDS = [
    ("v_add_u32_e32", [A] * 8),
    ("v_pk_add_u16", [P] * 8),
    ("pat_a_p_alternate", [A, P] * 4),
    ("pat_aa_pp", [A, A, P, P] * 2),
    ("pat_aaaa_pppp", [A] * 4 + [P] * 4),
    ("pat_aaa_p", [A, A, A, P] * 2),
    ("pat_a_ppp", [A, P, P, P] * 2),
    ("pat_m16_p_alternate", [M16, P] * 4),
    ("v_sub_u16_e64_clamp", ["v_sub_u16_e64 {d}, {d}, {s} clamp"] * 8),
    ("v_sub_u16_e32", ["v_sub_u16_e32 {d}, {d}, {s}"] * 8),
    ("v_min_u16_e32", ["v_min_u16_e32 {d}, {d}, {s}"] * 8),
    ("v_max_i16_e32", ["v_max_i16_e32 {d}, {d}, {s}"] * 8),
    ("v_min_u32_e32", ["v_min_u32_e32 {d}, {d}, {s}"] * 8),
    ("v_max_i32_e32", ["v_max_i32_e32 {d}, {d}, {s}"] * 8),
    ("v_or_b32_e32", ["v_or_b32_e32 {d}, {d}, {s}"] * 8),
    ("v_xor_b32_e32", ["v_xor_b32_e32 {d}, {d}, {s}"] * 8),
    ("v_not_b32", ["v_not_b32_e32 {d}, {d}"] * 8),
    ("v_subrev_u32", ["v_subrev_u32_e32 {d}, {d}, {s}"] * 8),
    ("v_lshrrev_b32", ["v_lshrrev_b32_e32 {d}, 1, {d}"] * 8),
    ("v_ashrrev_i32", ["v_ashrrev_i32_e32 {d}, 1, {d}"] * 8),
    ("v_lshlrev_b16", ["v_lshlrev_b16_e32 {d}, 1, {d}"] * 8),
    ("v_lshl_add_u32", ["v_lshl_add_u32 {d}, {d}, 1, {s}"] * 8),
    ("v_add_lshl_u32", ["v_add_lshl_u32 {d}, {d}, {s}, 1"] * 8),
    ("v_bfi_b32", ["v_bfi_b32 {d}, {s}, {d}, {s}"] * 8),
    ("v_mad_u16", ["v_mad_u16 {d}, {d}, {s}, {s}"] * 8),
    ("v_mul_lo_u16", ["v_mul_lo_u16_e32 {d}, {d}, {s}"] * 8),
    ("v_add_co_u32", ["v_add_co_u32_e32 {d}, vcc, {d}, {s}"] * 8),
    ("v_addc_co_u32", ["v_addc_co_u32_e32 {d}, vcc, {d}, {s}, vcc"] * 8),
    ("v_cmp_eq_sdwa_byte", ["v_cmp_eq_u32_sdwa s[20:21], {d}, {s} src0_sel:BYTE_1 src1_sel:DWORD"] * 8),
    ("v_cmp_gt_u16_e32", ["v_cmp_gt_u16_e32 vcc, {d}, {s}"] * 8),
    ("cmp_e32_then_cndmask_e32", ["v_cmp_eq_u32_e32 vcc, {d}, {s}\n v_cndmask_b32_e32 {d}, {d}, {s}, vcc"] * 8),
    ("cmp_e64_then_cndmask_e64", ["v_cmp_eq_u32_e64 s[20:21], {d}, {s}\n v_cndmask_b32_e64 {d}, {d}, {s}, s[20:21]"] * 8),
    ("cndmask_e32_spaced_by_adds", ["v_cndmask_b32_e32 {d}, {d}, {s}, vcc\n v_add_u32_e32 {d}, {d}, {s}"] * 8),
    ("v_cndmask_e32_alone", ["v_cndmask_b32_e32 {d}, {d}, {s}, vcc"] * 8),
    ("v_max3_u16?", None),
    ("v_sat_sub_via_max_sub", ["v_max_u16_e32 {d}, {d}, {s}\n v_sub_u32_e32 {d}, {d}, {s}"] * 8),
    ("v_pk_mul_lo_u16", ["v_pk_mul_lo_u16 {d}, {d}, {s}"] * 8),
    ("v_and_b32_lit", ["v_and_b32_e32 {d}, 0x10001, {d}"] * 8),
    ("v_mov_b32_sdwa", ["v_mov_b32_sdwa {d}, {s} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0"] * 8),
    ("ds_bpermute", ["ds_bpermute_b32 {d}, {s}, {d}\n s_waitcnt lgkmcnt(0)"] * 8),
    ("ds_swizzle", ["ds_swizzle_b32 {d}, {d} offset:0x801f\n s_waitcnt lgkmcnt(0)"] * 8),
    ("v_readlane_then_use", ["v_readlane_b32 s22, {d}, 5\n v_add_u32_e32 {d}, s22, {d}"] * 8),
]
KINDS = [k for k in KINDS if k[1] is not None]
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <cstring>', '#include <cstdlib>',
       'template <int KIND> __global__ __launch_bounds__(64) void k(unsigned *out, int iters)', '{',
       '    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = (a0 * 3 + 1) & 0xff;',
       '    asm volatile("s_mov_b64 s[20:21], 0x5555\\n s_mov_b32 s22, 0\\n s_mov_b64 vcc, 0x3333" ::: "s20", "s21", "s22", "vcc");',
       '    for (int i = 0; i < iters; i++) {', '#pragma unroll', '        for (int u = 0; u < 8; u++) {']
counts = []
for idx, (name, fmts) in enumerate(KINDS):
    body = "\\n ".join(f.replace("\n", "\\n").format(d="%%%d" % r, s="%8") for r, f in enumerate(fmts))
    counts.append(sum(f.count("\n") + 1 for f in fmts) - sum(f.count("s_waitcnt") for f in fmts))
    src.append('            if (KIND == %d) asm volatile("%s" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "vcc");' % (idx, body))
src += ['        }', '    }',
        '    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;', '}',
        'template <int KIND> void run(const char *name, unsigned *d, int per8)', '{',
        '    const int iters = 1000;',
        '    for (int wps : {8, 4}) {',
        '        int blocks = 256 * 4 * wps;',
        '        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);',
        '        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 10); (void)hipDeviceSynchronize();',
        '        float best = 1e9f;',
        '        for (int it = 0; it < 3; it++) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters); (void)hipEventRecord(e1);',
        '        (void)hipDeviceSynchronize(); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }',
        '        double instr = (double)iters * 8.0 * per8;',
        '        printf("%-30s w/SIMD %d: %.3f ns per instruction and SIMD (%d instructions per group of 8 registers)\\n", name, wps, best * 1e6 / (instr * wps), per8);',
        '    }', '}', 'int main(int argc, char **argv)', '{',
        '    unsigned *d; (void)hipMalloc(&d, 2097152 * 4 + 8192 * 8 * 2);']
for idx, (name, fmts) in enumerate(KINDS):
    src.append('    if (argc < 2 || strstr("%s", argv[1])) run<%d>("%s", d, %d);' % (name, idx, name, counts[idx]))
src += ['    return 0;', '}']
open(__file__.replace("gen_issue_bench2.py", "issue_bench2.hip"), "w").write("\n".join(src) + "\n")
