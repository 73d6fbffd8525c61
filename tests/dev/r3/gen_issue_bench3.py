"""Development aid (round 3): issue cost of more VALU instruction kinds and of fast/slow instruction PATTERNS on gfx950.
Generates issue_bench3.hip; build: hipcc --offload-arch=gfx950 -O3 -o issue_bench2 issue_bench3.hip.  8 waves per SIMD only."""
A = "v_add_u32_e32 {d}, {d}, {s}"
P = "v_pk_add_u16 {d}, {d}, {s}"
M16 = "v_max_u16_e32 {d}, {d}, {s}"
P = "v_pk_add_u16 {d}, {d}, {s}"
KINDS = [
    ("pk_then_cnd_e64", [P + "\n v_cndmask_b32_e64 {d}, {d}, {s}, s[20:21]"] * 8),
    ("pk_then_smov_cnd_e32", [P + "\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32_e32 {d}, {d}, {s}, vcc"] * 8),
    ("pk_then_cnd_e32_vcc_static", [P + "\n v_cndmask_b32_e32 {d}, {d}, {s}, vcc"] * 8),
    ("pk_cmp_e64_cnd_e64", [P + "\n v_cmp_eq_u32_e64 s[20:21], {d}, {s}\n v_cndmask_b32_e64 {d}, {d}, {s}, s[20:21]"] * 8),
    ("pk_cmp_e32_cnd_e32", [P + "\n v_cmp_eq_u32_e32 vcc, {d}, {s}\n v_cndmask_b32_e32 {d}, {d}, {s}, vcc"] * 8),
    ("pk_cmp_e32_pk_cnd_e32", [P + "\n v_cmp_eq_u32_e32 vcc, {d}, {s}\n " + P + "\n v_cndmask_b32_e32 {d}, {d}, {s}, vcc"] * 8),
    ("add_then_smov_cnd_e32", ["v_add_u32_e32 {d}, {d}, {s}\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32_e32 {d}, {d}, {s}, vcc"] * 8),
    ("add_then_cnd_e64", ["v_add_u32_e32 {d}, {d}, {s}\n v_cndmask_b32_e64 {d}, {d}, {s}, s[20:21]"] * 8),
    ("pk_only", [P] * 8),
    ("v_max_u16_then_pk", ["v_max_u16_e32 {d}, {d}, {s}\n " + P] * 8),
    ("v_or3", ["v_or3_b32 {d}, {d}, {s}, {s}"] * 8),
    ("two_v_or", ["v_or_b32_e32 {d}, {d}, {s}\n v_or_b32_e32 {d}, {d}, {s}"] * 8),
    ("lshl_or", ["v_lshl_or_b32 {d}, {d}, 8, {s}"] * 8),
    ("lshlrev_b32_then_or", ["v_lshlrev_b32_e32 {d}, 8, {d}\n v_or_b32_e32 {d}, {d}, {s}"] * 8),
    ("v_max_u32_e32", ["v_max_u32_e32 {d}, {d}, {s}"] * 8),
    ("v_max_f32_e32", ["v_max_f32_e32 {d}, {d}, {s}"] * 8),
    ("v_pk_max_i16", ["v_pk_max_i16 {d}, {d}, {s}"] * 8),
    ("v_cmp_gt_u32_e32", ["v_cmp_gt_u32_e32 vcc, {d}, {s}"] * 8),
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
open(__file__.replace("gen_issue_bench3.py", "issue_bench3.hip"), "w").write("\n".join(src) + "\n")
