// issue_mix.hip -- microbenchmark: what scalar instructions, exec-mask regions and branches cost next to a stream of fp32
// VALU instructions on gfx950, at 8 waves per SIMD.  Every variant runs the same 32 v_fma_f32 per iteration (independent
// chains); the variants add scalar ALU work, s_and_saveexec / s_or exec pairs, or uniform branches.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_mix.hip -o tools/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

#define V8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
           "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {                 // 32 VALU
            asm volatile(V8 V8 V8 V8 OPS);
        } else if (MODE == 1) {          // 32 VALU + 16 SALU (two after every four VALU)
            asm volatile(V8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         V8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         V8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         V8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         OPS : "s20", "s21", "s22", "s23", "scc");
        } else if (MODE == 2) {          // 32 VALU + 32 SALU
            asm volatile(V8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         V8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         V8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         V8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         OPS : "s20", "s21", "s22", "s23", "scc");
        } else if (MODE == 3) {          // 32 VALU + 4 exec-mask regions (saveexec ... or exec), no branch
            asm volatile("s_and_saveexec_b64 s[20:21], exec\n" V8 "s_or_b64 exec, exec, s[20:21]\n"
                         "s_and_saveexec_b64 s[20:21], exec\n" V8 "s_or_b64 exec, exec, s[20:21]\n"
                         "s_and_saveexec_b64 s[20:21], exec\n" V8 "s_or_b64 exec, exec, s[20:21]\n"
                         "s_and_saveexec_b64 s[20:21], exec\n" V8 "s_or_b64 exec, exec, s[20:21]\n"
                         OPS : "s20", "s21", "scc");
        } else if (MODE == 4) {          // 32 VALU + 4 regions each with a NOT-taken s_cbranch_execz
            asm volatile("s_and_saveexec_b64 s[20:21], exec\n s_cbranch_execz 1f\n" V8 "1: s_or_b64 exec, exec, s[20:21]\n"
                         "s_and_saveexec_b64 s[20:21], exec\n s_cbranch_execz 2f\n" V8 "2: s_or_b64 exec, exec, s[20:21]\n"
                         "s_and_saveexec_b64 s[20:21], exec\n s_cbranch_execz 3f\n" V8 "3: s_or_b64 exec, exec, s[20:21]\n"
                         "s_and_saveexec_b64 s[20:21], exec\n s_cbranch_execz 4f\n" V8 "4: s_or_b64 exec, exec, s[20:21]\n"
                         OPS : "s20", "s21", "scc");
        } else if (MODE == 5) {          // 32 VALU + 4 TAKEN uniform branches (s_branch over nothing)
            asm volatile(V8 "s_branch 1f\n s_nop 0\n 1:\n" V8 "s_branch 2f\n s_nop 0\n 2:\n" V8 "s_branch 3f\n s_nop 0\n 3:\n" V8 "s_branch 4f\n s_nop 0\n 4:\n" OPS);
        } else if (MODE == 6) {          // 32 VALU + 8 v_cmp (VOPC -> vcc) + 8 s_and on the masks
            asm volatile(V8 "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cmp_lt_f32 vcc, %2, %3\n s_and_b64 s[20:21], vcc, exec\n"
                         V8 "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cmp_lt_f32 vcc, %2, %3\n s_and_b64 s[20:21], vcc, exec\n"
                         V8 "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cmp_lt_f32 vcc, %2, %3\n s_and_b64 s[20:21], vcc, exec\n"
                         V8 "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cmp_lt_f32 vcc, %2, %3\n s_and_b64 s[20:21], vcc, exec\n"
                         OPS : "s20", "s21", "vcc", "scc");
        } else if (MODE == 7) {          // 32 VALU + 12 ds_read_b128 of one address (broadcast), waited once
            asm volatile(V8 "ds_read_b128 v[40:43], %10\n ds_read_b128 v[44:47], %10 offset:16\n ds_read_b128 v[48:51], %10 offset:32\n"
                         V8 "ds_read_b128 v[40:43], %10\n ds_read_b128 v[44:47], %10 offset:16\n ds_read_b128 v[48:51], %10 offset:32\n"
                         V8 "ds_read_b128 v[40:43], %10\n ds_read_b128 v[44:47], %10 offset:16\n ds_read_b128 v[48:51], %10 offset:32\n"
                         V8 "ds_read_b128 v[40:43], %10\n ds_read_b128 v[44:47], %10 offset:16\n ds_read_b128 v[48:51], %10 offset:32\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "v"(0)
                         : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
        } else if (MODE == 8) {          // same with ds_read_b96
            asm volatile(V8 "ds_read_b96 v[40:42], %10\n ds_read_b96 v[44:46], %10 offset:16\n ds_read_b96 v[48:50], %10 offset:32\n"
                         V8 "ds_read_b96 v[40:42], %10\n ds_read_b96 v[44:46], %10 offset:16\n ds_read_b96 v[48:50], %10 offset:32\n"
                         V8 "ds_read_b96 v[40:42], %10\n ds_read_b96 v[44:46], %10 offset:16\n ds_read_b96 v[48:50], %10 offset:32\n"
                         V8 "ds_read_b96 v[40:42], %10\n ds_read_b96 v[44:46], %10 offset:16\n ds_read_b96 v[48:50], %10 offset:32\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "v"(0)
                         : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(float* d, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048, blocks = 8192;   // 8 blocks/CU resident, 4 waves each = 8 waves per SIMD
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 64, 0, d, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd_iters = blocks * 4.0 * iters / 1024.0;
        if (rep) printf("mode %d  %-62s %8.3f ms  %7.1f cycles per iteration per SIMD (@2.4 GHz; 32 VALU alone = 64 at 2 cycles)\n", MODE, what, ms,
                        ms * 1e-3 * 2.4e9 / per_simd_iters);
    }
}

int main() {
    float* d; hipMalloc(&d, 256 * 8192 * 4);
    run<0>(d, "32 v_fma");
    run<1>(d, "32 v_fma + 16 s_add");
    run<2>(d, "32 v_fma + 32 s_add");
    run<3>(d, "32 v_fma + 4 x (s_and_saveexec, s_or exec)");
    run<4>(d, "32 v_fma + 4 x (s_and_saveexec, s_cbranch_execz not taken, s_or)");
    run<5>(d, "32 v_fma + 4 taken s_branch");
    run<6>(d, "32 v_fma + 8 v_cmp + 8 s_and");
    run<7>(d, "32 v_fma + 12 ds_read_b128 (broadcast) + 1 wait");
    run<8>(d, "32 v_fma + 12 ds_read_b96 (broadcast) + 1 wait");
    return 0;
}
