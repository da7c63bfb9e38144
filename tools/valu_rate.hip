// valu_rate.hip -- microbenchmark: issue rate of scalar vs packed fp32 VALU ops and of the IEEE
// divide sequence on gfx950.  Decides whether packing two triangle tests per lane (v_pk_*_f32)
// halves the VALU time of the traversal kernels.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 b0 = {a0, a1}, b1 = {a2, a3}, b2 = {a4, a5}, b3 = {a6, a7}, b4 = {a1, a0}, b5 = {a3, a2}, b6 = {a5, a4}, b7 = {a7, a6};
    const float m = 1.0000001f, c = 1e-7f;
    const f2 m2 = {m, m}, c2 = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {        // 16 scalar ops (8 mul + 8 add), independent chains
            a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
            a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
        } else if (MODE == 1) { // 16 packed ops = 32 flops-lanes
            b0 = b0 * m2; b1 = b1 * m2; b2 = b2 * m2; b3 = b3 * m2; b4 = b4 * m2; b5 = b5 * m2; b6 = b6 * m2; b7 = b7 * m2;
            b0 = b0 + c2; b1 = b1 + c2; b2 = b2 + c2; b3 = b3 + c2; b4 = b4 + c2; b5 = b5 + c2; b6 = b6 + c2; b7 = b7 + c2;
        } else {                // 8 IEEE divisions
            a0 = m / a0; a1 = m / a1; a2 = m / a2; a3 = m / a3; a4 = m / a4; a5 = m / a5; a6 = m / a6; a7 = m / a7;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0.x + b1.y + b2.x + b3.y + b4.x + b5.y + b6.x + b7.y;
}

int main() {
    float* d; hipMalloc(&d, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 8192;   // 8 blocks/CU resident, 4 waves each
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double waves = blocks * 4.0, ops = (mode == 2 ? 8.0 : 16.0) * iters;
            double per_simd = waves * ops / 1024.0;        // wave-instructions (or divisions) per SIMD
            printf("mode %d rep %d: %.3f ms  -> %.2f ns per wave-op per SIMD (%.2f cycles @2.4GHz)\n", mode, rep, ms,
                   ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
        }
    }
    return 0;
}
