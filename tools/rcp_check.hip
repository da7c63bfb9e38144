// rcp_check.hip -- exhaustive check: for which refinement of v_rcp_f32 is the result bit-identical to the
// IEEE division 1.0f/a for EVERY binary32 a?  (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int STEPS>
__device__ float rcp_refined(float a) {
    float r = __builtin_amdgcn_rcpf(a);
    for (int i = 0; i < STEPS; i++) {
        const float e = __builtin_fmaf(-a, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    return r;
}

template <int STEPS>
__global__ void check(unsigned long long* mismatches, unsigned* first_bad, unsigned lo_exp, unsigned hi_exp) {
    const unsigned long long total = 1ull << 32;
    unsigned long long bad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)i;
        const unsigned e = (bits >> 23) & 0xFF;
        if (e < lo_exp || e > hi_exp) continue;
        const float a = __uint_as_float(bits);
        const float ref = 1.0f / a;
        const float got = rcp_refined<STEPS>(a);
        if (__float_as_uint(ref) != __float_as_uint(got)) { bad++; atomicMin(first_bad, bits); }
    }
    if (bad) atomicAdd(mismatches, bad);
}

int main() {
    unsigned long long* d_m; unsigned* d_f;
    (void)hipMalloc(&d_m, 8); (void)hipMalloc(&d_f, 4);
    for (int steps = 0; steps <= 3; steps++) {
        for (int range = 0; range < 2; range++) {
            const unsigned lo = range == 0 ? 1 : 27, hi = range == 0 ? 254 : 227;
            unsigned long long z = 0; unsigned f = 0xFFFFFFFFu;
            (void)hipMemcpy(d_m, &z, 8, hipMemcpyHostToDevice); (void)hipMemcpy(d_f, &f, 4, hipMemcpyHostToDevice);
            if (steps == 0) hipLaunchKernelGGL(check<0>, dim3(4096), dim3(256), 0, 0, d_m, d_f, lo, hi);
            if (steps == 1) hipLaunchKernelGGL(check<1>, dim3(4096), dim3(256), 0, 0, d_m, d_f, lo, hi);
            if (steps == 2) hipLaunchKernelGGL(check<2>, dim3(4096), dim3(256), 0, 0, d_m, d_f, lo, hi);
            if (steps == 3) hipLaunchKernelGGL(check<3>, dim3(4096), dim3(256), 0, 0, d_m, d_f, lo, hi);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&z, d_m, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, d_f, 4, hipMemcpyDeviceToHost);
            printf("steps %d exponent range [%u,%u]: %llu mismatches (first bad bits 0x%08x)\n", steps, lo, hi, z, f);
        }
    }
    return 0;
}
