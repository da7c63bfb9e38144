// divpi_check.hip -- exhaustive check of x / PI_F computed as q = x*r; q' = fma(fma(-PI,q,x), r, q) with r = RN(1/PI)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void check(unsigned long long* mismatches, unsigned* first_bad, float PI, float R) {
    unsigned long long bad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)i;
        const unsigned e = (bits >> 23) & 0xFF;
        if (e < 30 || e > 220) continue;
        const float x = __uint_as_float(bits);
        const float ref = x / PI;
        const float q = x * R;
        const float got = __builtin_fmaf(__builtin_fmaf(-PI, q, x), R, q);
        if (__float_as_uint(ref) != __float_as_uint(got)) { bad++; atomicMin(first_bad, bits); }
    }
    if (bad) atomicAdd(mismatches, bad);
}
int main() {
    unsigned long long* d_m; unsigned* d_f;
    (void)hipMalloc(&d_m, 8); (void)hipMalloc(&d_f, 4);
    unsigned long long z = 0; unsigned f = 0xFFFFFFFFu;
    (void)hipMemcpy(d_m, &z, 8, hipMemcpyHostToDevice); (void)hipMemcpy(d_f, &f, 4, hipMemcpyHostToDevice);
    const float PI = 3.14159265359f;
    const float R = 1.0f / PI;
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d_m, d_f, PI, R);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&z, d_m, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, d_f, 4, hipMemcpyDeviceToHost);
    printf("x/PI via fma correction: %llu mismatches (first 0x%08x)\n", z, f);
    return 0;
}
