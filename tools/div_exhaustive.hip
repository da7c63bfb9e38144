// div_exhaustive.hip -- PROOF by enumeration that div_exact(a,b) == a/b (IEEE, round to nearest even) for all
// normal operands away from over/underflow: rounding depends only on the two 24-bit significands, so all
// 2^23 x 2^23 significand pairs at one fixed exponent pair cover every case.
//   usage: div_exhaustive <first_b_chunk> <n_chunks>   (chunk = 4096 denominators; 2048 chunks in all)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ float rcp_exact(float a) {
    const unsigned e = (__float_as_uint(a) >> 23) & 0xFFu;
    if (e - 1u < 252u) { const float r = __builtin_amdgcn_rcpf(a); return __builtin_fmaf(__builtin_fmaf(-a, r, 1.0f), r, r); }
    return 1.0f / a;
}
__device__ __forceinline__ float div_fast(float a, float b) {
    const float y = rcp_exact(b);
    const float q = a * y;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
}
__global__ void check(unsigned b_base, unsigned long long* bad, unsigned* first) {
    const unsigned mb = b_base + blockIdx.y;
    const float b = __uint_as_float((127u << 23) | mb);
    unsigned long long nb = 0;
    for (unsigned m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
        const float a = __uint_as_float((127u << 23) | m);
        if (__float_as_uint(a / b) != __float_as_uint(div_fast(a, b))) { nb++; atomicMin(first, mb); }
    }
    if (nb) atomicAdd(bad, nb);
}
int main(int argc, char** argv) {
    const unsigned first_chunk = argc > 1 ? atoi(argv[1]) : 0, n_chunks = argc > 2 ? atoi(argv[2]) : 2048;
    unsigned long long* d_b; unsigned* d_f;
    (void)hipMalloc(&d_b, 8); (void)hipMalloc(&d_f, 4);
    unsigned long long z = 0; unsigned f = 0xFFFFFFFFu;
    (void)hipMemcpy(d_b, &z, 8, hipMemcpyHostToDevice); (void)hipMemcpy(d_f, &f, 4, hipMemcpyHostToDevice);
    for (unsigned c = first_chunk; c < first_chunk + n_chunks && c < 2048; c++) {
        hipLaunchKernelGGL(check, dim3(32, 4096), dim3(256), 0, 0, c * 4096u, d_b, d_f);
        if ((c - first_chunk) % 64 == 63 || c + 1 == first_chunk + n_chunks) {
            (void)hipDeviceSynchronize();
            unsigned long long b; (void)hipMemcpy(&b, d_b, 8, hipMemcpyDeviceToHost);
            printf("denominator significands [%u, %u): %llu mismatches so far\n", first_chunk * 4096u, (c + 1) * 4096u, b); fflush(stdout);
        }
    }
    (void)hipDeviceSynchronize();
    unsigned long long b; (void)hipMemcpy(&b, d_b, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, d_f, 4, hipMemcpyDeviceToHost);
    printf("RESULT chunks [%u,%u): %llu mismatches over %llu significand pairs (first bad denominator significand 0x%06x)\n",
           first_chunk, first_chunk + n_chunks, b, (unsigned long long)n_chunks * 4096ull * (1ull << 23), f);
    return b ? 1 : 0;
}
