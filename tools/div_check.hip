// div_check.hip -- a/b via correctly rounded reciprocal + one FMA correction vs the IEEE divide.
// y = rcp_exact(b) (proven = RN(1/b)); q = a*y; r = fma(-b,q,a); q' = fma(r,y,q).
// Markstein's theorem says q' = RN(a/b) when y = RN(1/b) and nothing over/underflows; this tool hammers
// it with random and structured operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float rcp_exact(float a) {
    const unsigned e = (__float_as_uint(a) >> 23) & 0xFFu;
    if (e - 1u < 252u) { const float r = __builtin_amdgcn_rcpf(a); return __builtin_fmaf(__builtin_fmaf(-a, r, 1.0f), r, r); }
    return 1.0f / a;
}
__device__ __forceinline__ bool in_range(float a, float b) {
    const int ea = (__float_as_uint(a) >> 23) & 0xFF, eb = (__float_as_uint(b) >> 23) & 0xFF;
    return ea >= 40 && ea <= 214 && eb >= 40 && eb <= 214 && (ea - eb) < 80 && (eb - ea) < 80;
}
__device__ __forceinline__ float div_fast(float a, float b) {
    const float y = rcp_exact(b);
    const float q = a * y;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
}
__device__ __forceinline__ uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

// mode 0: random bit patterns; mode 1: b mantissa from a small structured set x all a mantissas of one binade
__global__ void check(int mode, unsigned long long iters, unsigned long long* bad, unsigned long long* tested, unsigned* first) {
    uint64_t s = 0x1234567ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    unsigned long long nb = 0, nt = 0;
    const unsigned special[16] = {0x000000, 0x000001, 0x7FFFFF, 0x7FFFFE, 0x400000, 0x3FFFFF, 0x400001, 0x555555,
                                  0x2AAAAA, 0x7FFFF0, 0x00000F, 0x490FDB, 0x600000, 0x200000, 0x7F0000, 0x0000FF};
    for (unsigned long long i = 0; i < iters; i++) {
        uint64_t z = splitmix(s);
        float a, b;
        if (mode == 0) { a = __uint_as_float((unsigned)z); b = __uint_as_float((unsigned)(z >> 32)); }
        else {
            const unsigned mb = special[(z >> 60) & 15] ^ (((z >> 40) & 1) ? 0 : (unsigned)((z >> 41) & 0x7));   // near-special mantissas
            b = __uint_as_float((127u + (unsigned)((z >> 50) & 7)) << 23 | (mb & 0x7FFFFF));
            a = __uint_as_float((127u << 23) | (unsigned)(z & 0x7FFFFF));
        }
        if (!in_range(a, b)) continue;
        nt++;
        const float ref = a / b, got = div_fast(a, b);
        if (__float_as_uint(ref) != __float_as_uint(got)) { nb++; atomicMin(first, __float_as_uint(b)); }
    }
    atomicAdd(bad, nb); atomicAdd(tested, nt);
}
// mode 2: exhaustive over all mantissa pairs would be 2^46: instead exhaustive a-mantissa (2^23) for 4096 b's
__global__ void check_grid(unsigned long long* bad, unsigned long long* tested, unsigned* first, unsigned b_seed) {
    const unsigned bi = blockIdx.y;                       // which b
    uint64_t s = 0xABCDEFull * (bi + 1) + b_seed;
    const unsigned mb = (unsigned)splitmix(s) & 0x7FFFFF;
    const float b = __uint_as_float((127u << 23) | mb);
    unsigned long long nb = 0, nt = 0;
    for (unsigned m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
        const float a = __uint_as_float((127u << 23) | m);
        nt++;
        if (__float_as_uint(a / b) != __float_as_uint(div_fast(a, b))) { nb++; atomicMin(first, mb); }
    }
    atomicAdd(bad, nb); atomicAdd(tested, nt);
}
int main() {
    unsigned long long *d_b, *d_t; unsigned* d_f;
    (void)hipMalloc(&d_b, 8); (void)hipMalloc(&d_t, 8); (void)hipMalloc(&d_f, 4);
    for (int mode = 0; mode < 3; mode++) {
        unsigned long long z = 0; unsigned f = 0xFFFFFFFFu;
        (void)hipMemcpy(d_b, &z, 8, hipMemcpyHostToDevice); (void)hipMemcpy(d_t, &z, 8, hipMemcpyHostToDevice); (void)hipMemcpy(d_f, &f, 4, hipMemcpyHostToDevice);
        if (mode < 2) hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, mode, 16384ull, d_b, d_t, d_f);
        else hipLaunchKernelGGL(check_grid, dim3(64, 4096), dim3(256), 0, 0, d_b, d_t, d_f, 7u);
        (void)hipDeviceSynchronize();
        unsigned long long b, t;
        (void)hipMemcpy(&b, d_b, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, d_f, 4, hipMemcpyDeviceToHost);
        printf("mode %d: %llu mismatches in %llu tested (first b bits 0x%08x)\n", mode, b, t, f);
    }
    return 0;
}
