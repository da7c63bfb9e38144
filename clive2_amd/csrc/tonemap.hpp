// tonemap.hpp -- the log-average tone map of `camera.py:73-82` on the device, for the three pictures of
// `renderer.py:293-316` (`image`, `unweighted_image`, `unidirectional_image`), straight from the planar accumulators.
//
// The reference computes (numpy): luma = sum_c(image_c * tone_vector_c) in float64, Lw = exp(sum(log(0.1 + luma)) / (H*W)),
// result = image * exposure / Lw, out = uint8(255 * result / (result + white_point^2)).  The dtypes follow numpy's
// promotion rules, reproduced here: `image` and `unweighted_image` are float32 pictures, so `image * exposure` is a float32
// product which the division by the float64 scalar Lw then widens; `unidirectional_image` divides a float32 array by an
// int32 array, which makes the whole chain float64.  Everything is a deterministic function of the accumulators (fixed
// reduction tree, no atomics).  What is NOT reproduced bit for bit is the float64 sum over the pixels -- numpy adds
// pairwise, this adds per thread, per wave, per workgroup -- and libm's log: the sum, and with it Lw, can differ from
// numpy's in its last bits, which moves an output byte only where 255*x/(x+w) lies within ~1e-13 of an integer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cl2 {

constexpr int TONE_BLOCKS = 1024;

// np.nan_to_num(x, neginf=0, posinf=0)
__device__ __forceinline__ float tone_scrub(float x) { return (__builtin_fabsf(x) < __builtin_inff()) ? x : 0.0f; }
__device__ __forceinline__ double tone_scrub(double x) { return (__builtin_fabs(x) < __builtin_inf()) ? x : 0.0; }

// One pixel of picture `which` (0 image, 1 unweighted_image, 2 unidirectional_image), channel order b, g, r:
//   base[c]  the picture's value as the float64 that `image * tone_vector` sees
//   pre[c]   `image * exposure` widened to float64 (a float32 product for the float32 pictures)
__device__ __forceinline__ void tone_pixel(const float* __restrict__ acc, size_t B, size_t p, int which, double exposure,
                                           double (&base)[3], double (&pre)[3]) {
    if (which == 2) {
        const double cnt = (double)acc[7 * B + p];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            base[c] = tone_scrub((double)acc[(4 + c) * B + p] / cnt);
            pre[c] = base[c] * exposure;
        }
    } else {
        const float w = acc[3 * B + p];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = acc[(size_t)c * B + p];
            const float f = tone_scrub(which == 0 ? v / w : v);
            base[c] = (double)f;
            pre[c] = (double)(f * (float)exposure);
        }
    }
}

// sum over the pixels of log(0.1 + luma): per-thread sums, wave shuffles, one partial per workgroup
__global__ __launch_bounds__(256) void k_tone_logsum(const float* __restrict__ acc, int B, int which, double* __restrict__ partial) {
    __shared__ double s_wave[4];
    double sum = 0.0;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)B; p += (size_t)gridDim.x * 256) {
        double base[3], pre[3];
        tone_pixel(acc, (size_t)B, p, which, 1.0, base, pre);
        const double luma = (base[0] * 0.0722 + base[1] * 0.7152) + base[2] * 0.2126;     // np.sum(image * tone_vector, axis=2)
        sum += log(0.1 + luma);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
}

__global__ __launch_bounds__(256) void k_tone_logsum_final(const double* __restrict__ partial, int n, double* __restrict__ out) {
    __shared__ double s_wave[4];
    double sum = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) sum += partial[i];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) *out = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
}

// (255 * result / (result + white_point^2)).astype(np.uint8) with result = image * exposure / Lw; the cast is C's
// float64 -> integer truncation as numpy performs it on x86-64 (through int32: out-of-range and NaN give 0)
__global__ __launch_bounds__(256) void k_tone_apply(const float* __restrict__ acc, int B, int which, double exposure, double wp2, double Lw,
                                                    uint8_t* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)B) return;
    double base[3], pre[3];
    tone_pixel(acc, (size_t)B, p, which, exposure, base, pre);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double result = pre[c] / Lw;
        const double v = 255.0 * result / (result + wp2);
        const int iv = (v > -2147483648.0 && v < 2147483648.0) ? (int)v : 0;
        out[3 * p + c] = (uint8_t)(iv & 0xFF);
    }
}

}  // namespace cl2
