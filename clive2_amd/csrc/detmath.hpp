// detmath.hpp -- deterministic float32 elementary functions for the device code.
//
// The reference's kernels call Metal's sin/cos/acos/atan/exp (trace.metal:213-233, :564-567),
// which are only specified to a few ulp.  A path tracer amplifies last-bit differences into
// different paths, so this library fixes ONE definition: classic single-precision argument
// reduction + minimax polynomials (the published Cephes sinf/cosf/asinf/atanf/expf schemes),
// evaluated with IEEE binary32 +,-,*,/ and sqrt in a fixed order.  Built with
// -ffp-contract=off every operation rounds once, so results are reproducible across
// compilers and can be checked exactly against any other IEEE evaluation of the same scheme.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cl2 {

constexpr float DM_FOPI = 1.27323954473516f;
constexpr float DM_DP1 = 0.78515625f;
constexpr float DM_DP2 = 2.4187564849853515625e-4f;
constexpr float DM_DP3 = 3.77489497744594108e-8f;
constexpr float DM_PIO2 = 1.5707963267948966192f;
constexpr float DM_PIO4 = 0.7853981633974483096f;
constexpr float DM_PI = 3.14159265358979323846f;

__device__ __forceinline__ float dm_sin_poly(float x, float z) {
    float y = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x;
    return y + x;
}
__device__ __forceinline__ float dm_cos_poly(float z) {
    float y = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}

// sin and cos of the same argument share the octant reduction.
__device__ __forceinline__ void det_sincosf(float xx, float& s_out, float& c_out) {
    float x = __builtin_fabsf(xx);
    if (!(x <= 8192.0f)) { s_out = x - x; c_out = x - x; return; }
    int j = (int)(DM_FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    bool sneg = xx < 0.0f, cneg = false;
    if (j > 3) { sneg = !sneg; cneg = !cneg; j -= 4; }
    if (j > 1) cneg = !cneg;
    x = ((x - y * DM_DP1) - y * DM_DP2) - y * DM_DP3;
    float z = x * x;
    float ps = dm_sin_poly(x, z), pc = dm_cos_poly(z);
    bool swap = (j == 1 || j == 2);
    float s = swap ? pc : ps;
    float c = swap ? ps : pc;
    s_out = sneg ? -s : s;
    c_out = cneg ? -c : c;
}
__device__ __forceinline__ float det_sinf(float x) { float s, c; det_sincosf(x, s, c); return s; }
__device__ __forceinline__ float det_cosf(float x) { float s, c; det_sincosf(x, s, c); return c; }

__device__ __forceinline__ float det_asinf(float xx) {
    // |x| by comparison, not by clearing the sign bit: asin(-0) must stay -0 like the CPU statement
    bool neg = xx < 0.0f, flag = false;
    float a = neg ? -xx : xx, x, z;
    if (!(a <= 1.0f)) return a - a + (a - a) / (a - a);
    if (a < 1.0e-4f) return neg ? -a : a;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); x = __builtin_sqrtf(z); flag = true; }
    else { x = a; z = x * x; }
    z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z
         + 1.6666752422E-1f) * z * x + x;
    if (flag) { z = z + z; z = DM_PIO2 - z; }
    return neg ? -z : z;
}

__device__ __forceinline__ float det_acosf(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return (x - x) / (x - x);
    if (x > 0.5f) return 2.0f * det_asinf(__builtin_sqrtf(0.5f * (1.0f - x)));
    if (x < -0.5f) return DM_PI - 2.0f * det_asinf(__builtin_sqrtf(0.5f * (1.0f + x)));
    return DM_PIO2 - det_asinf(x);
}

__device__ __forceinline__ float det_atanf(float xx) {
    float x = __builtin_fabsf(xx), y;
    bool neg = xx < 0.0f;
    if (x > 2.414213562373095f) { y = DM_PIO2; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = DM_PIO4; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x);
    return neg ? -y : y;
}

__device__ __forceinline__ float det_expf(float xx) {
    float x = xx;
    if (x != x) return x;
    if (x > 88.0f) return __builtin_inff();
    if (x < -87.0f) return 0.0f;
    float fz = __builtin_floorf(1.44269504088896341f * x + 0.5f);
    x = x - fz * 0.693359375f;
    x = x - fz * -2.12194440e-4f;
    int n = (int)fz;
    float z = x * x;
    z = (((((1.9875691500E-4f * x + 1.3981999507E-3f) * x + 8.3334519073E-3f) * x + 4.1665795894E-2f) * x
          + 1.6666665459E-1f) * x + 5.0000001201E-1f) * z + x + 1.0f;
    return z * __uint_as_float((uint32_t)(n + 127) << 23);
}

}  // namespace cl2
