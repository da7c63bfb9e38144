// vecmath.hpp -- float3 algebra with a pinned evaluation order (IEEE binary32, no contraction).
//
// Pinned reading of the Metal vector built-ins used by trace.metal:
//   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z        cross = the textbook component formula
//   normalize(v) = v * (1 / sqrt(dot(v,v)))          min(x,y) = y<x ? y : x,  max(x,y) = x<y ? y : x
#pragma once
#include <hip/hip_runtime.h>

namespace cl2 {

struct V3 { float x, y, z; };

// Correctly rounded reciprocal and x/pi at roughly half the cost of the compiler's IEEE divide
// sequence (41 cycles per wave on gfx950, tools/valu_rate.hip).
//   rcp_exact(a): v_rcp_f32 + one FMA Newton step.  Bit-identical to 1.0f/a for EVERY binary32 a whose
//   exponent field is in [1,252] (a and 1/a both normal) -- verified exhaustively over all 2^32 inputs
//   on MI355X (tools/rcp_check.hip; cl2_selftest_exact_math re-runs the proof through these very
//   functions).  Everything else (0, denormals, |a| >= 2^126, inf, NaN) takes the IEEE divide.
//   div_pi(x): q = x*RN(1/pi), then q + RN(1/pi)*fma(-pi, q, x); bit-identical to x / PI_F for every x
//   with exponent field in [30,220] (exhaustive, tools/divpi_check.hip); IEEE divide otherwise.
// Because the results are the correctly rounded ones, every evaluation of the same formulas with IEEE
// division agrees with them bit for bit.
__device__ __forceinline__ float rcp_exact(float a) {
    const unsigned e = (__float_as_uint(a) >> 23) & 0xFFu;
    if (e - 1u < 252u) {
        const float r = __builtin_amdgcn_rcpf(a);
        return __builtin_fmaf(__builtin_fmaf(-a, r, 1.0f), r, r);
    }
    return 1.0f / a;
}
constexpr float PI_CONST = 3.14159265359f;           // trace.metal:4
__device__ __forceinline__ float div_pi(float x) {
    const unsigned e = (__float_as_uint(x) >> 23) & 0xFFu;
    if (e - 30u <= 190u) {
        const float r = 1.0f / PI_CONST;             // constant-folded: RN(1/pi)
        const float q = x * r;
        return __builtin_fmaf(__builtin_fmaf(-PI_CONST, q, x), r, q);
    }
    return x / PI_CONST;
}

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 v3(const float4& f) { return V3{f.x, f.y, f.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 rcp3(V3 a) { return V3{rcp_exact(a.x), rcp_exact(a.y), rcp_exact(a.z)}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float length3(V3 a) { return __builtin_sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) { float inv = rcp_exact(length3(a)); return a * inv; }
__device__ __forceinline__ float min_msl(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float max_msl(float x, float y) { return x < y ? y : x; }
__device__ __forceinline__ float4 f4(V3 a, float w) { return make_float4(a.x, a.y, a.z, w); }

}  // namespace cl2
