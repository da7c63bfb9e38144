// vecmath.hpp -- float3 algebra with a pinned evaluation order (IEEE binary32, no contraction).
//
// Pinned reading of the Metal vector built-ins used by trace.metal:
//   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z        cross = the textbook component formula
//   normalize(v) = v * (1 / sqrt(dot(v,v)))          min(x,y) = y<x ? y : x,  max(x,y) = x<y ? y : x
#pragma once
#include <hip/hip_runtime.h>

namespace cl2 {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 v3(const float4& f) { return V3{f.x, f.y, f.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 rcp3(V3 a) { return V3{1.0f / a.x, 1.0f / a.y, 1.0f / a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float length3(V3 a) { return __builtin_sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) { float inv = 1.0f / length3(a); return a * inv; }
__device__ __forceinline__ float min_msl(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float max_msl(float x, float y) { return x < y ? y : x; }
__device__ __forceinline__ float4 f4(V3 a, float w) { return make_float4(a.x, a.y, a.z, w); }

}  // namespace cl2
