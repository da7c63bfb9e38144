// bsdf.hpp -- sampling and microfacet terms of the bounce step.
//
// Device-side statement of the scalar helpers of the reference (`src/trace.metal:200-379`):
// orthonormal frame, cosine / uniform hemisphere sampling, GGX normal sampling, dielectric
// Fresnel, GGX D/G, half-vector Jacobians, and the three bounce routines that return the sampled
// direction `wo`, the BSDF*cos value `f` and the forward / reverse solid-angle pdfs (swapped by
// `from_camera`).  Operation order is the reference's, literal by literal (float literals: MSL
// has no double).
#pragma once
#include "vecmath.hpp"
#include "detmath.hpp"

namespace cl2 {

constexpr float PI_F = PI_CONST;         // trace.metal:4
constexpr float DELTA_F = 0.0001f;       // trace.metal:5

// trace.metal:87-93
__device__ __forceinline__ float xorshift_random(uint32_t& s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return (float)s / (float)0xFFFFFFFFu;
}

// trace.metal:200-211
__device__ __forceinline__ void orthonormal(V3 n, V3& x, V3& y) {
    float ax = __builtin_fabsf(n.x), ay = __builtin_fabsf(n.y), az = __builtin_fabsf(n.z);
    V3 v;
    if (ax <= ay && ax <= az) v = v3(1, 0, 0);
    else if (ay <= az) v = v3(0, 1, 0);
    else v = v3(0, 0, 1);
    x = normalize(v - dot(v, n) * n);
    y = normalize(cross(n, x));
}

// trace.metal:213-217
__device__ __forceinline__ V3 random_hemisphere_cosine(V3 xa, V3 ya, V3 za, float rx, float ry) {
    float theta = det_acosf(__builtin_sqrtf(rx));
    float phi = 2 * PI_F * ry;
    float st, ct, sp, cp;
    det_sincosf(theta, st, ct);
    det_sincosf(phi, sp, cp);
    return normalize(((st * cp) * xa + (st * sp) * ya) + ct * za);
}

// trace.metal:219-224
__device__ __forceinline__ V3 random_hemisphere_uniform(V3 xa, V3 ya, V3 za, float rx, float ry) {
    float z = rx;
    float r = __builtin_sqrtf(max_msl(0.0f, 1.0f - z * z));
    float phi = 2 * PI_F * ry;
    float sp, cp;
    det_sincosf(phi, sp, cp);
    return normalize(((r * cp) * xa + (r * sp) * ya) + z * za);
}

// trace.metal:226-233
__device__ __forceinline__ V3 GGX_sample(V3 n, float rx, float ry, float alpha) {
    V3 x, y;
    orthonormal(n, x, y);
    float theta = 2 * PI_F * rx;
    float phi = det_atanf(alpha * __builtin_sqrtf(ry) / __builtin_sqrtf(1.0f - ry));
    float sp, cp, st, ct;
    det_sincosf(phi, sp, cp);
    det_sincosf(theta, st, ct);
    return normalize(((sp * ct) * x + (sp * st) * y) + cp * n);
}

// trace.metal:235-237
__device__ __forceinline__ V3 specular_reflection(V3 i, V3 m) { return normalize((2 * dot(i, m)) * m - i); }

// trace.metal:243-248
__device__ __forceinline__ V3 GGX_transmit(V3 i, V3 m, float ni, float no) {
    float ci = dot(i, m);
    float eta = ni / no;
    float ct = __builtin_sqrtf(1 + eta * (ci * ci - 1));
    return normalize((eta * ci - ct) * m - eta * i);
}

// trace.metal:250-252
__device__ __forceinline__ V3 transmit_half_direction(V3 i, V3 o, float ni, float no) {
    return normalize(no * o + ni * i);
}

// trace.metal:254-264
__device__ __forceinline__ float degreve_fresnel(V3 i, V3 m, float ni, float nt) {
    float ci = __builtin_fabsf(dot(i, m));
    float eta = ni / nt;
    float st2 = eta * eta * (1.0f - ci * ci);
    if (st2 >= 1.0f) return 1.0f;
    float ct = __builtin_sqrtf(1.0f - st2);
    float rpar = (nt * ci - ni * ct) / (nt * ci + ni * ct);
    float rper = (ni * ci - nt * ct) / (ni * ci + nt * ct);
    return 0.5f * (rpar * rpar + rper * rper);
}

// trace.metal:266-271
__device__ __forceinline__ float GGX_G1(V3 v, V3 m, float alpha) {
    float mv = dot(m, v);
    float sin2 = 1.0f - mv * mv;
    float tan2 = sin2 / (mv * mv);
    return 2.0f / (1.0f + __builtin_sqrtf(1.0f + alpha * alpha * tan2));
}

// trace.metal:273-277
__device__ __forceinline__ float GGX_G(V3 i, V3 o, V3 m, V3 n, float alpha) {
    if (dot(i, m) * dot(i, n) <= 0.0f) return 0.0f;
    if (dot(o, m) * dot(o, n) <= 0.0f) return 0.0f;
    return GGX_G1(i, m, alpha) * GGX_G1(o, m, alpha);
}

// trace.metal:279-288
__device__ __forceinline__ float GGX_D(V3 m, V3 n, float alpha) {
    if (alpha == 0.0f) return 1.0f;
    float a2 = alpha * alpha;
    float c = dot(m, n);
    float c2 = c * c;
    float denom = c2 * (a2 - 1.0f) + 1.0f;
    return a2 / (PI_F * denom * denom);
}

// trace.metal:290-292
__device__ __forceinline__ float reflect_jacobian(V3 m, V3 o) { return rcp_exact(4.0f * __builtin_fabsf(dot(m, o))); }

// trace.metal:294-301 (the `m` argument is unused there too)
__device__ __forceinline__ float transmit_jacobian(V3 i, V3 o, float ni, float no) {
    V3 h = transmit_half_direction(i, o, ni, no);
    float ci = dot(i, h);
    float co = dot(o, h);
    float num = no * no * __builtin_fabsf(co);
    float den = (ni * ci + no * co) * (ni * ci + no * co);
    return num / den;
}

// trace.metal:303-309
__device__ __forceinline__ float GGX_BRDF_reflect(V3 i, V3 o, V3 m, V3 n, float ni, float no, float alpha) {
    float D = GGX_D(m, n, alpha);
    float G = GGX_G(i, o, m, n, alpha);
    float F = degreve_fresnel(i, m, ni, no);
    return (D * G * F) / (4.0f * __builtin_fabsf(dot(i, m)));
}

// trace.metal:311-328
__device__ __forceinline__ float GGX_BRDF_transmit(V3 i, V3 o, V3 m, V3 n, float ni, float no, float alpha) {
    V3 h = transmit_half_direction(i, o, ni, no);
    float D = GGX_D(m, n, alpha);
    float G = GGX_G(i, o, m, n, alpha);
    float F = degreve_fresnel(i, m, ni, no);
    float im = dot(i, h), om = dot(o, h), in = dot(i, n), on = dot(o, n);
    float coeff = (im * om) / (in * on);
    float num = no * no * D * G * (1.0f - F);
    float den = (ni * im + no * om) * (ni * im + no * om);
    return coeff * num / den;
}

struct Bounce { V3 wo; float f, c_p, l_p; };

// trace.metal:334-346
__device__ __forceinline__ Bounce diffuse_bounce(V3 wi, V3 n, bool from_camera, float rx, float ry) {
    V3 x, y;
    orthonormal(n, x, y);
    Bounce b;
    b.wo = random_hemisphere_cosine(x, y, n, rx, ry);
    b.f = div_pi(__builtin_fabsf(dot(n, b.wo)));
    float p_o = div_pi(__builtin_fabsf(dot(n, b.wo)));
    float p_i = div_pi(__builtin_fabsf(dot(n, wi)));
    b.c_p = from_camera ? p_o : p_i;
    b.l_p = from_camera ? p_i : p_o;
    return b;
}

// trace.metal:348-362
__device__ __forceinline__ Bounce reflect_bounce(V3 wi, V3 n, V3 m, float ni, float no, float alpha, bool from_camera) {
    Bounce b;
    b.wo = specular_reflection(wi, m);
    b.f = GGX_BRDF_reflect(wi, b.wo, m, n, ni, no, alpha);
    float pf = degreve_fresnel(wi, m, ni, no);
    float pm = __builtin_fabsf(dot(m, n)) * GGX_D(m, n, alpha);
    float p_o = pf * pm * reflect_jacobian(m, b.wo);
    float p_i = pf * pm * reflect_jacobian(m, wi);
    b.c_p = from_camera ? p_o : p_i;
    b.l_p = from_camera ? p_i : p_o;
    return b;
}

// trace.metal:364-379
__device__ __forceinline__ Bounce transmit_bounce(V3 wi, V3 n, V3 m, float ni, float no, float alpha, bool from_camera) {
    Bounce b;
    b.wo = GGX_transmit(wi, m, ni, no);
    b.f = GGX_BRDF_transmit(wi, b.wo, m, n, ni, no, alpha);
    float pf = 1.0f - degreve_fresnel(wi, m, ni, no);
    float pm = __builtin_fabsf(dot(m, n)) * GGX_D(m, n, alpha);
    float p_fwd = pf * pm * transmit_jacobian(wi, b.wo, ni, no);
    float p_rev = pf * pm * transmit_jacobian(b.wo, wi, no, ni);
    b.c_p = from_camera ? p_fwd : p_rev;
    b.l_p = from_camera ? p_rev : p_fwd;
    return b;
}

}  // namespace cl2
