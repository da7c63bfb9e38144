/* bdpt_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's device code.
 *
 * This file is the parity ORACLE for the MI355X path tracer: a plain-C, IEEE binary32,
 * statement-by-statement restatement of the eight Metal kernels of the reference
 * (`/root/reference/src/trace.metal`), operating on the reference's own AoS records
 * (`src/struct_types.py`).  Each function cites the lines it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (clive2_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned by the reference" -- the reference ships no tests, golden
 * vectors or runnable device code.  The oracle is pinned instead by
 * (a) RNG known-answer vectors computed from trace.metal:87-93, (b) analytic cases
 * (tests/test_oracle_*.py), (c) an independent numpy restatement of the ray generators, the
 * hemisphere samplers, detmath and traverse_bvh (oracle/np_kernels.py), (d) the reference's own BDPT-vs-unidirectional
 * self-consistency check.
 *
 * Pinned interpretation of Metal semantics (all float32, no FMA contraction):
 *   dot(a,b)        = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   cross(a,b)      = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 *   length(v)       = sqrt(dot(v,v));  normalize(v) = v * (1 / length(v))
 *   min(x,y)        = y < x ? y : x;   max(x,y) = x < y ? y : x   (MSL spec wording)
 *   sin/cos/acos/atan/exp = detmath.h
 *   unsuffixed literals are float (MSL has no double)
 * Undefined behaviour of the reference is DEFINED here:
 *   Q1 light index clamped to count-1; Q3/Q4 Path / new_ray / next_ray zero-initialised;
 *   Q5 p_ratios/p_values zero-initialised; Q7 round() = half away from zero, pixel index
 *   may exceed the frame (dropped by the host glue); NaN->int conversions never occur
 *   because the guarded values are finite.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "detmath.h"

#ifdef ORACLE_LIBM
/* Cross-check build: libm transcendentals instead of detmath (statistical comparison only). */
#define SINF sinf
#define COSF cosf
#define ACOSF acosf
#define ATANF atanf
#define EXPF expf
#else
#define SINF det_sinf
#define COSF det_cosf
#define ACOSF det_acosf
#define ATANF det_atanf
#define EXPF det_expf
#endif

#define PI 3.14159265359f            /* trace.metal:4 */
#define DELTA 0.0001f                /* trace.metal:5 */
#define BOUNCES 6                    /* trace.metal:407 */

/* ---- records: byte-identical to trace.metal:7-85 / struct_types.py ---- */
typedef struct { float x, y, z, w; } f3;   /* float3 padded to 16 B; .w is padding */

typedef struct {
    f3 origin, direction, inv_direction, color, normal;
    int32_t material, triangle;
    float c_importance, l_importance, tot_importance;
    int32_t hit_light, from_camera, hit_camera, pixel_idx;
    int32_t pad[3];
} Ray;                                        /* 128 B */

typedef struct { float weights[3][3]; float pad0[3]; f3 total_contribution; float contrib_weight_sum; float pad1[15]; } WeightAggregator; /* 128 B stride */

typedef struct { Ray rays[8]; int32_t length, from_camera, pad[2]; } Path;          /* 1040 B */
typedef struct { f3 min, max; int32_t left, right, pad[2]; } Box;                   /* 48 B */
typedef struct { f3 v0, v1, v2, n0, n1, n2, normal; int32_t material, is_light, is_camera, pad; } Triangle; /* 128 B */
typedef struct { f3 color, emission; int32_t type; float alpha, ior; int32_t transmissive; } Material;     /* 48 B */
typedef struct { f3 center, focal_point, direction, dx, dy; int32_t pixel_width, pixel_height;
                 float phys_width, phys_height, h_fov, v_fov; int32_t pad[2]; } Camera;                     /* 112 B */

_Static_assert(sizeof(Ray) == 128, "Ray");
_Static_assert(sizeof(Path) == 1040, "Path");
_Static_assert(sizeof(Box) == 48, "Box");
_Static_assert(sizeof(Triangle) == 128, "Triangle");
_Static_assert(sizeof(Material) == 48, "Material");
_Static_assert(sizeof(Camera) == 112, "Camera");
_Static_assert(sizeof(WeightAggregator) == 128, "WeightAggregator");

/* counters: [0] rays (traverse_bvh calls), [1] box tests, [2] triangle tests */
typedef struct { uint64_t rays, box_tests, tri_tests; } Counters;

int orc_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(Ray); case 1: return sizeof(Path); case 2: return sizeof(Box);
        case 3: return sizeof(Triangle); case 4: return sizeof(Material); case 5: return sizeof(Camera);
        case 6: return sizeof(WeightAggregator);
    }
    return -1;
}

/* ---- float3 algebra with pinned evaluation order ---- */
static inline f3 V3(float x, float y, float z) { f3 r = {x, y, z, 0.0f}; return r; }
static inline f3 vadd(f3 a, f3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 vsub(f3 a, f3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 vmul(f3 a, f3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 vscale(f3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }      /* a * s and s * a */
static inline f3 vdivs(f3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline f3 vneg(f3 a) { return V3(-a.x, -a.y, -a.z); }
static inline f3 vrcp(f3 a) { return V3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
static inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross(f3 a, f3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(f3 a) { return sqrtf(dot(a, a)); }
static inline f3 normalize(f3 a) { float inv = 1.0f / length3(a); return vscale(a, inv); }
static inline float fmin_msl(float x, float y) { return y < x ? y : x; }
static inline float fmax_msl(float x, float y) { return x < y ? y : x; }

/* trace.metal:87-93 */
static inline float xorshift_random(uint32_t *seed) {
    uint32_t s = *seed;
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    *seed = s;
    return (float)s / (float)0xFFFFFFFFu;
}

float orc_xorshift(uint32_t *seed) { return xorshift_random(seed); }

/* trace.metal:106-115.  t is in/out: in = current best_t, out = entry distance. */
static inline int ray_box_intersect(const Ray *ray, const Box *box, float *t) {
    f3 t0s = vmul(vsub(box->min, ray->origin), ray->inv_direction);
    f3 t1s = vmul(vsub(box->max, ray->origin), ray->inv_direction);
    f3 tmin = V3(fmin_msl(t0s.x, t1s.x), fmin_msl(t0s.y, t1s.y), fmin_msl(t0s.z, t1s.z));
    f3 tmax = V3(fmax_msl(t0s.x, t1s.x), fmax_msl(t0s.y, t1s.y), fmax_msl(t0s.z, t1s.z));
    float tmin_final = fmax_msl(fmax_msl(tmin.x, tmin.y), fmax_msl(tmin.z, 0.0f));
    float tmax_final = fmin_msl(fmin_msl(tmax.x, tmax.y), fmin_msl(tmax.z, *t));
    *t = tmin_final;
    return tmin_final <= tmax_final;
}

/* trace.metal:117-142 (Moller-Trumbore; no parallel-ray guard: f may be +-inf/NaN) */
static inline int ray_triangle_intersect(const Ray *ray, const Triangle *tr, float *t_out, float *u, float *v) {
    f3 edge1 = vsub(tr->v1, tr->v0);
    f3 edge2 = vsub(tr->v2, tr->v0);
    f3 h = cross(ray->direction, edge2);
    float a = dot(edge1, h);
    float f = 1.0f / a;
    f3 s = vsub(ray->origin, tr->v0);
    *u = f * dot(s, h);
    if (*u < 0 || *u > 1) return 0;
    f3 q = cross(s, edge1);
    *v = f * dot(ray->direction, q);
    if (*v < 0 || *u + *v > 1) return 0;
    float t = f * dot(edge2, q);
    if (t > DELTA) { *t_out = t; return 1; }
    return 0;
}

/* trace.metal:144-176: 64-entry stack, unordered, right child popped first. */
static void traverse_bvh(const Ray *ray, const Box *boxes, const Triangle *triangles,
                         int *best_i, float *best_t, float *u_out, float *v_out, Counters *cnt) {
    int stack[64];
    int stack_ptr = 0;
    stack[stack_ptr++] = 0;
    cnt->rays++;
    while (stack_ptr > 0 && stack_ptr < 64) {
        int box_id = stack[--stack_ptr];
        Box box = boxes[box_id];
        float t = INFINITY;
        float u, v;
        cnt->box_tests++;
        /* NB the reference passes t = INFINITY into ray_box_intersect (trace.metal:153-155), so the
         * slab clip is against +inf, and the prune is the separate `t < best_t` test. */
        int hit = ray_box_intersect(ray, &box, &t);
        if (hit && t < *best_t) {
            if (box.right == 0) {
                stack[stack_ptr++] = box.left;
                stack[stack_ptr++] = box.left + 1;
            } else {
                for (int i = box.left; i < box.right; i++) {
                    t = INFINITY;
                    cnt->tri_tests++;
                    hit = ray_triangle_intersect(ray, &triangles[i], &t, &u, &v);
                    if (hit && t < *best_t) {
                        *best_i = i;
                        *best_t = t;
                        *u_out = u;
                        *v_out = v;
                    }
                }
            }
        }
    }
}

/* trace.metal:178-197 */
static int visibility_test(const Ray *a, const Ray *b, const Box *boxes, const Triangle *triangles, Counters *cnt) {
    Ray test_ray;
    memset(&test_ray, 0, sizeof test_ray);
    test_ray.origin = a->origin;
    f3 direction = normalize(vsub(b->origin, a->origin));
    test_ray.direction = direction;
    test_ray.inv_direction = vrcp(direction);
    test_ray.triangle = a->triangle;
    int best_i = -1;
    float best_t = INFINITY, u = 0, v = 0;
    traverse_bvh(&test_ray, boxes, triangles, &best_i, &best_t, &u, &v, cnt);
    if (best_i == -1) return 0;
    if (best_i == a->triangle) return 0;
    if (best_i == b->triangle) return 1;
    return 0;
}

/* trace.metal:200-211 */
static inline void orthonormal(f3 n, f3 *x, f3 *y) {
    f3 v;
    if (fabsf(n.x) <= fabsf(n.y) && fabsf(n.x) <= fabsf(n.z)) v = V3(1, 0, 0);
    else if (fabsf(n.y) <= fabsf(n.z)) v = V3(0, 1, 0);
    else v = V3(0, 0, 1);
    *x = normalize(vsub(v, vscale(n, dot(v, n))));
    *y = normalize(cross(n, *x));
}

/* trace.metal:213-217 */
static inline f3 random_hemisphere_cosine(f3 x_axis, f3 y_axis, f3 z_axis, float rx, float ry) {
    float theta = ACOSF(sqrtf(rx));
    float phi = 2 * PI * ry;
    float st = SINF(theta), ct = COSF(theta), sp = SINF(phi), cp = COSF(phi);
    return normalize(vadd(vadd(vscale(x_axis, st * cp), vscale(y_axis, st * sp)), vscale(z_axis, ct)));
}

/* trace.metal:219-224 */
static inline f3 random_hemisphere_uniform(f3 x_axis, f3 y_axis, f3 z_axis, float rx, float ry) {
    float z = rx;
    float r = sqrtf(fmax_msl(0.0f, 1.0f - z * z));
    float phi = 2 * PI * ry;
    return normalize(vadd(vadd(vscale(x_axis, r * COSF(phi)), vscale(y_axis, r * SINF(phi))), vscale(z_axis, z)));
}

/* trace.metal:226-233 */
static inline f3 GGX_sample(f3 n, float rx, float ry, float alpha) {
    f3 x, y;
    orthonormal(n, &x, &y);
    float theta = 2 * PI * rx;
    float phi = ATANF(alpha * sqrtf(ry) / sqrtf(1.0f - ry));
    float sp = SINF(phi), cp = COSF(phi), st = SINF(theta), ct = COSF(theta);
    return normalize(vadd(vadd(vscale(x, sp * ct), vscale(y, sp * st)), vscale(n, cp)));
}

/* trace.metal:235-237 */
static inline f3 specular_reflection(f3 i, f3 m) {
    return normalize(vsub(vscale(m, 2 * dot(i, m)), i));
}

/* trace.metal:243-248 */
static inline f3 GGX_transmit(f3 i, f3 m, float ni, float no) {
    float cosTheta_i = dot(i, m);
    float eta = ni / no;
    float cosTheta_t = sqrtf(1 + eta * (cosTheta_i * cosTheta_i - 1));
    return normalize(vsub(vscale(m, eta * cosTheta_i - cosTheta_t), vscale(i, eta)));
}

/* trace.metal:250-252 */
static inline f3 specular_transmit_half_direction(f3 i, f3 o, float ni, float no) {
    return normalize(vadd(vscale(o, no), vscale(i, ni)));
}

/* trace.metal:254-264 */
static inline float degreve_fresnel(f3 i, f3 m, float ni, float nt) {
    float cosTheta_i = fabsf(dot(i, m));
    float eta = ni / nt;
    float sinTheta_t2 = eta * eta * (1.0f - cosTheta_i * cosTheta_i);
    if (sinTheta_t2 >= 1.0f) return 1.0f;
    float cosTheta_t = sqrtf(1.0f - sinTheta_t2);
    float r_parallel = (nt * cosTheta_i - ni * cosTheta_t) / (nt * cosTheta_i + ni * cosTheta_t);
    float r_perpendicular = (ni * cosTheta_i - nt * cosTheta_t) / (ni * cosTheta_i + nt * cosTheta_t);
    return 0.5f * (r_parallel * r_parallel + r_perpendicular * r_perpendicular);
}

/* trace.metal:266-271 */
static inline float GGX_G1(f3 v, f3 m, float alpha) {
    float mv = dot(m, v);
    float sin2 = 1.0f - mv * mv;
    float tan2 = sin2 / (mv * mv);
    return 2.0f / (1.0f + sqrtf(1.0f + alpha * alpha * tan2));
}

/* trace.metal:273-277 */
static inline float GGX_G(f3 i, f3 o, f3 m, f3 n, float alpha) {
    if (dot(i, m) * dot(i, n) <= 0.0f) return 0.0f;
    if (dot(o, m) * dot(o, n) <= 0.0f) return 0.0f;
    return GGX_G1(i, m, alpha) * GGX_G1(o, m, alpha);
}

/* trace.metal:279-288 */
static inline float GGX_D(f3 m, f3 n, float alpha) {
    if (alpha == 0.0f) return 1.0f;
    float alpha2 = alpha * alpha;
    float cosTheta = dot(m, n);
    float cosTheta2 = cosTheta * cosTheta;
    float denom = cosTheta2 * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (PI * denom * denom);
}

/* trace.metal:290-292 */
static inline float reflect_jacobian(f3 m, f3 o) { return 1.0f / (4.0f * fabsf(dot(m, o))); }

/* trace.metal:294-301 */
static inline float transmit_jacobian(f3 i, f3 o, f3 m, float ni, float no) {
    (void)m;
    f3 h = specular_transmit_half_direction(i, o, ni, no);
    float cosTheta_i = dot(i, h);
    float cosTheta_o = dot(o, h);
    float numerator = no * no * fabsf(cosTheta_o);
    float denominator = (ni * cosTheta_i + no * cosTheta_o) * (ni * cosTheta_i + no * cosTheta_o);
    return numerator / denominator;
}

/* trace.metal:303-309 */
static inline float GGX_BRDF_reflect(f3 i, f3 o, f3 m, f3 n, float ni, float no, float alpha) {
    float D = GGX_D(m, n, alpha);
    float G = GGX_G(i, o, m, n, alpha);
    float F = degreve_fresnel(i, m, ni, no);
    return (D * G * F) / (4.0f * fabsf(dot(i, m)));
}

/* trace.metal:311-328 */
static inline float GGX_BRDF_transmit(f3 i, f3 o, f3 m, f3 n, float ni, float no, float alpha) {
    f3 h = specular_transmit_half_direction(i, o, ni, no);
    float D = GGX_D(m, n, alpha);
    float G = GGX_G(i, o, m, n, alpha);
    float F = degreve_fresnel(i, m, ni, no);
    float im = dot(i, h);
    float om = dot(o, h);
    float in = dot(i, n);
    float on = dot(o, n);
    float coeff = (im * om) / (in * on);
    float num = no * no * D * G * (1.0f - F);
    float denom = (ni * im + no * om) * (ni * im + no * om);
    return coeff * num / denom;
}

/* trace.metal:330-332 */
static inline f3 sample_normal(const Triangle *tr, float u, float v) {
    return normalize(vadd(vadd(vscale(tr->n0, 1 - u - v), vscale(tr->n1, u)), vscale(tr->n2, v)));
}

/* trace.metal:334-346 */
static inline void diffuse_bounce(f3 wi, f3 n, int from_camera, float rx, float ry, f3 *wo, float *f, float *c_p, float *l_p) {
    f3 x, y;
    orthonormal(n, &x, &y);
    *wo = random_hemisphere_cosine(x, y, n, rx, ry);
    *f = fabsf(dot(n, *wo)) / PI;
    if (from_camera) { *c_p = fabsf(dot(n, *wo)) / PI; *l_p = fabsf(dot(n, wi)) / PI; }
    else { *c_p = fabsf(dot(n, wi)) / PI; *l_p = fabsf(dot(n, *wo)) / PI; }
}

/* trace.metal:348-362 */
static inline void reflect_bounce(f3 wi, f3 n, f3 m, float ni, float no, float alpha, int from_camera, f3 *wo, float *f, float *c_p, float *l_p) {
    *wo = specular_reflection(wi, m);
    *f = GGX_BRDF_reflect(wi, *wo, m, n, ni, no, alpha);
    float pf = degreve_fresnel(wi, m, ni, no);
    float pm = fabsf(dot(m, n)) * GGX_D(m, n, alpha);
    if (from_camera) { *c_p = pf * pm * reflect_jacobian(m, *wo); *l_p = pf * pm * reflect_jacobian(m, wi); }
    else { *c_p = pf * pm * reflect_jacobian(m, wi); *l_p = pf * pm * reflect_jacobian(m, *wo); }
}

/* trace.metal:364-379 */
static inline void transmit_bounce(f3 wi, f3 n, f3 m, float ni, float no, float alpha, int from_camera, f3 *wo, float *f, float *c_p, float *l_p) {
    *wo = GGX_transmit(wi, m, ni, no);
    *f = GGX_BRDF_transmit(wi, *wo, m, n, ni, no, alpha);
    float pf = 1.0f - degreve_fresnel(wi, m, ni, no);
    float pm = fabsf(dot(m, n)) * GGX_D(m, n, alpha);
    if (from_camera) {
        *c_p = pf * pm * transmit_jacobian(wi, *wo, m, ni, no);
        *l_p = pf * pm * transmit_jacobian(*wo, wi, vneg(m), no, ni);
    } else {
        *c_p = pf * pm * transmit_jacobian(*wo, wi, vneg(m), no, ni);
        *l_p = pf * pm * transmit_jacobian(wi, *wo, m, ni, no);
    }
}

static inline void counters_merge(Counters *dst, const Counters *src) {
    if (!dst) return;
#pragma omp atomic
    dst->rays += src->rays;
#pragma omp atomic
    dst->box_tests += src->box_tests;
#pragma omp atomic
    dst->tri_tests += src->tri_tests;
}

/* ---- K3 generate_paths, trace.metal:381-532 ---- */
static void generate_paths_one(uint32_t id, const Ray *rays, const Box *boxes, const Triangle *triangles,
                               const Material *materials, uint32_t *random_buffer, float *out4,
                               Path *output_paths, float *float_debug4, Counters *cnt) {
    Path path;
    memset(&path, 0, sizeof path);          /* Q3: defined as zero-filled */
    path.length = 0;
    Ray ray, new_ray, next_ray;
    memset(&new_ray, 0, sizeof new_ray);    /* Q4 */
    memset(&next_ray, 0, sizeof next_ray);
    ray = rays[id];
    path.from_camera = ray.from_camera;
    out4[4 * id + 0] = 0; out4[4 * id + 1] = 0; out4[4 * id + 2] = 0; out4[4 * id + 3] = 0;

    uint32_t seed0 = random_buffer[2 * id];
    uint32_t seed1 = random_buffer[2 * id + 1];

    if (path.from_camera == 0) new_ray.l_importance = 1.0f / (2.0f * PI);
    else new_ray.c_importance = ray.c_importance;

    for (int i = 0; i < BOUNCES; i++) {
        int best_i = -1;
        float best_t = INFINITY;
        float u = 0, v = 0;
        traverse_bvh(&ray, boxes, triangles, &best_i, &best_t, &u, &v, cnt);
        if (best_i == -1) break;

        Triangle triangle = triangles[best_i];
        Material material = materials[triangle.material];

        f3 n;
        float ni, no;
        float alpha = material.alpha;
        f3 sampled_normal = sample_normal(&triangle, u, v);
        float facing = dot(vneg(ray.direction), triangle.normal);
        if (facing > 0) { n = sampled_normal; ni = 1.0f; no = material.ior; }
        else if (facing < 0) { n = vneg(sampled_normal); ni = material.ior; no = 1.0f; }
        else break;

        new_ray.origin = vadd(ray.origin, vscale(ray.direction, best_t));
        new_ray.material = triangle.material;
        new_ray.triangle = best_i;

        if (triangle.is_light && dot(ray.direction, triangle.normal) < 0.0f) new_ray.hit_light = best_i;
        else new_ray.hit_light = -1;
        if (triangle.is_camera) new_ray.hit_camera = best_i;
        else new_ray.hit_camera = -1;

        f3 wi = vneg(ray.direction);

        float rand_x_a = xorshift_random(&seed0);
        float rand_y_a = xorshift_random(&seed1);
        float rand_x_b = xorshift_random(&seed0);
        float rand_y_b = xorshift_random(&seed1);

        f3 wo = V3(0, 0, 0);
        float f = 1.0f, c_p = 1.0f, l_p = 1.0f;

        f3 m = GGX_sample(n, rand_x_a, rand_y_a, alpha);
        if (dot(wi, m) < 0.0f) break;
        if (dot(m, n) < 0.0f) break;
        new_ray.normal = n;

        float fresnel = degreve_fresnel(wi, m, ni, no);
        if (material.type == 0) {
            diffuse_bounce(wi, n, path.from_camera, rand_x_b, rand_y_b, &wo, &f, &c_p, &l_p);
        } else if (material.type == 1) {
            if (rand_x_b <= fresnel) reflect_bounce(wi, n, m, ni, no, alpha, path.from_camera, &wo, &f, &c_p, &l_p);
            else transmit_bounce(wi, n, m, ni, no, alpha, path.from_camera, &wo, &f, &c_p, &l_p);
        } else if (material.type == 2) {
            if (rand_x_b <= fresnel) reflect_bounce(wi, n, m, ni, no, alpha, path.from_camera, &wo, &f, &c_p, &l_p);
            else diffuse_bounce(wi, n, path.from_camera, rand_x_b, rand_y_b, &wo, &f, &c_p, &l_p);
        } else {
            reflect_bounce(wi, n, m, ni, no, alpha, path.from_camera, &wo, &f, &c_p, &l_p);
        }

        float wi_n = dot(wi, triangle.normal), wo_n = dot(wo, triangle.normal);
        if (wi_n > 0.0f && wo_n > 0.0f) new_ray.color = vmul(vscale(ray.color, f), material.color);       /* external reflection */
        else if (wi_n < 0.0f && wo_n > 0.0f) new_ray.color = vmul(vscale(ray.color, f), material.color);  /* egress */
        else new_ray.color = vscale(ray.color, f);                                                       /* internal reflection, ingress */

        new_ray.direction = wo;
        new_ray.inv_direction = vrcp(wo);

        if (path.from_camera) {
            next_ray.c_importance = c_p;
            ray.l_importance = l_p;
            new_ray.tot_importance = ray.tot_importance * new_ray.c_importance;
        } else {
            next_ray.l_importance = l_p;
            ray.c_importance = c_p;
            new_ray.tot_importance = ray.tot_importance * new_ray.l_importance;
        }

        if (f == 0.0f) break;

        path.rays[i] = ray;
        path.length = i + 1;

        ray = new_ray;
        new_ray = next_ray;
    }

    output_paths[id] = path;
    float_debug4[4 * id + 0] = 100.0f; float_debug4[4 * id + 1] = 100.0f;
    float_debug4[4 * id + 2] = 100.0f; float_debug4[4 * id + 3] = 100.0f;

    for (int i = 0; i < path.length; i++) {
        if (path.rays[i].hit_light >= 0) {
            if (i > 0) {   /* rays[0] always has hit_light = -1 (both generators), so i >= 1 */
                f3 c = vdivs(path.rays[i - 1].color, path.rays[i].tot_importance);
                out4[4 * id + 0] = c.x; out4[4 * id + 1] = c.y; out4[4 * id + 2] = c.z; out4[4 * id + 3] = 1.0f;
            }
            break;
        }
    }
    random_buffer[2 * id] = seed0;
    random_buffer[2 * id + 1] = seed1;
}

void orc_generate_paths(int n_threads, const Ray *rays, const Box *boxes, const Triangle *triangles,
                        const Material *materials, uint32_t *random_buffer, float *out4,
                        Path *output_paths, float *float_debug4, Counters *counters) {
#pragma omp parallel
    {
        Counters local = {0, 0, 0};
#pragma omp for schedule(dynamic, 256)
        for (int id = 0; id < n_threads; id++)
            generate_paths_one((uint32_t)id, rays, boxes, triangles, materials, random_buffer, out4,
                               output_paths, float_debug4, &local);
        counters_merge(counters, &local);
    }
}

/* trace.metal:539-544 (uses the vertices' STORED outgoing directions, Q9) */
static inline float cosine_geometry_term(const Ray *a, const Ray *b) {
    float dist = length3(vsub(b->origin, a->origin));
    float cos_a = fabsf(dot(a->direction, a->normal));
    float cos_b = fabsf(dot(b->direction, b->normal));
    return cos_a * cos_b / (dist * dist);
}

/* trace.metal:546-549 */
static inline const Ray *get_ray(const Path *camera_path, const Path *light_path, int t, int s, int i) {
    if (i < s) return &light_path->rays[i];
    return &camera_path->rays[t + s - i - 1];
}

/* trace.metal:551-562 (no +0.5: Q8) */
static inline f3 pixel_center(const Camera *camera, int x, int y) {
    float x_normalized = (x - 0.5f * camera->pixel_width) / (float)camera->pixel_width;
    float y_normalized = (y - 0.5f * camera->pixel_height) / (float)camera->pixel_height;
    f3 x_vector = vscale(camera->dx, x_normalized * camera->phys_width);
    f3 y_vector = vscale(camera->dy, y_normalized * camera->phys_height);
    return vadd(vadd(camera->center, x_vector), y_vector);
}

/* trace.metal:564-567 */
static inline float gaussian_weight(f3 p, f3 q, float sigma) {
    float dist = length3(vsub(p, q));
    return EXPF(-dist * dist / (2.0f * sigma * sigma));
}

/* trace.metal:569-617 */
static void world_ray_to_camera_ray(const Box *boxes, const Triangle *triangles, const Material *materials,
                                    const Camera *camera, const Ray *world_ray, int *pixel_idx, Ray *camera_ray,
                                    Counters *cnt) {
    if (materials[triangles[world_ray->triangle].material].type > 0) return;

    Ray test_ray;
    memset(&test_ray, 0, sizeof test_ray);
    test_ray.origin = world_ray->origin;
    test_ray.direction = normalize(vsub(camera->focal_point, world_ray->origin));
    if (dot(test_ray.direction, camera->direction) > 0.0f) return;
    test_ray.inv_direction = vrcp(test_ray.direction);
    test_ray.triangle = world_ray->triangle;
    test_ray.normal = world_ray->normal;

    int best_i = -1;
    float best_t = INFINITY, u = 0, v = 0;
    traverse_bvh(&test_ray, boxes, triangles, &best_i, &best_t, &u, &v, cnt);
    if (best_i == -1) return;
    if (!triangles[best_i].is_camera) return;

    f3 camera_point = vadd(test_ray.origin, vscale(test_ray.direction, best_t));
    float x = dot(vsub(camera_point, camera->center), camera->dx);
    float y = dot(vsub(camera_point, camera->center), camera->dy);
    int pixel_x = (int)roundf((x / camera->phys_width + 0.5f) * camera->pixel_width);
    int pixel_y = (int)roundf((y / camera->phys_height + 0.5f) * camera->pixel_height);

    *pixel_idx = pixel_y * camera->pixel_width + pixel_x;

    camera_ray->origin = camera_point;
    camera_ray->direction = normalize(vsub(camera->focal_point, camera_point));
    camera_ray->inv_direction = vrcp(camera_ray->direction);
    camera_ray->normal = camera->direction;
    camera_ray->material = 7;
    camera_ray->color = V3(1.0f, 1.0f, 1.0f);
    camera_ray->triangle = best_i;
    camera_ray->tot_importance = 1.0f;
    camera_ray->hit_light = -1;
    camera_ray->hit_camera = best_i;
}

/* Strategy log (tests only): when set, connect_paths_one records for every CONNECTED pair (t,s) of every pixel what the
 * MIS stage computed -- record (id*7 + t)*7 + s of STRATEGY_LOG_STRIDE floats:
 *   [0] 1.0 (pair produced a weight)  [1] w  [2] p_s  [3] sum of p_values  [4] g  [5..7] color  [8] s+t
 *   [9..9+13) p_values[0..12] after the specular zeroing and p_values[s+t] = 0 (trace.metal:745-771).
 * The log has no influence on any result. */
#define STRATEGY_LOG_STRIDE 24
static float *g_strategy_log = NULL;
void orc_set_strategy_log(float *buf) { g_strategy_log = buf; }
int orc_strategy_log_stride(void) { return STRATEGY_LOG_STRIDE; }

/* ---- K5 connect_paths, trace.metal:620-869 ---- */
static void connect_paths_one(uint32_t id, const Path *camera_paths, const Path *light_paths,
                              const Triangle *triangles, const Material *materials, const Box *boxes,
                              const Camera *camera, WeightAggregator *weight_aggregators, float *out4,
                              int32_t *light_pixel_indices, int32_t *light_path_indices,
                              int32_t *light_ray_indices, float *light_weights, float *light_shade,
                              Counters *cnt) {
    Path camera_path = camera_paths[id];
    Path light_path = light_paths[id];
    const Ray cached_camera_zero = camera_path.rays[0];
    Camera c = camera[0];

    WeightAggregator aggregator;
    memset(&aggregator, 0, sizeof aggregator);
    int pixel_idx = cached_camera_zero.pixel_idx;
    int light_pixel_idx = -1;
    int total_pixels = c.pixel_width * c.pixel_height;
    float contrib_weight_sum = 0.0f;

    for (int t = 1; t < camera_path.length + 1; t++) {
        for (int s = 0; s < light_path.length + 1; s++) {
            if (t + s < 2) continue;

            Ray light_ray, camera_ray;
            memset(&light_ray, 0, sizeof light_ray);
            memset(&camera_ray, 0, sizeof camera_ray);
            light_ray.triangle = -1;
            camera_ray.triangle = -1;
            camera_path.rays[0] = cached_camera_zero;
            f3 dir_l_to_c = V3(0, 0, 0);
            light_pixel_idx = -1;

            if (s == 0) {
                camera_ray = camera_path.rays[t - 1];
                if (camera_ray.hit_light < 0) continue;
            } else if (t == 1) {
                light_ray = light_path.rays[s - 1];
                world_ray_to_camera_ray(boxes, triangles, materials, &c, &light_ray, &light_pixel_idx, &camera_path.rays[0], cnt);
                if (light_pixel_idx == -1) continue;
                camera_ray = camera_path.rays[0];
                dir_l_to_c = normalize(vsub(camera_ray.origin, light_ray.origin));
            } else {
                camera_ray = camera_path.rays[t - 1];
                light_ray = light_path.rays[s - 1];
                if (materials[light_ray.material].type > 0) continue;
                if (materials[camera_ray.material].type > 0) continue;
                dir_l_to_c = normalize(vsub(camera_ray.origin, light_ray.origin));
                if (dot(light_ray.normal, dir_l_to_c) < DELTA) continue;
                if (dot(camera_ray.normal, vneg(dir_l_to_c)) < DELTA) continue;
                if (!visibility_test(&light_ray, &camera_ray, boxes, triangles, cnt)) continue;
            }

            float p_ratios[32];
            float p_values[32];
            memset(p_ratios, 0, sizeof p_ratios);   /* Q5 */
            memset(p_values, 0, sizeof p_values);

            for (int i = 0; i < s + t; i++) {
                float num, denom;
                if (i == 0) {
                    const Ray *a = get_ray(&camera_path, &light_path, t, s, 0);
                    const Ray *b = get_ray(&camera_path, &light_path, t, s, 1);
                    num = a->l_importance;
                    denom = a->c_importance * cosine_geometry_term(a, b);
                } else if (i == s + t - 1) {
                    const Ray *a = get_ray(&camera_path, &light_path, t, s, s + t - 1);
                    const Ray *b = get_ray(&camera_path, &light_path, t, s, s + t - 2);
                    num = a->l_importance * cosine_geometry_term(a, b);
                    denom = a->c_importance;
                } else {
                    const Ray *a = get_ray(&camera_path, &light_path, t, s, i - 1);
                    const Ray *b = get_ray(&camera_path, &light_path, t, s, i);
                    const Ray *cc = get_ray(&camera_path, &light_path, t, s, i + 1);
                    num = b->l_importance * cosine_geometry_term(a, b);
                    denom = b->c_importance * cosine_geometry_term(b, cc);
                }
                p_ratios[i] = num / denom;
            }

            float prior_camera_importance = camera_ray.tot_importance;
            float prior_light_importance;
            if (s == 0) prior_light_importance = 1.0f;
            else prior_light_importance = light_ray.tot_importance;
            float p_s = prior_camera_importance * prior_light_importance;

            float p_i = p_s;
            for (int i = s; i < s + t + 1; i++) {
                p_values[i + 1] = p_ratios[i] * p_i;
                p_i = p_values[i + 1];
            }
            p_i = p_s;
            for (int i = s - 1; i >= 0; i--) {
                p_values[i] = p_i / p_ratios[i];
                p_i = p_values[i];
            }
            p_values[s] = p_s;

            for (int i = 0; i < s + t; i++) {
                if (materials[get_ray(&camera_path, &light_path, t, s, i)->material].type > 0) {
                    p_values[i] = 0.0f;
                    p_values[i + 1] = 0.0f;
                }
            }
            p_values[s + t] = 0.0f;

            float sum = 0.0f;
            for (int i = 0; i < s + t + 1; i++) sum += p_values[i];

            float w;
            if (p_values[s] > 0.0f && sum > 0.0f) w = p_values[s] / sum;
            else continue;

            f3 color = V3(1.0f, 1.0f, 1.0f);
            float g = 1.0f;
            float new_light_f = 1.0f;
            float new_camera_f = 1.0f;

            if (s == 0) {
                f3 prior_color = camera_path.rays[t - 2].color;   /* t >= 2 here since t + s >= 2 */
                f3 emission = materials[camera_ray.material].emission;
                color = vmul(prior_color, emission);
            } else if (t == 1) {
                int prior_light_ind = (s - 2) > 0 ? (s - 2) : 0;
                f3 prior_color = light_path.rays[prior_light_ind].color;
                if (s > 1) new_light_f = fabsf(dot(dir_l_to_c, light_ray.normal)) / PI;
                color = vmul(vscale(prior_color, new_light_f), materials[light_ray.material].color);
                g = cosine_geometry_term(&light_ray, &camera_ray);
            } else {
                f3 prior_camera_color = camera_path.rays[t - 2].color;
                Material camera_material = materials[camera_ray.material];
                new_camera_f = fabsf(dot(vneg(dir_l_to_c), camera_ray.normal)) / PI;
                f3 camera_color = vmul(vscale(prior_camera_color, new_camera_f), camera_material.color);
                f3 light_color;
                if (s == 1) {
                    light_color = materials[light_ray.material].emission;
                } else {
                    f3 prior_light_color = light_path.rays[s - 2].color;
                    Material light_material = materials[light_ray.material];
                    new_light_f = fabsf(dot(dir_l_to_c, light_ray.normal)) / PI;
                    light_color = vmul(vscale(prior_light_color, new_light_f), light_material.color);
                }
                color = vmul(camera_color, light_color);
                g = cosine_geometry_term(&camera_ray, &light_ray);
            }
            if (g_strategy_log) {
                float *rec = g_strategy_log + ((size_t)(id * 7u + (uint32_t)t) * 7u + (uint32_t)s) * STRATEGY_LOG_STRIDE;
                rec[0] = 1.0f; rec[1] = w; rec[2] = p_s; rec[3] = sum; rec[4] = g;
                rec[5] = color.x; rec[6] = color.y; rec[7] = color.z; rec[8] = (float)(s + t);
                for (int i = 0; i < 13; i++) rec[9 + i] = p_values[i];
            }
            if (t != 1) {
                aggregator.total_contribution = vadd(aggregator.total_contribution, vdivs(vscale(color, w * g), p_s));
                contrib_weight_sum += w;
            } else {
                size_t slot = (size_t)id + (size_t)s * (size_t)total_pixels;
                light_pixel_indices[slot] = light_pixel_idx;
                light_path_indices[slot] = (int32_t)id;
                light_ray_indices[slot] = s - 1;
                light_weights[slot] = w;
                light_shade[slot] = new_light_f * g / p_s;
            }
        }
    }

    float weight_sum = 0.0f;
    float pixel_phys_width = c.phys_width / c.pixel_width;
    float pixel_phys_height = c.phys_height / c.pixel_height;
    float sigma = 0.5f * sqrtf(pixel_phys_width * pixel_phys_width + pixel_phys_height * pixel_phys_height);

    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) aggregator.weights[i][j] = 0.0f;

    for (int i = -1; i < 2; i++) {
        for (int j = -1; j < 2; j++) {
            int new_sample_x = (pixel_idx % c.pixel_width) + i;
            int new_sample_y = (pixel_idx / c.pixel_width) + j;
            if (new_sample_x < 0 || new_sample_x >= c.pixel_width || new_sample_y < 0 || new_sample_y >= c.pixel_height) continue;
            int new_sample_index = new_sample_y * c.pixel_width + new_sample_x;
            if (new_sample_index < 0 || new_sample_index >= c.pixel_width * c.pixel_height) continue;
            float weight = gaussian_weight(pixel_center(&c, new_sample_x, new_sample_y), camera_paths[id].rays[0].origin, sigma);
            aggregator.weights[i + 1][j + 1] = weight;
            weight_sum += weight;
        }
    }
    if (weight_sum != 0.0f)
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) aggregator.weights[i][j] = aggregator.weights[i][j] / weight_sum;

    aggregator.contrib_weight_sum = contrib_weight_sum;
    out4[4 * id + 0] = aggregator.total_contribution.x; out4[4 * id + 1] = aggregator.total_contribution.y;
    out4[4 * id + 2] = aggregator.total_contribution.z; out4[4 * id + 3] = 1.0f;
    weight_aggregators[id] = aggregator;
}

void orc_connect_paths(int n_threads, const Path *camera_paths, const Path *light_paths, const Triangle *triangles,
                       const Material *materials, const Box *boxes, const Camera *camera,
                       WeightAggregator *weight_aggregators, float *out4, int32_t *light_pixel_indices,
                       int32_t *light_path_indices, int32_t *light_ray_indices, float *light_weights,
                       float *light_shade, Counters *counters) {
#pragma omp parallel
    {
        Counters local = {0, 0, 0};
#pragma omp for schedule(dynamic, 64)
        for (int id = 0; id < n_threads; id++)
            connect_paths_one((uint32_t)id, camera_paths, light_paths, triangles, materials, boxes, camera,
                              weight_aggregators, out4, light_pixel_indices, light_path_indices,
                              light_ray_indices, light_weights, light_shade, &local);
        counters_merge(counters, &local);
    }
}

/* ---- K7 light_sort, trace.metal:872-934: one bitonic pass; threads are independent ---- */
void orc_light_sort(int64_t n_threads, int32_t *light_pixel_indices, int32_t *light_path_indices,
                    int32_t *light_ray_indices, float *light_weights, float *light_shade,
                    uint32_t stage, uint32_t passOfStage, uint32_t n, uint32_t pairs_per_thread) {
#pragma omp parallel for schedule(static)
    for (int64_t id = 0; id < n_threads; id++) {
        for (uint32_t p = 0; p < pairs_per_thread; ++p) {
            uint32_t global_pair_id = (uint32_t)id * pairs_per_thread + p;
            uint32_t pairDistance = 1u << (passOfStage - 1);
            uint32_t blockWidth = 1u << stage;
            uint32_t leftId = (global_pair_id / pairDistance) * pairDistance * 2u + (global_pair_id % pairDistance);
            uint32_t rightId = leftId + pairDistance;
            if (rightId >= n || leftId >= n) continue;
            int ascending = ((global_pair_id & (blockWidth >> 1)) == 0u);
            int32_t l = light_pixel_indices[leftId], r = light_pixel_indices[rightId];
            if ((ascending && l > r) || (!ascending && l < r)) {
                int32_t ti; float tf;
                light_pixel_indices[leftId] = r; light_pixel_indices[rightId] = l;
                ti = light_path_indices[leftId]; light_path_indices[leftId] = light_path_indices[rightId]; light_path_indices[rightId] = ti;
                ti = light_ray_indices[leftId]; light_ray_indices[leftId] = light_ray_indices[rightId]; light_ray_indices[rightId] = ti;
                tf = light_weights[leftId]; light_weights[leftId] = light_weights[rightId]; light_weights[rightId] = tf;
                tf = light_shade[leftId]; light_shade[leftId] = light_shade[rightId]; light_shade[rightId] = tf;
            }
        }
    }
}

/* The full launch schedule of renderer.py:213-231 in one call (same passes, same order). */
void orc_light_sort_all(int32_t *light_pixel_indices, int32_t *light_path_indices, int32_t *light_ray_indices,
                        float *light_weights, float *light_shade, uint32_t n) {
    int log_n = 0;
    while ((1u << log_n) < n) log_n++;
    for (int stage = 1; stage <= log_n; stage++)
        for (int pass = stage; pass > 0; pass--)
            orc_light_sort(n / 8, light_pixel_indices, light_path_indices, light_ray_indices, light_weights,
                           light_shade, (uint32_t)stage, (uint32_t)pass, n, 4);
}

/* ---- K8 light_image_gather, trace.metal:937-964 ---- */
void orc_light_image_gather(int n_threads, const Path *light_paths, const Material *materials,
                            const int32_t *path_indices, const int32_t *ray_indices, const int32_t *bins,
                            uint32_t offset, const float *weights, const float *shades, float *light_image4,
                            float *sum_weights) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int id = 0; id < n_threads; id++) {
        int start_idx = bins[id];
        int end_idx = bins[id + 1];
        f3 total_contribution = V3(0, 0, 0);
        float weight_sum = 0.0f;
        for (int64_t i = (int64_t)start_idx + offset; i < (int64_t)end_idx + offset; i++) {
            int path_idx = path_indices[i];
            int ray_idx = ray_indices[i];
            const Path *path = &light_paths[path_idx];
            const Ray *ray = &path->rays[ray_idx];
            const Ray *prior_ray = &path->rays[ray_idx - 1 > 0 ? ray_idx - 1 : 0];
            const Material *mat = &materials[ray->material];
            total_contribution = vadd(total_contribution, vmul(vscale(prior_ray->color, weights[i] * shades[i]), mat->color));
            weight_sum += weights[i];
        }
        light_image4[4 * id + 0] = total_contribution.x; light_image4[4 * id + 1] = total_contribution.y;
        light_image4[4 * id + 2] = total_contribution.z; light_image4[4 * id + 3] = 1.0f;
        sum_weights[id] += weight_sum;
    }
}

/* ---- K4 reset_light_indices, trace.metal:967-978 ---- */
void orc_reset_light_indices(int64_t n, int32_t *light_pixel_indices, int32_t *light_path_indices,
                             int32_t *light_ray_indices, float *light_weights, float *light_shade) {
#pragma omp parallel for schedule(static)
    for (int64_t id = 0; id < n; id++) {
        light_pixel_indices[id] = -1;
        light_path_indices[id] = 0;
        light_ray_indices[id] = 0;
        light_weights[id] = 0.0f;
        light_shade[id] = 0.0f;
    }
}

/* ---- K6 adaptive_finalize_samples, trace.metal:981-1018 ---- */
void orc_adaptive_finalize_samples(int n_threads, const WeightAggregator *weight_aggregators, const Camera *camera_buffer,
                                   float *out4, uint32_t *sample_counts, const uint32_t *sample_bin_offsets,
                                   float *sample_weights) {
    Camera camera = camera_buffer[0];
#pragma omp parallel for schedule(static)
    for (int id = 0; id < n_threads; id++) {
        f3 total_sample = V3(0, 0, 0);
        float weight_sum = 0.0f;
        for (int i = -1; i < 2; i++) {
            for (int j = -1; j < 2; j++) {
                int sample_x = (id % camera.pixel_width) + i;
                int sample_y = (id / camera.pixel_width) + j;
                if (sample_x < 0 || sample_x >= camera.pixel_width || sample_y < 0 || sample_y >= camera.pixel_height) continue;
                int sample_index = sample_y * camera.pixel_width + sample_x;
                if (sample_index < 0 || sample_index >= camera.pixel_width * camera.pixel_height) continue;
                for (uint32_t k = sample_bin_offsets[sample_index]; k < sample_bin_offsets[sample_index + 1]; k++) {
                    const WeightAggregator *wa = &weight_aggregators[k];
                    float weight = wa->weights[1 - i][1 - j];
                    total_sample = vadd(total_sample, vscale(wa->total_contribution, weight));
                    weight_sum += weight * wa->contrib_weight_sum;
                }
            }
        }
        sample_counts[id] = sample_bin_offsets[id + 1] - sample_bin_offsets[id];
        out4[4 * id + 0] = total_sample.x; out4[4 * id + 1] = total_sample.y; out4[4 * id + 2] = total_sample.z; out4[4 * id + 3] = 1.0f;
        sample_weights[id] = weight_sum;
    }
}

/* ---- K2 generate_camera_rays, trace.metal:1020-1067 ---- */
void orc_generate_camera_rays(int n_threads, const Camera *camera, uint32_t *random_buffer, const uint32_t *indices, Ray *out) {
    Camera c = camera[0];
#pragma omp parallel for schedule(static)
    for (int id = 0; id < n_threads; id++) {
        Ray ray;
        memset(&ray, 0, sizeof ray);
        uint32_t seed0 = random_buffer[2 * id];
        uint32_t seed1 = random_buffer[2 * id + 1];
        float x_offset = xorshift_random(&seed0);
        float y_offset = xorshift_random(&seed1);
        int pixel_idx = (int)indices[id];
        int pixel_x = pixel_idx % c.pixel_width;
        int pixel_y = pixel_idx / c.pixel_width;
        float x_normalized = (pixel_x + x_offset - 0.5f * c.pixel_width) / (float)c.pixel_width;
        float y_normalized = (pixel_y + y_offset - 0.5f * c.pixel_height) / (float)c.pixel_height;
        f3 x_vector = vscale(vscale(c.dx, x_normalized), c.phys_width);
        f3 y_vector = vscale(vscale(c.dy, y_normalized), c.phys_height);
        f3 origin = vadd(vadd(c.center, x_vector), y_vector);
        f3 direction = normalize(vsub(c.focal_point, origin));
        ray.origin = origin;
        ray.direction = direction;
        ray.normal = c.direction;
        ray.inv_direction = vrcp(direction);
        ray.color = V3(1.0f, 1.0f, 1.0f);
        ray.material = 7;
        ray.triangle = -1;
        ray.hit_light = -1;
        ray.hit_camera = -1;
        ray.from_camera = 1;
        ray.c_importance = 1.0f / (c.phys_width * c.phys_height);
        ray.l_importance = 1.0f;
        ray.tot_importance = ray.c_importance;
        ray.pixel_idx = pixel_idx;
        out[id] = ray;
        random_buffer[2 * id] = seed0;
        random_buffer[2 * id + 1] = seed1;
    }
}

/* ---- K1 generate_light_rays, trace.metal:1070-1124 ---- */
void orc_generate_light_rays(int n_threads, const Triangle *light_triangles, const float *surface_areas,
                             const int32_t *light_triangle_indices, const Material *materials,
                             uint32_t *random_buffer, Ray *out, const int32_t *counts) {
    int light_count = counts[0];
#pragma omp parallel for schedule(static)
    for (int id = 0; id < n_threads; id++) {
        Ray ray;
        memset(&ray, 0, sizeof ray);
        ray.from_camera = 0;
        ray.hit_light = -1;
        ray.hit_camera = -1;
        uint32_t seed0 = random_buffer[2 * id];
        uint32_t seed1 = random_buffer[2 * id + 1];
        int light_index = (int)(xorshift_random(&seed0) * light_count);
        if (light_index > light_count - 1) light_index = light_count - 1;    /* Q1 */
        Triangle light_triangle = light_triangles[light_index];
        float surface_area = surface_areas[light_index];
        float u = xorshift_random(&seed0);
        float v = xorshift_random(&seed1);
        if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
        float w = 1.0f - u - v;
        ray.normal = light_triangle.normal;
        ray.origin = vadd(vadd(vadd(vscale(light_triangle.v0, u), vscale(light_triangle.v1, v)), vscale(light_triangle.v2, w)),
                          vscale(ray.normal, DELTA));
        f3 x, y;
        orthonormal(ray.normal, &x, &y);
        float rand_x = xorshift_random(&seed0);
        float rand_y = xorshift_random(&seed1);
        ray.direction = random_hemisphere_uniform(x, y, ray.normal, rand_x, rand_y);
        ray.inv_direction = vrcp(ray.direction);
        ray.material = light_triangle.material;
        ray.color = materials[ray.material].emission;
        ray.color.w = 0.0f;
        ray.triangle = light_triangle_indices[light_index];
        ray.c_importance = 1.0f;
        ray.l_importance = 1.0f / (light_count * surface_area);
        ray.tot_importance = ray.l_importance;
        out[id] = ray;
        random_buffer[2 * id] = seed0;
        random_buffer[2 * id + 1] = seed1;
    }
}

/* ---- standalone probes used by the tests ---- */
void orc_traverse(int n, const Ray *rays, const Box *boxes, const Triangle *triangles,
                  int32_t *best_i, float *best_t, float *u, float *v, Counters *counters) {
#pragma omp parallel
    {
        Counters local = {0, 0, 0};
#pragma omp for schedule(dynamic, 256)
        for (int id = 0; id < n; id++) {
            int bi = -1; float bt = INFINITY, uu = 0, vv = 0;
            traverse_bvh(&rays[id], boxes, triangles, &bi, &bt, &uu, &vv, &local);
            best_i[id] = bi; best_t[id] = bt; u[id] = uu; v[id] = vv;
        }
        counters_merge(counters, &local);
    }
}

void orc_math(int which, int n, const float *in, float *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        float x = in[i];
        switch (which) {
            case 0: out[i] = det_sinf(x); break;
            case 1: out[i] = det_cosf(x); break;
            case 2: out[i] = det_acosf(x); break;
            case 3: out[i] = det_atanf(x); break;
            case 4: out[i] = det_expf(x); break;
            case 5: out[i] = det_asinf(x); break;
        }
    }
}

/* Scalar BSDF probes (analytic tests: Fresnel, GGX normalisation, bounce pdfs). */
float orc_fresnel(const float *i, const float *m, float ni, float nt) {
    return degreve_fresnel(V3(i[0], i[1], i[2]), V3(m[0], m[1], m[2]), ni, nt);
}
float orc_ggx_d(const float *m, const float *n, float alpha) {
    return GGX_D(V3(m[0], m[1], m[2]), V3(n[0], n[1], n[2]), alpha);
}
void orc_ggx_sample(const float *n, float rx, float ry, float alpha, float *out) {
    f3 m = GGX_sample(V3(n[0], n[1], n[2]), rx, ry, alpha);
    out[0] = m.x; out[1] = m.y; out[2] = m.z;
}
/* kind: 0 diffuse, 1 reflect, 2 transmit.  out = wo[3], f, c_p, l_p */
void orc_bounce(int kind, const float *wi, const float *n, const float *m, float ni, float no, float alpha,
                int from_camera, float rx, float ry, float *out) {
    f3 wo = V3(0, 0, 0);
    float f = 1, c_p = 1, l_p = 1;
    f3 WI = V3(wi[0], wi[1], wi[2]), N = V3(n[0], n[1], n[2]), M = V3(m[0], m[1], m[2]);
    if (kind == 0) diffuse_bounce(WI, N, from_camera, rx, ry, &wo, &f, &c_p, &l_p);
    else if (kind == 1) reflect_bounce(WI, N, M, ni, no, alpha, from_camera, &wo, &f, &c_p, &l_p);
    else transmit_bounce(WI, N, M, ni, no, alpha, from_camera, &wo, &f, &c_p, &l_p);
    out[0] = wo.x; out[1] = wo.y; out[2] = wo.z; out[3] = f; out[4] = c_p; out[5] = l_p;
}

/* Batch form of the bounce probe, same item layout as the product's cl2_probe_bounce:
 * in 12 floats {wi.xyz, n.xyz, rx, ry, ni, no, alpha, kind}; out 8 floats {wo.xyz, f, c_p, l_p, fresnel, m.x}. */
void orc_bounce_batch(int n, int from_camera, const float *in, float *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        const float *p = in + 12 * i;
        f3 wi = V3(p[0], p[1], p[2]), nn = V3(p[3], p[4], p[5]);
        float rx = p[6], ry = p[7], ni = p[8], no = p[9], alpha = p[10];
        int kind = (int)p[11];
        f3 m = GGX_sample(nn, rx, ry, alpha);
        f3 wo = V3(0, 0, 0);
        float f = 1.0f, c_p = 1.0f, l_p = 1.0f;
        if (kind == 0) diffuse_bounce(wi, nn, from_camera, rx, ry, &wo, &f, &c_p, &l_p);
        else if (kind == 1) reflect_bounce(wi, nn, m, ni, no, alpha, from_camera, &wo, &f, &c_p, &l_p);
        else if (kind == 2) transmit_bounce(wi, nn, m, ni, no, alpha, from_camera, &wo, &f, &c_p, &l_p);
        else wo = m;
        float *q = out + 8 * i;
        q[0] = wo.x; q[1] = wo.y; q[2] = wo.z; q[3] = f; q[4] = c_p; q[5] = l_p;
        q[6] = degreve_fresnel(wi, m, ni, no); q[7] = m.x;
    }
}

/* ---- research probe (not part of the restatement): a front-to-back walk over the same tree, to
 * measure what ordering would save and how often it disagrees with the reference's walk.  Children
 * are entered nearest-first; everything else (tests, strict `<`, leaf order) as in traverse_bvh. ---- */
void orc_traverse_ordered(int n, const Ray *rays, const Box *boxes, const Triangle *triangles,
                          int32_t *best_i_out, float *best_t_out, Counters *counters) {
#pragma omp parallel
    {
        Counters local = {0, 0, 0};
#pragma omp for schedule(dynamic, 256)
        for (int id = 0; id < n; id++) {
            const Ray *ray = &rays[id];
            int stack[128]; float tstack[128];
            int sp = 0, best_i = -1;
            float best_t = INFINITY, u, v;
            local.rays++;
            float t0 = INFINITY;
            local.box_tests++;
            if (ray_box_intersect(ray, &boxes[0], &t0)) { stack[sp] = 0; tstack[sp++] = t0; }
            while (sp > 0) {
                int node = stack[--sp];
                float tn = tstack[sp];
                if (!(tn < best_t)) continue;
                const Box *b = &boxes[node];
                if (b->right == 0) {
                    float tl = INFINITY, tr = INFINITY;
                    local.box_tests += 2;
                    int hl = ray_box_intersect(ray, &boxes[b->left], &tl) && tl < best_t;
                    int hr = ray_box_intersect(ray, &boxes[b->left + 1], &tr) && tr < best_t;
                    if (hl && hr) {
                        if (tl <= tr) { stack[sp] = b->left + 1; tstack[sp++] = tr; stack[sp] = b->left; tstack[sp++] = tl; }
                        else { stack[sp] = b->left; tstack[sp++] = tl; stack[sp] = b->left + 1; tstack[sp++] = tr; }
                    } else if (hl) { stack[sp] = b->left; tstack[sp++] = tl; }
                    else if (hr) { stack[sp] = b->left + 1; tstack[sp++] = tr; }
                } else {
                    for (int i = b->left; i < b->right; i++) {
                        float t = INFINITY;
                        local.tri_tests++;
                        if (ray_triangle_intersect(ray, &triangles[i], &t, &u, &v) && t < best_t) { best_i = i; best_t = t; }
                    }
                }
            }
            best_i_out[id] = best_i; best_t_out[id] = best_t;
        }
        counters_merge(counters, &local);
    }
}
