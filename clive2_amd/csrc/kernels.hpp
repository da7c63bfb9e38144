// kernels.hpp -- bidirectional path tracing kernels for gfx950 (MI355X).
//
// The reference runs the per-sample pipeline as two megakernels (`generate_paths`,
// `connect_paths`, src/trace.metal:381-532, :620-869) that keep whole 1040-byte Path records in
// thread-private memory, followed by a 300-launch bitonic sort + host bincount to splat the light
// image.  Here the same arithmetic runs as a short sequence of launches over SoA state:
//
//   k_gen_light_rays / k_gen_camera_rays     vertex slot 0 of each subpath                   (K1, K2)
//   k_trace_subpath<cam> levels [first,end)  closest hit + bounce, walking state in registers; the
//                                            survivors of a launch are compacted into the next
//                                            launch's queue with wave ballots                   (K3)
//     large trees: k_traverse_persistent (lane-level ray replacement) + k_trace_subpath<EXT_HIT>
//   k_connect_setup                          enumerate (t,s) strategy pairs, cull, emit a compacted
//                                            queue of connection-ray tags                (K5, culls)
//   k_traverse_conn / k_traverse_persistent  closest hit for every connection ray
//   k_connect_resolve (connect_resolve.hpp)  MIS weights + contributions per pixel in the
//                                            reference's (t,s) order; t=1 splats by float atomics
//                                            (replaces K4, K7 x300, host bincount, K8) -- or, with
//                                            cl2_set_reproducible, as the reference's records, sorted
//                                            stably by target and summed in order by k_det_gather
//                                            (det_splat.hpp)
//   k_finalize / k_accumulate                3x3 reconstruction filter, on-device accumulators (K6,
//                                            renderer.py:253-278)
//   k_traverse_paths                         closest-hit probe (cl2_probe_traverse)
//
// Path-vertex SoA (per subpath kind, slot v in [0,6), pixel p; index v*B + p):
//   P0 = {origin.xyz, c_importance}  P1 = {direction.xyz, l_importance}
//   P2 = {normal.xyz, meta}          P3 = {color.xyz, tot_importance}     tri = triangle index
//   meta = material | hit_light<<8 | hit_camera<<9
// Sample streams (cl2_set_sample_streams): a handle may carry K independent samples of the frame per pass.  Every per-pixel
// array then has K x W x H entries, entry k*W*H + p being pixel p of stream k -- exactly what K Renderers with K seed buffers
// (renderer.py:86-87; the ranks of the sample split) would hold -- and every launch covers all of them: `B` below is that
// entry count (the array stride), the frame's own pixel count is camera.pixel_width x pixel_height.  Only four places
// look at a pixel's POSITION and take `entry % (W*H)`: the camera-ray generator, the reconstruction-filter weights, the
// filter gather of K6 and the light-image splat; the accumulators are per pixel (K6 adds the streams in order).
// Every arithmetic statement follows the cited reference lines in the same operation order
// (IEEE binary32, -ffp-contract=off), so per-stage results can be compared exactly with a CPU
// evaluation of the same statements.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vecmath.hpp"
#include "detmath.hpp"
#include "bsdf.hpp"
#include "bvh_traverse.hpp"
#include "bvh_wide.hpp"

namespace cl2 {

constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / 64;
constexpr int MAX_VERTS = 6;          // bounce loop bound, trace.metal:407
constexpr int CONN_SLOTS = 36;        // (t in 1..6) x (s in 1..6) strategy pairs that need a ray
constexpr int TAG_PID_BITS = 26;
constexpr int SHADE_LDS_CAP = 128;   // shading triangles staged in LDS by the subpath kernel (64 B each)
constexpr int LDS_MAT_CAP = 32;      // materials staged in LDS by the subpath kernel
constexpr int META_HIT_LIGHT = 1 << 8;
constexpr int META_HIT_CAMERA = 1 << 9;

struct CameraRec {   // byte-identical to struct Camera, trace.metal:72-85
    float center[4], focal_point[4], direction[4], dx[4], dy[4];
    int pixel_width, pixel_height;
    float phys_width, phys_height, h_fov, v_fov;
    int pad[2];
};

struct MaterialDev { float4 color_type; float4 emission_alpha; float ior; float pad[3]; };  // 48 B

struct PathBufs {
    float4 *P0, *P1, *P2, *P3;
    int* tri;
    int* len;
    float* carry;
};

struct Stats {   // device-side tallies
    unsigned long long rays, box_tests, tri_tests, conn_rays, counted_rays;
    // cl2_set_counting(2): what the 4-wide walk ITSELF fetched (not the reference's walk, whose tallies the fields above hold):
    // [0] subpath launches, [1] connection launches; {rays, wide-node visits, triangle records, stack entries spilled to the
    // global overflow array, binary records visited by rays with a non-finite 1/d}
    unsigned long long walk[2][5];
};

__device__ __forceinline__ V3 cam3(const float* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Block-level stream compaction: wave ballots + one global atomic per block.
// Returns the output index for threads with `pred`, garbage otherwise.  All threads must call.
__device__ __forceinline__ unsigned block_compact_index(bool pred, unsigned* counter) {
    __shared__ unsigned s_wave_total[WAVES_PER_BLOCK];
    __shared__ unsigned s_base;
    const unsigned long long mask = __ballot(pred);
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const unsigned before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_total[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < WAVES_PER_BLOCK; w++) tot += s_wave_total[w];
        s_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    unsigned off = s_base + before;
    for (int w = 0; w < wave; w++) off += s_wave_total[w];
    return off;
}

// ---------------------------------------------------------------- K1: generate_light_rays
// trace.metal:1070-1124.  Writes vertex slot 0 of the light subpath.
__device__ __forceinline__ void gen_light_ray(int id, const float4* __restrict__ light_tris /*5 float4 per light: v0, v1, v2, normal, {material}*/,
                                              const float* __restrict__ light_areas, const int* __restrict__ light_tri_index,
                                              const MaterialDev* __restrict__ mats, int light_count, uint32_t& seed0, uint32_t& seed1,
                                              const PathBufs& pb) {
    int li = (int)(xorshift_random(seed0) * light_count);
    if (li > light_count - 1) li = light_count - 1;     // the draw can be exactly 1.0
    const float4 a0 = light_tris[5 * li], a1 = light_tris[5 * li + 1], a2 = light_tris[5 * li + 2],
                 an = light_tris[5 * li + 3], am = light_tris[5 * li + 4];
    const float area = light_areas[li];
    float u = xorshift_random(seed0);
    float v = xorshift_random(seed1);
    if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
    const float w = 1.0f - u - v;
    const V3 normal = v3(an);
    const V3 origin = ((v3(a0) * u + v3(a1) * v) + v3(a2) * w) + DELTA_F * normal;
    V3 x, y;
    orthonormal(normal, x, y);
    const float rx = xorshift_random(seed0);
    const float ry = xorshift_random(seed1);
    const V3 dir = random_hemisphere_uniform(x, y, normal, rx, ry);
    const int material = __float_as_int(am.x);
    const float4 em = mats[material].emission_alpha;
    const float l_imp = 1.0f / (light_count * area);
    pb.P0[id] = f4(origin, 1.0f);                 // c_importance "filled in later"
    pb.P1[id] = f4(dir, l_imp);
    pb.P2[id] = f4(normal, __int_as_float(material));
    pb.P3[id] = make_float4(em.x, em.y, em.z, l_imp);
    pb.tri[id] = light_tri_index[li];
    pb.len[id] = 0;
    pb.carry[id] = 1.0f / (2.0f * PI_F);          // new_ray.l_importance, trace.metal:401
}

__global__ __launch_bounds__(BLOCK) void k_gen_light_rays(
        int B, const float4* __restrict__ light_tris, const float* __restrict__ light_areas, const int* __restrict__ light_tri_index,
        const MaterialDev* __restrict__ mats, int light_count, uint2* __restrict__ seeds, PathBufs pb) {
    const int id = blockIdx.x * BLOCK + threadIdx.x;
    if (id >= B) return;
    uint2 sd = seeds[id];
    uint32_t seed0 = sd.x, seed1 = sd.y;
    gen_light_ray(id, light_tris, light_areas, light_tri_index, mats, light_count, seed0, seed1, pb);
    seeds[id] = make_uint2(seed0, seed1);
}

// ---------------------------------------------------------------- K2: generate_camera_rays
// trace.metal:1020-1067 with indices[id] == id (renderer.py:92-94).
__device__ __forceinline__ void gen_camera_ray(int id, const CameraRec& c, uint32_t& seed0, uint32_t& seed1, const PathBufs& pb) {
    const float x_offset = xorshift_random(seed0);
    const float y_offset = xorshift_random(seed1);
    const int pix = id % (c.pixel_width * c.pixel_height);          // the pixel of entry `id` (sample streams: see the header)
    const int pixel_x = pix % c.pixel_width, pixel_y = pix / c.pixel_width;
    const float xn = (pixel_x + x_offset - 0.5f * c.pixel_width) / (float)c.pixel_width;
    const float yn = (pixel_y + y_offset - 0.5f * c.pixel_height) / (float)c.pixel_height;
    const V3 xv = (xn * cam3(c.dx)) * c.phys_width;
    const V3 yv = (yn * cam3(c.dy)) * c.phys_height;
    const V3 origin = (cam3(c.center) + xv) + yv;
    const V3 dir = normalize(cam3(c.focal_point) - origin);
    const float c_imp = 1.0f / (c.phys_width * c.phys_height);
    pb.P0[id] = f4(origin, c_imp);
    pb.P1[id] = f4(dir, 1.0f);                    // l_importance "filled in later"
    pb.P2[id] = f4(cam3(c.direction), __int_as_float(7));
    pb.P3[id] = make_float4(1.0f, 1.0f, 1.0f, c_imp);
    pb.tri[id] = -1;
    pb.len[id] = 0;
    pb.carry[id] = c_imp;                         // new_ray.c_importance, trace.metal:404
}

__global__ __launch_bounds__(BLOCK) void k_gen_camera_rays(int B, CameraRec c, uint2* __restrict__ seeds, PathBufs pb) {
    const int id = blockIdx.x * BLOCK + threadIdx.x;
    if (id >= B) return;
    uint2 sd = seeds[id];
    uint32_t seed0 = sd.x, seed1 = sd.y;
    gen_camera_ray(id, c, seed0, seed1, pb);
    seeds[id] = make_uint2(seed0, seed1);
}

// K1 then K2 for the same pixel in one launch (cl2_run_samples): the pixel's RNG state goes from the light-ray draws to
// the camera-ray draws in registers, as make_light_rays followed by make_camera_rays leaves it (renderer.py:281-283).
__global__ __launch_bounds__(BLOCK) void k_gen_rays(
        int B, const float4* __restrict__ light_tris, const float* __restrict__ light_areas, const int* __restrict__ light_tri_index,
        const MaterialDev* __restrict__ mats, int light_count, CameraRec c, uint2* __restrict__ seeds, PathBufs lp, PathBufs cp) {
    const int id = blockIdx.x * BLOCK + threadIdx.x;
    if (id >= B) return;
    uint2 sd = seeds[id];
    uint32_t seed0 = sd.x, seed1 = sd.y;
    gen_light_ray(id, light_tris, light_areas, light_tri_index, mats, light_count, seed0, seed1, lp);
    gen_camera_ray(id, c, seed0, seed1, cp);
    seeds[id] = make_uint2(seed0, seed1);
}

// ---------------------------------------------------------------- traversal of subpath rays
// One closest-hit query (trace.metal:144-176) per queued path; ray = (P0.xyz, P1.xyz) of `level`.
template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void k_traverse_paths(
        BvhView bvh, const int* __restrict__ queue, const unsigned* __restrict__ count,
        const float4* __restrict__ P0v, const float4* __restrict__ P1v, float4* __restrict__ hit, Stats* stats) {
    BvhLds lds{nullptr, nullptr};
    stage_bvh(lds, bvh);
    const unsigned n = *count;
    const unsigned j = blockIdx.x * BLOCK + threadIdx.x;
    unsigned nb = 0, nt = 0;
    if (j < n) {
        const int pid = queue ? queue[j] : (int)j;
        const V3 o = v3(P0v[pid]), d = v3(P1v[pid]);
        const Hit h = closest_hit<COUNT>(lds, bvh, o, d, rcp3(d), nb, nt);
        hit[pid] = make_float4(__int_as_float(h.tri), h.t, h.u, h.v);
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { nb += __shfl_down(nb, off); nt += __shfl_down(nt, off); }
        if (lane_id() == 0) {
            atomicAdd(&stats->box_tests, (unsigned long long)nb);
            atomicAdd(&stats->tri_tests, (unsigned long long)nt);
        }
    }
    if (j == 0) {
        atomicAdd(&stats->rays, (unsigned long long)n);
        if (COUNT) atomicAdd(&stats->counted_rays, (unsigned long long)n);
    }
}

// ---------------------------------------------------------------- persistent traversal kernels (large scenes)
// Ray sources of the persistent walks: load(j) fetches ray j of the launch and returns the TOKEN the lane keeps while it
// walks (pixel id / connection tag), store(token, hit) writes the result -- without going back to the queue for it.
struct PathRaySource {          // subpath rays: queue entry -> pixel -> (P0.xyz, P1.xyz) of one level
    const int* queue; const float4* P0v; const float4* P1v; float4* hit;
    __device__ __forceinline__ int load(unsigned j, V3& o, V3& d) const {
        const int p = queue ? queue[j] : (int)j;
        o = v3(P0v[p]); d = v3(P1v[p]);
        return p;
    }
    __device__ __forceinline__ void store(int pid, const Hit& h) const {
        hit[pid] = make_float4(__int_as_float(h.tri), h.t, h.u, h.v);
    }
};

// Two level-0 launches in one: the first rays of the light subpaths (rays [0, B)) and of the camera subpaths (rays [B, 2B)).
// Traversal needs no random numbers, so the camera subpath's first closest hit does not have to wait for the light subpath's
// last bounce (its BOUNCE does: one RNG stream per pixel, renderer.py:281-291); the hits of the camera rays go to a buffer of
// their own and are consumed by the camera subpath's level-0 bounce launch six levels later.  Token: bit 31 = camera.
struct DualPathRaySource {
    const float4 *LP0, *LP1, *CP0, *CP1; float4 *hit_light, *hit_camera; int B;
    __device__ __forceinline__ int load(unsigned j, V3& o, V3& d) const {
        const bool cam = j >= (unsigned)B;
        const int p = (int)(cam ? j - (unsigned)B : j);
        o = v3((cam ? CP0 : LP0)[p]); d = v3((cam ? CP1 : LP1)[p]);
        return p | (cam ? (int)0x80000000 : 0);
    }
    __device__ __forceinline__ void store(int token, const Hit& h) const {
        float4* dst = token < 0 ? hit_camera : hit_light;
        dst[token & 0x7FFFFFFF] = make_float4(__int_as_float(h.tri), h.t, h.u, h.v);
    }
};

// Closest-hit results of the connection rays: the hit triangle for every slot (all the t >= 2 pairs need,
// visibility_test compares triangles only) and the distance for the six t = 1 slots (the film projection,
// world_ray_to_camera_ray, needs it).  One allocation: int tri[36][B] followed by float t1[6][B].
__device__ __forceinline__ void chit_store(float2* chit, int B, int slot, int pid, int tri, float t) {
    reinterpret_cast<int*>(chit)[(size_t)slot * B + pid] = tri;
    if (slot < MAX_VERTS) (reinterpret_cast<float*>(chit) + (size_t)CONN_SLOTS * B)[(size_t)slot * B + pid] = t;
}
__device__ __forceinline__ float2 chit_load(const float2* chit, int B, int t, int s, int pid) {
    const int slot = (t - 1) * 6 + (s - 1);
    const int tri = reinterpret_cast<const int*>(chit)[(size_t)slot * B + pid];
    const float dist = (t == 1) ? (reinterpret_cast<const float*>(chit) + (size_t)CONN_SLOTS * B)[(size_t)slot * B + pid] : 0.0f;
    return make_float2(__int_as_float(tri), dist);
}

struct ConnRaySource {          // connection rays: tag {slot, pixel} -> light vertex s-1 toward focal point / camera vertex t-1
    const int* ctag; const float4* LP0; const float4* CP0; float2* chit; V3 focal; int B;
    __device__ __forceinline__ int load(unsigned j, V3& o, V3& d) const {
        const int tag = ctag[j];
        const int pid = tag & ((1 << TAG_PID_BITS) - 1), slot = (unsigned)tag >> TAG_PID_BITS;
        const int t = slot / 6 + 1, s = slot % 6 + 1;
        o = v3(LP0[(size_t)(s - 1) * B + pid]);
        V3 target = focal;
        if (t > 1) target = v3(CP0[(size_t)(t - 1) * B + pid]);
        d = normalize(target - o);
        return tag;
    }
    __device__ __forceinline__ void store(int tag, const Hit& h) const {
        const int pid = tag & ((1 << TAG_PID_BITS) - 1), slot = (unsigned)tag >> TAG_PID_BITS;
        chit_store(chit, B, slot, pid, h.tri, h.t);
    }
};

// SGPR budget: 256-thread workgroups are admitted per CU up to floor(800 / (ceil(sgprs / 16) * 16 + 16)) -- 8 up to 80
// SGPRs, 7 from 81 (MI355X_MICROARCH.md, "Residency").  Left alone the compiler takes 81 for the connection-ray
// instantiation: one workgroup in eight of the persistent grid then never becomes resident beside the others.
template <bool COUNT, bool TWO_TRIS, class Source>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(78))) void k_traverse_persistent(BvhView bvh, const unsigned* __restrict__ count,
                                                              unsigned* __restrict__ work_counter, Source src, Stats* stats,
                                                              int is_conn) {
    BvhLds lds{nullptr, nullptr};
    stage_bvh(lds, bvh);
    const unsigned n = *count;
    unsigned nb = 0, nt = 0;
    traverse_persistent<COUNT, TWO_TRIS>(lds, bvh, n, work_counter, src, nb, nt);
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { nb += __shfl_down(nb, off); nt += __shfl_down(nt, off); }
        if (lane_id() == 0) {
            atomicAdd(&stats->box_tests, (unsigned long long)nb);
            atomicAdd(&stats->tri_tests, (unsigned long long)nt);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&stats->rays, (unsigned long long)n);
        if (is_conn) atomicAdd(&stats->conn_rays, (unsigned long long)n);
        if (COUNT) atomicAdd(&stats->counted_rays, (unsigned long long)n);
    }
}

// The same launch shape over the 4-wide collapse of the tree (bvh_wide.hpp): exact, half the dependent fetches.
// (The ray tally is added BEFORE the walk: `threadIdx.x == 0` after it would keep the thread index alive through the whole loop,
// in a kernel held to 64 VGPRs -- round 4's builds spilled exactly that register to scratch at entry and reloaded it at exit.)
// The tallying variant (cl2_set_counting(2), never timed) carries four more counters per lane and takes 6 waves per SIMD.
template <int TRI_REPS, class Source, bool TALLY = false, bool SPEC = false, bool PACK = false, bool ORDER = false>
__global__ __launch_bounds__(BLOCK, TALLY ? 6 : 8) __attribute__((amdgpu_num_sgpr(80))) void k_traverse_wide(WideView wide, BvhView bvh, const unsigned* __restrict__ count,
                                                        unsigned* __restrict__ work_counter, Source src, Stats* stats, int is_conn) {
    const unsigned n = *count;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&stats->rays, (unsigned long long)n);
        if (is_conn) atomicAdd(&stats->conn_rays, (unsigned long long)n);
        if (TALLY) atomicAdd(&stats->walk[is_conn ? 1 : 0][0], (unsigned long long)n);
    }
    WalkTally tally;
    traverse_wide_persistent<TRI_REPS, TALLY, SPEC, PACK, ORDER>(wide, bvh, n, work_counter, src, tally);
    if (TALLY) {
        unsigned v[4] = {tally.visits, tally.tri_records, tally.spills, tally.bin_nodes};
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 4; k++) v[k] += __shfl_down(v[k], off);
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)
            for (int k = 0; k < 4; k++) atomicAdd(&stats->walk[is_conn ? 1 : 0][1 + k], (unsigned long long)v[k]);
    }
}

// ---------------------------------------------------------------- K3: generate_paths, levels [first, end)
// trace.metal:407-516 for the iterations first..end-1 of the path loop, one thread per queued path.
//
// The launch keeps the walking state (current vertex, RNG, carried pdf) in registers across its
// levels: a vertex is written to HBM once, complete, when its reverse pdf becomes known one level
// later (the reference's `path.rays[i] = ray`, :512), instead of being written, re-read and patched
// by consecutive launches.  Survivors of the last level are compacted into `queue_out` (wave
// ballots, one atomic per workgroup) for the next launch.  The host chooses the segmentation:
// [0,6) in one launch when nearly every path survives (closed scenes: compaction buys nothing),
// one level per launch when paths die quickly (open scenes, glass).
//
// Vertices `first` (written by the generator or the previous launch) is patched in place; the vertex
// created at level end-1 is written complete (reverse pdf still open) when the path goes on.
struct ShadeLds {
    float4 tri_shade[4 * SHADE_LDS_CAP];
    MaterialDev mats[LDS_MAT_CAP];
};

// Where the shading records of a launch live: LDS copies for small scenes, else global memory.
struct ShadeSrc {
    const ShadeLds* lds; bool shade_lds, mats_lds;
    const float4* tri_shade_g; const MaterialDev* mats_g;
};

// One iteration of the path loop AFTER the closest hit is known (trace.metal:417-516): shading normal,
// side of the surface, microfacet normal, the bounce, throughput.  In: the current vertex's outgoing ray
// (ro, rd), its throughput colour and total importance, the carried forward pdf `fwd`, the RNG state.
// Out (go_on): the next vertex (the reference's new_ray) complete but for its own reverse pdf, the reverse pdf
// of the CURRENT vertex (`rev`: l_importance of a camera vertex, c_importance of a light vertex, :501/:505)
// and the forward pdf the next iteration carries.  `from_camera` is a compile-time constant in the
// per-level kernels and a per-lane value in the persistent whole-subpath kernel; the arithmetic is the same.
struct BounceOut {
    bool go_on;
    V3 o, d, n, col;
    float tot, next_fwd, rev;
    int meta, tri;
};
__device__ __forceinline__ BounceOut shade_and_bounce(bool from_camera, const Hit& hh, V3 ro, V3 rd, V3 rcol, float r_tot,
                                                      float fwd, uint2& sd, bool& seeds_dirty, const ShadeSrc& src) {
    BounceOut out;
    out.go_on = false;
    out.o = ro; out.d = rd; out.n = ro; out.col = rcol;
    out.tot = 0.0f; out.next_fwd = 0.0f; out.rev = 0.0f; out.meta = 0; out.tri = -1;
    const int best_i = hh.tri;
    if (best_i == -1) return out;
    const float best_t = hh.t, u = hh.u, v = hh.v;
    float4 s0, s1, s2, s3;
    if (src.shade_lds) { s0 = src.lds->tri_shade[4 * best_i]; s1 = src.lds->tri_shade[4 * best_i + 1]; s2 = src.lds->tri_shade[4 * best_i + 2]; s3 = src.lds->tri_shade[4 * best_i + 3]; }
    else { s0 = src.tri_shade_g[4 * best_i]; s1 = src.tri_shade_g[4 * best_i + 1]; s2 = src.tri_shade_g[4 * best_i + 2]; s3 = src.tri_shade_g[4 * best_i + 3]; }
    const int material = __float_as_int(s0.w);
    const bool is_light = __float_as_int(s1.w) != 0, is_camera = __float_as_int(s2.w) != 0;
    const MaterialDev mat = src.mats_lds ? src.lds->mats[material] : src.mats_g[material];
    const int mtype = __float_as_int(mat.color_type.w);
    const float alpha = mat.emission_alpha.w;
    const V3 tn = v3(s3);

    const V3 sn = normalize((v3(s0) * (1 - u - v) + v3(s1) * u) + v3(s2) * v);   // sample_normal :330-332
    const float facing = dot(-rd, tn);
    V3 nrm = sn;
    float ni = 1.0f, no = mat.ior;
    if (facing > 0) { }
    else if (facing < 0) { nrm = -sn; ni = mat.ior; no = 1.0f; }
    else return out;                                                           // :433-435
    const bool hit_light = is_light && dot(rd, tn) < 0.0f;
    const V3 wi = -rd;
    const float rxa = xorshift_random(sd.x);
    const float rya = xorshift_random(sd.y);
    const float rxb = xorshift_random(sd.x);
    const float ryb = xorshift_random(sd.y);
    seeds_dirty = true;

    // Microfacet normal (:466).  For a smooth Lambertian surface (alpha == 0, type 0) GGX_sample
    // reduces exactly to normalize(n): phi = atan(0) = 0, so m = normalize(0*x + 0*y + 1*n), and m
    // only feeds the two sign tests below, which ignore the sign of a zero component.  The draw
    // ry == 1 (0/0 -> NaN) keeps the general route.
    V3 m;
    if (mtype == 0 && alpha == 0.0f && rya != 1.0f) m = normalize(nrm);
    else m = GGX_sample(nrm, rxa, rya, alpha);
    if (dot(wi, m) < 0.0f || dot(m, nrm) < 0.0f) return out;
    float fresnel = 0.0f;                                  // only types 1 and 2 read it (:476-485)
    if (mtype != 0) fresnel = degreve_fresnel(wi, m, ni, no);
    Bounce b;
    if (mtype == 0) b = diffuse_bounce(wi, nrm, from_camera, rxb, ryb);
    else if (mtype == 1) {
        if (rxb <= fresnel) b = reflect_bounce(wi, nrm, m, ni, no, alpha, from_camera);
        else b = transmit_bounce(wi, nrm, m, ni, no, alpha, from_camera);
    } else if (mtype == 2) {
        if (rxb <= fresnel) b = reflect_bounce(wi, nrm, m, ni, no, alpha, from_camera);
        else b = diffuse_bounce(wi, nrm, from_camera, rxb, ryb);
    } else b = reflect_bounce(wi, nrm, m, ni, no, alpha, from_camera);

    const float wi_n = dot(wi, tn), wo_n = dot(b.wo, tn);
    V3 ncol = b.f * rcol;
    if ((wi_n > 0.0f && wo_n > 0.0f) || (wi_n < 0.0f && wo_n > 0.0f)) ncol = ncol * v3(mat.color_type);

    if (b.f == 0.0f) return out;                           // :509
    out.rev = from_camera ? b.l_p : b.c_p;                 // :501 / :505
    // new_ray (:437-449, :496-506) becomes the current vertex of the next level
    out.o = ro + rd * best_t;
    out.d = b.wo; out.n = nrm; out.col = ncol;
    out.tot = r_tot * fwd;                                 // :502 / :506
    out.meta = material | (hit_light ? META_HIT_LIGHT : 0) | (is_camera ? META_HIT_CAMERA : 0);
    out.tri = best_i;
    out.next_fwd = from_camera ? b.c_p : b.l_p;            // next_ray, :500 / :504
    out.go_on = true;
    return out;
}

// (EXT_HIT, the bounce launch of the per-level organisation: five waves per SIMD.  At six -- 80 VGPRs -- it spilled one temporary
// of the bounce arithmetic; round 4 measured 5 and 6 waves equal: 1.14 ms per sample on the glass scene, 7 and 8 slower.)
template <bool FROM_CAMERA, bool COUNT, bool EXT_HIT>
__global__ __launch_bounds__(BLOCK, EXT_HIT ? 5 : 6) void k_trace_subpath(
        BvhView bvh, Stats* stats, int first, int end, const int* __restrict__ queue_in,
        const unsigned* __restrict__ count_in, int* __restrict__ queue_out, unsigned* __restrict__ count_out, int B,
        PathBufs pb, uint2* __restrict__ seeds, const float4* __restrict__ tri_shade_g,
        const MaterialDev* __restrict__ mats_g, int n_mats, unsigned long long* __restrict__ block_stats,
        const float4* __restrict__ ext_hit) {
    // EXT_HIT: the closest hits of the (single) level were produced by k_traverse_persistent (large
    // scenes: traversal with ray replacement runs as its own launch); the BVH is not staged here.
    BvhLds lds{nullptr, nullptr};
    __shared__ ShadeLds sh;
    const bool shade_lds = bvh.n_tris <= SHADE_LDS_CAP, mats_lds = n_mats <= LDS_MAT_CAP;
    if (shade_lds) for (int i = threadIdx.x; i < 4 * bvh.n_tris; i += BLOCK) sh.tri_shade[i] = tri_shade_g[i];
    if (mats_lds) for (int i = threadIdx.x; i < n_mats; i += BLOCK) sh.mats[i] = mats_g[i];
    if (!EXT_HIT) stage_bvh(lds, bvh);                     // ends with the barrier
    else __syncthreads();
    const ShadeSrc shade_src{&sh, shade_lds, mats_lds, tri_shade_g, mats_g};

    const unsigned n = *count_in;
    const unsigned j = blockIdx.x * BLOCK + threadIdx.x;
    bool alive = j < n;
    int pid = 0;
    // current vertex (the reference's `ray`) and what it needs to be stored complete
    V3 ro = v3(0, 0, 0), rd = ro, rn = ro, rcol = ro;
    float r_c = 0.0f, r_l = 0.0f, r_tot = 0.0f, fwd = 0.0f;
    int r_meta = 0, r_tri = -1;
    uint2 sd = make_uint2(0, 0);
    bool seeds_dirty = false;
    int stored = -1;                                       // index of the last vertex stored by this launch
    if (alive) {
        pid = queue_in ? queue_in[j] : (int)j;
        const size_t cur = (size_t)first * B + pid;
        const float4 p0 = pb.P0[cur], p1 = pb.P1[cur], p3 = pb.P3[cur];
        ro = v3(p0); rd = v3(p1); rcol = v3(p3);
        r_c = p0.w; r_l = p1.w; r_tot = p3.w;
        fwd = pb.carry[pid];
        sd = seeds[pid];
    }
    unsigned nb = 0, nt = 0, nrays = 0;

    for (int level = first; level < end; level++) {
        if (!__any(alive)) break;
        BounceOut bo;
        bo.go_on = false;
        if (alive) {
            Hit hh;
            if (EXT_HIT) {
                const float4 h4 = ext_hit[pid];
                hh = Hit{__float_as_int(h4.x), h4.y, h4.z, h4.w};
            } else {
                nrays++;
                hh = closest_hit<COUNT>(lds, bvh, ro, rd, rcp3(rd), nb, nt);             // :409-415
            }
            bo = shade_and_bounce(FROM_CAMERA, hh, ro, rd, rcol, r_tot, fwd, sd, seeds_dirty, shade_src);
            if (bo.go_on) {
                // path.rays[level] = ray, its reverse pdf now known (:501 / :505, :512)
                if (FROM_CAMERA) r_l = bo.rev; else r_c = bo.rev;
                const size_t cur = (size_t)level * B + pid;
                if (level == first) {
                    if (FROM_CAMERA) pb.P1[cur] = f4(rd, r_l); else pb.P0[cur] = f4(ro, r_c);
                } else {
                    pb.P0[cur] = f4(ro, r_c);
                    pb.P1[cur] = f4(rd, r_l);
                    pb.P2[cur] = f4(rn, __int_as_float(r_meta));
                    pb.P3[cur] = f4(rcol, r_tot);
                    pb.tri[cur] = r_tri;
                }
                stored = level;
            }
        }
        if (bo.go_on) {
            ro = bo.o; rd = bo.d; rn = bo.n; rcol = bo.col;
            r_c = FROM_CAMERA ? fwd : 0.0f;
            r_l = FROM_CAMERA ? 0.0f : fwd;
            r_tot = bo.tot; r_meta = bo.meta; r_tri = bo.tri;
            fwd = bo.next_fwd;
        }
        alive = bo.go_on;
    }

    if (stored >= 0) pb.len[pid] = stored + 1;
    if (seeds_dirty) seeds[pid] = sd;
    const bool hand_over = alive && end < MAX_VERTS;      // the path goes on in a later launch
    if (hand_over) {
        const size_t nxt = (size_t)end * B + pid;
        pb.P0[nxt] = f4(ro, r_c);
        pb.P1[nxt] = f4(rd, r_l);
        pb.P2[nxt] = f4(rn, __int_as_float(r_meta));
        pb.P3[nxt] = f4(rcol, r_tot);
        pb.tri[nxt] = r_tri;
        pb.carry[pid] = fwd;
    }
    // tallies: wave shuffle -> LDS -> one plain read-modify-write per workgroup into its own slot
    // (a same-address atomic per wave costs ~0.4 ms per launch at 32k waves; launches on one stream
    // are ordered, so the slot needs no atomic).  The host sums the slots.
    for (int off = 32; off > 0; off >>= 1) {
        nrays += __shfl_down(nrays, off);
        if (COUNT) { nb += __shfl_down(nb, off); nt += __shfl_down(nt, off); }
    }
    __shared__ unsigned s_tally[WAVES_PER_BLOCK][3];
    if (lane_id() == 0) { s_tally[threadIdx.x >> 6][0] = nrays; s_tally[threadIdx.x >> 6][1] = nb; s_tally[threadIdx.x >> 6][2] = nt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tr = 0, tb = 0, tt = 0;
        for (int w = 0; w < WAVES_PER_BLOCK; w++) { tr += s_tally[w][0]; tb += s_tally[w][1]; tt += s_tally[w][2]; }
        unsigned long long* slot = block_stats + (size_t)blockIdx.x * 4;
        slot[0] += tr;
        if (COUNT) { slot[1] += tb; slot[2] += tt; slot[3] += tr; }
    }
    if (end >= MAX_VERTS) return;                          // uniform: no queue after the final bounce
    const unsigned idx = block_compact_index(hand_over, count_out);
    if (hand_over) queue_out[idx] = pid;
}

// ---------------------------------------------------------------- K3 for large trees: whole subpaths, persistent
// Both subpaths of a pixel -- light first, then camera: the camera walk continues the pixel's RNG stream where
// the light walk left it (one seed pair per pixel, renderer.py:86-87, used in that order by run_sample) -- walked
// by ONE lane of a persistent launch.  A lane is a small state machine:
//     TRAV  one node visit / triangle test per step, exactly the steps of traverse_persistent
//     PEND  closest hit known, waiting for the wave's next bounce phase
//     IDLE  takes the next pixel of the launch (chunks handed to waves by one global atomic)
// The bounce (shade_and_bounce: ~10x the instructions of a step, divergent by material) is not run for the
// one or two lanes that finish in a step but for all lanes that have gathered in PEND, when there are
// `bounce_lanes` of them, when the oldest has waited `bounce_wait` steps, or when no lane is left walking.
// Between bounces the walking state lives where the per-level launches keep it: the current vertex is stored
// complete but for its reverse pdf when it is created and patched when the next bounce is known, so a lane
// carries only the ray, the pixel, the level, the RNG state and the forward pdf across its traversal steps.
//
// Against one traversal launch + one bounce launch per level and subpath kind (24 launches per sample, each
// ending in a tail of a few long rays, about half of their time on a 2 M-ray level) this is one launch per
// sample with one tail.  Same arithmetic on the same operands in the same order per path: identical results.
enum { LANE_IDLE = 0, LANE_TRAV = 1, LANE_PEND = 2 };

// WIDE: the lanes walk the exact 4-wide collapse of the tree (bvh_wide.hpp: wide node block, per-lane stack in LDS, two
// triangle pairs per pass) -- except the rays with a non-finite 1/d, which keep the binary walk (records read through
// the caches), as in traverse_wide_persistent.
template <bool COUNT, bool TWO_TRIS, int WAVES_PER_SIMD, bool WIDE, int WIDE_REPS = WIDE_TRI_REPS>
__global__ __launch_bounds__(BLOCK, WAVES_PER_SIMD) void k_subpaths_persistent(
        BvhView bvh, WideView wide, int B, unsigned* __restrict__ work_counter, PathBufs lp, PathBufs cp, uint2* __restrict__ seeds,
        const float4* __restrict__ tri_shade_g, const MaterialDev* __restrict__ mats_g, int n_mats, Stats* stats,
        int bounce_lanes, int bounce_wait, int kinds /* 1 light subpaths only, 2 camera only, 3 both (light first) */) {
    BvhLds s{nullptr, nullptr};
    BvhView b = bvh;
    // WIDE: LDS holds the top of the wide tree and the per-lane stacks instead of the binary window
    extern __shared__ float4 cl2_tree_lds[];
    const int tid = threadIdx.x;
    const int S = wide.stack_lds;
    float4* s_wnodes = cl2_tree_lds;
    int2* s_stack = reinterpret_cast<int2*>(cl2_tree_lds + 8 * wide.n_lds_nodes) + tid;     // entry e of this lane: s_stack[e * BLOCK] (bvh_wide.hpp)
    volatile int* ovf = nullptr;                            // volatile: never merged with the LDS access into a flat one
    if (WIDE) {
        for (int i = tid; i < 8 * wide.n_lds_nodes; i += BLOCK) s_wnodes[i] = wide.nodes[i];
        __syncthreads();
        b.n_lds_nodes = 0; b.lds_tris = 0; b.n_fast_nodes = 0;   // the binary records (rays with a non-finite 1/d) come through the caches
        ovf = reinterpret_cast<volatile int*>(wide.overflow + ((size_t)blockIdx.x * BLOCK + tid) * wide.ovf_stride);
    } else {
        stage_bvh(s, bvh);                                 // ends with the barrier
    }
    // shading records and the (tiny) material table are read through the caches: this launch is for trees
    // that do not fit LDS, whose shading triangles do not either
    const ShadeSrc src{nullptr, false, false, tri_shade_g, mats_g};
    const unsigned n = (unsigned)B;
    const unsigned waves = gridDim.x * (blockDim.x >> 6);
    unsigned chunk = n / (waves * 4u);
    chunk = chunk < 64u ? 64u : (chunk > (unsigned)RAY_CHUNK_MAX ? (unsigned)RAY_CHUNK_MAX : chunk);
    unsigned w_next = 0, w_end = 0;
    bool dry = false;
    int waited = 0;                                        // wave-uniform: steps since a lane first went PEND

    // per-lane state
    int state = LANE_IDLE, pid = 0, level = 0, plen = 0;
    bool cam = false, fast = true, seeds_dirty = false;
    uint2 sd = make_uint2(0, 0);
    float fwd = 0.0f;
    V3 o = v3(0, 0, 0), d = o, inv = o;
    Hit best{-1, __builtin_inff(), 0.0f, 0.0f};
    const int n_nodes = b.n_nodes;
    int node = n_nodes, tri_i = 0, tri_end = 0;
    int cur = -1, sp = 0;                                  // WIDE: wide node to visit next (-1 none), stack depth
    bool wlane = false;                                    // WIDE: this ray walks the wide tree
    unsigned n_box = 0, n_tri = 0, n_rays = 0;

    // (re)start the walk of the current subpath at vertex `level` (its outgoing ray was stored with the vertex)
    auto start_ray = [&](V3 ro, V3 rd) {
        o = ro; d = rd;
        inv = rcp3(d);
        fast = finite3(inv);
        best = Hit{-1, __builtin_inff(), 0.0f, 0.0f};
        node = 0; tri_i = 0; tri_end = 0;
        cur = -1; sp = 0; wlane = false;
        if (WIDE && fast) {
            wlane = true;
            node = n_nodes;
            // the root box, trace.metal:150-156 with best_t = inf
            const float t0x = (wide.root_lo.x - o.x) * inv.x, t0y = (wide.root_lo.y - o.y) * inv.y, t0z = (wide.root_lo.z - o.z) * inv.z;
            const float t1x = (wide.root_hi.x - o.x) * inv.x, t1y = (wide.root_hi.y - o.y) * inv.y, t1z = (wide.root_hi.z - o.z) * inv.z;
            const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                               __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
            const float tmax = __builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fminf(__builtin_fmaxf(t0y, t1y), __builtin_fmaxf(t0z, t1z)));
            if (tmin <= tmax) cur = 0;
        }
        state = LANE_TRAV;
        n_rays++;
    };
    auto push = [&](int ref, float tmin) {
        if (sp < S) s_stack[sp * BLOCK] = make_int2(ref, __float_as_int(tmin));
        else { ovf[2 * (sp - S)] = ref; ovf[2 * (sp - S) + 1] = __float_as_int(tmin); }
        sp++;
    };
    auto pop_next = [&]() {
        while (sp > 0 && cur < 0 && tri_i >= tri_end) {
            sp--;
            int ref; float tmin;
            if (sp < S) { const int2 e = s_stack[sp * BLOCK]; ref = e.x; tmin = __int_as_float(e.y); }
            else { ref = ovf[2 * (sp - S)]; tmin = __int_as_float(ovf[2 * (sp - S) + 1]); }
            if (!(tmin < best.t)) continue;
            if (ref >= 0) cur = ref;
            else { const int info = ~ref; tri_i = info >> 4; tri_end = tri_i + (info & 15) + 1; }
        }
    };

    while (true) {
        // ---- idle lanes take the next pixels of the launch (at once: gathering them first, which pays in the traversal
        // launches -- REFILL_MIN_*, bvh_traverse.hpp -- costs here: 17.0 -> 18.5 / 21.1 ms at 16 / 32 lanes on the 1M-triangle
        // scene, an idle lane being one that could be walking its next subpath) ----
        unsigned long long idle = __ballot(state == LANE_IDLE);
        while (idle && !dry) {
            if (w_next >= w_end) {
                unsigned base = 0;
                if (wave_lane() == 0) base = atomicAdd(work_counter, chunk);
                base = __shfl(base, 0);
                if (base >= n) { dry = true; break; }
                w_next = base;
                w_end = base + chunk < n ? base + chunk : n;
            }
            const unsigned avail = w_end - w_next;
            const unsigned rank = rank_below(idle);
            if (state == LANE_IDLE && rank < avail) {
                pid = (int)(w_next + rank);
                cam = !(kinds & 1); level = 0; plen = 0; seeds_dirty = false;
                sd = seeds[pid];
                fwd = cam ? cp.carry[pid] : lp.carry[pid];
                if (cam) start_ray(v3(cp.P0[pid]), v3(cp.P1[pid])); else start_ray(v3(lp.P0[pid]), v3(lp.P1[pid]));
            }
            const unsigned taken = __popcll(idle) < avail ? __popcll(idle) : avail;
            w_next += taken;
            idle = __ballot(state == LANE_IDLE);
        }
        const unsigned long long pend = __ballot(state == LANE_PEND), trav = __ballot(state == LANE_TRAV);
        if (!pend && !trav) break;

        // ---- bounce phase for the lanes that have gathered ----
        waited = pend ? waited + 1 : 0;
        if (pend && (__popcll(pend) >= bounce_lanes || waited > bounce_wait || !trav)) {
            waited = 0;
            if (state == LANE_PEND) {
                PathBufs& pb = cam ? cp : lp;
                const size_t cur = (size_t)level * B + pid;
                bool go_on = false;
                if (best.tri != -1) {
                    const float4 p3 = pb.P3[cur];
                    const BounceOut bo = shade_and_bounce(cam, best, o, d, v3(p3), p3.w, fwd, sd, seeds_dirty, src);
                    if (bo.go_on) {
                        // the current vertex is complete now: patch its reverse pdf (:501 / :505, :512)
                        if (cam) pb.P1[cur] = f4(d, bo.rev); else pb.P0[cur] = f4(o, bo.rev);
                        plen = level + 1;
                        if (level + 1 < MAX_VERTS) {
                            // the next vertex, complete but for its own reverse pdf; its forward pdf is the one carried here
                            const size_t nxt = cur + B;
                            pb.P0[nxt] = f4(bo.o, cam ? fwd : 0.0f);
                            pb.P1[nxt] = f4(bo.d, cam ? 0.0f : fwd);
                            pb.P2[nxt] = f4(bo.n, __int_as_float(bo.meta));
                            pb.P3[nxt] = f4(bo.col, bo.tot);
                            pb.tri[nxt] = bo.tri;
                            fwd = bo.next_fwd;
                            level++;
                            start_ray(bo.o, bo.d);
                            go_on = true;
                        }
                    }
                }
                if (!go_on) {
                    // this subpath is complete
                    pb.len[pid] = plen;
                    if (!cam && (kinds & 2)) {
                        cam = true; level = 0; plen = 0;
                        fwd = cp.carry[pid];
                        start_ray(v3(cp.P0[pid]), v3(cp.P1[pid]));
                    } else {
                        if (seeds_dirty) seeds[pid] = sd;
                        state = LANE_IDLE;
                    }
                }
            }
            continue;                                      // refill the lanes that went idle before the next step
        }

        // ---- one traversal step per walking lane (the step of traverse_persistent, or of traverse_wide_persistent) ----
        const bool all_fast = !WIDE && __all(state != LANE_TRAV || fast);    // WIDE: the binary block only sees non-finite rays
        if (state == LANE_TRAV) {
            if (WIDE && wlane) {
                pop_next();
                // round 5 (bvh_wide.hpp, SPEC): a lane that is busy with a leaf expands the wide node on top of its stack in the same
                // step; the entry leaves the stack pruned or replaced by ALL its passing children
                bool spec = false;
                int spec_ref = -1;
                if (cur < 0 && tri_i < tri_end && sp > 0 && sp + 3 <= S) {
                    const int2 e = s_stack[(sp - 1) * BLOCK];
                    if (e.x >= 0) { sp--; spec = __int_as_float(e.y) < best.t; spec_ref = e.x; }
                }
                if (cur >= 0 || spec) {
                    const int vnode = spec ? spec_ref : cur;
                    float4 lx, ly, lz, hx, hy, hz, rf;
                    if (wide.n_lds_nodes > 0 && vnode < wide.n_lds_nodes) {    // the window's lanes read LDS in a branch of their own: no flat loads
                        const float4* nd = s_wnodes + 8 * vnode;
                        lx = nd[0]; ly = nd[1]; lz = nd[2]; hx = nd[3]; hy = nd[4]; hz = nd[5]; rf = nd[6];
                        asm volatile("" ::: "memory");
                    } else {
                        const float4* __restrict__ nd = wide.nodes + (size_t)8 * vnode;
                        lx = nd[0]; ly = nd[1]; lz = nd[2]; hx = nd[3]; hy = nd[4]; hz = nd[5]; rf = nd[6];
                    }
                    cur = -1;
                    const float lox[4] = {lx.x, lx.y, lx.z, lx.w}, loy[4] = {ly.x, ly.y, ly.z, ly.w}, loz[4] = {lz.x, lz.y, lz.z, lz.w};
                    const float hix[4] = {hx.x, hx.y, hx.z, hx.w}, hiy[4] = {hy.x, hy.y, hy.z, hy.w}, hiz[4] = {hz.x, hz.y, hz.z, hz.w};
                    const int ref[4] = {__float_as_int(rf.x), __float_as_int(rf.y), __float_as_int(rf.z), __float_as_int(rf.w)};
                    // round 4, as traverse_wide_persistent: four unconditional slab tests (an empty slot's box lies at +inf), the
                    // candidate found so far written unconditionally above the top of the stack with `sp += pushed` while every
                    // visiting lane has room for three entries in the LDS part; the slot that would pop first skips the stack
                    float tm[4];
                    bool pass[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float t0x = (lox[k] - o.x) * inv.x, t0y = (loy[k] - o.y) * inv.y, t0z = (loz[k] - o.z) * inv.z;
                        const float t1x = (hix[k] - o.x) * inv.x, t1y = (hiy[k] - o.y) * inv.y, t1z = (hiz[k] - o.z) * inv.z;
                        const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                                           __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
                        const float tmax = __builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fminf(__builtin_fmaxf(t0y, t1y), __builtin_fmaxf(t0z, t1z)));
                        tm[k] = tmin;
                        pass[k] = tmin <= tmax && tmin < best.t;
                    }
                    int next_ref = pass[3] ? ref[3] : WIDE_EMPTY;
                    float next_tmin = tm[3];
                    if (!__any(sp + 3 > S)) {
#pragma unroll
                        for (int k = 2; k >= 0; k--) {
                            s_stack[sp * BLOCK] = make_int2(next_ref, __float_as_int(next_tmin));
                            sp += (pass[k] && next_ref != WIDE_EMPTY) ? 1 : 0;
                            next_ref = pass[k] ? ref[k] : next_ref;
                            next_tmin = pass[k] ? tm[k] : next_tmin;
                        }
                    } else {
#pragma unroll
                        for (int k = 2; k >= 0; k--) {
                            if (pass[k]) {
                                if (next_ref != WIDE_EMPTY) push(next_ref, next_tmin);
                                next_ref = ref[k]; next_tmin = tm[k];
                            }
                        }
                    }
                    if (spec) {
                        s_stack[sp * BLOCK] = make_int2(next_ref, __float_as_int(next_tmin));      // sp < S: room for four was required
                        sp += next_ref != WIDE_EMPTY ? 1 : 0;
                    }
                    else if (next_ref == WIDE_EMPTY) pop_next();
                    else if (next_ref >= 0) cur = next_ref;
                    else { const int info = ~next_ref; tri_i = info >> 4; tri_end = tri_i + (info & 15) + 1; }
                }
            } else if (tri_i >= tri_end && node < n_nodes) {
                float4 lo, hi;
                if (node < b.n_lds_nodes) { lo = s.nodes[2 * node]; hi = s.nodes[2 * node + 1]; }
                else { lo = b.nodes[2 * node]; hi = b.nodes[2 * node + 1]; }
                const int next = __float_as_int(lo.w);
                if (COUNT) n_box++;
                const float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
                const float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
                float tmin, tmax;
                if (all_fast) {
                    tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                           __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
                    tmax = __builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fminf(__builtin_fmaxf(t0y, t1y), __builtin_fmaxf(t0z, t1z)));
                } else {
                    tmin = max_msl(max_msl(min_msl(t0x, t1x), min_msl(t0y, t1y)), max_msl(min_msl(t0z, t1z), 0.0f));
                    tmax = min_msl(min_msl(max_msl(t0x, t1x), max_msl(t0y, t1y)), min_msl(max_msl(t0z, t1z), __builtin_inff()));
                }
                node = next;
                if (tmin <= tmax && tmin < best.t) {
                    const int info = __float_as_int(hi.w);
                    if (info < 0) node = ~info;
                    else { tri_i = info >> 4; tri_end = tri_i + (info & 15) + 1; }
                }
            }
#pragma unroll
            for (int rep = 0; rep < (WIDE ? WIDE_REPS : 1); rep++)       // triangle pairs per pass: 2 while the tree is cache-resident, 1 above (bvh_wide.hpp)
            if (tri_i < tri_end && (rep == 0 || wlane)) {
                const int i0 = tri_i;
                const bool two = TWO_TRIS && i0 + 1 < tri_end;
                const int i1 = two ? i0 + 1 : i0;
                tri_i = i1 + 1;
                float4 a0, a1, a2, c0, c1, c2;
                if (WIDE && TWO_TRIS && wide.tris36) {
                    // 36-byte records, the pair as one run of 72 bytes: five L1 look-ups instead of six (bvh_wide.hpp: PACK); the
                    // record behind a leaf's last odd triangle is loaded and its test masked out
                    const float* __restrict__ ta = wide.tris36 + (size_t)9 * i0;
                    float a[18];
#pragma unroll
                    for (int k = 0; k < 18; k++) a[k] = ta[k];
                    if (COUNT) n_tri += two ? 2 : 1;
                    tri_test_branchless(o, d, make_float4(a[0], a[1], a[2], 0.0f), make_float4(a[3], a[4], a[5], 0.0f), make_float4(a[6], a[7], a[8], 0.0f), i0, best);
                    Hit second = best;
                    tri_test_branchless(o, d, make_float4(a[9], a[10], a[11], 0.0f), make_float4(a[12], a[13], a[14], 0.0f), make_float4(a[15], a[16], a[17], 0.0f), i0 + 1, second);
                    if (two) best = second;
                    continue;
                }
                if (b.lds_tris) {
                    a0 = s.tris[3 * i0]; a1 = s.tris[3 * i0 + 1]; a2 = s.tris[3 * i0 + 2];
                    if (TWO_TRIS) { c0 = s.tris[3 * i1]; c1 = s.tris[3 * i1 + 1]; c2 = s.tris[3 * i1 + 2]; }
                } else {
                    a0 = b.tris[3 * i0]; a1 = b.tris[3 * i0 + 1]; a2 = b.tris[3 * i0 + 2];
                    if (TWO_TRIS) { c0 = b.tris[3 * i1]; c1 = b.tris[3 * i1 + 1]; c2 = b.tris[3 * i1 + 2]; }
                }
                if (COUNT) n_tri += two ? 2 : 1;
                if (WIDE) {
                    // no early exits, one predicated update (bvh_wide.hpp); an odd leaf re-tests its last triangle, which cannot
                    // pass `t < best_t` a second time
                    tri_test_branchless(o, d, a0, a1, a2, i0, best);
                    if (TWO_TRIS) tri_test_branchless(o, d, c0, c1, c2, i1, best);
                } else {
                    tri_test(o, d, a0, a1, a2, i0, best);
                    if (TWO_TRIS && two) tri_test(o, d, c0, c1, c2, i1, best);
                }
            }
            if (tri_i >= tri_end && ((WIDE && wlane) ? (cur < 0 && sp == 0) : node >= n_nodes)) state = LANE_PEND;
        }
    }

    for (int off = 32; off > 0; off >>= 1) {
        n_rays += __shfl_down(n_rays, off);
        if (COUNT) { n_box += __shfl_down(n_box, off); n_tri += __shfl_down(n_tri, off); }
    }
    if (wave_lane() == 0) {
        atomicAdd(&stats->rays, (unsigned long long)n_rays);
        if (COUNT) {
            atomicAdd(&stats->box_tests, (unsigned long long)n_box);
            atomicAdd(&stats->tri_tests, (unsigned long long)n_tri);
            atomicAdd(&stats->counted_rays, (unsigned long long)n_rays);
        }
    }
}

// ---------------------------------------------------------------- connection stage
// cosine_geometry_term, trace.metal:539-544: uses the STORED outgoing directions.
__device__ __forceinline__ float geom_term(float cos_a, float cos_b, V3 oa, V3 ob) {
    const float dist = length3(ob - oa);
    return cos_a * cos_b / (dist * dist);
}

__device__ __forceinline__ int conn_slot(int t, int s) { return (t - 1) * 6 + (s - 1); }

// Cull test for one (t,s) pair, trace.metal:667-688 + :577-584; on success returns the ray.
struct ConnVtx { V3 o, n; int meta; };
// `l_specular` / `c_specular`: material type > 0 at the light / camera vertex (looked up once per vertex by the caller: as loads
// inside the 36-pair loop they were 68 dependent single-dword fetches per thread, each with its own wait).
__device__ __forceinline__ bool conn_ray(int t, const ConnVtx& lv, const ConnVtx& cv, bool l_specular, bool c_specular,
                                         V3 focal, V3 cam_dir, V3& dir) {
    if (l_specular) return false;
    if (t == 1) {
        dir = normalize(focal - lv.o);
        return !(dot(dir, cam_dir) > 0.0f);
    }
    if (c_specular) return false;
    dir = normalize(cv.o - lv.o);
    if (dot(lv.n, dir) < DELTA_F) return false;
    if (dot(cv.n, -dir) < DELTA_F) return false;
    return true;
}

// The `pid >= B` contract of kernels whose loads are decided per WAVE (`if (ballot(slot < len)) load slot`): the lanes behind
// the end of a frame that is not a multiple of 256 pixels take part in such a load although they own no pixel, and slot v of
// pixel >= B lies behind the allocation for v = MAX_VERTS - 1 (the out-of-bounds READ of round 3, found by the fuzz).  Such a
// kernel either indexes every wave-decided load with clamped_pid() -- the lanes without a pixel read pixel 0's records and
// never use them -- or returns for pid >= B before the first of them (k_connect_resolve).
__device__ __forceinline__ size_t clamped_pid(bool valid, int pid) { return valid ? (size_t)pid : (size_t)0; }

// Enumerate strategy pairs per pixel, emit connection rays into a compacted queue of 4-byte tags
// {slot, pixel}.  Tags are ordered wave-by-wave, slot-major inside a wave, so consecutive queue
// entries are the same (t,s) strategy of neighbouring pixels: the vertex gathers of
// k_traverse_conn are coalesced and its rays coherent.
__global__ __launch_bounds__(BLOCK, 6) void k_connect_setup(
        int B, PathBufs lp, PathBufs cp, const MaterialDev* __restrict__ mats, int n_mats, CameraRec cam,
        int* __restrict__ ctag, unsigned* __restrict__ ccount, unsigned long long* __restrict__ cmask) {
    __shared__ unsigned s_wave_total[WAVES_PER_BLOCK];
    __shared__ unsigned s_base;
    __shared__ int s_mtype[256];                           // material types (the table has at most 256 entries, cl2_upload_scene)
    for (int i = threadIdx.x; i < n_mats; i += BLOCK) s_mtype[i] = __float_as_int(mats[i].color_type.w);
    __syncthreads();
    const int pid = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = pid < B;
    const int Lc = valid ? cp.len[pid] : 0, Ll = valid ? lp.len[pid] : 0;
    const V3 focal = cam3(cam.focal_point), cam_dir = cam3(cam.direction);
    const int lane = lane_id(), wave = threadIdx.x >> 6;

    // The vertex loads are issued in batches -- the whole light subpath, then the camera subpath three vertices at a time -- and
    // not one vertex per `if (v < L)`: inside its own conditional every vertex was a memory round trip of its own (13 per pixel
    // in a kernel that does nothing but wait for memory; now 4).  A vertex slot is fetched when SOME lane of the wave has it.
    // The fetch of a slot is decided per WAVE, so lanes behind the end of the frame (the last workgroup of a frame that is not a
    // multiple of 256 pixels) take part in it: they read pixel 0's records (slot 5 of pixel >= B would lie behind the buffer).
    const size_t vB = (size_t)B, lpid = clamped_pid(valid, pid);
    float4 la[MAX_VERTS], lc[MAX_VERTS];
#pragma unroll
    for (int s = 0; s < MAX_VERTS; s++) {
        if (__builtin_amdgcn_ballot_w64(s < Ll) != 0ull) { la[s] = lp.P0[s * vB + lpid]; lc[s] = lp.P2[s * vB + lpid]; }
        else la[s] = lc[s] = make_float4(0, 0, 0, 0);
    }
    ConnVtx lv[MAX_VERTS];
    unsigned l_spec = 0;
#pragma unroll
    for (int s = 0; s < MAX_VERTS; s++) {
        if (s < Ll) {
            lv[s] = ConnVtx{v3(la[s]), v3(lc[s]), __float_as_int(lc[s].w)};
            if (s_mtype[lv[s].meta & 0xFF] > 0) l_spec |= 1u << s;
        } else lv[s] = ConnVtx{v3(0, 0, 0), v3(0, 0, 0), 0};
    }
    // pass 1: predicates
    unsigned long long mine = 0;
    unsigned wave_total = 0;
#pragma unroll
    for (int tb = 0; tb < MAX_VERTS; tb += 3) {
        float4 ca[3], cc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (__builtin_amdgcn_ballot_w64(tb + k < Lc) != 0ull) { ca[k] = cp.P0[(tb + k) * vB + lpid]; cc[k] = cp.P2[(tb + k) * vB + lpid]; }
            else ca[k] = cc[k] = make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int t = tb + k + 1;
            ConnVtx cv{v3(0, 0, 0), v3(0, 0, 0), 0};
            bool c_specular = false;
            if (t <= Lc) {
                cv = ConnVtx{v3(ca[k]), v3(cc[k]), __float_as_int(cc[k].w)};
                c_specular = s_mtype[cv.meta & 0xFF] > 0;
            }
#pragma unroll
            for (int s = 1; s <= MAX_VERTS; s++) {
                V3 dir;
                const bool pred = (t <= Lc) && (s <= Ll) && conn_ray(t, lv[s - 1], cv, (l_spec >> (s - 1)) & 1u, c_specular, focal, cam_dir, dir);
                if (pred) mine |= 1ull << conn_slot(t, s);
                wave_total += __popcll(__ballot(pred));
            }
        }
    }
    if (lane == 0) s_wave_total[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < WAVES_PER_BLOCK; w++) tot += s_wave_total[w];
        s_base = tot ? atomicAdd(ccount, tot) : 0u;
    }
    __syncthreads();
    unsigned running = s_base;
    for (int w = 0; w < wave; w++) running += s_wave_total[w];
    if (valid) cmask[pid] = mine;
    // pass 2: write the ray tags {slot, pixel}; the traversal kernels rebuild the ray from the two vertices.  (Round 5 measured
    // the alternative -- this pass writing {direction, tag}, 16 B per ray, so that the persistent walks' refill need not gather
    // the second vertex and normalise again: the walk gained 1.5 %, this kernel took 0.39 instead of 0.21 ms per sample for its
    // extra 12 B per ray; profiles/r05_direction_queue.patch.)
#pragma unroll
    for (int t = 1; t <= MAX_VERTS; t++) {
#pragma unroll
        for (int s = 1; s <= MAX_VERTS; s++) {
            const int slot = conn_slot(t, s);
            const bool pred = (mine >> slot) & 1ull;
            const unsigned long long m = __ballot(pred);
            if (pred) ctag[running + __popcll(m & ((1ull << lane) - 1ull))] = (slot << TAG_PID_BITS) | pid;
            running += __popcll(m);
        }
    }
}

// Closest hit for each connection ray; result scattered to chit[slot*B + pid] = {tri, t}.
// The ray of tag {slot=(t,s), pixel} starts at light vertex s-1 and points at the focal point
// (t == 1: projection onto the film, trace.metal:580-585) or at camera vertex t-1
// (visibility_test, trace.metal:180-184).
template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void k_traverse_conn(
        BvhView bvh, int B, const unsigned* __restrict__ count, const int* __restrict__ ctag,
        const float4* __restrict__ LP0, const float4* __restrict__ CP0, CameraRec cam,
        float2* __restrict__ chit, Stats* stats) {
    BvhLds lds{nullptr, nullptr};
    stage_bvh(lds, bvh);
    const unsigned n = *count;
    const V3 focal = cam3(cam.focal_point);
    unsigned nb = 0, nt = 0;
    for (unsigned j = blockIdx.x * BLOCK + threadIdx.x; j < n; j += gridDim.x * BLOCK) {
        const int tag = ctag[j];
        const int pid = tag & ((1 << TAG_PID_BITS) - 1), slot = (unsigned)tag >> TAG_PID_BITS;
        const int t = slot / 6 + 1, s = slot % 6 + 1;
        const V3 o = v3(LP0[(size_t)(s - 1) * B + pid]);
        V3 target = focal;
        if (t > 1) target = v3(CP0[(size_t)(t - 1) * B + pid]);
        const V3 d = normalize(target - o);
        const Hit h = closest_hit<COUNT>(lds, bvh, o, d, rcp3(d), nb, nt);
        chit_store(chit, B, slot, pid, h.tri, h.t);
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { nb += __shfl_down(nb, off); nt += __shfl_down(nt, off); }
        if (lane_id() == 0) {
            atomicAdd(&stats->box_tests, (unsigned long long)nb);
            atomicAdd(&stats->tri_tests, (unsigned long long)nt);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&stats->rays, (unsigned long long)n);
        atomicAdd(&stats->conn_rays, (unsigned long long)n);
        if (COUNT) atomicAdd(&stats->counted_rays, (unsigned long long)n);
    }
}

// Aggregator SoA rows: 0..8 weights[i][j] (row i*3+j), 9..11 total_contribution, 12 contrib_weight_sum.
constexpr int AGG_ROWS = 13;

}  // namespace cl2
#include "connect_resolve.hpp"
#ifdef CL2_TEST_VARIANT
#include "../../tests/connect_resolve_wide.hpp"      // test code: a second implementation of the resolve stage (cross-check only)
#endif
namespace cl2 {

// ---------------------------------------------------------------- K6: adaptive_finalize_samples
// trace.metal:981-1018 with identity sample bins (renderer.py:92-94).
__global__ __launch_bounds__(BLOCK) void k_finalize(int B, int W, int H, const float* __restrict__ agg,
                                                    float4* __restrict__ finalized, float* __restrict__ sample_w) {
    const int id = blockIdx.x * BLOCK + threadIdx.x;
    if (id >= B) return;
    const int pix = id % (W * H);                                     // entry -> pixel; its neighbours are entries of the same stream
    const size_t base = (size_t)(id - pix);
    V3 total = v3(0, 0, 0);
    float wsum = 0.0f;
    for (int i = -1; i < 2; i++) {
        for (int j = -1; j < 2; j++) {
            const int sx = (pix % W) + i, sy = (pix / W) + j;
            if (sx < 0 || sx >= W || sy < 0 || sy >= H) continue;
            const size_t k = base + (size_t)sy * W + sx;
            const float weight = agg[(size_t)((1 - i) * 3 + (1 - j)) * B + k];
            total = total + weight * v3(agg[(size_t)9 * B + k], agg[(size_t)10 * B + k], agg[(size_t)11 * B + k]);
            wsum += weight * agg[(size_t)12 * B + k];
        }
    }
    finalized[id] = make_float4(total.x, total.y, total.z, 1.0f);
    sample_w[id] = wsum;
}

// ---------------------------------------------------------------- reproducible light image (det_splat.hpp)
// `keys` / `slots`: the n records of one pass sorted stably by target entry, i.e. by (target, s, source pixel); slots without a
// contribution carry DET_NO_KEY and sort to the end.  The thread that finds the FIRST record of a target sums that target's run
// front to back and adds it to the light image: one writer per pixel, one order.  Runs are short (<= 6 contributions per source
// pixel, spread over the film).
__global__ __launch_bounds__(BLOCK) void k_det_gather(const unsigned* __restrict__ keys, const unsigned* __restrict__ slots, size_t n,
                                                      const float4* __restrict__ vals, float4* __restrict__ light_image) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const unsigned pix = keys[i];
    if (pix == ~0u) return;
    if (i > 0 && keys[i - 1] == pix) return;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t j = i; j < n && keys[j] == pix; j++) {
        const float4 v = vals[slots[j]];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4 l = light_image[pix];
    l.x += acc.x; l.y += acc.y; l.z += acc.z; l.w += acc.w;
    light_image[pix] = l;
}

__global__ __launch_bounds__(BLOCK) void k_iota(unsigned* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = (unsigned)i;
}

__device__ __forceinline__ float scrub(float x) {   // np.nan_to_num(x, posinf=0, neginf=0)
    return (x != x || __builtin_isinf(x)) ? 0.0f : x;
}

// ---------------------------------------------------------------- process_images, renderer.py:253-278
// Accumulators stay on the device: acc rows = summed_image rgb (0..2), summed_sample_weights (3),
// unidirectional rgb (4..6), summed_sample_counts as float (7).  The light image is zeroed for the
// next sample.
// `FB` = pixels of the frame, `streams` = samples per pixel in this pass (entries s*FB + id): added in stream order, each
// exactly as one process_images call adds one sample.
__global__ __launch_bounds__(BLOCK) void k_accumulate(int FB, int streams, const float4* __restrict__ finalized,
                                                      const float* __restrict__ sample_w, float4* __restrict__ light_image,
                                                      const float4* __restrict__ uni, float* __restrict__ acc) {
    const int id = blockIdx.x * BLOCK + threadIdx.x;
    if (id >= FB) return;
    float a[8];
#pragma unroll
    for (int c = 0; c < 8; c++) a[c] = acc[(size_t)c * FB + id];
#pragma unroll 1
    for (int s = 0; s < streams; s++) {
        const size_t e = (size_t)s * FB + id;
        const float4 f = finalized[e], l = light_image[e], u = uni[e];
        a[0] += scrub(l.x + f.x);
        a[1] += scrub(l.y + f.y);
        a[2] += scrub(l.z + f.z);
        a[3] += sample_w[e] + l.w;                      // K8's `sum_weights[id] += weight_sum`, :963
        a[4] += scrub(u.x);
        a[5] += scrub(u.y);
        a[6] += scrub(u.z);
        a[7] += 1.0f;
        light_image[e] = make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 8; c++) acc[(size_t)c * FB + id] = a[c];
}

// K6 + process_images in one launch for cl2_run_samples: the filtered sample goes straight from registers into the
// accumulators (the per-sample images finalized / sample_weights are not written: they are what the stage calls
// cl2_finalize_samples / cl2_process_images exchange, 36 B per pixel and a launch boundary per sample).
// Same statements in the same order as k_finalize followed by k_accumulate.
// `B` = entries (streams x W x H, the stride of the aggregator rows); one thread per PIXEL adds its streams in order.
__global__ __launch_bounds__(BLOCK) void k_finalize_accumulate(int B, int W, int H, const float* __restrict__ agg,
                                                               float4* __restrict__ light_image, const float4* __restrict__ uni,
                                                               float* __restrict__ acc) {
    const int id = blockIdx.x * BLOCK + threadIdx.x;
    const int FB = W * H;
    if (id >= FB) return;
    float a[8];
#pragma unroll
    for (int c = 0; c < 8; c++) a[c] = acc[(size_t)c * FB + id];
#pragma unroll 1
    for (size_t base = 0; base < (size_t)B; base += (size_t)FB) {
        V3 total = v3(0, 0, 0);
        float wsum = 0.0f;
        // one row of three neighbours in flight at a time: unrolled all the way the nine gathers take 136 VGPRs
#pragma unroll 1
        for (int i = -1; i < 2; i++) {
#pragma unroll
            for (int j = -1; j < 2; j++) {
                const int sx = (id % W) + i, sy = (id / W) + j;
                if (sx < 0 || sx >= W || sy < 0 || sy >= H) continue;
                const size_t k = base + (size_t)sy * W + sx;
                const float weight = agg[(size_t)((1 - i) * 3 + (1 - j)) * B + k];
                total = total + weight * v3(agg[(size_t)9 * B + k], agg[(size_t)10 * B + k], agg[(size_t)11 * B + k]);
                wsum += weight * agg[(size_t)12 * B + k];
            }
        }
        const float4 l = light_image[base + id], u = uni[base + id];
        a[0] += scrub(l.x + total.x);
        a[1] += scrub(l.y + total.y);
        a[2] += scrub(l.z + total.z);
        a[3] += wsum + l.w;
        a[4] += scrub(u.x);
        a[5] += scrub(u.y);
        a[6] += scrub(u.z);
        a[7] += 1.0f;
        light_image[base + id] = make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 8; c++) acc[(size_t)c * FB + id] = a[c];
}

// ---------------------------------------------------------------- exactness self-test
// Re-runs the proof behind rcp_exact / div_pi on this device: every one of the 2^32 binary32 inputs
// must give the bits of the IEEE operation.  out[0], out[1] = mismatch counts.
__global__ void k_selftest_exact_math(unsigned long long* out) {
    unsigned long long bad_rcp = 0, bad_pi = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float r0 = 1.0f / x, r1 = rcp_exact(x);
        const float p0 = x / PI_CONST, p1 = div_pi(x);
        const bool nan_r = r0 != r0 && r1 != r1, nan_p = p0 != p0 && p1 != p1;
        if (!nan_r && __float_as_uint(r0) != __float_as_uint(r1)) bad_rcp++;
        if (!nan_p && __float_as_uint(p0) != __float_as_uint(p1)) bad_pi++;
    }
    if (bad_rcp) atomicAdd(out, bad_rcp);
    if (bad_pi) atomicAdd(out + 1, bad_pi);
}

// Elementwise probe of the deterministic elementary functions (detmath.hpp) and of the exact
// reciprocal: which = 0 sin, 1 cos, 2 acos, 3 atan, 4 exp, 5 asin, 6 rcp_exact, 7 div_pi.
__global__ void k_probe_math(int which, size_t n, const float* __restrict__ in, float* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = in[i];
        float y;
        switch (which) {
            case 0: y = det_sinf(x); break;
            case 1: y = det_cosf(x); break;
            case 2: y = det_acosf(x); break;
            case 3: y = det_atanf(x); break;
            case 4: y = det_expf(x); break;
            case 5: y = det_asinf(x); break;
            case 6: y = rcp_exact(x); break;
            default: y = div_pi(x); break;
        }
        out[i] = y;
    }
}

// Elementwise probe of the bounce routines (bsdf.hpp).  in: 12 floats per item {wi.xyz, n.xyz, rx, ry,
// ni, no, alpha, kind} with kind 0 diffuse, 1 reflect, 2 transmit, 3 GGX_sample (the microfacet normal m is
// GGX_sample(n, rx, ry, alpha) in every case, as in generate_paths); out: 8 floats {wo.xyz, f, c_p, l_p,
// fresnel(wi,m), m.x}.  `from_camera` selects the pdf direction convention.
__global__ void k_probe_bounce(size_t n, int from_camera, const float* __restrict__ in, float* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float* p = in + 12 * i;
        const V3 wi = v3(p[0], p[1], p[2]), nn = v3(p[3], p[4], p[5]);
        const float rx = p[6], ry = p[7], ni = p[8], no = p[9], alpha = p[10];
        const int kind = (int)p[11];
        const V3 m = GGX_sample(nn, rx, ry, alpha);
        Bounce b{v3(0, 0, 0), 1.0f, 1.0f, 1.0f};
        if (kind == 0) b = diffuse_bounce(wi, nn, from_camera != 0, rx, ry);
        else if (kind == 1) b = reflect_bounce(wi, nn, m, ni, no, alpha, from_camera != 0);
        else if (kind == 2) b = transmit_bounce(wi, nn, m, ni, no, alpha, from_camera != 0);
        else b.wo = m;
        float* q = out + 8 * i;
        q[0] = b.wo.x; q[1] = b.wo.y; q[2] = b.wo.z; q[3] = b.f; q[4] = b.c_p; q[5] = b.l_p;
        q[6] = degreve_fresnel(wi, m, ni, no); q[7] = m.x;
    }
}

// ---------------------------------------------------------------- debug exports (reference AoS)
struct RayRec {   // struct Ray, trace.metal:7-23
    float origin[4], direction[4], inv_direction[4], color[4], normal[4];
    int material, triangle;
    float c_importance, l_importance, tot_importance;
    int hit_light, from_camera, hit_camera, pixel_idx;
    int pad[3];
};
static_assert(sizeof(RayRec) == 128, "Ray record");

__device__ __forceinline__ void fill_ray(RayRec& r, const PathBufs& pb, size_t k, int from_camera, int pixel_idx) {
    const float4 a = pb.P0[k], b = pb.P1[k], c = pb.P2[k], d = pb.P3[k];
    const int meta = __float_as_int(c.w), tri = pb.tri[k];
    r.origin[0] = a.x; r.origin[1] = a.y; r.origin[2] = a.z; r.origin[3] = 0;
    r.direction[0] = b.x; r.direction[1] = b.y; r.direction[2] = b.z; r.direction[3] = 0;
    r.inv_direction[0] = 1.0f / b.x; r.inv_direction[1] = 1.0f / b.y; r.inv_direction[2] = 1.0f / b.z; r.inv_direction[3] = 0;
    r.color[0] = d.x; r.color[1] = d.y; r.color[2] = d.z; r.color[3] = 0;
    r.normal[0] = c.x; r.normal[1] = c.y; r.normal[2] = c.z; r.normal[3] = 0;
    r.material = meta & 0xFF; r.triangle = tri;
    r.c_importance = a.w; r.l_importance = b.w; r.tot_importance = d.w;
    r.hit_light = (meta & META_HIT_LIGHT) ? tri : -1;
    r.from_camera = from_camera;
    r.hit_camera = (meta & META_HIT_CAMERA) ? tri : -1;
    r.pixel_idx = pixel_idx;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
}

// exports of ONE sample stream: `n` = pixels of the frame, `off` = first entry of the stream, `B` = entries (row stride)
__global__ void k_export_rays(int n, int off, PathBufs pb, int from_camera, RayRec* out) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    RayRec r;
    fill_ray(r, pb, (size_t)off + id, from_camera, from_camera ? id : 0);
    out[id] = r;
}

// Path record = 8 RayRec + length + from_camera + pad[2] (1040 B).  Slots >= length are zero.
__global__ void k_export_paths(int n, int off, int B, PathBufs pb, int from_camera, unsigned char* out) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    unsigned char* rec = out + (size_t)id * 1040;
    const int len = pb.len[off + id];
    for (int v = 0; v < 8; v++) {
        RayRec r;
        if (v < len) fill_ray(r, pb, (size_t)v * B + off + id, v == 0 ? from_camera : 0, (v == 0 && from_camera) ? id : 0);
        else memset(&r, 0, sizeof r);
        *reinterpret_cast<RayRec*>(rec + 128 * v) = r;
    }
    int* tail = reinterpret_cast<int*>(rec + 1024);
    tail[0] = len; tail[1] = from_camera; tail[2] = 0; tail[3] = 0;
}

// WeightAggregator records at the reference's 128-byte host stride (renderer.py:71).
__global__ void k_export_aggregators(int n, int off, int B, const float* __restrict__ agg, float* out) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    float* rec = out + (size_t)id * 32;
    const size_t e = (size_t)off + id;
    for (int i = 0; i < 32; i++) rec[i] = 0.0f;
    for (int r = 0; r < 9; r++) rec[r] = agg[(size_t)r * B + e];
    rec[12] = agg[(size_t)9 * B + e]; rec[13] = agg[(size_t)10 * B + e]; rec[14] = agg[(size_t)11 * B + e];
    rec[16] = agg[(size_t)12 * B + e];
}

}  // namespace cl2
