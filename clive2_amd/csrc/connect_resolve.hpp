// connect_resolve.hpp -- MIS weights and contributions of every (t,s) strategy pair of one pixel.
//
// Reference: the strategy loop of `connect_paths` (src/trace.metal:649-825) with its two BVH
// queries replaced by the results of k_traverse_conn, the filter-weight epilogue (:827-868), the
// light-image gather K8 (:937-964, here a float-atomic splat) and the unidirectional estimate of
// `generate_paths` (:523-528).
//
// Formulation.  For a pair (t,s) the unified path is x_0..x_k, k = s+t-1, x_i = light[i] for i < s
// and camera[t+s-i-1] otherwise (get_ray, :546-549).  The reference rebuilds all k+1 pdf ratios
//     r_0 = l_0 / (c_0 G_01)      r_i = (l_i G_{i-1,i}) / (c_i G_{i,i+1})      r_k = (l_k G_{k,k-1}) / c_k
// for every pair (:709-735).  All but the (at most) two ratios that touch the connection edge
// depend on one subpath only, so they are evaluated ONCE per pixel with exactly the reference's
// operations -- RL[i] for the light side, RC[m] for the camera side -- and each pair only computes
// the junction geometry term, the two junction ratios and the two running products (:745-755).
// Same float operations on the same operands in the same order => the aggregator is reproduced
// exactly; per-pixel divisions drop from ~400 to ~250 and every path vertex is read from HBM once
// (light subpath: registers, statically indexed because the s-loop is unrolled) or twice (camera).
#pragma once
#include "vecmath.hpp"
#include "bsdf.hpp"

namespace cl2 {

// The triangles of the camera quad (is_camera, scene.py): the t = 1 pairs ask whether the triangle their ray hit is one of
// them.  As a look-up in the shading records (tri_shade[4 i + 2].w) that was a dependent load -- a memory round trip of its
// own -- inside each of the six t = 1 pairs; the reference's scenes have two such triangles, which travel as kernel arguments.
// More than CAM_TRI_ARGS of them: n < 0 and the look-up stays.
constexpr int CAM_TRI_ARGS = 4;
struct CamTris { int n; int idx[CAM_TRI_ARGS]; };
__device__ __forceinline__ bool is_camera_tri(const CamTris& ct, const float4* __restrict__ tri_shade, int i) {
    if (ct.n < 0) return __float_as_int(tri_shade[4 * i + 2].w) != 0;
    bool hit = false;
#pragma unroll
    for (int k = 0; k < CAM_TRI_ARGS; k++) hit = hit || (k < ct.n && i == ct.idx[k]);
    return hit;
}

struct LightVtx {          // the part of a light vertex every pair touches: registers
    V3 o;
    float c, l, tot, cosv;
    int tri, meta;
};
// Normal and colour of the light vertices are read once per contributing pair only: they live in LDS
// ([word][thread], conflict-free) so the kernel fits 3 waves per SIMD instead of 2.
__device__ __forceinline__ V3 lds_v3(const float* base, int v) {
    return v3(base[(3 * v + 0) * BLOCK + threadIdx.x], base[(3 * v + 1) * BLOCK + threadIdx.x], base[(3 * v + 2) * BLOCK + threadIdx.x]);
}

// One strategy pair with s = S (compile time).  Returns true when a contribution was produced.
// DET (cl2_set_reproducible): a t = 1 contribution is not added to the light image with float atomics but WRITTEN as a record
// {target pixel, source slot} / {c.xyz, w} at slot (S-1) * B + pid -- the reference's own scatter `id + s * total_pixels`
// (trace.metal:817-823) -- and summed per target pixel in a fixed order afterwards (det_splat.hpp).
template <int S, bool DET>
__device__ __forceinline__ void resolve_pair(
        int t, int B, int pid, const LightVtx (&lv)[MAX_VERTS], const float (&GL)[MAX_VERTS], const float (&RL)[MAX_VERTS],
        unsigned l_spec, unsigned c_spec, bool spec7,
        // camera junction vertex t-1 and per-path camera tables (LDS)
        V3 c_o_in, V3 c_n_in, float c_c, float c_l, float c_tot_in, float c_cos_in, int c_tri, int c_meta,
        V3 prior_camera_color, const float* GCs, const float* RCs /* [m*BLOCK + tid] */, const float* LNs, const float* LCs,
        unsigned long long mask, float2 h, const float4* __restrict__ tri_shade, const CamTris& cam_tris,
        const MaterialDev* __restrict__ mats, const CameraRec& cam, V3 focal, V3 cam_dir,
        V3& total, float& contrib_weight_sum, float4* __restrict__ light_image, float* splat_tab, int debug_flags,
        unsigned* __restrict__ det_keys, float4* __restrict__ det_vals) {
    const int tid = threadIdx.x;
    V3 c_o = c_o_in, c_n = c_n_in;
    float c_tot = c_tot_in, c_cos = c_cos_in;
    V3 dir_l_to_c = v3(0, 0, 0);
    int light_pixel_idx = -1;
    float Gj = 0.0f;

    if (S == 0) {
        if (!(c_meta & META_HIT_LIGHT)) return;                               // :665
    } else {
        if (!((mask >> conn_slot(t, S)) & 1ull)) return;                      // culled in k_connect_setup
        const LightVtx& a = lv[S - 1];
        const int best_i = __float_as_int(h.x);
        if (best_i == -1) return;                                             // :193 / :593
        if (t == 1) {
            // world_ray_to_camera_ray, :595-616
            if (!is_camera_tri(cam_tris, tri_shade, best_i)) return;         // !is_camera
            const V3 tdir = normalize(focal - a.o);
            const V3 camera_point = a.o + h.y * tdir;
            const float x = dot(camera_point - cam3(cam.center), cam3(cam.dx));
            const float y = dot(camera_point - cam3(cam.center), cam3(cam.dy));
            const int pixel_x = (int)__builtin_roundf((x / cam.phys_width + 0.5f) * cam.pixel_width);
            const int pixel_y = (int)__builtin_roundf((y / cam.phys_height + 0.5f) * cam.pixel_height);
            light_pixel_idx = pixel_y * cam.pixel_width + pixel_x;
            if (light_pixel_idx == -1) return;                                // :671
            c_o = camera_point;
            const V3 cdir = normalize(focal - camera_point);
            c_n = cam_dir;
            c_cos = __builtin_fabsf(dot(cdir, cam_dir));
            c_tot = 1.0f;
        } else {
            if (best_i == a.tri) return;                                      // visibility_test, :194-196
            if (best_i != c_tri) return;
        }
        dir_l_to_c = normalize(c_o - a.o);
        Gj = geom_term(a.cosv, c_cos, a.o, c_o);
    }

    // ---- p_s and the two running products (:737-757) ----
    const float p_s = c_tot * ((S == 0) ? 1.0f : lv[S > 0 ? S - 1 : 0].tot);
    // backward (light side): p[i] = p[i+1] / r_i for i = S-1 .. 0
    float pb[MAX_VERTS > 0 ? MAX_VERTS : 1];
    if (S > 0) {
        const LightVtx& a = lv[S - 1];
        const float r_junc = (S == 1) ? a.l / (a.c * Gj) : (a.l * GL[S - 2]) / (a.c * Gj);
        float v = p_s / r_junc;
        pb[S - 1] = v;
#pragma unroll
        for (int i = S - 2; i >= 0; i--) { v = v / RL[i]; pb[i] = v; }
    }
    // specular zeroing (:759-764): p[i] is zeroed when x_i or x_{i-1} is specular
    auto spec_at = [&](int i) -> bool {        // unified index -> material type > 0
        if (i < S) return (l_spec >> i) & 1u;
        const int m = t + S - i - 1;
        if (t == 1 && S > 0) return spec7;       // projected camera vertex carries material 7 (:611)
        return (c_spec >> m) & 1u;
    };
    float sum = 0.0f;
    bool prev_spec = false;
#pragma unroll
    for (int i = 0; i < S; i++) {
        const bool sp = (l_spec >> i) & 1u;
        sum += (sp || prev_spec) ? 0.0f : pb[i];
        prev_spec = sp;
    }
    // i = S: p[S] = p_s
    bool sp_s = spec_at(S);
    const float p_at_s = (sp_s || prev_spec) ? 0.0f : p_s;
    sum += p_at_s;
    prev_spec = sp_s;
    // forward (camera side): p[i+1] = r_i * p[i], i = S .. S+t-2 useful (p[S+t] is overwritten by 0, :766)
    {
        float v = p_s;
        for (int i = S; i < S + t - 1; i++) {
            float r;
            if (i == S) {
                if (S == 0) r = c_l / (c_c * GCs[(t - 2) * BLOCK + tid]);            // i == 0 form, x_1 = camera[t-2]
                else r = (c_l * Gj) / (c_c * GCs[(t - 2) * BLOCK + tid]);           // interior form at the junction
            } else {
                r = RCs[(t + S - i - 1) * BLOCK + tid];
            }
            v = r * v;
            const bool sp = spec_at(i + 1);
            sum += (sp || prev_spec) ? 0.0f : v;
            prev_spec = sp;
        }
    }
    sum += 0.0f;                                                              // p[S+t] = 0
    if (!(p_at_s > 0.0f && sum > 0.0f)) return;                               // :773-776
    const float w = p_at_s / sum;

    if (S == 0) {                                                             // :783-786
        const V3 emission = v3(mats[c_meta & 0xFF].emission_alpha);
        const V3 color = prior_camera_color * emission;
        total = total + ((w * 1.0f) * color) / p_s;
        contrib_weight_sum += w;
    } else if (t == 1) {                                                      // :787-793, :817-823, K8 :952-961
        const LightVtx& a = lv[S - 1];
        const V3 prior_color = lds_v3(LCs, (S - 2) > 0 ? (S - 2) : 0);
        float new_light_f = 1.0f;
        if (S > 1) new_light_f = div_pi(__builtin_fabsf(dot(dir_l_to_c, lds_v3(LNs, S - 1))));
        const V3 mcol = v3(mats[a.meta & 0xFF].color_type);
        const float shade = new_light_f * Gj / p_s;
#ifdef CL2_TEST_VARIANT
        const bool splat_on = !(debug_flags & 1);          // timing dissection (test variant only): no splat = invalid render
#else
        const bool splat_on = true;
#endif
        const int frame_pixels = cam.pixel_width * cam.pixel_height;
        if (light_pixel_idx >= 0 && light_pixel_idx < frame_pixels && splat_on) {
            const V3 c = ((w * shade) * prior_color) * mcol;
            // the light image of THIS entry's sample stream (computed here, in the rare branch, not carried through the kernel:
            // it runs at 162 of 168 VGPRs)
            const int splat_idx = (pid - pid % frame_pixels) + light_pixel_idx;
            if (DET) {
                const size_t rec = (size_t)(S > 0 ? S - 1 : 0) * B + pid;
                det_vals[rec] = make_float4(c.x, c.y, c.z, w);
                det_keys[rec] = (unsigned)splat_idx;
                return;
            }
            // Splat {c.xyz, w} into light_image[pixel] (float4).  The lanes that reach this point
            // exchange their 4 values through a per-wave LDS table so that four CONSECUTIVE lanes
            // add the four components of one pixel: each atomic wave-instruction then carries whole
            // 16-byte segments instead of 64 unrelated dwords (4x fewer memory-side requests).
            float* tab = splat_tab + (threadIdx.x >> 6) * (5 * 64);
            const unsigned long long here = __ballot(true);
            const int n_here = __popcll(here), k = __popcll(here & ((1ull << (threadIdx.x & 63)) - 1ull));
            tab[0 * 64 + k] = __int_as_float(splat_idx);
            tab[1 * 64 + k] = c.x; tab[2 * 64 + k] = c.y; tab[3 * 64 + k] = c.z; tab[4 * 64 + k] = w;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int item = k + j * n_here, slot = item >> 2, comp = item & 3;
                const int pix = __float_as_int(tab[slot]);
                const float val = tab[(1 + comp) * 64 + slot];
                atomicAdd(reinterpret_cast<float*>(&light_image[pix]) + comp, val);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    } else {                                                                  // :794-816
        const LightVtx& a = lv[S - 1];
        const MaterialDev cmat = mats[c_meta & 0xFF];
        const float new_camera_f = div_pi(__builtin_fabsf(dot(-dir_l_to_c, c_n)));
        const V3 camera_color = (prior_camera_color * new_camera_f) * v3(cmat.color_type);
        V3 light_color;
        if (S == 1) light_color = v3(mats[a.meta & 0xFF].emission_alpha);
        else {
            const V3 prior_light_color = lds_v3(LCs, S >= 2 ? S - 2 : 0);
            const float new_light_f = div_pi(__builtin_fabsf(dot(dir_l_to_c, lds_v3(LNs, S - 1))));
            light_color = (prior_light_color * new_light_f) * v3(mats[a.meta & 0xFF].color_type);
        }
        const V3 color = camera_color * light_color;
        total = total + ((w * Gj) * color) / p_s;
        contrib_weight_sum += w;
    }
}

template <int WAVES_PER_SIMD, bool MATS_LDS, bool DET = false>
__global__ __launch_bounds__(BLOCK, WAVES_PER_SIMD) void k_connect_resolve(
        int B, PathBufs lp, PathBufs cp, const MaterialDev* __restrict__ mats_g, int n_mats,
        const float4* __restrict__ tri_shade, CamTris cam_tris, CameraRec cam, const unsigned long long* __restrict__ cmask,
        const float2* __restrict__ chit, float* __restrict__ agg, float4* __restrict__ light_image,
        float4* __restrict__ uni_out, Stats* stats, int debug_flags,
        unsigned* __restrict__ det_keys, float4* __restrict__ det_vals) {
    __shared__ float GCs[(MAX_VERTS - 1) * BLOCK];     // GC[v] = G(camera[v], camera[v+1])
    __shared__ float RCs[(MAX_VERTS - 1) * BLOCK];     // RC[m]: ratio of camera vertex m with both neighbours on the camera side
    __shared__ float LNs[3 * MAX_VERTS * BLOCK];       // light vertex normals
    __shared__ float LCs[3 * MAX_VERTS * BLOCK];       // light vertex colours (throughput numerators)
    __shared__ float splat_tab[WAVES_PER_BLOCK * 5 * 64];   // per-wave exchange table of the light-image splat
    // The material table in LDS (up to LDS_MAT_CAP entries; the reference ships 8): every contributing pair reads a colour or an
    // emission by the material index of a vertex -- as global loads those were two dependent fetches inside each pair's code.
    __shared__ MaterialDev s_mats[LDS_MAT_CAP];
    const int tid = threadIdx.x;
    if (MATS_LDS) {
        for (int i = tid; i < n_mats; i += BLOCK) s_mats[i] = mats_g[i];
        __syncthreads();                               // before any thread leaves
    }
    const MaterialDev* __restrict__ mats = MATS_LDS ? s_mats : mats_g;     // compile-time choice: plain LDS reads, not generic ones
    const int pid = blockIdx.x * BLOCK + tid;           // entry: pixel pid % (W*H) of sample stream pid / (W*H) (kernels.hpp, header)
    // The `pid >= B` contract (clamped_pid, kernels.hpp): the light-subpath fetches below are decided per WAVE, so every lane
    // that reaches them must own a pixel.  This return guarantees it -- it must stay in FRONT of the first such fetch, and no
    // block-wide barrier may follow it (the LDS rows are private per thread, the splat exchange is per wave).
    if (pid >= B) return;
    __builtin_assume(pid < B);
    const int Lc = cp.len[pid], Ll = lp.len[pid];
    const unsigned long long mask = cmask[pid];
    const V3 focal = cam3(cam.focal_point), cam_dir = cam3(cam.direction);
    const bool spec7 = __float_as_int(mats[7].color_type.w) > 0;

    // ---- camera subpath: its per-path tables (GC, RC, specular bits) are built incrementally inside the
    // t loop below from the vertex that iteration loads anyway -- pair (t,s) only needs GC[0..t-2],
    // RC[0..t-2] and the specular bits of camera[0..t-1] -- so the camera subpath is read once. ----
    unsigned c_spec = 0;
    bool light_hit_seen = false;                     // a camera vertex v >= 1 with hit_light was met
    V3 cam_prev_o = v3(0, 0, 0);
    float cam_prev_cos = 0.0f, cam_prev_l = 0.0f, cam_prev_c = 0.0f, cam_prev_G = 0.0f;

    // ---- the whole light subpath is fetched back to back (with each vertex's loads inside its own `if (v < Ll)` the kernel
    // made a memory round trip per light vertex), and each iteration of the t loop fetches its camera vertex and the
    // closest-hit results of its pairs together (they were two round trips).  With the is_camera flags as kernel arguments
    // that is 8 dependent round trips per pixel instead of 25.  Measured: resolve 0.838 -> 0.822 ms -- the kernel waits for
    // its dependent ARITHMETIC (division chains at three waves per SIMD) more than for memory; fetching the first camera
    // vertex up here as well costs 168 VGPRs + 16 B of scratch and is slower (0.87 ms).  A vertex slot is fetched when SOME
    // lane of the wave has it (open scenes: most subpaths are short). ----
    float4 La[MAX_VERTS], Lb[MAX_VERTS], Lcn[MAX_VERTS], Ld[MAX_VERTS];
    int Lt[MAX_VERTS];
#pragma unroll
    for (int v = 0; v < MAX_VERTS; v++) {
        if (__builtin_amdgcn_ballot_w64(v < Ll) != 0ull) {
            const size_t k = (size_t)v * B + pid;
            La[v] = lp.P0[k]; Lb[v] = lp.P1[k]; Lcn[v] = lp.P2[k]; Ld[v] = lp.P3[k]; Lt[v] = lp.tri[k];
        } else {
            La[v] = Lb[v] = Lcn[v] = Ld[v] = make_float4(0, 0, 0, 0); Lt[v] = -1;
        }
    }
    float4 cP0 = make_float4(0, 0, 0, 0), cP1 = cP0, cP2 = cP0, cP3 = cP0;
    int c_tri = -1;
    float2 hits[MAX_VERTS + 1];
    auto load_camera_vertex = [&](int t) {         // camera vertex t-1 and the results of the connection rays that end at it
        const size_t ck = (size_t)(t - 1) * B + pid;
        cP0 = cp.P0[ck]; cP1 = cp.P1[ck]; cP2 = cp.P2[ck]; cP3 = cp.P3[ck];
        c_tri = cp.tri[ck];
#pragma unroll
        for (int s = 1; s <= MAX_VERTS; s++) hits[s] = chit_load(chit, B, t, s, pid);
    };
    hits[0] = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int s = 1; s <= MAX_VERTS; s++) hits[s] = make_float2(0.0f, 0.0f);

    // ---- light subpath -> registers; adjacent geometry terms and interior ratios ----
    LightVtx lv[MAX_VERTS];
    float GL[MAX_VERTS], RL[MAX_VERTS];
    unsigned l_spec = 0;
#pragma unroll
    for (int v = 0; v < MAX_VERTS; v++) {
        lv[v] = LightVtx{v3(0, 0, 0), 0.0f, 0.0f, 0.0f, 0.0f, -1, 0};
        GL[v] = 0.0f; RL[v] = 0.0f;
        if (v < Ll) {
            const float4 a = La[v], b = Lb[v], c = Lcn[v], d = Ld[v];
            lv[v].o = v3(a);
            LNs[(3 * v + 0) * BLOCK + tid] = c.x; LNs[(3 * v + 1) * BLOCK + tid] = c.y; LNs[(3 * v + 2) * BLOCK + tid] = c.z;
            LCs[(3 * v + 0) * BLOCK + tid] = d.x; LCs[(3 * v + 1) * BLOCK + tid] = d.y; LCs[(3 * v + 2) * BLOCK + tid] = d.z;
            lv[v].c = a.w; lv[v].l = b.w; lv[v].tot = d.w;
            lv[v].cosv = __builtin_fabsf(dot(v3(b), v3(c)));
            lv[v].tri = Lt[v];
            lv[v].meta = __float_as_int(c.w);
            if (__float_as_int(mats[lv[v].meta & 0xFF].color_type.w) > 0) l_spec |= 1u << v;
        }
    }
#pragma unroll
    for (int v = 0; v + 1 < MAX_VERTS; v++)
        if (v + 1 < Ll) GL[v] = geom_term(lv[v].cosv, lv[v + 1].cosv, lv[v].o, lv[v + 1].o);
#pragma unroll
    for (int i = 0; i + 1 < MAX_VERTS; i++) {
        if (i + 1 < Ll) {
            if (i == 0) RL[0] = lv[0].l / (lv[0].c * GL[0]);
            else RL[i] = (lv[i].l * GL[i - 1]) / (lv[i].c * GL[i]);
        }
    }

    V3 total = v3(0, 0, 0);
    float contrib_weight_sum = 0.0f;
    float4 uni = make_float4(0, 0, 0, 0);

    V3 prior_camera_color = v3(0, 0, 0);            // rays[t-2].color: the previous iteration's P3 (read once, not again)
    for (int t = 1; t < Lc + 1; t++) {
        prior_camera_color = v3(cP3); load_camera_vertex(t);
        const int c_meta = __float_as_int(cP2.w);
        const V3 c_o = v3(cP0), c_n = v3(cP2);
        const float c_cos = __builtin_fabsf(dot(v3(cP1), c_n));
        // per-path camera tables, vertex v = t-1 (same statements as the light side above)
        {
            const int v = t - 1;
            if (__float_as_int(mats[c_meta & 0xFF].color_type.w) > 0) c_spec |= 1u << v;
            if (v > 0) {
                const float G = geom_term(cam_prev_cos, c_cos, cam_prev_o, c_o);           // GC[v-1]
                GCs[(v - 1) * BLOCK + tid] = G;
                const int m = v - 1;                       // ratio of camera vertex m, its far neighbour now known
                RCs[m * BLOCK + tid] = (m == 0) ? (cam_prev_l * G) / cam_prev_c : (cam_prev_l * G) / (cam_prev_c * cam_prev_G);
                cam_prev_G = G;
            }
            cam_prev_o = c_o; cam_prev_cos = c_cos; cam_prev_l = cP1.w; cam_prev_c = cP0.w;
        }
        // unidirectional estimate of generate_paths (camera pass), trace.metal:523-528: first stored
        // vertex v >= 1 with hit_light -> rays[v-1].color / rays[v].tot_importance; both are this iteration's loads
        if (t >= 2 && !light_hit_seen && (c_meta & META_HIT_LIGHT)) {
            light_hit_seen = true;
            const V3 c = prior_camera_color / cP3.w;
            uni = make_float4(c.x, c.y, c.z, 1.0f);
        }
#ifdef CL2_TEST_VARIANT
        if ((debug_flags & 2) && t >= 2) continue;      // timing dissection (test variant only): skip the t >= 2 / t == 1 pairs
        if ((debug_flags & 4) && t == 1) continue;
#endif
#define CL2_PAIR(S)                                                                                             \
        if ((S) <= Ll && t + (S) >= 2)                                                                          \
            resolve_pair<S, DET>(t, B, pid, lv, GL, RL, l_spec, c_spec, spec7, c_o, c_n, cP0.w, cP1.w, cP3.w, c_cos, \
                            c_tri, c_meta, prior_camera_color, GCs, RCs, LNs, LCs, mask, hits[S], tri_shade, cam_tris, mats, cam, \
                            focal, cam_dir, total, contrib_weight_sum, light_image, splat_tab, debug_flags, det_keys, det_vals)
        CL2_PAIR(0); CL2_PAIR(1); CL2_PAIR(2); CL2_PAIR(3); CL2_PAIR(4); CL2_PAIR(5); CL2_PAIR(6);
#undef CL2_PAIR
    }

    // ---- reconstruction-filter weights, trace.metal:827-862.  A zero-length camera path is the
    // reference's zero-filled Path: pixel 0, film point (0,0,0). ----
    const int pixel_idx = (Lc > 0) ? pid % (cam.pixel_width * cam.pixel_height) : 0;
    V3 film = v3(0, 0, 0);
    if (Lc > 0) film = v3(cp.P0[pid]);
    const float ppw = cam.phys_width / cam.pixel_width, pph = cam.phys_height / cam.pixel_height;
    const float sigma = 0.5f * __builtin_sqrtf(ppw * ppw + pph * pph);
    float wts[9];
    float weight_sum = 0.0f;
#pragma unroll
    for (int i = -1; i < 2; i++) {
#pragma unroll
        for (int j = -1; j < 2; j++) {
            wts[(i + 1) * 3 + (j + 1)] = 0.0f;
            const int nx = (pixel_idx % cam.pixel_width) + i, ny = (pixel_idx / cam.pixel_width) + j;
            if (nx < 0 || nx >= cam.pixel_width || ny < 0 || ny >= cam.pixel_height) continue;
            // pixel_center, trace.metal:551-562 (no +0.5)
            const float xn = (nx - 0.5f * cam.pixel_width) / (float)cam.pixel_width;
            const float yn = (ny - 0.5f * cam.pixel_height) / (float)cam.pixel_height;
            const V3 pc = (cam3(cam.center) + (xn * cam.phys_width) * cam3(cam.dx)) + (yn * cam.phys_height) * cam3(cam.dy);
            const float dist = length3(pc - film);
            const float wgt = det_expf(-dist * dist / (2.0f * sigma * sigma));
            wts[(i + 1) * 3 + (j + 1)] = wgt;
            weight_sum += wgt;
        }
    }
#pragma unroll
    for (int r = 0; r < 9; r++) agg[(size_t)r * B + pid] = (weight_sum != 0.0f) ? wts[r] / weight_sum : wts[r];
    agg[(size_t)9 * B + pid] = total.x;
    agg[(size_t)10 * B + pid] = total.y;
    agg[(size_t)11 * B + pid] = total.z;
    agg[(size_t)12 * B + pid] = contrib_weight_sum;

    uni_out[pid] = uni;
}

}  // namespace cl2
