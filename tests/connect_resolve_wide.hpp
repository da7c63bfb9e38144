// connect_resolve_wide.hpp -- the resolve stage with one WAVE per camera vertex.
//
// Same arithmetic as connect_resolve.hpp (see there for the formulation: per-subpath pdf ratios
// evaluated once, per pair only the junction terms and the two running products), different
// distribution of the work.  The one-thread-per-pixel kernel walks its 42 strategy pairs one after
// the other, holds the whole light subpath in registers (167 VGPRs) and 204 bytes of LDS per thread:
// three waves per SIMD and long dependent divide chains -- PMC: 55 % of its wave cycles wait, the
// VALUs are half idle.  Here a workgroup of six waves handles 64 pixels; wave w owns camera vertex
// t = w + 1 of those pixels and evaluates its seven pairs (t, s = 0..6).  The light subpath of the 64
// pixels sits in LDS, shared by the six waves (23 KB per workgroup), the per-subpath tables are built
// in parallel (one geometry term and one ratio per wave and side), and a wave needs registers for ONE
// camera vertex only (95 VGPRs, 5 waves per SIMD, 42 KB of LDS per workgroup).
//
// MEASURED SLOWER than the one-thread-per-pixel kernel (1.14 vs 0.93 ms at 1080p, 15 % more VALU
// instructions, waits unchanged at 58 % of the wave cycles): the six waves of a workgroup move through
// their phases in lockstep -- eight workgroup barriers -- so a CU holds three phase streams instead of
// twelve independent waves, and that costs more latency hiding than the parallel chains win.  Kept as
// a second, structurally different implementation of the same arithmetic: it agrees with the oracle
// bit for bit (tests run it through debug_flags bits 4-6 = 7) and is not used by default.
//
// Exactness.  The reference adds the contributions of the pairs to ONE running total in (t, s) order
// (trace.metal:783-816); float addition does not associate, so the waves do not add partial sums:
// every wave keeps the values its pairs would add, and the running total is relayed through LDS from
// wave 0 to wave 5, each adding its values in s order.  Same additions, same order, same result.
#pragma once


namespace cl2 {

constexpr int RW_PIX = 64;                     // pixels per workgroup = lanes of a wave
constexpr int RW_BLOCK = RW_PIX * MAX_VERTS;   // 384 threads: wave w <-> camera vertex t = w + 1

enum RwWord { RW_OX = 0, RW_OY, RW_OZ, RW_C, RW_L, RW_TOT, RW_COS, RW_TRI, RW_META, RW_NX, RW_NY, RW_NZ, RW_KX, RW_KY, RW_KZ, RW_WORDS };

struct ResolveLds {
    float lv[RW_WORDS * MAX_VERTS * RW_PIX];       // light vertices: [word][vertex][pixel]
    float GL[(MAX_VERTS - 1) * RW_PIX], RL[(MAX_VERTS - 1) * RW_PIX];     // light-side tables
    float GC[(MAX_VERTS - 1) * RW_PIX], RC[(MAX_VERTS - 1) * RW_PIX];     // camera-side tables
    float cv[4 * MAX_VERTS * RW_PIX];              // camera vertices, what a neighbour needs: [o.xyz, cos][vertex][pixel]
    int lflag[MAX_VERTS * RW_PIX], cflag[MAX_VERTS * RW_PIX];   // specular (bit 0) / hit_light (bit 1, camera side) per vertex
    float relay[4 * RW_PIX];                       // running {total.xyz, contrib_weight_sum}
    float splat_tab[5 * 64];                       // exchange table of the light-image splat (wave 0 only)
};

__device__ __forceinline__ float rw_f(const ResolveLds& L, int word, int v, int lane) { return L.lv[(word * MAX_VERTS + v) * RW_PIX + lane]; }
__device__ __forceinline__ int rw_i(const ResolveLds& L, int word, int v, int lane) { return __float_as_int(rw_f(L, word, v, lane)); }
__device__ __forceinline__ V3 rw_v3(const ResolveLds& L, int word0, int v, int lane) {
    return v3(rw_f(L, word0, v, lane), rw_f(L, word0 + 1, v, lane), rw_f(L, word0 + 2, v, lane));
}

// One strategy pair (t, S): connect_resolve.hpp's resolve_pair with the light vertex read from LDS.
// Returns true when {add_v, add_w} are to be added to the pixel's running total / weight sum.
template <int S>
__device__ __forceinline__ bool resolve_pair_wide(
        int t, int B, int lane, const ResolveLds& L, int Ll, unsigned l_spec, unsigned c_spec, bool spec7,
        V3 c_o_in, V3 c_n_in, float c_c, float c_l, float c_tot_in, float c_cos_in, int c_tri, int c_meta,
        V3 prior_camera_color, unsigned long long mask, float2 h, const float4* __restrict__ tri_shade,
        const MaterialDev* __restrict__ mats, const CameraRec& cam, V3 focal, V3 cam_dir,
        V3& add_v, float& add_w, float4* __restrict__ light_image, float* splat_tab, int debug_flags) {
    constexpr int SV = S > 0 ? S - 1 : 0;          // index of the light junction vertex
    V3 c_o = c_o_in, c_n = c_n_in;
    float c_tot = c_tot_in, c_cos = c_cos_in;
    V3 dir_l_to_c = v3(0, 0, 0);
    int light_pixel_idx = -1;
    float Gj = 0.0f;
    V3 a_o = v3(0, 0, 0);
    float a_c = 0.0f, a_l = 0.0f, a_tot = 0.0f, a_cos = 0.0f;
    int a_meta = 0;

    if (S == 0) {
        if (!(c_meta & META_HIT_LIGHT)) return false;                         // :665
    } else {
        if (S > Ll) return false;
        if (!((mask >> conn_slot(t, S)) & 1ull)) return false;                // culled in k_connect_setup
        const int best_i = __float_as_int(h.x);
        if (best_i == -1) return false;                                       // :193 / :593
        a_o = rw_v3(L, RW_OX, SV, lane);
        if (t == 1) {
            // world_ray_to_camera_ray, :595-616
            if (__float_as_int(tri_shade[4 * best_i + 2].w) == 0) return false;   // !is_camera
            const V3 tdir = normalize(focal - a_o);
            const V3 camera_point = a_o + h.y * tdir;
            const float x = dot(camera_point - cam3(cam.center), cam3(cam.dx));
            const float y = dot(camera_point - cam3(cam.center), cam3(cam.dy));
            const int pixel_x = (int)__builtin_roundf((x / cam.phys_width + 0.5f) * cam.pixel_width);
            const int pixel_y = (int)__builtin_roundf((y / cam.phys_height + 0.5f) * cam.pixel_height);
            light_pixel_idx = pixel_y * cam.pixel_width + pixel_x;
            if (light_pixel_idx == -1) return false;                          // :671
            c_o = camera_point;
            const V3 cdir = normalize(focal - camera_point);
            c_n = cam_dir;
            c_cos = __builtin_fabsf(dot(cdir, cam_dir));
            c_tot = 1.0f;
        } else {
            if (best_i == rw_i(L, RW_TRI, SV, lane)) return false;            // visibility_test, :194-196
            if (best_i != c_tri) return false;
        }
        a_c = rw_f(L, RW_C, SV, lane); a_l = rw_f(L, RW_L, SV, lane); a_tot = rw_f(L, RW_TOT, SV, lane);
        a_cos = rw_f(L, RW_COS, SV, lane); a_meta = rw_i(L, RW_META, SV, lane);
        dir_l_to_c = normalize(c_o - a_o);
        Gj = geom_term(a_cos, c_cos, a_o, c_o);
    }

    // ---- p_s and the two running products (:737-757) ----
    const float p_s = c_tot * ((S == 0) ? 1.0f : a_tot);
    // backward (light side): p[i] = p[i+1] / r_i for i = S-1 .. 0
    float pb[MAX_VERTS > 0 ? MAX_VERTS : 1];
    if (S > 0) {
        const float r_junc = (S == 1) ? a_l / (a_c * Gj) : (a_l * L.GL[(S >= 2 ? S - 2 : 0) * RW_PIX + lane]) / (a_c * Gj);
        float v = p_s / r_junc;
        pb[S - 1] = v;
#pragma unroll
        for (int i = S - 2; i >= 0; i--) { v = v / L.RL[i * RW_PIX + lane]; pb[i] = v; }
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
                if (S == 0) r = c_l / (c_c * L.GC[(t - 2) * RW_PIX + lane]);            // i == 0 form, x_1 = camera[t-2]
                else r = (c_l * Gj) / (c_c * L.GC[(t - 2) * RW_PIX + lane]);           // interior form at the junction
            } else {
                r = L.RC[(t + S - i - 1) * RW_PIX + lane];
            }
            v = r * v;
            const bool sp = spec_at(i + 1);
            sum += (sp || prev_spec) ? 0.0f : v;
            prev_spec = sp;
        }
    }
    sum += 0.0f;                                                              // p[S+t] = 0
    if (!(p_at_s > 0.0f && sum > 0.0f)) return false;                         // :773-776
    const float w = p_at_s / sum;

    if (S == 0) {                                                             // :783-786
        const V3 emission = v3(mats[c_meta & 0xFF].emission_alpha);
        const V3 color = prior_camera_color * emission;
        add_v = ((w * 1.0f) * color) / p_s;
        add_w = w;
        return true;
    } else if (t == 1) {                                                      // :787-793, :817-823, K8 :952-961
        const V3 prior_color = rw_v3(L, RW_KX, (S - 2) > 0 ? (S - 2) : 0, lane);
        float new_light_f = 1.0f;
        if (S > 1) new_light_f = div_pi(__builtin_fabsf(dot(dir_l_to_c, rw_v3(L, RW_NX, SV, lane))));
        const V3 mcol = v3(mats[a_meta & 0xFF].color_type);
        const float shade = new_light_f * Gj / p_s;
        if (light_pixel_idx >= 0 && light_pixel_idx < B && !(debug_flags & 1)) {
            const V3 c = ((w * shade) * prior_color) * mcol;
            // transposed splat through the wave's LDS table: see connect_resolve.hpp
            float* tab = splat_tab;
            const unsigned long long here = __ballot(true);
            const int n_here = __popcll(here), k = __popcll(here & ((1ull << lane) - 1ull));
            tab[0 * 64 + k] = __int_as_float(light_pixel_idx);
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
        return false;
    } else {                                                                  // :794-816
        const MaterialDev cmat = mats[c_meta & 0xFF];
        const float new_camera_f = div_pi(__builtin_fabsf(dot(-dir_l_to_c, c_n)));
        const V3 camera_color = (prior_camera_color * new_camera_f) * v3(cmat.color_type);
        V3 light_color;
        if (S == 1) light_color = v3(mats[a_meta & 0xFF].emission_alpha);
        else {
            const V3 prior_light_color = rw_v3(L, RW_KX, S >= 2 ? S - 2 : 0, lane);
            const float new_light_f = div_pi(__builtin_fabsf(dot(dir_l_to_c, rw_v3(L, RW_NX, SV, lane))));
            light_color = (prior_light_color * new_light_f) * v3(mats[a_meta & 0xFF].color_type);
        }
        const V3 color = camera_color * light_color;
        add_v = ((w * Gj) * color) / p_s;
        add_w = w;
        return true;
    }
}

__global__ __launch_bounds__(RW_BLOCK) void k_connect_resolve_wide(
        int B, PathBufs lp, PathBufs cp, const MaterialDev* __restrict__ mats,
        const float4* __restrict__ tri_shade, CameraRec cam, const unsigned long long* __restrict__ cmask,
        const float2* __restrict__ chit, float* __restrict__ agg, float4* __restrict__ light_image,
        float4* __restrict__ uni_out, Stats* stats, int debug_flags) {
    __shared__ ResolveLds L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pid_raw = blockIdx.x * RW_PIX + lane;
    const bool valid = pid_raw < B;
    const int pid = valid ? pid_raw : 0;           // out-of-range lanes shadow pixel 0 and write nothing
    const int Lc = valid ? cp.len[pid] : 0, Ll = valid ? lp.len[pid] : 0;
    const V3 focal = cam3(cam.focal_point), cam_dir = cam3(cam.direction);
    const bool spec7 = __float_as_int(mats[7].color_type.w) > 0;
    const int t = wave + 1, v = wave;              // this wave: camera vertex v = t-1 in registers, light vertex v staged

    // ---- phase 1: stage light vertex v; load camera vertex v, publish what its neighbours need ----
    {
        float4 a = make_float4(0, 0, 0, 0), b = a, c = a, d = a;
        int tri = -1, flag = 0;
        if (v < Ll) {
            const size_t k = (size_t)v * B + pid;
            a = lp.P0[k]; b = lp.P1[k]; c = lp.P2[k]; d = lp.P3[k];
            tri = lp.tri[k];
            flag = __float_as_int(mats[__float_as_int(c.w) & 0xFF].color_type.w) > 0 ? 1 : 0;
        }
        float* base = L.lv + v * RW_PIX + lane;
        base[RW_OX * MAX_VERTS * RW_PIX] = a.x; base[RW_OY * MAX_VERTS * RW_PIX] = a.y; base[RW_OZ * MAX_VERTS * RW_PIX] = a.z;
        base[RW_C * MAX_VERTS * RW_PIX] = a.w; base[RW_L * MAX_VERTS * RW_PIX] = b.w; base[RW_TOT * MAX_VERTS * RW_PIX] = d.w;
        base[RW_COS * MAX_VERTS * RW_PIX] = __builtin_fabsf(dot(v3(b), v3(c)));
        base[RW_TRI * MAX_VERTS * RW_PIX] = __int_as_float(tri);
        base[RW_META * MAX_VERTS * RW_PIX] = c.w;
        base[RW_NX * MAX_VERTS * RW_PIX] = c.x; base[RW_NY * MAX_VERTS * RW_PIX] = c.y; base[RW_NZ * MAX_VERTS * RW_PIX] = c.z;
        base[RW_KX * MAX_VERTS * RW_PIX] = d.x; base[RW_KY * MAX_VERTS * RW_PIX] = d.y; base[RW_KZ * MAX_VERTS * RW_PIX] = d.z;
        L.lflag[v * RW_PIX + lane] = flag;
    }
    float4 cP0 = make_float4(0, 0, 0, 0), cP1 = cP0, cP2 = cP0, cP3 = cP0;
    int c_tri = -1;
    V3 prior_camera_color = v3(0, 0, 0);
    const bool have_t = t <= Lc;
    if (have_t) {
        const size_t ck = (size_t)v * B + pid;
        cP0 = cp.P0[ck]; cP1 = cp.P1[ck]; cP2 = cp.P2[ck]; cP3 = cp.P3[ck];
        c_tri = cp.tri[ck];
        if (t >= 2) prior_camera_color = v3(cp.P3[ck - B]);
    }
    const int c_meta = __float_as_int(cP2.w);
    const V3 c_o = v3(cP0), c_n = v3(cP2);
    const float c_cos = __builtin_fabsf(dot(v3(cP1), c_n));
    {
        float* cb = L.cv + v * RW_PIX + lane;
        cb[0 * MAX_VERTS * RW_PIX] = c_o.x; cb[1 * MAX_VERTS * RW_PIX] = c_o.y; cb[2 * MAX_VERTS * RW_PIX] = c_o.z;
        cb[3 * MAX_VERTS * RW_PIX] = c_cos;
        int flag = 0;
        if (have_t) {
            if (__float_as_int(mats[c_meta & 0xFF].color_type.w) > 0) flag |= 1;
            if (c_meta & META_HIT_LIGHT) flag |= 2;
        }
        L.cflag[v * RW_PIX + lane] = flag;
    }
    // all closest-hit results of this t in flight at once (entries of culled pairs are never used)
    float2 hits[MAX_VERTS + 1];
    hits[0] = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int s = 1; s <= MAX_VERTS; s++) hits[s] = have_t ? chit_load(chit, B, t, s, pid) : make_float2(0.0f, 0.0f);
    const unsigned long long mask = valid ? cmask[pid] : 0ull;
    __syncthreads();

    // ---- phase 2a: adjacent geometry terms, one per wave and side: GL[v] (light v, v+1), GC[v-1] (camera v-1, v) ----
    unsigned l_spec = 0, c_spec = 0, c_hitl = 0;
#pragma unroll
    for (int k = 0; k < MAX_VERTS; k++) {
        if (L.lflag[k * RW_PIX + lane]) l_spec |= 1u << k;
        const int f = L.cflag[k * RW_PIX + lane];
        if (f & 1) c_spec |= 1u << k;
        if (f & 2) c_hitl |= 1u << k;
    }
    if (v + 1 < MAX_VERTS) {
        float G = 0.0f;
        if (v + 1 < Ll) G = geom_term(rw_f(L, RW_COS, v, lane), rw_f(L, RW_COS, v + 1, lane), rw_v3(L, RW_OX, v, lane), rw_v3(L, RW_OX, v + 1, lane));
        L.GL[v * RW_PIX + lane] = G;
    }
    if (v >= 1) {
        float G = 0.0f;
        if (have_t) {
            const float* cb = L.cv + (v - 1) * RW_PIX + lane;
            const V3 po = v3(cb[0 * MAX_VERTS * RW_PIX], cb[1 * MAX_VERTS * RW_PIX], cb[2 * MAX_VERTS * RW_PIX]);
            G = geom_term(cb[3 * MAX_VERTS * RW_PIX], c_cos, po, c_o);                                   // GC[v-1]
        }
        L.GC[(v - 1) * RW_PIX + lane] = G;
    }
    // unidirectional estimate of generate_paths (camera pass), trace.metal:523-528: first stored vertex
    // k >= 1 with hit_light -> rays[k-1].color / rays[k].tot_importance; the wave that holds it writes it
    {
        const unsigned first = c_hitl & ~1u;                   // vertex 0 does not count
        if (first == 0u) { if (wave == 0 && valid) uni_out[pid] = make_float4(0, 0, 0, 0); }
        else if ((int)__builtin_ctz(first) == v && valid) {
            const V3 c = prior_camera_color / cP3.w;
            uni_out[pid] = make_float4(c.x, c.y, c.z, 1.0f);
        }
    }
    __syncthreads();

    // ---- phase 2b: interior ratios, one per wave and side: RL[v], RC[v] ----
    if (v + 1 < MAX_VERTS) {
        float R = 0.0f;
        if (v + 1 < Ll) {
            const float lv_l = rw_f(L, RW_L, v, lane), lv_c = rw_f(L, RW_C, v, lane), G = L.GL[v * RW_PIX + lane];
            R = (v == 0) ? lv_l / (lv_c * G) : (lv_l * L.GL[(v - 1) * RW_PIX + lane]) / (lv_c * G);
        }
        L.RL[v * RW_PIX + lane] = R;
        float RCv = 0.0f;
        if (v + 1 < Lc) {                                      // camera vertex m = v with its far neighbour v+1
            const float G = L.GC[v * RW_PIX + lane];
            RCv = (v == 0) ? (cP1.w * G) / cP0.w : (cP1.w * G) / (cP0.w * L.GC[(v - 1) * RW_PIX + lane]);
        }
        L.RC[v * RW_PIX + lane] = RCv;
    }
    __syncthreads();

    // ---- phase 3: the seven pairs (t, s = 0..6) of this wave's camera vertex ----
    V3 addv[MAX_VERTS + 1];
    float addw[MAX_VERTS + 1];
    unsigned produced = 0;
    if (have_t) {
        const bool skip = ((debug_flags & 2) && t >= 2) || ((debug_flags & 4) && t == 1);   // experiment switches
#define CL2_PAIR(S)                                                                                                   \
        addv[S] = v3(0, 0, 0); addw[S] = 0.0f;                                                                        \
        if (!skip && (S) <= Ll && t + (S) >= 2 &&                                                                     \
            resolve_pair_wide<S>(t, B, lane, L, Ll, l_spec, c_spec, spec7, c_o, c_n, cP0.w, cP1.w, cP3.w, c_cos, c_tri, \
                                 c_meta, prior_camera_color, mask, hits[S], tri_shade, mats, cam, focal, cam_dir,     \
                                 addv[S], addw[S], light_image, L.splat_tab, debug_flags))                            \
            produced |= 1u << (S)
        CL2_PAIR(0); CL2_PAIR(1); CL2_PAIR(2); CL2_PAIR(3); CL2_PAIR(4); CL2_PAIR(5); CL2_PAIR(6);
#undef CL2_PAIR
    }

    // ---- phase 4: the running total goes from wave to wave, each adds its values in s order ----
    for (int w = 0; w < MAX_VERTS; w++) {
        if (wave == w) {
            V3 total = v3(0, 0, 0);
            float cws = 0.0f;
            if (w > 0) {
                total = v3(L.relay[0 * RW_PIX + lane], L.relay[1 * RW_PIX + lane], L.relay[2 * RW_PIX + lane]);
                cws = L.relay[3 * RW_PIX + lane];
            }
#pragma unroll
            for (int S = 0; S <= MAX_VERTS; S++) {
                if ((produced >> S) & 1u) { total = total + addv[S]; cws += addw[S]; }
            }
            if (w + 1 < MAX_VERTS) {
                L.relay[0 * RW_PIX + lane] = total.x; L.relay[1 * RW_PIX + lane] = total.y; L.relay[2 * RW_PIX + lane] = total.z;
                L.relay[3 * RW_PIX + lane] = cws;
            } else if (valid) {
                agg[(size_t)9 * B + pid] = total.x;
                agg[(size_t)10 * B + pid] = total.y;
                agg[(size_t)11 * B + pid] = total.z;
                agg[(size_t)12 * B + pid] = cws;
            }
        }
        if (w + 1 < MAX_VERTS) __syncthreads();
    }

    // ---- reconstruction-filter weights, trace.metal:827-862 (independent of the pairs): wave 0, which is
    // through with the relay first.  A zero-length camera path is the reference's zero-filled Path: pixel 0,
    // film point (0,0,0). ----
    if (wave == 0) {
        const int pixel_idx = (Lc > 0) ? pid : 0;
        V3 film = v3(0, 0, 0);
        if (Lc > 0) film = v3(cP0);                            // camera vertex 0 is this wave's
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
        if (valid) {
#pragma unroll
            for (int r = 0; r < 9; r++) agg[(size_t)r * B + pid] = (weight_sum != 0.0f) ? wts[r] / weight_sum : wts[r];
        }
    }
}

}  // namespace cl2
