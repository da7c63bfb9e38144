// bvh_traverse.hpp -- closest-hit BVH query, stackless, nodes in visit order, top of the array in LDS.
//
// Reference semantics (`src/trace.metal:106-176`): slab box test that returns the entry distance,
// Moller-Trumbore triangle test (reject t <= DELTA, inclusive u,v bounds, no parallel-ray guard),
// depth-first walk that visits the RIGHT child (left+1) before the left one, prunes a node when
// its entry distance is not < best_t, scans leaf triangles in ascending order and keeps a hit
// only on strict t < best_t.
//
// MI355X formulation.  The reference's 64-entry per-thread stack would live in scratch memory.
// Instead the nodes are re-ordered at scene upload into the reference's own VISIT order (node, right
// subtree, left subtree), so that
//     hit on an inner node  -> next node is index + 1            (same or next cache line)
//     miss / leaf finished  -> next node is `skip` = index + size of this subtree
// which visits exactly the nodes the stack version visits, in the same order, testing each against
// the same best_t -- hits (including exact-t ties) are identical -- with no stack, no scratch
// traffic, ONE 32-byte record per step and sequential addresses while a ray descends.  Records
// [0, n_lds_nodes) and, when they fit, all intersection triangles are copied into LDS by each
// workgroup (the whole Cornell box: 5 nodes + 16 triangles).
//
// Device layouts (built by cl2_upload_scene):
//   node : float4 lo = {min.xyz, as_float(skip)}, float4 hi = {max.xyz, as_float(info)}      (32 B)
//          info < 0: inner node.  info >= 0: leaf over triangles [info >> 4, (info >> 4) + (info & 15) + 1).
//          A reference leaf with more than 16 triangles (only possible past the builder's depth
//          limit) continues in follow-up records whose box is (-inf, +inf): always entered.
//   tri  : 3 x float4 = {v0.xyz,-}, {e1.xyz,-}, {e2.xyz,-} with e1 = v1-v0, e2 = v2-v0 (the same
//          float32 subtractions the reference performs per test, done once); reference order.
#pragma once
#include "vecmath.hpp"
#include "bsdf.hpp"

namespace cl2 {

struct BvhView {
    const float4* nodes;     // 2 float4 per node, visit order
    const float4* tris;      // 3 float4 per triangle
    int n_nodes;             // records (>= reference boxes when oversized leaves were split)
    int n_tris;
    int n_lds_nodes;         // records staged in LDS (<= LDS_NODE_CAP)
    int lds_tris;            // 1: all triangles staged in LDS (n_tris <= LDS_TRI_CAP)
    int n_fast_nodes;        // records of the pruned table (0: none); only for trees that are wholly staged
    const float4* fast_nodes;   // the tree without the inner records whose test is not worth its cost (cl2_upload_scene):
                                // same hits for rays with finite 1/d, fewer box tests
    int fast_flat;              // 1: the pruned table is a plain list of leaves (every record a leaf, skip = index + 1): every lane
                                // visits the same records in the same order, so the walk's control flow is wave-uniform
};

constexpr int LDS_NODE_CAP = 512;   // records in the LDS window: at most 512 * 32 B = 16 KB
constexpr int LDS_TRI_CAP = 512;    // triangles staged in LDS when the whole scene has no more: at most 512 * 48 B = 24 KB
constexpr int LEAF_PACK_MAX = 16;   // triangles per leaf record

// The staged part of the tree lives in DYNAMIC shared memory sized by the scene (bvh_lds_bytes on the
// host): the Cornell box takes 1 KB, a 500-triangle scene 40 KB -- one kernel serves both without the
// small scene paying for the large one's reservation.
struct BvhLds {
    const float4* nodes;     // 2 float4 per staged record
    const float4* tris;      // 3 float4 per triangle (only when b.lds_tris)
    const float4* fast_nodes;   // 2 float4 per record of the pruned table (only when b.n_fast_nodes)
};

// Cooperative copy of the staged part of the tree; every thread of the block must call it.
__device__ __forceinline__ void stage_bvh(BvhLds& s, const BvhView& b) {
    extern __shared__ float4 cl2_tree_lds[];
    float4* nodes = cl2_tree_lds;
    float4* tris = cl2_tree_lds + 2 * b.n_lds_nodes;
    const int nt = blockDim.x, t = threadIdx.x;
    for (int i = t; i < 2 * b.n_lds_nodes; i += nt) nodes[i] = b.nodes[i];
    float4* fast = tris + (b.lds_tris ? 3 * b.n_tris : 0);
    if (b.lds_tris)
        for (int i = t; i < 3 * b.n_tris; i += nt) tris[i] = b.tris[i];
    for (int i = t; i < 2 * b.n_fast_nodes; i += nt) fast[i] = b.fast_nodes[i];
    s.nodes = nodes; s.tris = tris; s.fast_nodes = fast;
    __syncthreads();
}

struct Hit { int tri; float t, u, v; };

// Closest hit along (o, d) with inv = 1/d.
//   COUNT    adds node/triangle test tallies.
//   ALL_LDS  the whole tree and all triangles are staged in LDS (small scenes): plain ds_read, no
//            per-access LDS/global pointer select.
//   FAST_MINMAX  the slab test uses v_min/v_max instead of the compare+select form of the MSL
//            min/max.  The two differ only when an operand is NaN (0 * inf, i.e. a direction
//            component is exactly 0 and the origin lies on a slab plane) or in the sign of a zero,
//            and a zero's sign never reaches a decision (the slab values only feed comparisons);
//            the caller selects FAST_MINMAX only for waves whose rays all have finite 1/d.
template <bool COUNT, bool ALL_LDS, bool FAST_MINMAX>
__device__ __forceinline__ Hit closest_hit_impl(const BvhLds& s, const BvhView& b, V3 o, V3 d, V3 inv,
                                               unsigned& n_box, unsigned& n_tri, const float4* lds_nodes, int n_nodes) {
    Hit best{-1, __builtin_inff(), 0.0f, 0.0f};
    int node = 0;
    while (node < n_nodes) {
        float4 lo, hi;
        if (ALL_LDS || node < b.n_lds_nodes) { lo = lds_nodes[2 * node]; hi = lds_nodes[2 * node + 1]; }
        else { lo = b.nodes[2 * node]; hi = b.nodes[2 * node + 1]; }
        int next = __float_as_int(lo.w);                   // skip: first record after this subtree
        if (COUNT) n_box++;
        // ray_box_intersect, trace.metal:106-115 (called with t = INFINITY, :153-155)
        float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
        float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
        float tmin, tmax;
        if (FAST_MINMAX) {
            tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                   __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
            tmax = __builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fminf(__builtin_fmaxf(t0y, t1y), __builtin_fmaxf(t0z, t1z)));
        } else {
            tmin = max_msl(max_msl(min_msl(t0x, t1x), min_msl(t0y, t1y)), max_msl(min_msl(t0z, t1z), 0.0f));
            tmax = min_msl(min_msl(max_msl(t0x, t1x), max_msl(t0y, t1y)), min_msl(max_msl(t0z, t1z), __builtin_inff()));
        }
        if (tmin <= tmax && tmin < best.t) {
            const int info = __float_as_int(hi.w);
            if (info < 0) {
                next = ~info;                          // right child first (trace.metal:158-159)
            } else {
                const int left = info >> 4, right = left + (info & 15) + 1;
                for (int i = left; i < right; i++) {   // trace.metal:161-172
                    float4 a0, a1, a2;
                    if (ALL_LDS || b.lds_tris) { a0 = s.tris[3 * i]; a1 = s.tris[3 * i + 1]; a2 = s.tris[3 * i + 2]; }
                    else { a0 = b.tris[3 * i]; a1 = b.tris[3 * i + 1]; a2 = b.tris[3 * i + 2]; }
                    if (COUNT) n_tri++;
                    // ray_triangle_intersect, trace.metal:117-142
                    V3 e1 = v3(a1), e2 = v3(a2);
                    V3 h = cross(d, e2);
                    float a = dot(e1, h);
                    float f = rcp_exact(a);
                    V3 sv = o - v3(a0);
                    float u = f * dot(sv, h);
                    if (u < 0 || u > 1) continue;
                    V3 q = cross(sv, e1);
                    float v = f * dot(d, q);
                    if (v < 0 || u + v > 1) continue;
                    float t = f * dot(e2, q);
                    if (t > DELTA_F && t < best.t) { best.tri = i; best.t = t; best.u = u; best.v = v; }
                }
            }
        }
        node = next;
    }
    return best;
}

// ray_triangle_intersect, trace.metal:117-142, against a fetched record {v0, e1, e2}; keeps the hit on strict t < best.t
__device__ __forceinline__ void tri_test(V3 o, V3 d, float4 a0, float4 a1, float4 a2, int index, Hit& best) {
    const V3 e1 = v3(a1), e2 = v3(a2);
    const V3 h = cross(d, e2);
    const float f = rcp_exact(dot(e1, h));
    const V3 sv = o - v3(a0);
    const float u = f * dot(sv, h);
    if (!(u < 0 || u > 1)) {
        const V3 q = cross(sv, e1);
        const float v = f * dot(d, q);
        if (!(v < 0 || u + v > 1)) {
            const float t = f * dot(e2, q);
            if (t > DELTA_F && t < best.t) { best.tri = index; best.t = t; best.u = u; best.v = v; }
        }
    }
}

// The same test without early exits and with one predicated update: for waves whose lanes test DIFFERENT triangles (the
// persistent walks of the mesh scenes) some lane always passes the `u` test, so the exits skip nothing and cost three
// exec-mask regions of scalar instructions.  Same decisions as tri_test, NaN cases included.
__device__ __forceinline__ void tri_test_branchless(V3 o, V3 d, const float4& p0, const float4& p1, const float4& p2, int index, Hit& best) {
    const V3 e1 = v3(p1), e2 = v3(p2);
    const V3 h = cross(d, e2);
    const float f = rcp_exact(dot(e1, h));
    const V3 sv = o - v3(p0);
    const float u = f * dot(sv, h);
    const V3 q = cross(sv, e1);
    const float v = f * dot(d, q);
    const float t = f * dot(e2, q);
    const bool ok = !(u < 0 || u > 1) && !(v < 0 || u + v > 1) && (t > DELTA_F && t < best.t);
    best.tri = ok ? index : best.tri;
    best.t = ok ? t : best.t;
    best.u = ok ? u : best.u;
    best.v = ok ? v : best.v;
}

// tri_test_branchless for the nearest-first walk (bvh_wide.hpp, ORDER): the reference keeps the FIRST triangle it meets at a given t
// (`t < best_t`, trace.metal:170).  A walk that meets the leaves in another order gets the same winner by letting a hit at exactly
// best_t replace a held triangle that the reference would have met later (rank[]: position in the reference's visit order, built by
// cl2_upload_scene; rank[-1] = INT_MIN for "nothing held" -- t = best_t = +inf, a degenerate triangle -- and one entry behind the
// last triangle).  Ties are a few rays in 1e8: the walk tests a pair with tri_test_branchless_tie and comes here, inside a wave-level
// branch, only when one of the two tied -- repeating a test is harmless (a triangle that holds best_t has no lower rank than itself),
// and so is the rule where the order IS the reference's (the rays of the binary walk inside the same loop): what is held then has
// the lower rank.
__device__ __forceinline__ void tri_test_tie_rule(V3 o, V3 d, const float4& p0, const float4& p1, const float4& p2, int index, Hit& best,
                                                  const int* __restrict__ rank) {
    const V3 e1 = v3(p1), e2 = v3(p2);
    const V3 h = cross(d, e2);
    const float f = rcp_exact(dot(e1, h));
    const V3 sv = o - v3(p0);
    const float u = f * dot(sv, h);
    const V3 q = cross(sv, e1);
    const float v = f * dot(d, q);
    const float t = f * dot(e2, q);
    const bool inside = !(u < 0 || u > 1) && !(v < 0 || u + v > 1) && t > DELTA_F;
    bool ok = inside && t < best.t;
    if (inside && t == best.t) ok = rank[index] < rank[best.tri];
    best.tri = ok ? index : best.tri;
    best.t = ok ? t : best.t;
    best.u = ok ? u : best.u;
    best.v = ok ? v : best.v;
}

// tri_test_branchless that also says whether the triangle is hit at EXACTLY the best_t it was tested against (the nearest-first walk
// then repeats the pair's tests under tri_test_tie_rule: one wave-level branch per pair, taken for a few rays in 1e8).
__device__ __forceinline__ bool tri_test_branchless_tie(V3 o, V3 d, const float4& p0, const float4& p1, const float4& p2, int index, Hit& best) {
    const V3 e1 = v3(p1), e2 = v3(p2);
    const V3 h = cross(d, e2);
    const float f = rcp_exact(dot(e1, h));
    const V3 sv = o - v3(p0);
    const float u = f * dot(sv, h);
    const V3 q = cross(sv, e1);
    const float v = f * dot(d, q);
    const float t = f * dot(e2, q);
    const bool inside = !(u < 0 || u > 1) && !(v < 0 || u + v > 1) && t > DELTA_F;
    const bool ok = inside && t < best.t;
    const bool tie = inside && t == best.t;
    best.tri = ok ? index : best.tri;
    best.t = ok ? t : best.t;
    best.u = ok ? u : best.u;
    best.v = ok ? v : best.v;
    return tie;
}

// The pruned table of a tiny scene (the Cornell box: three leaves, 16 triangles) has no inner records left: every ray
// visits record 0, 1, 2, ... in that order, whatever it hits.  The per-lane walk above then spends vector instructions on
// bookkeeping that is the same in every lane (record index, triangle index, loop tests, LDS addresses) and waits for each
// triangle's LDS reads right before it uses them.  Here the two loops run on the scalar unit, a lane that fails a leaf's
// box test is masked for that leaf's triangles, and the next triangle's record is fetched while the current one is tested.
// Per lane the sequence of box tests, triangle tests and comparisons is exactly that of closest_hit_impl<.., true, true>
// on the same table.  Only for waves whose rays all have finite 1/d (v_min / v_max slab test), never while counting.
__device__ __forceinline__ Hit closest_hit_flat(const BvhLds& s, const BvhView& b, V3 o, V3 d, V3 inv) {
    Hit best{-1, __builtin_inff(), 0.0f, 0.0f};
    const int n = b.n_fast_nodes;
    for (int node = 0; node < n; node++) {
        const float4 lo = s.fast_nodes[2 * node], hi = s.fast_nodes[2 * node + 1];
        const float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
        const float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
        const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                           __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
        const float tmax = __builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fminf(__builtin_fmaxf(t0y, t1y), __builtin_fmaxf(t0z, t1z)));
        const bool in = tmin <= tmax && tmin < best.t;
        if (!__any(in)) continue;
        const int info = __builtin_amdgcn_readfirstlane(__float_as_int(hi.w));      // the same word in every lane
        const int left = info >> 4, right = left + (info & 15) + 1;
        // Two register sets in turn: the record of triangle i + 1 is on its way while triangle i is tested.  The fetch is
        // unconditional (the leaf's last triangle fetches itself again): a branch around it makes the compiler wait for ALL
        // outstanding LDS reads at the join instead of counting them.  keep4() makes the unused fourth word of a record live,
        // so that each float4 comes by one ds_read_b128 (4 LDS cycles) and not by the ds_read_b96 the compiler would narrow
        // it to (8 LDS cycles, MI355X_MICROARCH.md "LDS").
        auto keep4 = [](float4& v) { asm volatile("" : "+v"(v.w)); };
        auto fetch = [&](int k, float4& r0, float4& r1, float4& r2) {
            r0 = s.tris[3 * k]; r1 = s.tris[3 * k + 1]; r2 = s.tris[3 * k + 2];
        };
        const int last = right - 1;
        float4 a0, a1, a2, c0, c1, c2;
        fetch(left, a0, a1, a2);
        for (int i = left;;) {
            fetch(i < last ? i + 1 : last, c0, c1, c2);
            if (in) tri_test(o, d, a0, a1, a2, i, best);
            keep4(a0); keep4(a1); keep4(a2);        // here, where the record has been consumed: the asm waits for its operand
            if (++i > last) break;
            fetch(i < last ? i + 1 : last, a0, a1, a2);
            if (in) tri_test(o, d, c0, c1, c2, i, best);
            keep4(c0); keep4(c1); keep4(c2);
            if (++i > last) break;
        }
    }
    return best;
}

// Lane index and "set bits of a wave mask below this lane" straight from v_mbcnt: two instructions where they are needed instead
// of a lane index and a 64-bit lane mask kept alive through a persistent loop (round 4's wide kernels spilled exactly that mask
// to scratch and reloaded it at the top of every pass).
__device__ __forceinline__ unsigned wave_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ unsigned rank_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ bool finite3(V3 a) {
    return __builtin_fabsf(a.x) < __builtin_inff() && __builtin_fabsf(a.y) < __builtin_inff() && __builtin_fabsf(a.z) < __builtin_inff();
}

// Dispatch: wave-uniform choice of the slab-test form, block-uniform choice of the staging mode.
template <bool COUNT>
__device__ __forceinline__ Hit closest_hit(const BvhLds& s, const BvhView& b, V3 o, V3 d, V3 inv,
                                          unsigned& n_box, unsigned& n_tri) {
    const bool all_lds = b.lds_tris && b.n_nodes <= b.n_lds_nodes;
    const bool fast = __all(finite3(inv));
    if (all_lds) {
        // the pruned table: not while counting (the tallies are those of the full walk)
        if (fast && !COUNT && b.n_fast_nodes && b.fast_flat) return closest_hit_flat(s, b, o, d, inv);
        if (fast && !COUNT && b.n_fast_nodes) return closest_hit_impl<COUNT, true, true>(s, b, o, d, inv, n_box, n_tri, s.fast_nodes, b.n_fast_nodes);
        if (fast) return closest_hit_impl<COUNT, true, true>(s, b, o, d, inv, n_box, n_tri, s.nodes, b.n_nodes);
        return closest_hit_impl<COUNT, true, false>(s, b, o, d, inv, n_box, n_tri, s.nodes, b.n_nodes);
    }
    if (fast) return closest_hit_impl<COUNT, false, true>(s, b, o, d, inv, n_box, n_tri, s.nodes, b.n_nodes);
    return closest_hit_impl<COUNT, false, false>(s, b, o, d, inv, n_box, n_tri, s.nodes, b.n_nodes);
}

}  // namespace cl2

// ---------------------------------------------------------------------------------------------
// Persistent traversal with lane-level ray replacement (large scenes).
//
// On a 1M-triangle scene the reference's unordered walk visits 58 nodes + 23 triangles per ray on
// average with a long tail (hundreds for rays that graze many objects).  With one ray per lane for
// the lifetime of a wave, a wave runs as long as its slowest ray and every instruction is issued
// for a handful of live lanes (measured SIMD efficiency ~8 %).  Here each lane is a small state
// machine -- one node visit or one triangle test per step -- and a lane whose ray is finished
// immediately fetches the next ray of the launch (chunks of the ray range are handed to waves by
// one global atomic per chunk), so all 64 lanes stay busy until the launch runs dry.  The sequence
// of node visits, triangle tests and comparisons of each ray is exactly that of closest_hit_impl.
namespace cl2 {

constexpr int WORK_STRIDE = 16;     // unsigned words per launch slot of the work counters (one 64-byte line)
constexpr int RAY_CHUNK_MAX = 512;  // rays handed to a wave per global atomic: 64..512, about a quarter of a wave's fair share
// Lanes whose ray is finished take a new one only when at least REFILL_MIN of them are idle (or nobody is walking).  The
// refill is all-wave code for a handful of lanes -- a tag load, two dependent vertex gathers, a normalisation and three
// exact reciprocals, and the wave waits for those loads before its next node fetch -- so running it in EVERY pass for the
// one or two lanes that just finished puts two extra memory round trips on every pass's critical path.  Same-box A/B
// (tools/exp_ab_scene.py, connection launch alone, ms): 1M triangles, binary walk 15.7 -> 14.5 / 14.0 / 14.0 / 15.8 at
// 16 / 24 / 32 / 48 lanes; glass, 4-wide walk 3.90 -> 3.81 / 3.86 / 3.80 / 4.14; blob 5.12 -> 5.03 / 5.04 / 5.10 / 6.10.
#ifndef CL2_REFILL_MIN_WIDE
#define CL2_REFILL_MIN_WIDE 16
#endif
constexpr int REFILL_MIN_BINARY = 24;
constexpr int REFILL_MIN_WIDE = CL2_REFILL_MIN_WIDE;

template <bool COUNT, bool TWO_TRIS, class Source>
__device__ __forceinline__ void traverse_persistent(const BvhLds& s, const BvhView& b, unsigned n, unsigned* work_counter,
                                                    const Source& src, unsigned& n_box, unsigned& n_tri) {
    // chunk size: small launches (one subpath level = one ray per pixel) must still spread over every
    // wave and leave rays for replacement; big launches amortise the atomic
    const unsigned waves = gridDim.x * (blockDim.x >> 6);
    unsigned chunk = n / (waves * 4u);
    chunk = chunk < 64u ? 64u : (chunk > (unsigned)RAY_CHUNK_MAX ? (unsigned)RAY_CHUNK_MAX : chunk);
    unsigned w_next = 0, w_end = 0;            // wave-uniform: current chunk [w_next, w_end)
    bool dry = false;                          // wave-uniform: the launch has no rays left
    // per-lane ray state
    bool active = false, fast = true;
    V3 o = v3(0, 0, 0), d = o, inv = o;
    Hit best{-1, __builtin_inff(), 0.0f, 0.0f};
    const int n_nodes = b.n_nodes;
    int node = n_nodes, tri_i = 0, tri_end = 0;
    int key = 0;                               // the source's token of the lane's ray (pixel id / tag): store() needs it again

    while (true) {
        // ---- refill idle lanes ----
        unsigned long long idle = __ballot(!active);
        if (__popcll(idle) < REFILL_MIN_BINARY && __popcll(idle) < 64) idle = 0;               // not yet: see REFILL_MIN_BINARY
        while (idle && !dry) {
            if (w_next >= w_end) {             // wave-uniform branch: fetch a new chunk
                unsigned base = 0;
                if (wave_lane() == 0) base = atomicAdd(work_counter, chunk);
                base = __shfl(base, 0);
                if (base >= n) { dry = true; break; }
                w_next = base;
                w_end = base + chunk < n ? base + chunk : n;
            }
            const unsigned avail = w_end - w_next;
            const unsigned rank = rank_below(idle);
            const bool take = !active && rank < avail;
            if (take) {
                key = src.load(w_next + rank, o, d);
                inv = rcp3(d);
                fast = finite3(inv);
                best = Hit{-1, __builtin_inff(), 0.0f, 0.0f};
                node = 0; tri_i = 0; tri_end = 0;
                active = true;
            }
            const unsigned taken = __popcll(idle) < avail ? __popcll(idle) : avail;
            w_next += taken;
            idle = __ballot(!active);
        }
        if (!__any(active)) break;

        // ---- one step per live lane: at most one node visit, then at most one triangle test (a lane
        // that has just entered a leaf tests its first triangle in the same step), then retire ----
        const bool all_fast = __all(!active || fast);
        if (active) {
            if (tri_i >= tri_end && node < n_nodes) {
                // one node: ray_box_intersect + descend / skip (trace.metal:150-160)
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
            if (tri_i < tri_end) {
                // one triangle of the current leaf -- two when TWO_TRIS (both fetched together, tested in order; pays
                // while the tree is cache-resident and the step is issue-bound, not when it streams from memory):
                // ray_triangle_intersect, trace.metal:117-142
                const int i0 = tri_i;
                const bool two = TWO_TRIS && i0 + 1 < tri_end;
                const int i1 = two ? i0 + 1 : i0;
                tri_i = i1 + 1;
                float4 a0, a1, a2, c0, c1, c2;
                if (b.lds_tris) {
                    a0 = s.tris[3 * i0]; a1 = s.tris[3 * i0 + 1]; a2 = s.tris[3 * i0 + 2];
                    if (TWO_TRIS) { c0 = s.tris[3 * i1]; c1 = s.tris[3 * i1 + 1]; c2 = s.tris[3 * i1 + 2]; }
                } else {
                    a0 = b.tris[3 * i0]; a1 = b.tris[3 * i0 + 1]; a2 = b.tris[3 * i0 + 2];
                    if (TWO_TRIS) { c0 = b.tris[3 * i1]; c1 = b.tris[3 * i1 + 1]; c2 = b.tris[3 * i1 + 2]; }
                }
                if (COUNT) n_tri += two ? 2 : 1;
                {
                    const V3 e1 = v3(a1), e2 = v3(a2);
                    const V3 h = cross(d, e2);
                    const float f = rcp_exact(dot(e1, h));
                    const V3 sv = o - v3(a0);
                    const float u = f * dot(sv, h);
                    if (!(u < 0 || u > 1)) {
                        const V3 q = cross(sv, e1);
                        const float v = f * dot(d, q);
                        if (!(v < 0 || u + v > 1)) {
                            const float t = f * dot(e2, q);
                            if (t > DELTA_F && t < best.t) { best.tri = i0; best.t = t; best.u = u; best.v = v; }
                        }
                    }
                }
                if (TWO_TRIS && two) {
                    const V3 e1 = v3(c1), e2 = v3(c2);
                    const V3 h = cross(d, e2);
                    const float f = rcp_exact(dot(e1, h));
                    const V3 sv = o - v3(c0);
                    const float u = f * dot(sv, h);
                    if (!(u < 0 || u > 1)) {
                        const V3 q = cross(sv, e1);
                        const float v = f * dot(d, q);
                        if (!(v < 0 || u + v > 1)) {
                            const float t = f * dot(e2, q);
                            if (t > DELTA_F && t < best.t) { best.tri = i1; best.t = t; best.u = u; best.v = v; }
                        }
                    }
                }
            }
            if (tri_i >= tri_end && node >= n_nodes) {
                src.store(key, best);
                active = false;
            }
        }
    }
}

}  // namespace cl2
