// bvh_wide.hpp -- persistent closest-hit walk over a 4-wide collapse of the reference's binary tree, with the
// reference's decisions (trace.metal:144-176) reproduced exactly.
//
// Why it is exact.  The reference pops a box, tests it (`tmin <= tmax && tmin < best_t`, tmin/tmax from the slab test),
// and either pushes its two children (the one at left+1 is popped first) or tests the leaf's triangles in order.  A box
// nests its children exactly: bounds are min/max over nested triangle sets, rounded float64 -> float32 monotonically
// (cl2_upload_scene verifies it box by box and disables this walk otherwise).  For a ray with finite 1/d every slab
// value is a monotonic function of the bound it comes from, so a child's tmin is >= and its tmax <= its parent's, and
// best_t only ever shrinks.  Hence: if a box passes its test at the time it is popped, every ancestor passed its own
// test earlier; if an inner box fails, every box below it fails too.  The inner tests therefore decide nothing that
// the tests of the boxes below do not decide again: they only prune.  A wide node stores the boxes of the (up to four)
// grandchildren of a reference box in the reference's visit order; the walk computes their slab values in one visit,
// pushes the ones that pass with their entry distance, and re-applies `tmin < best_t` when an entry is popped -- the
// reference's own test of that box at that moment.  Leaves are visited in the reference's order against the
// reference's best_t, so hits (exact-t ties included) are identical; what changes is the number of dependent fetches
// (one 128-byte line per two levels) and of loop passes.
//
// Rays with a non-finite 1/d (a direction component is exactly 0: 0 * inf = NaN breaks the monotonicity argument)
// do not walk the wide tree: their lanes run the binary stackless walk of traverse_persistent inside the same loop
// (records read through the caches, MSL min/max), so a launch needs no second pass for them.
//
// Wide node = 8 x float4: {lo.x[4]} {lo.y[4]} {lo.z[4]} {hi.x[4]} {hi.y[4]} {hi.z[4]} {ref[4] as int} {pad}
//   ref >= 0: wide node index; ref < 0 and != WIDE_EMPTY: leaf, ~ref = first_triangle << 4 | count - 1; WIDE_EMPTY: no slot.
// Per-lane stack of {ref, tmin}: the first WIDE_STACK_LDS entries in LDS ([entry][thread], conflict-free), deeper ones
// in a per-lane global array (rare: the dynamic depth of a 4-wide walk is mostly 2..6).
#pragma once
#include "bvh_traverse.hpp"

namespace cl2 {

constexpr int WIDE_EMPTY = (int)0x80000000;
constexpr int WIDE_STACK_LDS = 8;
// Triangle pairs a lane in a leaf tests per pass.  The pass -- its refill check, its node block for the other lanes, its
// bookkeeping, its dependent fetch -- is the unit of cost of the persistent walks, so a leaf of three or four
// triangles should not take two of them: a second pair in the same pass (fetched after the first is tested) took the
// connection launch from 4.78 to 3.78 ms on the glass scene and from 5.94 to 4.99 ms on the blob (three pairs: 4.15 /
// 5.45, four: 3.69 / 5.14).
#ifndef WIDE_TRI_REPS
#define WIDE_TRI_REPS 2
#endif
// (Two node visits per pass, by the same reasoning, do NOT pay: 3.80 -> 4.12 ms; nor does a second triangle round in the
// binary walks, whose pass already tests the first triangles of a leaf together with the node visit that enters it:
// whole-subpath launch 5.87 -> 6.30 ms on the glass scene, connection launch of the 1M-triangle scene 15.8 -> 16.3 ms.)
constexpr int WIDE_STACK_OVERFLOW = 136;      // 2 x the reference's 64-entry stack bound + slack: cannot be exceeded (Q18 check at upload)

struct WideView {
    const float4* nodes;       // 8 float4 per wide node
    const float4* tris;        // 3 float4 per triangle (the binary walk's array)
    float4 root_lo, root_hi;   // the root box: tested once per ray, as the reference does
    int2* overflow;            // [lanes of the launch][ovf_stride]
    int ovf_stride;            // entries per lane: the deepest stack the uploaded tree can produce (<= WIDE_STACK_OVERFLOW)
    int stack_lds;             // stack entries kept in LDS per lane (<= WIDE_STACK_LDS)
    int n_lds_nodes;           // wide nodes [0, n_lds_nodes) (breadth-first numbering: the top of the tree) staged in LDS
};

template <bool COUNT, bool TWO_TRIS, class Source>
__device__ __forceinline__ void traverse_wide_persistent(const WideView& w, const BvhView& b, unsigned n, unsigned* work_counter, const Source& src,
                                                         unsigned& n_box, unsigned& n_tri) {
    extern __shared__ float4 cl2_tree_lds[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int S = w.stack_lds;
    float4* s_nodes = cl2_tree_lds;                                                     // [8 * n_lds_nodes]
    int* s_ref = reinterpret_cast<int*>(cl2_tree_lds + 8 * w.n_lds_nodes);              // [S][blockDim.x]
    float* s_tmin = reinterpret_cast<float*>(s_ref) + S * nt;
    for (int i = tid; i < 8 * w.n_lds_nodes; i += nt) s_nodes[i] = w.nodes[i];
    __syncthreads();
    int2* ovf = w.overflow + ((size_t)blockIdx.x * nt + tid) * w.ovf_stride;
    const int lane = tid & 63;
    const unsigned waves = gridDim.x * (blockDim.x >> 6);
    unsigned chunk = n / (waves * 4u);
    chunk = chunk < 64u ? 64u : (chunk > (unsigned)RAY_CHUNK_MAX ? (unsigned)RAY_CHUNK_MAX : chunk);
    unsigned w_next = 0, w_end = 0;
    bool dry = false;
    // per-lane ray state
    bool active = false;
    V3 o = v3(0, 0, 0), d = o, inv = o;
    Hit best{-1, __builtin_inff(), 0.0f, 0.0f};
    int cur = -1, tri_i = 0, tri_end = 0, sp = 0;
    int node = 0x7fffffff;                     // binary walk of a ray with a non-finite 1/d (wlane == false)
    bool wlane = true;
    const int n_nodes = b.n_nodes;
    int key = 0;                               // the source's token of the lane's ray (pixel id / tag): store() needs it again

    auto push = [&](int ref, float tmin) {
        if (sp < S) { s_ref[sp * nt + tid] = ref; s_tmin[sp * nt + tid] = tmin; }
        else ovf[sp - S] = make_int2(ref, __float_as_int(tmin));
        sp++;
    };
    // next work item of the lane: pops until an entry still passes `tmin < best_t` (the reference's test of that box
    // at this moment) or the stack is empty
    auto pop_next = [&]() {
        while (sp > 0 && cur < 0 && tri_i >= tri_end) {
            sp--;
            int ref; float tmin;
            if (sp < S) { ref = s_ref[sp * nt + tid]; tmin = s_tmin[sp * nt + tid]; }
            else { const int2 e = ovf[sp - S]; ref = e.x; tmin = __int_as_float(e.y); }
            if (!(tmin < best.t)) continue;
            if (ref >= 0) cur = ref;
            else { const int info = ~ref; tri_i = info >> 4; tri_end = tri_i + (info & 15) + 1; }
        }
    };

    while (true) {
        // ---- refill idle lanes ----
        unsigned long long idle = __ballot(!active);
        if (__popcll(idle) < REFILL_MIN_WIDE && __popcll(idle) < 64) idle = 0;                 // not yet: see REFILL_MIN_WIDE (bvh_traverse.hpp)
        while (idle && !dry) {
            if (w_next >= w_end) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(work_counter, chunk);
                base = __shfl(base, 0);
                if (base >= n) { dry = true; break; }
                w_next = base;
                w_end = base + chunk < n ? base + chunk : n;
            }
            const unsigned avail = w_end - w_next;
            const unsigned rank = __popcll(idle & ((1ull << lane) - 1ull));
            if (!active && rank < avail) {
                key = src.load(w_next + rank, o, d);
                inv = rcp3(d);
                best = Hit{-1, __builtin_inff(), 0.0f, 0.0f};
                cur = -1; tri_i = 0; tri_end = 0; sp = 0;
                wlane = finite3(inv);
                node = wlane ? n_nodes : 0;
                active = true;
                if (wlane) {
                    // the root box, trace.metal:150-156 with best_t = inf
                    const float t0x = (w.root_lo.x - o.x) * inv.x, t0y = (w.root_lo.y - o.y) * inv.y, t0z = (w.root_lo.z - o.z) * inv.z;
                    const float t1x = (w.root_hi.x - o.x) * inv.x, t1y = (w.root_hi.y - o.y) * inv.y, t1z = (w.root_hi.z - o.z) * inv.z;
                    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                                       __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
                    const float tmax = __builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fminf(__builtin_fmaxf(t0y, t1y), __builtin_fmaxf(t0z, t1z)));
                    if (COUNT) n_box++;
                    if (tmin <= tmax && tmin < best.t) cur = 0;
                }
            }
            const unsigned taken = __popcll(idle) < avail ? __popcll(idle) : avail;
            w_next += taken;
            idle = __ballot(!active);
        }
        if (!__any(active)) break;

        if (active) {
            if (!wlane) {
                if (tri_i >= tri_end && node < n_nodes) {
                    const float4 lo = b.nodes[2 * node], hi = b.nodes[2 * node + 1];
                    const int next = __float_as_int(lo.w);
                    const float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
                    const float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
                    const float tmin = max_msl(max_msl(min_msl(t0x, t1x), min_msl(t0y, t1y)), max_msl(min_msl(t0z, t1z), 0.0f));
                    const float tmax = min_msl(min_msl(max_msl(t0x, t1x), max_msl(t0y, t1y)), min_msl(max_msl(t0z, t1z), __builtin_inff()));
                    node = next;
                    if (tmin <= tmax && tmin < best.t) {
                        const int info = __float_as_int(hi.w);
                        if (info < 0) node = ~info;
                        else { tri_i = info >> 4; tri_end = tri_i + (info & 15) + 1; }
                    }
                }
            } else {
            pop_next();                                                     // lanes that finished a leaf in the previous pass
            if (cur >= 0) {
                // one wide node: the slab tests of up to four boxes, in the reference's visit order (slot 0 first)
                const float4* nd = cur < w.n_lds_nodes ? s_nodes + 8 * cur : w.nodes + (size_t)8 * cur;
                const float4 lx = nd[0], ly = nd[1], lz = nd[2], hx = nd[3], hy = nd[4], hz = nd[5];
                const float4 rf = nd[6];
                cur = -1;
                const float lox[4] = {lx.x, lx.y, lx.z, lx.w}, loy[4] = {ly.x, ly.y, ly.z, ly.w}, loz[4] = {lz.x, lz.y, lz.z, lz.w};
                const float hix[4] = {hx.x, hx.y, hx.z, hx.w}, hiy[4] = {hy.x, hy.y, hy.z, hy.w}, hiz[4] = {hz.x, hz.y, hz.z, hz.w};
                const int ref[4] = {__float_as_int(rf.x), __float_as_int(rf.y), __float_as_int(rf.z), __float_as_int(rf.w)};
                // slots are tested last to first and pushed, so that the first one pops first -- except the one that WOULD pop
                // first: it is what the lane does next anyway (it has just passed `tmin < best_t`, and best_t has not moved),
                // so it never goes through the stack (one LDS write + read less per visit, one entry less of depth)
                int next_ref = WIDE_EMPTY;
                float next_tmin = 0.0f;
#pragma unroll
                for (int k = 3; k >= 0; k--) {
                    if (ref[k] == WIDE_EMPTY) continue;
                    const float t0x = (lox[k] - o.x) * inv.x, t0y = (loy[k] - o.y) * inv.y, t0z = (loz[k] - o.z) * inv.z;
                    const float t1x = (hix[k] - o.x) * inv.x, t1y = (hiy[k] - o.y) * inv.y, t1z = (hiz[k] - o.z) * inv.z;
                    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                                       __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
                    const float tmax = __builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fminf(__builtin_fmaxf(t0y, t1y), __builtin_fmaxf(t0z, t1z)));
                    if (COUNT) n_box++;
                    if (tmin <= tmax && tmin < best.t) {
                        if (next_ref != WIDE_EMPTY) push(next_ref, next_tmin);
                        next_ref = ref[k]; next_tmin = tmin;
                    }
                }
                if (next_ref == WIDE_EMPTY) pop_next();
                else if (next_ref >= 0) cur = next_ref;
                else { const int info = ~next_ref; tri_i = info >> 4; tri_end = tri_i + (info & 15) + 1; }
            }
            }
#pragma unroll
            for (int rep = 0; rep < (TWO_TRIS ? WIDE_TRI_REPS : 1); rep++)
            if (tri_i < tri_end && (rep == 0 || wlane)) {
                const int i0 = tri_i;
                const bool two = TWO_TRIS && i0 + 1 < tri_end;
                const int i1 = two ? i0 + 1 : i0;
                tri_i = i1 + 1;
                float4 a0, a1, a2, c0, c1, c2;
                a0 = w.tris[3 * i0]; a1 = w.tris[3 * i0 + 1]; a2 = w.tris[3 * i0 + 2];
                if (TWO_TRIS) { c0 = w.tris[3 * i1]; c1 = w.tris[3 * i1 + 1]; c2 = w.tris[3 * i1 + 2]; }
                if (COUNT) n_tri += two ? 2 : 1;
                tri_test(o, d, a0, a1, a2, i0, best);
                if (TWO_TRIS && two) tri_test(o, d, c0, c1, c2, i1, best);
            }
            if (tri_i >= tri_end && (wlane ? (cur < 0 && sp == 0) : node >= n_nodes)) {
                src.store(key, best);
                active = false;
            }
        }
    }
}

}  // namespace cl2
