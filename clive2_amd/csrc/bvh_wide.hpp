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
#ifndef CL2_WIDE_STACK_LDS
#define CL2_WIDE_STACK_LDS 8
#endif
constexpr int WIDE_STACK_LDS = CL2_WIDE_STACK_LDS;
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

#ifdef CL2_WALK_HISTO
// instrumentation build only (tools/exp_walk_histo.py): per pass of the wide walk, how many lanes visit a node / test triangles
// (first and second pair round) / are idle, and how many triangles a leaf has when it is entered
__device__ unsigned long long g_walk_histo[6][65];
#endif

struct WideView {
    const float4* nodes;       // 8 float4 per wide node
    const float4* tris;        // 3 float4 per triangle (the binary walk's array)
    const float* tris36;       // PACK: the same records without their padding words, 9 floats {v0, v1 - v0, v2 - v0} per triangle (nullptr: none)
    float4 root_lo, root_hi;   // the root box: tested once per ray, as the reference does
    int2* overflow;            // [lanes of the launch][ovf_stride]
    int ovf_stride;            // entries per lane: the deepest stack the uploaded tree can produce (<= WIDE_STACK_OVERFLOW)
    int stack_lds;             // stack entries kept in LDS per lane (<= WIDE_STACK_LDS)
    int n_lds_nodes;           // wide nodes [0, n_lds_nodes) (breadth-first numbering: the top of the tree) staged in LDS
    const int* tri_rank;       // ORDER only: each triangle's position in the reference's visit order; read when two hits tie at exactly one t
};

// Round 4: the pass rewritten for its instruction count.  The ISA of round 3's loop showed what a pass paid beside its arithmetic:
// the node record came by FLAT loads (a per-lane select between the LDS window and global memory is a generic pointer: such a load
// waits on BOTH memory counters) and in two dependent rounds (the fourth slot's bounds were loaded under `ref != EMPTY`); stack pops
// were flat loads too (LDS entry or global overflow entry, one selected pointer); every conditional push, every early exit of the
// triangle test and every skipped slot was an exec-mask region (3-4 scalar instructions each: ~310 scalar per ~600 vector
// instructions per pass, at the one-per-two ratio where the CU's single scalar unit binds, tools/issue_mix.hip); LDS addresses were
// computed with a multiply by blockDim.x; SGPRs spilled into VGPR lanes.  Here:
//   * a node visit is 7 x 16-byte global loads issued together, four unconditional slab tests (an empty slot holds a box at
//     +inf, which can never pass `tmin <= tmax && tmin < best_t`: cl2_upload_scene) and three UNCONDITIONAL 8-byte LDS writes
//     of the candidate found so far with `sp += pushed` -- no exec-mask region at all.  Lanes within three entries of the LDS
//     part's end take the conditional form with the global overflow array (rare, wave-level branch);
//   * the stack is int2 {ref, tmin} per entry ([entry][thread], one ds_read_b64 / ds_write_b64), 8 entries in LDS, addressed
//     by shift; the overflow accesses are `volatile` so that they cannot be merged with the LDS ones into a flat access;
//   * the triangle test has no early exits: in a wave of 40 lanes testing 40 different triangles some lane always passes the
//     `u` test, so the exits only cost scalar instructions; one predicated update at the end, identical decisions
//     (NaN included: `!(u < 0 || u > 1)` etc. are kept in that form).  The second triangle of a pair is tested
//     unconditionally: a leaf with an odd count re-tests its last triangle, which cannot pass `t < best_t` a second time;
//   * the LDS window of the top of the tree (launch_wide: 32 wide nodes while the tree is cache-resident, 64 when it streams
//     from memory) is read in a branch of its own, kept apart from the global loads of the nodes below it -- never through a
//     per-lane pointer select.
// Per ray the sequence of box tests, triangle tests and comparisons is unchanged (same parity tests, same fuzz).
constexpr int WIDE_NT = 256;                  // threads per workgroup of the wide launches (BLOCK)
constexpr int WIDE_S = WIDE_STACK_LDS;        // stack entries per lane in LDS

// TRI_REPS: triangle pairs a lane in a leaf tests per pass (WIDE_TRI_REPS = 2 while the tree is cache-resident; 1 when it streams
// from memory: round 4, same box, 1M triangles: connection launch 11.2 -> 10.7 ms, sample 22.3 -> 21.3 ms).
// TALLY (cl2_set_counting(2)): per-lane counts of what THIS walk fetches -- wide nodes visited, distinct triangle records read,
// stack entries that went to the global overflow array, binary records of the rays with a non-finite 1/d -- for the bytes the
// kernel itself asks of the memory system (bench.py: own bytes = 112 B per wide node + 48 B per triangle record + 48 B per ray).
struct WalkTally { unsigned visits = 0, tri_records = 0, spills = 0, bin_nodes = 0; };

// (Round 5 measured a next-line prefetch here -- a lane that tests triangles asking, by an LDS-direct load into a sink, for the next
// pair of its leaf or for what the top of its stack points at: subpath launches +7...9 %, connection launch +3 %, on the 5k- and the
// 1M-triangle scene alike; profiles/r05_prefetch_next_line.patch.  Loads return in order, so the request only moves the wait.)
// SPEC (round 5; debug bit 13 switches it off for A/B runs and tests): a lane that is busy with a leaf expands the wide node on top
// of its stack in the same pass -- the node block, which round 4's pass statistics showed running for 32 of 64 lanes, then runs
// for nearly all of them, and a ray needs fewer passes (same-box A/B, 8 sample streams, ms per sample: 5k triangles 6.64 -> 6.31-6.43,
// 82k 7.92 -> 7.66, 1M 18.1 -> 17.7-17.9; connection launch alone 2.69 -> 2.55 / 3.23 -> 3.06 / 8.85 -> 8.66;
// profiles/r05_spec_*.log).  Exact: the entry leaves the stack either
// way -- pruned now (`tmin >= best_t`, and best_t only shrinks: pruned later too) or replaced by its passing children, ALL of them
// pushed in the reference's order and each re-tested against best_t when it is popped; a child that passes under today's best_t
// but not under the one at pop time is dropped there, exactly as the reference, which tests the parent after the leaf, would never
// have pushed it (a child's tmin is >= its parent's).  Only while the four possible pushes fit the LDS part of the stack.
// PACK (round 6; debug bit 14 switches it off): trees that stream from beyond L2 (config 5: the launch moves 0.75 of the HBM peak
// across the fabric, and most of the lines that miss L2 are TRIANGLE lines) read their triangles as 36-byte records -- a leaf of n
// triangles is 36 n contiguous bytes instead of 48 n, i.e. fewer 128-byte lines per leaf visit and 12 MB less to keep in the caches
// per million triangles.  Same values, same operations, same order.
// ORDER (round 6; cl2_set_traversal_order(1), never the default and never the parity path): the passing slots of a node are taken
// NEAREST FIRST (by entry distance, a stable 4-element network on {tmin, ref}; ties keep the reference's slot order) instead of in
// the reference's fixed order.  A closest-hit query then finds a near hit early and prunes what lies behind it: fewer node visits
// and triangle tests per ray.  What the reference's result depends on its visit order for, and what this walk does about it:
//   * two triangles hit at exactly the same t: the first one the reference meets wins (`t < best_t`, trace.metal:170).  SETTLED the
//     same way here: boxes are pruned on `tmin > best_t` instead of `>=` (a leaf that may hold a hit at exactly best_t is still
//     entered), and a hit at exactly best_t replaces the held triangle when the reference would have met it first (tri_rank[]: each
//     triangle's position in the reference's visit order, read inside a wave-level branch that is taken for a few rays in 1e8);
//   * a hit that lies IN FRONT of its own leaf box's entry distance (rounding: a few ulp): the reference enters that leaf or not
//     depending on the best_t it holds when it gets there (trace.metal:152), and so does any other order.  NOT reproducible in
//     another order -- this is why the walk is not bit-exact by construction.
// For every other ray the two walks return the same (triangle, t, u, v): let x be the first triangle in the reference's order among
// the nearest hits of the leaves the ray's slabs enter; if t_x >= tmin of x's leaf, the reference finds x (when it gets to x's leaf it
// holds best_t > t_x >= tmin: entered) and so does this walk (best_t >= t_x >= tmin: entered under `<=`; the tie rule keeps x).
// tests/test_gpu_round6.py, test_gpu_fullsize.py: no differing ray is a tie, every differing ray is such a hit (2 / 3 / 5 rays of
// 3.4e8 on configs 3 / 4 / 5).
template <int TRI_REPS, bool TALLY, bool SPEC, bool PACK, bool ORDER, class Source>
__device__ __forceinline__ void traverse_wide_persistent(const WideView& w, const BvhView& b, unsigned n, unsigned* work_counter, const Source& src, WalkTally& tally) {
    constexpr bool TWO_TRIS = true;
    extern __shared__ float4 cl2_tree_lds[];
    const int tid = threadIdx.x;
    constexpr int NT = WIDE_NT;
    const int n_win = w.n_lds_nodes;
    float4* s_nodes = cl2_tree_lds;                                                     // [8 * n_win]
    int2* s_stack = reinterpret_cast<int2*>(cl2_tree_lds + 8 * n_win) + tid;            // entry e of this lane: s_stack[e * NT]
    if (n_win > 0) {
        for (int i = tid; i < 8 * n_win; i += NT) s_nodes[i] = w.nodes[i];
        __syncthreads();
    }
    volatile int* ovf = reinterpret_cast<volatile int*>(w.overflow + ((size_t)blockIdx.x * NT + tid) * w.ovf_stride);
    const unsigned waves = gridDim.x * (NT >> 6);
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

    auto take = [&](int ref) {                 // a stack entry / slot reference becomes the lane's next piece of work
        if (ref >= 0) cur = ref;
        else { const int info = ~ref; tri_i = info >> 4; tri_end = tri_i + (info & 15) + 1; }
    };
    // Lanes with nothing in hand pop until an entry still passes `tmin < best_t` (the reference's test of that box at this
    // moment) or their stack is empty.  A wave-level loop: one ds_read_b64 per lane and round.
    auto pop_loop = [&]() {
        bool need = active && wlane && cur < 0 && tri_i >= tri_end && sp > 0;
        while (__any(need)) {
            if (need) {
                sp--;
                int ref, tbits;
                if (sp < WIDE_S) { const int2 e = s_stack[sp * NT]; ref = e.x; tbits = e.y; }
                else { ref = ovf[2 * (sp - WIDE_S)]; tbits = ovf[2 * (sp - WIDE_S) + 1]; }
                if (ORDER ? __int_as_float(tbits) <= best.t : __int_as_float(tbits) < best.t) { take(ref); need = false; }
                else need = sp > 0;
            }
        }
    };

    while (true) {
        // ---- refill idle lanes ----
        unsigned long long idle = __ballot(!active);
        if (__popcll(idle) < REFILL_MIN_WIDE && __popcll(idle) < 64) idle = 0;                 // not yet: see REFILL_MIN_WIDE (bvh_traverse.hpp)
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
                    if (tmin <= tmax && tmin < best.t) cur = 0;
                }
            }
            const unsigned taken = __popcll(idle) < avail ? __popcll(idle) : avail;
            w_next += taken;
            idle = __ballot(!active);
        }
        if (!__any(active)) break;
#ifdef CL2_WALK_HISTO
        bool h_node = false, h_t0 = false, h_t1 = false;
        const int h_tri_before = tri_end - tri_i;
#endif

        // ---- rays with a non-finite 1/d: one node of the binary stackless walk (records through the caches, MSL min / max) ----
        if (__any(active && !wlane)) {
            if (active && !wlane && tri_i >= tri_end && node < n_nodes) {
                const float4 lo = b.nodes[2 * node], hi = b.nodes[2 * node + 1];
                const int next = __float_as_int(lo.w);
                if (TALLY) tally.bin_nodes++;
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
        }

        pop_loop();                                                         // lanes that finished a leaf in the previous pass
        bool spec = false;
        int spec_ref = -1;
        if (SPEC) {
            const bool cand = active && wlane && cur < 0 && tri_i < tri_end && sp > 0 && sp + 3 <= WIDE_S;
            if (cand) {
                const int2 e = s_stack[(sp - 1) * NT];
                if (e.x >= 0) {                                             // a wide node on top: it leaves the stack, pruned or expanded
                    sp--;
                    spec = ORDER ? __int_as_float(e.y) <= best.t : __int_as_float(e.y) < best.t;
                    spec_ref = e.x;
                }
            }
        }
        // ---- one wide node per lane that stands at one: the slab tests of its four boxes in the reference's visit order ----
        const bool visit = active && wlane && (cur >= 0 || (SPEC && spec));
#ifdef CL2_WALK_HISTO
        h_node = visit;
#endif
        if (__any(visit)) {
            if (visit) {
                float4 lx, ly, lz, hx, hy, hz, rf;
                const int vnode = (SPEC && spec) ? spec_ref : cur;
                if (n_win > 0 && vnode < n_win) {
                    const float4* nd = s_nodes + 8 * vnode;
                    lx = nd[0]; ly = nd[1]; lz = nd[2]; hx = nd[3]; hy = nd[4]; hz = nd[5]; rf = nd[6];
                    asm volatile("" ::: "memory");                          // keeps this branch's LDS reads apart from the global loads below
                } else {
                    const float4* __restrict__ nd = w.nodes + (size_t)8 * vnode;
                    lx = nd[0]; ly = nd[1]; lz = nd[2]; hx = nd[3]; hy = nd[4]; hz = nd[5]; rf = nd[6];
#ifdef CL2_EXTRA_NODE_LOADS
                    // measurement build only (round 6): N more 16-byte loads of the node's OWN line (its padding words, all zero) -- no new
                    // line, no new miss, only N more L1 look-ups per visit.  What the launch pays for them is what an L1 look-up costs.
                    float extra = 0.0f;
                    int vn[CL2_EXTRA_NODE_LOADS];
                    float ex[CL2_EXTRA_NODE_LOADS];
#pragma unroll
                    for (int e = 0; e < CL2_EXTRA_NODE_LOADS; e++) { vn[e] = vnode; asm("; index copy %1" : "+v"(vn[e]) : "i"(e)); }   // indices the compiler cannot prove equal: PLAIN global loads of their own
#pragma unroll
                    for (int e = 0; e < CL2_EXTRA_NODE_LOADS; e++) ex[e] = w.nodes[(size_t)8 * vn[e] + 7].x;
#pragma unroll
                    for (int e = 0; e < CL2_EXTRA_NODE_LOADS; e++) extra += ex[e];
                    lx.x += extra;
#endif
                }
                cur = -1;
                if (TALLY) tally.visits++;
                const float lox[4] = {lx.x, lx.y, lx.z, lx.w}, loy[4] = {ly.x, ly.y, ly.z, ly.w}, loz[4] = {lz.x, lz.y, lz.z, lz.w};
                const float hix[4] = {hx.x, hx.y, hx.z, hx.w}, hiy[4] = {hy.x, hy.y, hy.z, hy.w}, hiz[4] = {hz.x, hz.y, hz.z, hz.w};
                int ref[4] = {__float_as_int(rf.x), __float_as_int(rf.y), __float_as_int(rf.z), __float_as_int(rf.w)};
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
                    // an empty slot's box lies at +inf: never.  (ORDER: `<=`, so that a leaf that may hold a hit at exactly best_t is still
                    // entered -- the tie rule then decides as the reference does; an empty slot goes by its reference, below)
                    pass[k] = tmin <= tmax && (ORDER ? tmin <= best.t : tmin < best.t);
                }
                if (ORDER) {
                    // nearest first: slots that do not pass sort behind all that do (key +inf, no reference); adjacent
                    // exchanges only, on strict `>`, so equal keys keep the reference's slot order
#pragma unroll
                    for (int k = 0; k < 4; k++) { tm[k] = pass[k] ? tm[k] : __builtin_inff(); ref[k] = pass[k] ? ref[k] : WIDE_EMPTY; }
#define CL2_CSWAP(A, B) { const bool sw = tm[A] > tm[B]; const float ta = tm[A], tb = tm[B]; const int ra = ref[A], rb = ref[B]; \
                          tm[A] = sw ? tb : ta; tm[B] = sw ? ta : tb; ref[A] = sw ? rb : ra; ref[B] = sw ? ra : rb; }
                    CL2_CSWAP(0, 1) CL2_CSWAP(1, 2) CL2_CSWAP(2, 3) CL2_CSWAP(0, 1) CL2_CSWAP(1, 2) CL2_CSWAP(0, 1)
#undef CL2_CSWAP
#pragma unroll
                    for (int k = 0; k < 4; k++) pass[k] = ref[k] != WIDE_EMPTY;
                }
                // Slots are taken last to first, so that the first one pops first; the candidate found so far is pushed when
                // another slot passes in front of it, and the one left at the end -- the slot that WOULD pop first -- is what the
                // lane does next (it has just passed `tmin < best_t`, and best_t has not moved): it never goes through the stack.
                int next_ref = pass[3] ? ref[3] : WIDE_EMPTY;
                float next_tmin = tm[3];
                if (!__any(sp + 3 > WIDE_S)) {
                    // all three possible pushes fit the LDS part: the candidate is WRITTEN unconditionally above the top of the
                    // stack and only `sp` says whether it was pushed
#pragma unroll
                    for (int k = 2; k >= 0; k--) {
                        s_stack[sp * NT] = make_int2(next_ref, __float_as_int(next_tmin));
                        sp += (pass[k] && next_ref != WIDE_EMPTY) ? 1 : 0;
                        next_ref = pass[k] ? ref[k] : next_ref;
                        next_tmin = pass[k] ? tm[k] : next_tmin;
                    }
                } else {
#pragma unroll
                    for (int k = 2; k >= 0; k--) {
                        if (pass[k]) {
                            if (next_ref != WIDE_EMPTY) {
                                if (sp < WIDE_S) s_stack[sp * NT] = make_int2(next_ref, __float_as_int(next_tmin));
                                else { ovf[2 * (sp - WIDE_S)] = next_ref; ovf[2 * (sp - WIDE_S) + 1] = __float_as_int(next_tmin); if (TALLY) tally.spills++; }
                                sp++;
                            }
                            next_ref = ref[k]; next_tmin = tm[k];
                        }
                    }
                }
                if (SPEC && spec) {
                    // the lane is busy with its leaf: the slot that would pop first waits on the stack like the others
                    s_stack[sp * NT] = make_int2(next_ref, __float_as_int(next_tmin));
                    sp += next_ref != WIDE_EMPTY ? 1 : 0;
                } else if (next_ref != WIDE_EMPTY) take(next_ref);
            }
            pop_loop();                                                     // lanes whose visit left them empty-handed
        }
#ifdef CL2_WALK_HISTO
        if (active && wlane && tri_end - tri_i > 0 && tri_end - tri_i != h_tri_before) atomicAdd(&g_walk_histo[4][tri_end - tri_i], 1ull);   // a leaf was entered
#endif

        // ---- the triangles of the lane's leaf, a pair per round: ray_triangle_intersect (trace.metal:117-142) without early
        // exits, one predicated update; the second of the pair sees the first one's best_t, as in the reference's loop ----
#pragma unroll
        for (int rep = 0; rep < TRI_REPS; rep++) {
            const bool has = active && tri_i < tri_end;
            if (!__any(has)) break;
            if (has) {
#ifdef CL2_WALK_HISTO
                if (rep == 0) h_t0 = true; else h_t1 = true;
#endif
                const int i0 = tri_i;
                const int i1 = (TWO_TRIS && i0 + 1 < tri_end) ? i0 + 1 : i0;
                tri_i = i1 + 1;
                if (TALLY) tally.tri_records += (i1 != i0) ? 2u : 1u;
                if (PACK) {
                    // the pair as ONE run of 72 bytes (five loads instead of six: what binds this walk is the L1's look-up rate, one
                    // look-up per 16-byte load of a lane, profiles/r06_l1_lookup_cost.log).  A leaf's last odd triangle has no
                    // partner: the record behind it (the next leaf's first triangle; the array ends in one record of padding) is
                    // loaded and its test masked out -- where the 48-byte form re-tests the same triangle, which cannot pass twice.
                    const float* __restrict__ ta = w.tris36 + (size_t)9 * i0;
                    float a[18];
#pragma unroll
                    for (int k = 0; k < 18; k++) a[k] = ta[k];
                    if (ORDER) {
                        bool tie = tri_test_branchless_tie(o, d, make_float4(a[0], a[1], a[2], 0.0f), make_float4(a[3], a[4], a[5], 0.0f), make_float4(a[6], a[7], a[8], 0.0f), i0, best);
                        Hit second = best;
                        const bool tie2 = tri_test_branchless_tie(o, d, make_float4(a[9], a[10], a[11], 0.0f), make_float4(a[12], a[13], a[14], 0.0f), make_float4(a[15], a[16], a[17], 0.0f), i0 + 1, second);
                        if (i1 != i0) { best = second; tie = tie || tie2; }
                        if (__any(tie)) {
                            // a hit at exactly best_t: the pair again, under the reference's rule for ties (whatever the strict tests kept
                            // holds the same best_t: the repeated tests can only exchange triangles that tie at it)
                            if (tie) {
                                const float* __restrict__ tr = w.tris36 + (size_t)9 * i0;
                                tri_test_tie_rule(o, d, make_float4(tr[0], tr[1], tr[2], 0.0f), make_float4(tr[3], tr[4], tr[5], 0.0f), make_float4(tr[6], tr[7], tr[8], 0.0f), i0, best, w.tri_rank);
                                if (i1 != i0)
                                    tri_test_tie_rule(o, d, make_float4(tr[9], tr[10], tr[11], 0.0f), make_float4(tr[12], tr[13], tr[14], 0.0f), make_float4(tr[15], tr[16], tr[17], 0.0f), i1, best, w.tri_rank);
                            }
                        }
                    } else {
                        tri_test_branchless(o, d, make_float4(a[0], a[1], a[2], 0.0f), make_float4(a[3], a[4], a[5], 0.0f), make_float4(a[6], a[7], a[8], 0.0f), i0, best);
                        Hit second = best;
                        tri_test_branchless(o, d, make_float4(a[9], a[10], a[11], 0.0f), make_float4(a[12], a[13], a[14], 0.0f), make_float4(a[15], a[16], a[17], 0.0f), i0 + 1, second);
                        if (i1 != i0) best = second;
                    }
                } else {
                    const float4* __restrict__ ta = w.tris + (size_t)3 * i0;
                    const float4* __restrict__ tb = w.tris + (size_t)3 * i1;
                    const float4 a0 = ta[0], a1 = ta[1], a2 = ta[2];
                    float4 c0, c1, c2;
                    if (TWO_TRIS) { c0 = tb[0]; c1 = tb[1]; c2 = tb[2]; }
                    if (ORDER) {
                        bool tie = tri_test_branchless_tie(o, d, a0, a1, a2, i0, best);
                        if (TWO_TRIS) tie = (tri_test_branchless_tie(o, d, c0, c1, c2, i1, best) && i1 != i0) || tie;   // (an odd leaf's last triangle is tested twice: no tie)
                        if (__any(tie)) {
                            if (tie) {
                                tri_test_tie_rule(o, d, ta[0], ta[1], ta[2], i0, best, w.tri_rank);
                                if (TWO_TRIS) tri_test_tie_rule(o, d, tb[0], tb[1], tb[2], i1, best, w.tri_rank);
                            }
                        }
                    } else {
                        tri_test_branchless(o, d, a0, a1, a2, i0, best);
                        if (TWO_TRIS) tri_test_branchless(o, d, c0, c1, c2, i1, best);
                    }
                }
            }
        }
        // ---- retire: nothing in hand, nothing on the stack ----
        if (active && tri_i >= tri_end && (wlane ? (cur < 0 && sp == 0) : node >= n_nodes)) {
            src.store(key, best);
            active = false;
        }
#ifdef CL2_WALK_HISTO
        {
            const int a = __popcll(__ballot(h_node)), bb = __popcll(__ballot(h_t0)), c = __popcll(__ballot(h_t1)), dd = __popcll(__ballot(!h_node && !h_t0));
            int spmax = active ? sp : 0;
            for (int off = 32; off > 0; off >>= 1) { const int other = __shfl_xor(spmax, off); spmax = other > spmax ? other : spmax; }
            if (wave_lane() == 0) {
                atomicAdd(&g_walk_histo[0][a], 1ull); atomicAdd(&g_walk_histo[1][bb], 1ull);
                atomicAdd(&g_walk_histo[2][c], 1ull); atomicAdd(&g_walk_histo[3][dd], 1ull);
                atomicAdd(&g_walk_histo[5][spmax > 64 ? 64 : spmax], 1ull);
            }
        }
#endif
    }
}

}  // namespace cl2
