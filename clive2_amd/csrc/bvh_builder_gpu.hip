// bvh_builder_gpu.hip -- BVH construction on the GPU (second half): a PLOC tree (round 3; the
// round-2 LBVH is kept behind CLIVE2_GPU_BVH=lbvh for comparison) whose output is
// the reference's flattened convention (`np_flatten_bvh`, src/bvh.py:329-389): Box records numbered breadth-first,
// the two children of an inner box adjacent at (left, left+1) with right == 0, a leaf holding the range
// [left, right) of the leaf-ordered triangle list, at most `max_members` triangles per leaf (bvh.py:294).
//
// The reference grows its tree top-down with a full-sweep SAH split per node in numpy (object_split,
// bvh.py:132-191): minutes for 1M triangles; the host builder of this library (bvh_builder.hpp) applies the same rule
// in O(n log n).  This builder trades tree quality for set-up time: triangles are sorted along a 63-bit Morton curve
// of their centroids (rocPRIM radix sort), the binary radix tree over the sorted keys is built in one launch (Karras
// 2012: every inner node finds its key range and split independently), boxes are fitted bottom-up (the second child
// to arrive at a parent continues), and subtrees of at most `max_members` triangles become leaves.  The tracer
// accepts any valid tree in the convention (its walk, like the reference's, does not depend on how the tree was
// built), so renders on a GPU-built tree are compared with the oracle run on the same Box[] (tests/test_gpu_bvh.py).
//
// Round 3: the hierarchy above the sorted triangles is built by PLOC (parallel locally-ordered clustering, Meister &
// Bittner 2018) instead of the radix tree: every cluster looks `PLOC_RADIUS` positions to either side along the Morton
// order for the neighbour whose union with it has the smallest surface area, mutual nearest neighbours merge, the
// merged list is compacted (order kept) and the round repeats until one cluster is left.  A radix tree splits space
// at Morton bit boundaries whatever lies there; the merges follow the geometry.  Measured on the 1M-triangle scene
// (node tests per ray, reference walk, 200k random rays through the room, oracle counters): host SAH 36.4, LBVH 47.2,
// PLOC radius 8 / 16 / 32: 38.3 / 39.8 / 39.2 (an SAH-swept tree over the LBVH's leaves: 38.0; extended Morton codes
// with size bits: 42.9).
//
// Two details follow from the reference's traversal (trace.metal:144-176):
//   * it pops box left+1 before box left and keeps the other one on a 64-entry stack: the SMALLER subtree is stored
//     at left+1, so the number of pending entries is bounded by log2(n) whatever the depth of the radix tree
//     (cl2_upload_scene refuses trees that could overflow that stack, quirk Q18);
//   * box bounds are the float32 roundings of the float64 triangle bounds, as np_flatten_bvh stores them; rounding
//     is monotonic, so min/max over rounded values equals the rounded min/max and every box contains its triangles'
//     float32 vertices exactly.
#include <cstring>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/clive2_amd.h"
#include "bvh_builder.hpp"      // HostBox

namespace cl2 {
namespace lbvh {

constexpr int THREADS = 256;

// order-preserving map float -> uint32 (for atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

struct Node {                // inner node i of the radix tree over the sorted keys
    int left, right;         // child ids: >= 0 inner node, < 0 leaf ~id (sorted position)
    int first, last;         // range of sorted positions covered
};

// float32 bounds (rounded from the float64 inputs), centroid, and the bounds of all centroids
__global__ void k_prepare(int n, const double* __restrict__ tmin, const double* __restrict__ tmax, float4* __restrict__ lo,
                          float4* __restrict__ hi, float4* __restrict__ centroid, unsigned* __restrict__ cbounds /*[6]*/) {
    const int i = blockIdx.x * THREADS + threadIdx.x;
    float c[3] = {0, 0, 0};
    if (i < n) {
        float a[3], b[3];
        for (int k = 0; k < 3; k++) {
            a[k] = (float)tmin[3 * (size_t)i + k];
            b[k] = (float)tmax[3 * (size_t)i + k];
            c[k] = (float)((tmin[3 * (size_t)i + k] + tmax[3 * (size_t)i + k]) * 0.5);
        }
        lo[i] = make_float4(a[0], a[1], a[2], 0.0f);
        hi[i] = make_float4(b[0], b[1], b[2], 0.0f);
        centroid[i] = make_float4(c[0], c[1], c[2], 0.0f);
    }
    // wave reduction, then one atomic per wave and bound
    for (int k = 0; k < 3; k++) {
        unsigned mn = i < n ? f2ord(c[k]) : 0xFFFFFFFFu, mx = i < n ? f2ord(c[k]) : 0u;
        for (int off = 32; off > 0; off >>= 1) {
            mn = min(mn, (unsigned)__shfl_down(mn, off));
            mx = max(mx, (unsigned)__shfl_down(mx, off));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&cbounds[k], mn); atomicMax(&cbounds[3 + k], mx); }
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned v) {   // 21 bits -> every third bit of 63
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(int n, const float4* __restrict__ centroid, const unsigned* __restrict__ cbounds,
                         unsigned long long* __restrict__ keys, unsigned* __restrict__ ids) {
    const int i = blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float4 c = centroid[i];
    const float v[3] = {c.x, c.y, c.z};
    unsigned q[3];
    for (int k = 0; k < 3; k++) {
        const float a = ord2f(cbounds[k]), b = ord2f(cbounds[3 + k]);
        const float ext = b - a;
        float t = ext > 0.0f ? (v[k] - a) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        q[k] = min((unsigned)(t * 2097152.0f), 2097151u);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    ids[i] = (unsigned)i;
}

// length of the common prefix of keys i and j (equal keys: continue with the positions, Karras 2012 section 4)
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll(a ^ b);
    return 64 + __clz((unsigned)i ^ (unsigned)j);
}

__global__ void k_hierarchy(int n, const unsigned long long* __restrict__ keys, Node* __restrict__ nodes,
                            int* __restrict__ parent_inner, int* __restrict__ parent_leaf) {
    const int i = blockIdx.x * THREADS + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    Node nd;
    nd.first = first; nd.last = last;
    if (first == gamma) { nd.left = ~gamma; parent_leaf[gamma] = i; } else { nd.left = gamma; parent_inner[gamma] = i; }
    if (last == gamma + 1) { nd.right = ~(gamma + 1); parent_leaf[gamma + 1] = i; } else { nd.right = gamma + 1; parent_inner[gamma + 1] = i; }
    nodes[i] = nd;
    if (i == 0) parent_inner[0] = -1;
}

// bottom-up fit: the first child to arrive at a node stops, the second one (both children complete) computes the box
__global__ void k_fit(int n, const unsigned* __restrict__ ids, const float4* __restrict__ lo, const float4* __restrict__ hi,
                      const Node* __restrict__ nodes, const int* __restrict__ parent_inner, const int* __restrict__ parent_leaf,
                      float4* __restrict__ node_lo, float4* __restrict__ node_hi, unsigned* __restrict__ arrived) {
    const int p = blockIdx.x * THREADS + threadIdx.x;
    if (p >= n) return;
    int node = parent_leaf[p];
    while (node >= 0) {
        __threadfence();                                   // this thread's box stores before its arrival
        if (atomicAdd(&arrived[node], 1u) == 0u) return;   // the sibling subtree is not complete yet
        __threadfence();                                   // the sibling's box stores after its arrival
        const Node nd = nodes[node];
        float4 a, b, c, d;
        if (nd.left < 0) { const unsigned t = ids[~nd.left]; a = lo[t]; b = hi[t]; }
        else { a = node_lo[nd.left]; b = node_hi[nd.left]; }
        if (nd.right < 0) { const unsigned t = ids[~nd.right]; c = lo[t]; d = hi[t]; }
        else { c = node_lo[nd.right]; d = node_hi[nd.right]; }
        node_lo[node] = make_float4(fminf(a.x, c.x), fminf(a.y, c.y), fminf(a.z, c.z), 0.0f);
        node_hi[node] = make_float4(fmaxf(b.x, d.x), fmaxf(b.y, d.y), fmaxf(b.z, d.z), 0.0f);
        node = parent_inner[node];
    }
}

// ---- PLOC ----------------------------------------------------------------------------------------------------
constexpr int PLOC_RADIUS = 8;

__device__ __forceinline__ float half_area(float4 lo, float4 hi) {
    const float x = hi.x - lo.x, y = hi.y - lo.y, z = hi.z - lo.z;
    return x * y + y * z + z * x;
}

// nearest neighbour of cluster i among positions [i - R, i + R]: the pair with the smallest key (union area, lower
// position, higher position).  The key is a total order on PAIRS, so the pair that is smallest overall is each other's
// choice: every round merges at least one pair, whatever ties the geometry holds.
__global__ void k_ploc_nn(int m, const int* __restrict__ cid, const float4* __restrict__ blo, const float4* __restrict__ bhi,
                          int* __restrict__ nn) {
    const int i = blockIdx.x * THREADS + threadIdx.x;
    if (i >= m) return;
    const float4 lo = blo[cid[i]], hi = bhi[cid[i]];
    float best = __builtin_inff();
    int arg = -1;
    for (int off = 1; off <= PLOC_RADIUS; off++) {
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            const int j = i + sgn * off;
            if (j < 0 || j >= m) continue;
            const float4 a = blo[cid[j]], b = bhi[cid[j]];
            const float area = half_area(make_float4(fminf(lo.x, a.x), fminf(lo.y, a.y), fminf(lo.z, a.z), 0.0f),
                                         make_float4(fmaxf(hi.x, b.x), fmaxf(hi.y, b.y), fmaxf(hi.z, b.z), 0.0f));
            // candidates come in the order i-1, i+1, i-2, i+2, ...: among equal areas the pair (min, max) that is
            // lexicographically smallest wins
            bool better = area < best;
            if (area == best && arg >= 0) {
                const int a0 = min(i, j), a1 = max(i, j), b0 = min(i, arg), b1 = max(i, arg);
                better = a0 < b0 || (a0 == b0 && a1 < b1);
            }
            if (better) { best = area; arg = j; }
        }
    }
    nn[i] = arg;
}

// flags for the scan: low word 1 = this position survives (it is not the second partner of a merge),
// high word 1 = this position is the first partner of a merge (a new node is made here)
__global__ void k_ploc_flags(int m, const int* __restrict__ nn, unsigned long long* __restrict__ flags) {
    const int i = blockIdx.x * THREADS + threadIdx.x;
    if (i >= m) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    const unsigned long long keep = (mutual && j < i) ? 0ull : 1ull, first = (mutual && i < j) ? 1ull : 0ull;
    flags[i] = keep | (first << 32);
}

__global__ void k_ploc_apply(int m, int next_node, const int* __restrict__ cid, const int* __restrict__ nn,
                             const unsigned long long* __restrict__ flags, const unsigned long long* __restrict__ offsets,
                             int* __restrict__ cid_out, float4* __restrict__ blo, float4* __restrict__ bhi,
                             int2* __restrict__ children, unsigned long long* __restrict__ totals) {
    const int i = blockIdx.x * THREADS + threadIdx.x;
    if (i >= m) return;
    const unsigned long long f = flags[i], o = offsets[i];
    if (i == m - 1) *totals = o + f;                       // survivors in the low word, merges in the high word
    if (!(f & 1ull)) return;                               // merged into its partner's new node
    int id = cid[i];
    if (f >> 32) {
        const int a = id, b = cid[nn[i]];
        id = next_node + (int)(o >> 32);
        const float4 la = blo[a], ha = bhi[a], lb = blo[b], hb = bhi[b];
        blo[id] = make_float4(fminf(la.x, lb.x), fminf(la.y, lb.y), fminf(la.z, lb.z), 0.0f);
        bhi[id] = make_float4(fmaxf(ha.x, hb.x), fmaxf(ha.y, hb.y), fmaxf(ha.z, hb.z), 0.0f);
        children[id] = make_int2(a, b);
    }
    cid_out[(int)(o & 0xFFFFFFFFull)] = id;
}

__global__ void k_ploc_init(int n, const unsigned* __restrict__ ids, const float4* __restrict__ lo, const float4* __restrict__ hi,
                            int* __restrict__ cid, float4* __restrict__ blo, float4* __restrict__ bhi) {
    const int i = blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    cid[i] = i;                                            // node ids 0..n-1: the triangles in sorted order
    blo[i] = lo[ids[i]];
    bhi[i] = hi[ids[i]];
}

struct Dev {                 // frees what it allocated, whatever the exit path
    std::vector<void*> p;
    template <class T> hipError_t alloc(T** out, size_t count) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) { p.push_back(q); *out = static_cast<T*>(q); }
        return e;
    }
    ~Dev() { for (void* q : p) (void)hipFree(q); }
};

}  // namespace lbvh
}  // namespace cl2

// error text of the entry points that have no handle (defined in renderer_api.hip)
extern "C" void cl2_set_create_error(const char* msg);

extern "C" int cl2_build_bvh_gpu(int device_ordinal, const double* tri_min, const double* tri_max, int64_t n_triangles,
                                 int max_members, void* out_boxes, int64_t box_capacity, int64_t* n_boxes_out,
                                 int64_t* out_perm) {
    using namespace cl2;
    using namespace cl2::lbvh;
    auto bad = [](int code, const std::string& msg) { cl2_set_create_error(("cl2_build_bvh_gpu: " + msg).c_str()); return code; };
    if (!tri_min || !tri_max || !out_boxes || !n_boxes_out || !out_perm || n_triangles < 1 || max_members < 1 ||
        n_triangles > ((int64_t)1 << 27))
        return bad(CL2_E_INVALID, "bad argument");
    const int n = (int)n_triangles;
#define LB_TRY(expr)                                                                              \
    do {                                                                                          \
        const hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return bad(CL2_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_ordinal < 0 || device_ordinal >= n_dev)
        return bad(CL2_E_HIP, "no such HIP device");
    LB_TRY(hipSetDevice(device_ordinal));
    HostBox* boxes = static_cast<HostBox*>(out_boxes);

    const char* method_env = std::getenv("CLIVE2_GPU_BVH");
    bool use_ploc = !(method_env && std::string(method_env) == "lbvh") && n > 2;
    std::vector<Node> nodes((size_t)std::max(n - 1, 1));
    std::vector<float4> nlo, nhi, tlo((size_t)n), thi((size_t)n);
    std::vector<unsigned> ids((size_t)n);
    std::vector<int2> children;                            // PLOC: children of node n + k
    {
        Dev dev;
        double *d_min = nullptr, *d_max = nullptr;
        float4 *d_lo = nullptr, *d_hi = nullptr, *d_c = nullptr, *d_nlo = nullptr, *d_nhi = nullptr;
        unsigned *d_cb = nullptr, *d_ids = nullptr, *d_ids2 = nullptr, *d_arrived = nullptr;
        unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
        Node* d_nodes = nullptr;
        int *d_pi = nullptr, *d_pl = nullptr;
        LB_TRY(dev.alloc(&d_min, 3 * (size_t)n)); LB_TRY(dev.alloc(&d_max, 3 * (size_t)n));
        LB_TRY(dev.alloc(&d_lo, (size_t)n)); LB_TRY(dev.alloc(&d_hi, (size_t)n)); LB_TRY(dev.alloc(&d_c, (size_t)n));
        LB_TRY(dev.alloc(&d_cb, (size_t)6));
        LB_TRY(dev.alloc(&d_keys, (size_t)n)); LB_TRY(dev.alloc(&d_keys2, (size_t)n));
        LB_TRY(dev.alloc(&d_ids, (size_t)n)); LB_TRY(dev.alloc(&d_ids2, (size_t)n));
        LB_TRY(hipMemcpy(d_min, tri_min, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        LB_TRY(hipMemcpy(d_max, tri_max, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        const unsigned cb0[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
        LB_TRY(hipMemcpy(d_cb, cb0, sizeof cb0, hipMemcpyHostToDevice));
        const int grid = (n + THREADS - 1) / THREADS;
        hipLaunchKernelGGL(k_prepare, dim3(grid), dim3(THREADS), 0, 0, n, d_min, d_max, d_lo, d_hi, d_c, d_cb);
        hipLaunchKernelGGL(k_morton, dim3(grid), dim3(THREADS), 0, 0, n, d_c, d_cb, d_keys, d_ids);
        LB_TRY(hipGetLastError());
        size_t tmp_bytes = 0;
        LB_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys2, d_ids, d_ids2, (size_t)n, 0, 63, 0));
        unsigned char* d_tmp = nullptr;
        LB_TRY(dev.alloc(&d_tmp, tmp_bytes));
        LB_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_ids, d_ids2, (size_t)n, 0, 63, 0));
        LB_TRY(hipMemcpy(ids.data(), d_ids2, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost));
        if (use_ploc) {
            // ---- PLOC rounds: nearest neighbours, flags, one scan (survivor offsets | new-node ranks), apply ----
            const size_t n_all = (size_t)2 * n - 1;
            int *d_cid_a = nullptr, *d_cid_b = nullptr, *d_nn = nullptr;
            unsigned long long *d_flags = nullptr, *d_off = nullptr, *d_tot = nullptr;
            int2* d_children = nullptr;
            LB_TRY(dev.alloc(&d_nlo, n_all)); LB_TRY(dev.alloc(&d_nhi, n_all));
            LB_TRY(dev.alloc(&d_cid_a, (size_t)n)); LB_TRY(dev.alloc(&d_cid_b, (size_t)n)); LB_TRY(dev.alloc(&d_nn, (size_t)n));
            LB_TRY(dev.alloc(&d_flags, (size_t)n)); LB_TRY(dev.alloc(&d_off, (size_t)n)); LB_TRY(dev.alloc(&d_tot, (size_t)1));
            LB_TRY(dev.alloc(&d_children, n_all));
            size_t scan_bytes = 0;
            LB_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_flags, d_off, 0ull, (size_t)n, rocprim::plus<unsigned long long>()));
            unsigned char* d_scan = nullptr;
            LB_TRY(dev.alloc(&d_scan, scan_bytes));
            hipLaunchKernelGGL(k_ploc_init, dim3(grid), dim3(THREADS), 0, 0, n, d_ids2, d_lo, d_hi, d_cid_a, d_nlo, d_nhi);
            int m = n, next = n, rounds = 0;
            // every round merges at least one pair (k_ploc_nn), a round of distinct geometry about 40 % of the clusters; a
            // scene of coincident triangles would take one round per merge: past the cap the radix tree is built instead
            const int max_rounds = 400;
            while (m > 1 && rounds < max_rounds) {
                const int g = (m + THREADS - 1) / THREADS;
                hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(THREADS), 0, 0, m, d_cid_a, d_nlo, d_nhi, d_nn);
                hipLaunchKernelGGL(k_ploc_flags, dim3(g), dim3(THREADS), 0, 0, m, d_nn, d_flags);
                size_t sb = scan_bytes;
                LB_TRY(rocprim::exclusive_scan(d_scan, sb, d_flags, d_off, 0ull, (size_t)m, rocprim::plus<unsigned long long>()));
                hipLaunchKernelGGL(k_ploc_apply, dim3(g), dim3(THREADS), 0, 0, m, next, d_cid_a, d_nn, d_flags, d_off, d_cid_b, d_nlo, d_nhi,
                                   d_children, d_tot);
                LB_TRY(hipGetLastError());
                unsigned long long tot = 0;
                LB_TRY(hipMemcpy(&tot, d_tot, sizeof tot, hipMemcpyDeviceToHost));
                const int keep = (int)(tot & 0xFFFFFFFFull), merged = (int)(tot >> 32);
                if (keep != m - merged) return bad(CL2_E_HIP, "internal error: a PLOC round lost clusters");
                // nothing merged: no cluster found a neighbour (every union area +inf or NaN -- huge or NaN coordinates).  Not an
                // error of the input: the radix tree below needs no areas
                if (merged < 1) break;
                next += merged; m = keep; rounds++;
                std::swap(d_cid_a, d_cid_b);
            }
            if (m == 1) {
                nlo.resize(n_all); nhi.resize(n_all); children.resize(n_all);
                LB_TRY(hipMemcpy(nlo.data(), d_nlo, n_all * sizeof(float4), hipMemcpyDeviceToHost));
                LB_TRY(hipMemcpy(nhi.data(), d_nhi, n_all * sizeof(float4), hipMemcpyDeviceToHost));
                LB_TRY(hipMemcpy(children.data(), d_children, n_all * sizeof(int2), hipMemcpyDeviceToHost));
            } else {
                use_ploc = false;
            }
        }
        if (!use_ploc) {
            nlo.resize(nodes.size()); nhi.resize(nodes.size());
            LB_TRY(dev.alloc(&d_nodes, nodes.size())); LB_TRY(dev.alloc(&d_nlo, nodes.size())); LB_TRY(dev.alloc(&d_nhi, nodes.size()));
            LB_TRY(dev.alloc(&d_pi, nodes.size())); LB_TRY(dev.alloc(&d_pl, (size_t)n)); LB_TRY(dev.alloc(&d_arrived, nodes.size()));
            LB_TRY(hipMemset(d_arrived, 0, nodes.size() * sizeof(unsigned)));
            if (n > 1) {
                hipLaunchKernelGGL(k_hierarchy, dim3((n - 1 + THREADS - 1) / THREADS), dim3(THREADS), 0, 0, n, d_keys2, d_nodes, d_pi, d_pl);
                hipLaunchKernelGGL(k_fit, dim3(grid), dim3(THREADS), 0, 0, n, d_ids2, d_lo, d_hi, d_nodes, d_pi, d_pl, d_nlo, d_nhi, d_arrived);
                LB_TRY(hipGetLastError());
                LB_TRY(hipMemcpy(nodes.data(), d_nodes, nodes.size() * sizeof(Node), hipMemcpyDeviceToHost));
                LB_TRY(hipMemcpy(nlo.data(), d_nlo, nodes.size() * sizeof(float4), hipMemcpyDeviceToHost));
                LB_TRY(hipMemcpy(nhi.data(), d_nhi, nodes.size() * sizeof(float4), hipMemcpyDeviceToHost));
            }
            LB_TRY(hipMemcpy(tlo.data(), d_lo, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
            LB_TRY(hipMemcpy(thi.data(), d_hi, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
        }
    }
#undef LB_TRY

    int64_t n_tri = 0;
    std::vector<int> queue;
    queue.reserve((size_t)2 * n / std::max(1, max_members / 2) + 16);
    if (use_ploc) {
        // ---- PLOC tree -> the reference's convention.  Node ids: 0..n-1 the sorted triangles, n..2n-2 the merges (children
        // have smaller ids than their parent); the root is the last one made.  Subtrees of at most max_members triangles
        // become leaves (their triangles in left-to-right order), numbering is np_flatten_bvh's breadth-first queue. ----
        const int root = 2 * n - 2;
        std::vector<int> size((size_t)2 * n - 1, 1);
        for (int i = n; i <= root; i++) size[i] = size[children[i].x] + size[children[i].y];
        queue.push_back(root);
        std::vector<int> stack;
        for (size_t head = 0; head < queue.size(); head++) {
            if ((int64_t)head >= box_capacity) return bad(CL2_E_INVALID, "box_capacity too small");
            const int id = queue[head];
            HostBox& b = boxes[head];
            std::memset(&b, 0, sizeof b);
            b.min[0] = nlo[id].x; b.min[1] = nlo[id].y; b.min[2] = nlo[id].z;
            b.max[0] = nhi[id].x; b.max[1] = nhi[id].y; b.max[2] = nhi[id].z;
            if (size[id] <= max_members) {
                b.left = (int32_t)n_tri; b.right = (int32_t)(n_tri + size[id]);
                stack.assign(1, id);
                while (!stack.empty()) {
                    const int y = stack.back(); stack.pop_back();
                    if (y < n) out_perm[n_tri++] = (int64_t)ids[y];
                    else { stack.push_back(children[y].y); stack.push_back(children[y].x); }
                }
            } else {
                // the traversal pops box left+1 first and leaves box `left` on its stack: smaller subtree first
                const int a = children[id].x, c = children[id].y;
                const bool a_is_smaller = size[a] < size[c];
                b.left = (int32_t)queue.size(); b.right = 0;
                queue.push_back(a_is_smaller ? c : a);
                queue.push_back(a_is_smaller ? a : c);
            }
        }
    } else {
    // ---- collapse + breadth-first numbering (np_flatten_bvh's queue, bvh.py:345-373): O(boxes) on the host ----
    // queue entries: child id in the radix tree (>= 0 inner, < 0 single-triangle leaf ~position)
    queue.push_back(n == 1 ? ~0 : 0);
    for (size_t head = 0; head < queue.size(); head++) {
        if ((int64_t)head >= box_capacity) return bad(CL2_E_INVALID, "box_capacity too small");
        const int id = queue[head];
        HostBox& b = boxes[head];
        std::memset(&b, 0, sizeof b);
        int first, last;
        float4 lo, hi;
        if (id < 0) { first = last = ~id; lo = tlo[ids[first]]; hi = thi[ids[first]]; }
        else { first = nodes[id].first; last = nodes[id].last; lo = nlo[id]; hi = nhi[id]; }
        b.min[0] = lo.x; b.min[1] = lo.y; b.min[2] = lo.z;
        b.max[0] = hi.x; b.max[1] = hi.y; b.max[2] = hi.z;
        const int count = last - first + 1;
        if (id < 0 || count <= max_members) {
            b.left = (int32_t)n_tri; b.right = (int32_t)(n_tri + count);
            for (int k = first; k <= last; k++) out_perm[n_tri++] = (int64_t)ids[k];
        } else {
            const Node& nd = nodes[id];
            auto size_of = [&](int c) { return c < 0 ? 1 : nodes[c].last - nodes[c].first + 1; };
            // the traversal pops box left+1 first and leaves box `left` on its stack: smaller subtree first
            const bool left_is_smaller = size_of(nd.left) < size_of(nd.right);
            b.left = (int32_t)queue.size(); b.right = 0;
            queue.push_back(left_is_smaller ? nd.right : nd.left);
            queue.push_back(left_is_smaller ? nd.left : nd.right);
        }
    }
    }
    if (n_tri != n) return bad(CL2_E_INVALID, "internal error: the leaves do not cover every triangle once");
    *n_boxes_out = (int64_t)queue.size();
    return CL2_OK;
}
