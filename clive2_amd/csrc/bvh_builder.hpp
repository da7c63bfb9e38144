// bvh_builder.hpp -- native host BVH builder.
//
// Same algorithm, same decisions and same output convention as the reference's host builder
// (`src/bvh.py`): top-down full-sweep SAH object split over the three centroid orders
// (`object_split`, :132-191) with its cost `A_left(k+1 tris)*k + A_right*(n-1-k)` (left count short
// by one), first-minimum tie-breaking, axis order x,y,z with strict `<`; LIFO work list, right
// child pushed first, leaf when <= max_members triangles or when more than max_depth nodes are
// pending (`construct_BVH`, :288-313); breadth-first numbering with children at (left, left+1) and
// leaves holding [left,right) of the leaf-ordered triangle list, a leaf's triangles in the order of
// the sweep that created it (`np_flatten_bvh`, :329-389).
//
// What differs is cost: the reference re-sorts every node three times in numpy (O(n log^2 n) with
// Python-level node objects -- minutes for 1M triangles).  Here the three centroid orders are sorted
// ONCE and every split partitions them stably: O(n log n), float64 arithmetic as numpy uses for mesh
// scenes.  Equal centroids are ordered by triangle id (numpy's unstable quicksort orders them by an
// implementation detail), so on tie-free input the tree is identical to the reference's
// (tests/test_native_bvh.py) and on ties it is a valid tree built by the same rule.
#pragma once
#include <algorithm>
#include <cstdint>
#include <limits>
#include <numeric>
#include <vector>

namespace cl2 {

struct HostBox { float min[4], max[4]; int32_t left, right, pad[2]; };   // struct Box, 48 B

struct BvhBuildResult {
    std::vector<HostBox> boxes;
    std::vector<int64_t> perm;     // leaf-ordered triangle ids
    int64_t max_pending = 0;
};

namespace bvh_detail {
struct Node { int64_t begin, end, left, right; int order_axis; };
inline double area(const double* lo, const double* hi) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2 * (dx * dy + dy * dz + dz * dx);
}
inline void grow(double* lo, double* hi, const double* tmin, const double* tmax, int64_t t) {
    for (int c = 0; c < 3; c++) { lo[c] = std::min(lo[c], tmin[3 * t + c]); hi[c] = std::max(hi[c], tmax[3 * t + c]); }
}
}  // namespace bvh_detail

inline BvhBuildResult build_bvh_sah(const double* tmin, const double* tmax, int64_t n, int max_members, int max_depth) {
    using namespace bvh_detail;
    const double inf = std::numeric_limits<double>::infinity();
    // order[a] = triangle ids sorted by centroid along axis a (ties: id); order[3] = input order.
    // A node owns the same index range [begin,end) in all four arrays.
    std::vector<double> center(3 * (size_t)n);
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) center[3 * i + a] = (tmin[3 * i + a] + tmax[3 * i + a]) / 2;
    std::vector<int64_t> order[4], scratch((size_t)n);
    for (int a = 0; a < 4; a++) {
        order[a].resize((size_t)n);
        std::iota(order[a].begin(), order[a].end(), (int64_t)0);
        if (a < 3)
            std::stable_sort(order[a].begin(), order[a].end(),
                             [&](int64_t x, int64_t y) { return center[3 * x + a] < center[3 * y + a]; });
    }
    std::vector<char> side((size_t)n, 0);
    std::vector<double> area_r;
    std::vector<Node> nodes;
    nodes.push_back(Node{0, n, -1, -1, 3});
    std::vector<int64_t> pending{0};
    BvhBuildResult res;

    while (!pending.empty()) {
        const int64_t ni = pending.back();
        pending.pop_back();
        res.max_pending = std::max<int64_t>(res.max_pending, (int64_t)pending.size());
        const int64_t b = nodes[(size_t)ni].begin, e = nodes[(size_t)ni].end, cnt = e - b;
        if (cnt <= max_members || (int64_t)pending.size() > max_depth) continue;

        double best = inf;
        int best_axis = -1;
        int64_t best_k = 0;
        area_r.resize((size_t)cnt);
        for (int a = 0; a < 3; a++) {
            const int64_t* ord = order[a].data() + b;
            double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
            for (int64_t k = cnt - 1; k >= 1; k--) {           // area_r[k-1] = area of triangles k .. cnt-1
                grow(lo, hi, tmin, tmax, ord[k]);
                area_r[(size_t)k - 1] = area(lo, hi);
            }
            double llo[3] = {inf, inf, inf}, lhi[3] = {-inf, -inf, -inf};
            double axis_best = inf;
            int64_t axis_k = 0;
            for (int64_t k = 0; k + 1 < cnt; k++) {
                grow(llo, lhi, tmin, tmax, ord[k]);
                const double cost = area(llo, lhi) * (double)k + area_r[(size_t)k] * (double)((cnt - 1) - k);
                if (cost < axis_best) { axis_best = cost; axis_k = k; }       // np.argmin: first minimum
            }
            if (axis_best < best) { best = axis_best; best_axis = a; best_k = axis_k; }
        }
        if (best_axis < 0) continue;     // no finite cost (non-finite input): keep the node a leaf
        const int64_t n_left = best_k + 1;
        for (int64_t k = 0; k < cnt; k++) side[(size_t)order[best_axis][(size_t)(b + k)]] = k < n_left ? 0 : 1;
        for (int a = 0; a < 4; a++) {                          // stable partition of every order
            int64_t* ord = order[a].data() + b;
            int64_t l = 0, r = 0;
            for (int64_t k = 0; k < cnt; k++) {
                if (side[(size_t)ord[k]] == 0) ord[l++] = ord[k];
                else scratch[(size_t)r++] = ord[k];
            }
            std::copy(scratch.begin(), scratch.begin() + r, ord + l);
        }
        const int64_t li = (int64_t)nodes.size();
        nodes.push_back(Node{b, b + n_left, -1, -1, best_axis});
        nodes.push_back(Node{b + n_left, e, -1, -1, best_axis});
        nodes[(size_t)ni].left = li;
        nodes[(size_t)ni].right = li + 1;
        pending.push_back(li + 1);
        pending.push_back(li);
    }

    // breadth-first flatten
    res.boxes.resize(nodes.size());
    res.perm.reserve((size_t)n);
    std::vector<int64_t> bfs{0};
    bfs.reserve(nodes.size());
    for (size_t head = 0; head < bfs.size(); head++) {
        const Node& nd = nodes[(size_t)bfs[head]];
        HostBox& hb = res.boxes[head];
        double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        for (int64_t k = nd.begin; k < nd.end; k++) grow(lo, hi, tmin, tmax, order[3][(size_t)k]);
        for (int c = 0; c < 3; c++) { hb.min[c] = (float)lo[c]; hb.max[c] = (float)hi[c]; }
        hb.min[3] = hb.max[3] = 0.0f;
        hb.pad[0] = hb.pad[1] = 0;
        if (nd.left >= 0) {
            hb.left = (int32_t)bfs.size();                     // children land at the queue tail
            hb.right = 0;
            bfs.push_back(nd.left);
            bfs.push_back(nd.right);
        } else {
            hb.left = (int32_t)res.perm.size();
            for (int64_t k = nd.begin; k < nd.end; k++) res.perm.push_back(order[nd.order_axis][(size_t)k]);
            hb.right = (int32_t)res.perm.size();
        }
    }
    return res;
}

}  // namespace cl2
