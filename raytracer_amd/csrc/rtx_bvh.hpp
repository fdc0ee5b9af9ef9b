// rtx_bvh.hpp -- host-side construction of the candidate-finding BVH and of the
// inverse leaf index used to verify candidates against the reference's KD walk.
//
// The reference's KD tree (KDTree.cpp:68-126) duplicates every shape into dozens of
// leaves and mixes far-apart shapes in one leaf, so it is a poor structure for
// *finding* the shapes a ray can hit.  The library therefore also builds a
// conventional bounding-volume hierarchy over the shape boxes (binned surface-area
// heuristic, at most 4 shapes per leaf, depth-capped so the traversal stack is bounded).
// The BVH never decides a result on its own: a shape it finds is accepted only after
// the device code has shown that traverseKDTree would have visited a leaf holding it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <future>
#include <thread>
#include <limits>
#include <numeric>
#include <vector>

#include "rtx_device.hpp"

namespace rtx {

constexpr int kBvhMaxDepth = 22;        // traversal stack is 24 entries
constexpr uint32_t kBvhLeafSize = 4;       // at most 8: a leaf code keeps count - 1 in three bits

struct BvhBuild {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> prims;        // shape ids in leaf order
    int32_t root = 0;                   // >= 0 node, < 0 leaf code; kBvhEmpty when there are no shapes
    int depth = 0;
};

namespace bvh_detail {

struct Box {
    double mn[3], mx[3];
    void reset() { for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<double>::max(); mx[k] = -mn[k]; } }
    void grow(const PrimBox& b) { for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], b.bmin[k]); mx[k] = std::max(mx[k], b.bmax[k]); } }
    void grow(const Box& b) { for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], b.mn[k]); mx[k] = std::max(mx[k], b.mx[k]); } }
    void grow_point(const double* p) { for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); } }
    double area() const {
        double d0 = mx[0] - mn[0], d1 = mx[1] - mn[1], d2 = mx[2] - mn[2];
        if (d0 < 0 || d1 < 0 || d2 < 0) return 0.0;
        return 2.0 * (d0 * d1 + d1 * d2 + d2 * d0);
    }
};

// nearest floats that do not lie inside (x, x): rounding outwards
inline float float_below(double x) { float f = (float)x; return (double)f > x ? std::nextafterf(f, -std::numeric_limits<float>::infinity()) : f; }
inline float float_above(double x) { float f = (float)x; return (double)f < x ? std::nextafterf(f, std::numeric_limits<float>::infinity()) : f; }

struct Builder {
    const std::vector<PrimBox>& boxes;
    std::vector<double> centroid;       // 3 per prim
    std::vector<uint32_t> order;
    int spawn_depth = 0;                // levels (from the root) that may hand a subtree to another thread
    bool all_axes = true;               // split on the best of the three axes, not on the widest one only

    Builder(const std::vector<PrimBox>& b, uint32_t n) : boxes(b), centroid(3 * (size_t)n), order(n)
    {
        for (uint32_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) centroid[3 * (size_t)i + k] = 0.5 * (b[i].bmin[k] + b[i].bmax[k]);
        std::iota(order.begin(), order.end(), 0u);
    }

    Box bounds(uint32_t lo, uint32_t hi) const
    {
        Box b; b.reset();
        for (uint32_t i = lo; i < hi; ++i) b.grow(boxes[order[i]]);
        return b;
    }

    static int32_t make_leaf_in(BvhBuild& out, const std::vector<uint32_t>& order, uint32_t lo, uint32_t hi)
    {
        uint32_t code = (uint32_t)out.prims.size() << 3 | (hi - lo - 1);
        for (uint32_t i = lo; i < hi; ++i) out.prims.push_back(order[i]);
        return ~(int32_t)code;
    }

    // returns the split position in [lo+1, hi-1]
    uint32_t split(uint32_t lo, uint32_t hi, int depth)
    {
        const uint32_t n = hi - lo;
        // levels left must be able to hold n shapes in leaves of kBvhLeafSize by halving
        int need = 0;
        for (uint64_t cap = kBvhLeafSize; cap < n; cap *= 2) ++need;
        bool force_median = need >= kBvhMaxDepth - depth;
        Box cb; cb.reset();
        for (uint32_t i = lo; i < hi; ++i) cb.grow_point(&centroid[3 * (size_t)order[i]]);
        int axis = 0;
        double ext = -1.0;
        for (int k = 0; k < 3; ++k) if (cb.mx[k] - cb.mn[k] > ext) { ext = cb.mx[k] - cb.mn[k]; axis = k; }
        auto median = [&]() {
            uint32_t mid = lo + n / 2;
            std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](uint32_t a, uint32_t b) {
                double ca = centroid[3 * (size_t)a + axis], cb2 = centroid[3 * (size_t)b + axis];
                return ca < cb2 || (ca == cb2 && a < b);
            });
            return mid;
        };
        if (force_median || !(ext > 0.0)) return median();
        // binned SAH, the best split of the three axes (RTX_BVH_WIDEST_AXIS=1: of the widest centroid axis
        // only, a third of the work; on the reference's backpack model the traversal is then 4-8 % slower)
        constexpr int kBins = 16;
        double best = std::numeric_limits<double>::max();
        int best_b = -1, best_axis = axis;
        double best_scale = 0.0;
        for (int ax = 0; ax < 3; ++ax) {
            if (!all_axes && ax != axis) continue;
            const double e = cb.mx[ax] - cb.mn[ax];
            if (!(e > 0.0)) continue;
            Box bin_box[kBins];
            uint32_t bin_count[kBins] = {0};
            for (int b = 0; b < kBins; ++b) bin_box[b].reset();
            const double scale = kBins / e;
            for (uint32_t i = lo; i < hi; ++i) {
                int b = std::min(std::max((int)((centroid[3 * (size_t)order[i] + ax] - cb.mn[ax]) * scale), 0), kBins - 1);
                bin_box[b].grow(boxes[order[i]]); bin_count[b]++;
            }
            double left_area[kBins], right_area[kBins];
            uint32_t left_n[kBins], right_n[kBins];
            Box acc; acc.reset(); uint32_t cnt = 0;
            for (int b = 0; b < kBins; ++b) { if (bin_count[b]) acc.grow(bin_box[b]); cnt += bin_count[b]; left_area[b] = acc.area(); left_n[b] = cnt; }
            acc.reset(); cnt = 0;
            for (int b = kBins - 1; b >= 0; --b) { if (bin_count[b]) acc.grow(bin_box[b]); cnt += bin_count[b]; right_area[b] = acc.area(); right_n[b] = cnt; }
            for (int b = 0; b + 1 < kBins; ++b) {
                if (left_n[b] == 0 || right_n[b + 1] == 0) continue;
                double cost = left_area[b] * left_n[b] + right_area[b + 1] * right_n[b + 1];
                if (cost < best) { best = cost; best_b = b; best_axis = ax; best_scale = scale; }
            }
        }
        axis = best_axis;
        auto bin_of = [&](uint32_t prim) {
            int b = (int)((centroid[3 * (size_t)prim + axis] - cb.mn[axis]) * best_scale);
            return std::min(std::max(b, 0), kBins - 1);
        };
        if (best_b < 0) return median();
        auto it = std::partition(order.begin() + lo, order.begin() + hi, [&](uint32_t p) { return bin_of(p) <= best_b; });
        uint32_t mid = (uint32_t)(it - order.begin());
        if (mid == lo || mid == hi) return median();
        // keep the tree balanced enough for the depth budget
        uint32_t small = std::min(mid - lo, hi - mid);
        if ((uint64_t)small * 64 < n && need + 6 >= kBvhMaxDepth - depth) return median();
        return mid;
    }

    // appends a separately built subtree (disjoint range of `order`), renumbering it
    static int32_t splice(BvhBuild& out, const BvhBuild& sub, int32_t sub_root)
    {
        const int32_t node_off = (int32_t)out.nodes.size();
        const uint32_t prim_off = (uint32_t)out.prims.size();
        auto moved = [&](int32_t c) {
            if (c >= 0) return c + node_off;
            uint32_t code = (uint32_t)~c;
            return ~(int32_t)((((code >> 3) + prim_off) << 3) | (code & 7u));
        };
        for (BvhNode n : sub.nodes) { n.left = moved(n.left); n.right = moved(n.right); out.nodes.push_back(n); }
        out.prims.insert(out.prims.end(), sub.prims.begin(), sub.prims.end());
        out.depth = std::max(out.depth, sub.depth);
        return moved(sub_root);
    }

    int32_t build(BvhBuild& out, uint32_t lo, uint32_t hi, int depth)
    {
        out.depth = std::max(out.depth, depth);
        if (hi - lo <= kBvhLeafSize) return make_leaf_in(out, order, lo, hi);
        uint32_t mid = split(lo, hi, depth);
        uint32_t id = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        Box l = bounds(lo, mid), r = bounds(mid, hi);
        int32_t left, right;
        if (depth < spawn_depth && hi - lo >= 8192) {
            // the two halves of `order` are disjoint: the left one is built on another thread
            BvhBuild lsub, rsub;
            auto fut = std::async(std::launch::async, [&]() { return build(lsub, lo, mid, depth + 1); });
            int32_t rroot = build(rsub, mid, hi, depth + 1);
            int32_t lroot = fut.get();
            left = splice(out, lsub, lroot);
            right = splice(out, rsub, rroot);
        } else {
            left = build(out, lo, mid, depth + 1);
            right = build(out, mid, hi, depth + 1);
        }
        BvhNode& n = out.nodes[id];
        std::memset(&n, 0, sizeof n);
        for (int k = 0; k < 3; ++k) {
            n.lmin[k] = float_below(l.mn[k]); n.lmax[k] = float_above(l.mx[k]);
            n.rmin[k] = float_below(r.mn[k]); n.rmax[k] = float_above(r.mx[k]);
        }
        n.left = left; n.right = right;
        return (int32_t)id;
    }
};

}  // namespace bvh_detail

inline void build_bvh(const std::vector<PrimBox>& boxes, uint32_t n_shapes, BvhBuild& out)
{
    out = BvhBuild();
    if (n_shapes == 0 || n_shapes > kBvhMaxShapes) { out.root = kBvhEmpty; return; }
    bvh_detail::Builder b(boxes, n_shapes);
    if (const char* e = std::getenv("RTX_BVH_WIDEST_AXIS")) b.all_axes = e[0] != '1';
    unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    while ((1u << b.spawn_depth) < threads) ++b.spawn_depth;
    if (threads > 1) b.spawn_depth += 1;
    out.nodes.reserve(n_shapes / 2 + 1);
    out.root = b.build(out, 0, n_shapes, 0);
}

}  // namespace rtx
