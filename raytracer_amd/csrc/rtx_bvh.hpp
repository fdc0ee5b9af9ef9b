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

    const std::vector<uint32_t>* ref_shape = nullptr;   // with split references: the shape each box belongs to

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

    int32_t make_leaf_in(BvhBuild& out, const std::vector<uint32_t>& order, uint32_t lo, uint32_t hi) const
    {
        // (two pieces of one shape in a leaf: the shape is listed once)
        uint32_t ids[kBvhLeafSize], n = 0;
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t shape = ref_shape ? (*ref_shape)[order[i]] : order[i];
            bool seen = false;
            for (uint32_t k = 0; k < n; ++k) seen = seen || ids[k] == shape;
            if (!seen) ids[n++] = shape;
        }
        uint32_t code = (uint32_t)out.prims.size() << 3 | (n - 1);
        for (uint32_t k = 0; k < n; ++k) out.prims.push_back(ids[k]);
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

// References of large triangles, split before the build ("early split clipping").  The boxes of a
// model's few long or large triangles -- straps, panels, anything running diagonally -- overlap the
// boxes of thousands of small ones, and every ray through that space is sent to test them.  A
// triangle whose box has more than `factor` times the median box area of the scene is cut at the
// middle of the longest axis of its box, each part is given the box of the clipped polygon, and so on
// until the parts' boxes are small enough (or kMaxCuts deep).  Every part becomes a reference of its
// own in the BVH, all naming the same shape; a leaf lists a shape once.  The parts' boxes are padded
// against the rounding of the cuts and kept inside the triangle's own box.
//   ref_boxes / ref_shape      all references: whole shapes and parts
//   part_boxes / part_shape    the parts alone (what the walk of the reversed ray needs, see kBvhEmpty's note)
// Default `factor`: 0, nothing is split.  On the reference's backpack model (67 908 triangles, 1 816 of
// them with more than 16 medians of box area, half of the summed box area) the pieces halve the
// triangle tests of a bounce ray (18 -> 9) -- but the tests that go away are mostly those of triangles
// whose box surrounds the ray's origin, which is exactly where the reference accepts hits BEHIND the
// origin, so the walk with the reversed ray (kBvhEmpty's note) has to find them again: +15 node steps
// per bounce ray, and the 1080p 16-spp pathTrace frame goes from 16.7 ms to 18.3 / 20.7 / 26.6 ms at
// factor 64 / 16 / 4 (13.7 ms at factor 4 if hits behind the origin were simply dropped, which the
// parity tests refuse).  RTX_BVH_SPLIT_AREA=<factor> turns it on for a scene.
constexpr double kBvhSplitArea = 0.0;
constexpr int kMaxCuts = 10;
constexpr uint32_t kSplitMinShapes = 256;       // smaller scenes: nothing to gain
inline void split_large_triangles(const std::vector<PrimBox>& boxes, uint32_t n_shapes, const uint32_t* info, const double* tri_pos,
                                  double factor, std::vector<PrimBox>& ref_boxes, std::vector<uint32_t>& ref_shape,
                                  std::vector<PrimBox>& part_boxes, std::vector<uint32_t>& part_shape)
{
    auto area = [](const PrimBox& b) {
        const double x = b.bmax[0] - b.bmin[0], y = b.bmax[1] - b.bmin[1], z = b.bmax[2] - b.bmin[2];
        return 2.0 * (x * y + y * z + z * x);
    };
    ref_boxes.clear(); ref_shape.clear(); part_boxes.clear(); part_shape.clear();
    double limit = std::numeric_limits<double>::infinity();
    std::vector<double> areas(n_shapes);
    for (uint32_t i = 0; i < n_shapes; ++i) areas[i] = area(boxes[i]);
    if (factor > 0.0 && n_shapes >= kSplitMinShapes) {
        std::vector<double> sorted(areas);
        std::nth_element(sorted.begin(), sorted.begin() + n_shapes / 2, sorted.end());
        const double median = sorted[n_shapes / 2];
        if (median > 0.0 && std::isfinite(median)) limit = factor * median;
    }
    struct Poly { int n; double v[3 + kMaxCuts + 1][3]; };     // a cut adds at most one corner
    auto box_of = [](const Poly& p) {
        PrimBox b;
        for (int k = 0; k < 3; ++k) { b.bmin[k] = std::numeric_limits<double>::max(); b.bmax[k] = -b.bmin[k]; }
        for (int i = 0; i < p.n; ++i)
            for (int k = 0; k < 3; ++k) { b.bmin[k] = std::min(b.bmin[k], p.v[i][k]); b.bmax[k] = std::max(b.bmax[k], p.v[i][k]); }
        return b;
    };
    // the part of the convex polygon on one side of the plane (Sutherland-Hodgman)
    auto clip = [](const Poly& p, int axis, double value, bool keep_below) {
        Poly out; out.n = 0;
        const int cap = (int)(sizeof out.v / sizeof out.v[0]);
        for (int i = 0; i < p.n; ++i) {
            const double* a = p.v[i];
            const double* b = p.v[(i + 1) % p.n];
            const bool ia = keep_below ? a[axis] <= value : a[axis] >= value;
            const bool ib = keep_below ? b[axis] <= value : b[axis] >= value;
            if (ia && out.n < cap) { for (int k = 0; k < 3; ++k) out.v[out.n][k] = a[k]; out.n++; }
            if (ia != ib && out.n < cap) {
                const double t = (value - a[axis]) / (b[axis] - a[axis]);
                for (int k = 0; k < 3; ++k) out.v[out.n][k] = a[k] + t * (b[k] - a[k]);
                out.v[out.n][axis] = value;
                out.n++;
            }
        }
        return out;
    };
    std::vector<std::pair<Poly, int>> work;
    std::vector<PrimBox> parts;
    for (uint32_t i = 0; i < n_shapes; ++i) {
        bool whole = (info[i] & 3u) != RTX_PRIM_TRIANGLE || !(areas[i] > limit) || !std::isfinite(areas[i]);
        parts.clear();
        if (!whole) {
            Poly tri; tri.n = 3;
            const double* p = tri_pos + 9 * (size_t)(info[i] >> 2);
            for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) tri.v[v][k] = p[3 * v + k];
            work.clear();
            work.push_back({tri, 0});
            while (!work.empty()) {
                auto [poly, depth] = work.back();
                work.pop_back();
                if (poly.n < 3) continue;           // (a sliver the cut left nothing of: its neighbour's padded box has it)
                PrimBox b = box_of(poly);
                if (depth >= kMaxCuts || !(area(b) > limit)) {
                    for (int k = 0; k < 3; ++k) {
                        const double pad = 1e-9 * (std::fabs(b.bmin[k]) + std::fabs(b.bmax[k])) + 1e-12 * (boxes[i].bmax[k] - boxes[i].bmin[k]);
                        b.bmin[k] = std::max(boxes[i].bmin[k], b.bmin[k] - pad);
                        b.bmax[k] = std::min(boxes[i].bmax[k], b.bmax[k] + pad);
                    }
                    parts.push_back(b);
                    continue;
                }
                int axis = 0;
                for (int k = 1; k < 3; ++k) if (b.bmax[k] - b.bmin[k] > b.bmax[axis] - b.bmin[axis]) axis = k;
                const double mid = 0.5 * (b.bmin[axis] + b.bmax[axis]);
                work.push_back({clip(poly, axis, mid, true), depth + 1});
                work.push_back({clip(poly, axis, mid, false), depth + 1});
            }
            whole = parts.size() < 2;               // (nothing came of it: the shape stays whole)
        }
        if (whole) { ref_boxes.push_back(boxes[i]); ref_shape.push_back(i); continue; }
        for (const PrimBox& b : parts) {
            ref_boxes.push_back(b); ref_shape.push_back(i);
            part_boxes.push_back(b); part_shape.push_back(i);
        }
    }
}

// boxes[0 .. n_refs): one box per reference; ref_shape (optional): the shape of each reference, when
// they are not simply the shapes 0 .. n_refs-1
inline void build_bvh(const std::vector<PrimBox>& boxes, uint32_t n_refs, BvhBuild& out, const std::vector<uint32_t>* ref_shape = nullptr)
{
    out = BvhBuild();
    if (n_refs == 0 || n_refs > kBvhMaxShapes) { out.root = kBvhEmpty; return; }
    bvh_detail::Builder b(boxes, n_refs);
    b.ref_shape = ref_shape;
    if (const char* e = std::getenv("RTX_BVH_WIDEST_AXIS")) b.all_axes = e[0] != '1';
    unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    while ((1u << b.spawn_depth) < threads) ++b.spawn_depth;
    if (threads > 1) b.spawn_depth += 1;
    out.nodes.reserve(n_refs / 2 + 1);
    out.root = b.build(out, 0, n_refs, 0);
}

// half-precision bit patterns that do not lie inside (x, x): rounding outwards (|x| <= 65504)
inline uint16_t half_bits(float x, bool up)
{
    _Float16 h = (_Float16)x;                       // nearest
    uint16_t b;
    std::memcpy(&b, &h, 2);
    const float back = (float)h;
    if (up ? back < x : back > x) {
        // one step towards +inf (up) or -inf: on the bit pattern, away from zero for the matching sign
        const bool negative = (b & 0x8000u) != 0;
        if ((b & 0x7fffu) == 0) b = up ? 0x0001u : 0x8001u;             // from +-0 to the smallest subnormal
        else if (negative == !up) b = (uint16_t)(b + 1);                  // magnitude grows
        else b = (uint16_t)(b - 1);                                       // magnitude shrinks
    }
    return b;
}
inline float half_value(uint16_t b) { _Float16 h; std::memcpy(&h, &b, 2); return (float)h; }
// The nodes with their planes in half precision, rounded outwards -- if that serves the scene: half
// precision has 11 bits, so boxes far from the origin of coordinates grow by 2^-10 of their distance
// from it, which is nothing to a model that sits around the origin and ruinous to small shapes at
// coordinate 1000.  The measure is the summed surface area of all child boxes (what the expected
// number of node visits goes with); false, and `out` empty, when it grows by more than `tolerance` or
// a coordinate is beyond half precision's range.
inline bool nodes_to_half(const std::vector<BvhNode>& in, std::vector<BvhNode16>& out, double tolerance = 0.03)
{
    out.clear();
    if (in.empty()) return false;
    out.resize(in.size());
    auto area = [](const float* mn, const float* mx) {
        const double x = (double)mx[0] - mn[0], y = (double)mx[1] - mn[1], z = (double)mx[2] - mn[2];
        return x * y + y * z + z * x;
    };
    double exact = 0.0, loose = 0.0;
    for (size_t i = 0; i < in.size(); ++i) {
        const BvhNode& n = in[i];
        BvhNode16& o = out[i];
        float mn[2][3], mx[2][3];
        for (int k = 0; k < 3; ++k) {
            const float v[4] = {n.lmin[k], n.lmax[k], n.rmin[k], n.rmax[k]};
            for (float x : v) if (!(std::fabs(x) <= 65000.0f)) { out.clear(); return false; }
            o.planes[k] = half_bits(n.lmin[k], false); o.planes[3 + k] = half_bits(n.lmax[k], true);
            o.planes[6 + k] = half_bits(n.rmin[k], false); o.planes[9 + k] = half_bits(n.rmax[k], true);
            mn[0][k] = half_value(o.planes[k]); mx[0][k] = half_value(o.planes[3 + k]);
            mn[1][k] = half_value(o.planes[6 + k]); mx[1][k] = half_value(o.planes[9 + k]);
        }
        o.left = n.left; o.right = n.right;
        exact += area(n.lmin, n.lmax) + area(n.rmin, n.rmax);
        loose += area(mn[0], mx[0]) + area(mn[1], mx[1]);
    }
    if (!(loose <= exact * (1.0 + tolerance))) { out.clear(); return false; }
    return true;
}

// appends the tree `sub` to `out` (nodes and leaf runs renumbered) and returns its root there
inline int32_t append_bvh(BvhBuild& out, const BvhBuild& sub)
{
    if (sub.root == kBvhEmpty) return kBvhEmpty;
    return bvh_detail::Builder::splice(out, sub, sub.root);
}

}  // namespace rtx
