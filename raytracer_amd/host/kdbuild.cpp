// kdbuild.cpp -- the reference's KD tree (construction rules of KDTree.cpp:68-126) built
// straight into the flattened arrays.
//
// Produces exactly the tree that the reference's recursion with stable sorts and a pre-order
// flattening walk produce (same branches, same leaf contents in the same order, same
// numbering; tests check it against a restatement of that recursion), several times faster:
//   * the node's two std::stable_sort calls over Shape* with a comparator that chases
//     pointers become sorts of compact (key, key, position, id) records.  Stable sorts
//     compose: after "sort by min, then stable sort by max" the order is lexicographic in
//     (max, min, incoming position), and the lesser list is the prefix of the order
//     (min, incoming position).  Keys with a unique last component need no stable sort;
//   * only the shapes that are handed to a child are sorted: the lesser child takes the
//     shapes with min <= value, the greater child those with max >= value plus -- the
//     reference's scan stops one element late (KDTree.cpp:101-106) -- the last shape of
//     the max-order below the split value;
//   * no KDNode objects, no per-leaf vectors, no shape -> id hash map;
//   * the top levels build their lesser subtree on another thread; subtrees are built
//     with local numbering and spliced in afterwards, so the result does not depend on
//     the number of threads;
//   * near the root, where a node's two sorts are the longest serial stretches of the build,
//     each is itself cut into pieces sorted side by side and merged (sortWide).
#include <algorithm>
#include <future>
#include <thread>
#include <vector>

#include "scene.hpp"

namespace rth {
namespace {

struct BoxTable {
    std::vector<double> mn, mx, mid;        // 3 per shape
};

struct Rec {
    double k1, k2;
    uint32_t pos, id;
};
inline bool recLess(const Rec& a, const Rec& b) {
    if (a.k1 != b.k1) return a.k1 < b.k1;
    if (a.k2 != b.k2) return a.k2 < b.k2;
    return a.pos < b.pos;
}

// std::sort of [first, last) by recLess, `ways` pieces sorted side by side and merged pairwise (a
// strict total order -- the positions are unique -- so the result is std::sort's whatever the pieces)
void sortWide(std::vector<Rec>& rec, int ways) {
    const size_t n = rec.size();
    if (ways < 2 || n < (size_t(1) << 15)) { std::sort(rec.begin(), rec.end(), recLess); return; }
    std::vector<size_t> cut(ways + 1);
    for (int k = 0; k <= ways; ++k) cut[k] = n * (size_t)k / (size_t)ways;
    Rec* base = rec.data();
    std::vector<std::future<void>> jobs;
    for (int k = 1; k < ways; ++k)
        jobs.push_back(std::async(std::launch::async, [=]() { std::sort(base + cut[k], base + cut[k + 1], recLess); }));
    std::sort(base, base + cut[1], recLess);
    for (auto& j : jobs) j.get();
    for (int width = 1; width < ways; width *= 2) {
        jobs.clear();
        for (int k = 0; k + width < ways; k += 2 * width) {
            Rec *a = base + cut[k], *b = base + cut[k + width], *c = base + cut[std::min(k + 2 * width, ways)];
            jobs.push_back(std::async(std::launch::async, [=]() { std::inplace_merge(a, b, c, recLess); }));
        }
        for (auto& j : jobs) j.get();
    }
}

struct Builder {
    const BoxTable& t;
    int spawnLayers;                        // layers (from the root) that may hand a subtree to another thread
    int threads;

    // threads a list of a node of this layer may use for its sort: the node's two lists and the
    // subtrees beside it are at work at the same time
    int sortWays(int layer) const { return layer < 8 ? std::max(1, threads >> (layer + 1)) : 1; }

    // returns the node id local to `out`: >= 0 branch, < 0 leaf ~id
    int32_t build(FlatKD& out, std::vector<uint32_t>&& ids, AABB box, int axis, int rwc, int layer) const {
        if ((uint32_t)layer > out.stats.depth) out.stats.depth = (uint32_t)layer;
        const size_t n = ids.size();
        if (n < 7 || rwc > 3 || layer > 20) return leaf(out, ids);
        const double value = t.mid[3 * (size_t)ids[n / 2] + axis];
        const int nextAxis = (axis + 1) % 3;

        // lesser list: shapes with min <= value in the order (min, incoming position)
        std::vector<uint32_t> lesserIds;
        auto makeLesser = [&]() {
            std::vector<Rec> rec;
            rec.reserve(n);
            for (size_t i = 0; i < n; ++i) {
                double m = t.mn[3 * (size_t)ids[i] + axis];
                if (m <= value) rec.push_back({m, 0.0, (uint32_t)i, ids[i]});
            }
            sortWide(rec, sortWays(layer));
            lesserIds.resize(rec.size());
            for (size_t i = 0; i < rec.size(); ++i) lesserIds[i] = rec[i].id;
        };
        // greater list: order (max, min, incoming position); the shapes with max >= value and,
        // unless that is all of them, the last shape below the split value in that order
        std::vector<uint32_t> greaterIds;
        auto makeGreater = [&]() {
            std::vector<Rec> rec;
            rec.reserve(n);
            bool haveExtra = false;
            Rec extra{0.0, 0.0, 0u, 0u};
            for (size_t i = 0; i < n; ++i) {
                Rec r{t.mx[3 * (size_t)ids[i] + axis], t.mn[3 * (size_t)ids[i] + axis], (uint32_t)i, ids[i]};
                if (r.k1 >= value) rec.push_back(r);
                else if (!haveExtra || recLess(extra, r)) { extra = r; haveExtra = true; }
            }
            if (!rec.empty()) {
                sortWide(rec, sortWays(layer));
                greaterIds.reserve(rec.size() + 1);
                if (haveExtra) greaterIds.push_back(extra.id);
                for (const Rec& r : rec) greaterIds.push_back(r.id);
            }
        };
        // the two lists do not depend on each other: near the root, where a node's sorts are
        // the longest serial stretches of the build, they are made side by side
        if (layer < spawnLayers && n >= 4096) {
            auto fut = std::async(std::launch::async, makeLesser);
            makeGreater();
            fut.get();
        } else {
            makeLesser();
            makeGreater();
        }
        std::vector<uint32_t>().swap(ids);

        const uint32_t me = (uint32_t)out.value.size();
        for (int k = 0; k < 3; ++k) { out.bmin.push_back(box.min[k]); out.bmax.push_back(box.max[k]); }
        out.value.push_back(value);
        out.axis.push_back(axis);
        out.lesser.push_back(0);
        out.greater.push_back(0);

        AABB lesserBox = box, greaterBox = box;
        lesserBox.max[axis] = value;
        greaterBox.min[axis] = value;
        const int lesserRwc = lesserIds.size() == n ? rwc + 1 : 0;
        const int greaterRwc = greaterIds.size() == n ? rwc + 1 : 0;

        int32_t lesser, greater;
        if (layer < spawnLayers && n >= 4096 && !lesserIds.empty() && !greaterIds.empty()) {
            FlatKD left, right;
            auto fut = std::async(std::launch::async, [&]() {
                return build(left, std::move(lesserIds), lesserBox, nextAxis, lesserRwc, layer + 1);
            });
            int32_t rightRoot = build(right, std::move(greaterIds), greaterBox, nextAxis, greaterRwc, layer + 1);
            int32_t leftRoot = fut.get();
            lesser = splice(out, left, leftRoot);
            greater = splice(out, right, rightRoot);
        } else {
            // (an empty child is a KDLeaf({}) one layer down, KDTree.cpp:94,116)
            if (lesserIds.empty() || greaterIds.empty()) out.stats.depth = std::max(out.stats.depth, (uint32_t)layer + 1);
            lesser = lesserIds.empty() ? leaf(out, lesserIds)
                                       : build(out, std::move(lesserIds), lesserBox, nextAxis, lesserRwc, layer + 1);
            greater = greaterIds.empty() ? leaf(out, greaterIds)
                                         : build(out, std::move(greaterIds), greaterBox, nextAxis, greaterRwc, layer + 1);
        }
        out.lesser[me] = lesser;
        out.greater[me] = greater;
        return (int32_t)me;
    }

    static int32_t leaf(FlatKD& out, const std::vector<uint32_t>& ids) {
        uint32_t id = (uint32_t)out.leafFirst.size();
        out.leafFirst.push_back((uint32_t)out.leafRefs.size());
        out.leafCount.push_back((uint32_t)ids.size());
        out.leafRefs.insert(out.leafRefs.end(), ids.begin(), ids.end());
        if (ids.size() > out.stats.maxLeaf) out.stats.maxLeaf = (uint32_t)ids.size();
        return ~(int32_t)id;
    }

    // appends a separately built subtree, renumbering it; returns its root in `out`
    static int32_t splice(FlatKD& out, const FlatKD& sub, int32_t subRoot) {
        const int32_t branchOff = (int32_t)out.value.size();
        const int32_t leafOff = (int32_t)out.leafFirst.size();
        const uint32_t refOff = (uint32_t)out.leafRefs.size();
        auto moved = [&](int32_t node) { return node >= 0 ? node + branchOff : ~(~node + leafOff); };
        out.bmin.insert(out.bmin.end(), sub.bmin.begin(), sub.bmin.end());
        out.bmax.insert(out.bmax.end(), sub.bmax.begin(), sub.bmax.end());
        out.value.insert(out.value.end(), sub.value.begin(), sub.value.end());
        out.axis.insert(out.axis.end(), sub.axis.begin(), sub.axis.end());
        for (int32_t c : sub.lesser) out.lesser.push_back(moved(c));
        for (int32_t c : sub.greater) out.greater.push_back(moved(c));
        for (uint32_t f : sub.leafFirst) out.leafFirst.push_back(f + refOff);
        out.leafCount.insert(out.leafCount.end(), sub.leafCount.begin(), sub.leafCount.end());
        out.leafRefs.insert(out.leafRefs.end(), sub.leafRefs.begin(), sub.leafRefs.end());
        out.stats.maxLeaf = std::max(out.stats.maxLeaf, sub.stats.maxLeaf);
        out.stats.depth = std::max(out.stats.depth, sub.stats.depth);
        return moved(subRoot);
    }
};

}  // namespace

void buildFlatKD(const std::vector<Shape*>& shapes, const AABB& box, FlatKD& out, int threads, Axis axis,
                 int recursionWithoutChange, int layer) {
    out = FlatKD();
    BoxTable t;
    const size_t n = shapes.size();
    t.mn.resize(3 * n); t.mx.resize(3 * n); t.mid.resize(3 * n);
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; ++i) {
        const AABB& b = shapes[i]->boundingBox;
        for (int k = 0; k < 3; ++k) { t.mn[3 * i + k] = b.min[k]; t.mx[3 * i + k] = b.max[k]; t.mid[3 * i + k] = b.mid[k]; }
        ids[i] = (uint32_t)i;
    }
    if (threads <= 0) threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    int spawnLayers = 0;
    if (threads > 1) {
        while ((1 << spawnLayers) < threads) ++spawnLayers;
        spawnLayers += 2;                   // more tasks than threads: subtrees differ in size
    }
    Builder b{t, spawnLayers, threads};
    out.root = b.build(out, std::move(ids), box, axis, recursionWithoutChange, layer);
    out.stats.branches = (uint32_t)out.value.size();
    out.stats.leaves = (uint32_t)out.leafFirst.size();
    out.stats.leafRefs = (uint32_t)out.leafRefs.size();
}

}  // namespace rth
