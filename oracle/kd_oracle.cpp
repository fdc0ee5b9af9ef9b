// kd_oracle.cpp -- CPU restatement of the reference's KD-tree construction.  TEST INFRASTRUCTURE:
// only tests/ may load it; the product builds its trees in raytracer_amd/host/kdbuild.cpp.
//
// Follows Raytracer/KDTree.cpp:45-126 (sortShapesMax/Min, buildKDTree) on arrays of shape boxes
// instead of Shape objects, and flattens the result in the walk order of printKDTree
// (KDTree.cpp:137-155: lesser subtree first).  The one deliberate difference: the reference
// sorts with std::sort, whose order among equal keys is unspecified (libstdc++ and MSVC differ);
// std::stable_sort makes the tree a function of the input order, which is the tie-break the
// product's builder defines as well.
// Parity unpinned: the reference holds no fixture of a tree.
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {

struct Boxes { const double *mn, *mx, *mid; };      // 3 doubles per shape (Shape::boundingBox, Shape.hpp:13-32)

struct Node {
    bool leaf;
    std::vector<uint32_t> shapes;       // KDLeaf::shapes
    Node *lesser = nullptr, *greater = nullptr;
    double bmin[3], bmax[3], value = 0.0;
    int axis = 0;
    ~Node() { delete lesser; delete greater; }
};

// KDTree.cpp:45-60
void sort_max(const Boxes& b, std::vector<uint32_t>& s, int axis)
{
    std::stable_sort(s.begin(), s.end(), [&](uint32_t x, uint32_t y) { return b.mx[3 * (size_t)y + axis] > b.mx[3 * (size_t)x + axis]; });
}
void sort_min(const Boxes& b, std::vector<uint32_t>& s, int axis)
{
    std::stable_sort(s.begin(), s.end(), [&](uint32_t x, uint32_t y) { return b.mn[3 * (size_t)y + axis] > b.mn[3 * (size_t)x + axis]; });
}

Node* make_leaf(std::vector<uint32_t> s)
{
    Node* n = new Node;
    n->leaf = true;
    n->shapes = std::move(s);
    return n;
}

// KDTree.cpp:68-126
Node* build(const Boxes& b, std::vector<uint32_t> s, const double* bmin, const double* bmax, int axis, int unchanged, int layer)
{
    if (s.size() < 7 || unchanged > 3 || layer > 20) return make_leaf(std::move(s));           // :69
    const double value = b.mid[3 * (size_t)s[s.size() / 2] + axis];                             // :72, before any sort
    const int next = (axis + 1) % 3;                                                            // :73
    Node* n = new Node;
    n->leaf = false;
    n->axis = axis;
    n->value = value;
    for (int k = 0; k < 3; ++k) { n->bmin[k] = bmin[k]; n->bmax[k] = bmax[k]; }

    size_t upto = 0;                                                                            // :77-84
    sort_min(b, s, axis);
    while (upto < s.size() && b.mn[3 * (size_t)s[upto] + axis] <= value) upto++;
    if (upto > 0) {                                                                             // :85-93
        std::vector<uint32_t> part(s.begin(), s.begin() + upto);
        double cmax[3] = {bmax[0], bmax[1], bmax[2]};
        cmax[axis] = value;
        const int u = part.size() == s.size() ? unchanged + 1 : 0;
        n->lesser = build(b, std::move(part), bmin, cmax, next, u, layer + 1);
    } else {
        n->lesser = make_leaf({});                                                              // :94-96
    }

    int from = (int)s.size() - 1;                                                               // :99-106
    sort_max(b, s, axis);
    while (from > 0 && b.mx[3 * (size_t)s[from] + axis] >= value) from--;
    if (from < (int)s.size() - 1) {                                                             // :107-115
        std::vector<uint32_t> part(s.begin() + from, s.end());
        double cmin[3] = {bmin[0], bmin[1], bmin[2]};
        cmin[axis] = value;
        const int u = part.size() == s.size() ? unchanged + 1 : 0;
        n->greater = build(b, std::move(part), cmin, bmax, next, u, layer + 1);
    } else {
        n->greater = make_leaf({});                                                             // :116-118
    }
    return n;
}

struct Flat {
    std::vector<double> bmin, bmax, value;
    std::vector<int32_t> axis, lesser, greater;
    std::vector<uint32_t> leaf_first, leaf_count, leaf_refs;
    int32_t root = -1;
    uint32_t depth = 0;
};

int32_t flatten(const Node* n, Flat& f, uint32_t depth)
{
    if (depth > f.depth) f.depth = depth;
    if (n->leaf) {
        const uint32_t id = (uint32_t)f.leaf_first.size();
        f.leaf_first.push_back((uint32_t)f.leaf_refs.size());
        f.leaf_count.push_back((uint32_t)n->shapes.size());
        f.leaf_refs.insert(f.leaf_refs.end(), n->shapes.begin(), n->shapes.end());
        return ~(int32_t)id;
    }
    const uint32_t id = (uint32_t)f.value.size();
    for (int k = 0; k < 3; ++k) { f.bmin.push_back(n->bmin[k]); f.bmax.push_back(n->bmax[k]); }
    f.value.push_back(n->value);
    f.axis.push_back(n->axis);
    f.lesser.push_back(0);
    f.greater.push_back(0);
    const int32_t l = flatten(n->lesser, f, depth + 1);
    const int32_t g = flatten(n->greater, f, depth + 1);
    f.lesser[id] = l;
    f.greater[id] = g;
    return (int32_t)id;
}

}  // namespace

extern "C" {

// buildKDTree(shapes 0..n-1 in that order, scene box) flattened; returns a handle
void* orc_kd_build(uint32_t n, const double* box_min, const double* box_max, const double* box_mid, const double* scene_min,
                   const double* scene_max)
{
    Boxes b{box_min, box_max, box_mid};
    std::vector<uint32_t> s(n);
    for (uint32_t i = 0; i < n; ++i) s[i] = i;
    Node* root = build(b, std::move(s), scene_min, scene_max, 0, 0, 0);
    Flat* f = new Flat;
    f->root = flatten(root, *f, 0);
    delete root;
    return f;
}

// counts[0..4] = branches, leaves, leaf refs, depth, root
void orc_kd_counts(const void* h, int64_t* counts)
{
    const Flat* f = static_cast<const Flat*>(h);
    counts[0] = (int64_t)f->value.size(); counts[1] = (int64_t)f->leaf_first.size(); counts[2] = (int64_t)f->leaf_refs.size();
    counts[3] = f->depth; counts[4] = f->root;
}

void orc_kd_copy(const void* h, double* bmin, double* bmax, double* value, int32_t* axis, int32_t* lesser, int32_t* greater,
                 uint32_t* leaf_first, uint32_t* leaf_count, uint32_t* leaf_refs)
{
    const Flat* f = static_cast<const Flat*>(h);
    std::copy(f->bmin.begin(), f->bmin.end(), bmin);
    std::copy(f->bmax.begin(), f->bmax.end(), bmax);
    std::copy(f->value.begin(), f->value.end(), value);
    std::copy(f->axis.begin(), f->axis.end(), axis);
    std::copy(f->lesser.begin(), f->lesser.end(), lesser);
    std::copy(f->greater.begin(), f->greater.end(), greater);
    std::copy(f->leaf_first.begin(), f->leaf_first.end(), leaf_first);
    std::copy(f->leaf_count.begin(), f->leaf_count.end(), leaf_count);
    std::copy(f->leaf_refs.begin(), f->leaf_refs.end(), leaf_refs);
}

void orc_kd_free(void* h) { delete static_cast<Flat*>(h); }

}  // extern "C"
