// rtx_device.hpp -- device-side arithmetic of the trace loop (gfx950, FP64).
//
// Everything here is compiled with -ffp-contract=off: each FP64 multiply and add
// rounds separately, as in the reference's C++ (which uses glm's componentwise
// operators; no fused multiply-add appears in its arithmetic).  min/max are the
// explicit selects glm defines, never fmin/fmax, because their NaN behaviour
// decides slab tests for axis-parallel rays (Raytracer/Shape.cpp:14-24).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtx.h"

#define RTX_DEV __device__ __forceinline__

namespace rtx {

struct d3 { double x, y, z; };

RTX_DEV d3 mk(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
RTX_DEV d3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
RTX_DEV d3 operator+(d3 a, d3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RTX_DEV d3 operator-(d3 a, d3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RTX_DEV d3 operator-(d3 a) { return mk(-a.x, -a.y, -a.z); }
RTX_DEV d3 operator*(d3 a, d3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RTX_DEV d3 operator*(d3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
RTX_DEV d3 operator/(d3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
RTX_DEV double dot(d3 a, d3 b) { double tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
RTX_DEV d3 cross(d3 x, d3 y) { return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
RTX_DEV double length3(d3 a) { return sqrt(dot(a, a)); }
RTX_DEV double distance3(d3 a, d3 b) { return length3(b - a); }
RTX_DEV d3 normalize3(d3 a) { return a * (1.0 / sqrt(dot(a, a))); }
RTX_DEV double gmin(double x, double y) { return (y < x) ? y : x; }
RTX_DEV double gmax(double x, double y) { return (x < y) ? y : x; }
RTX_DEV double pick(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kInf = __builtin_huge_val();
constexpr double kEps = 2.220446049250313e-16;          // glm::epsilon<double>()
constexpr double kFar = 10000000000000.0;               // bigNumberButNotInfinity, DirectionalLight.hpp:12
// cos and sin of the double nearest pi, as a correctly rounded libm returns them
// (the reference passes glm::pi<double>() to glm::rotate, Raytracer.cpp:284)
constexpr double kCosPi = -1.0;
constexpr double kSinPi = 1.2246467991473532e-16;

// ---- RNG: Philox-4x32-10 keyed (seed, frame), counter (pixel, sample, branch, draw/4) ----
RTX_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One Philox block yields four consecutive draws: the block of the last draw is kept.
struct Rng {
    uint64_t seed; uint32_t frame, pixel, sample, branch, counter;
    uint32_t cached_block = 0xffffffffu;
    uint32_t o[4];
};

// uniform in [0,1] inclusive (rand()/RAND_MAX, Raytracer.cpp:65-67)
RTX_DEV double rng_uniform(uint64_t seed, uint32_t frame, uint32_t pixel, uint32_t sample, uint32_t branch, uint32_t draw)
{
    uint32_t o[4];
    philox4x32_10(pixel, sample, branch, draw >> 2, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (frame * 0x9E3779B9u), o);
    uint32_t w = draw & 3u;
    uint32_t v = w == 0 ? o[0] : (w == 1 ? o[1] : (w == 2 ? o[2] : o[3]));
    return (double)v / 4294967295.0;
}
RTX_DEV double rng_next(Rng& r)
{
    const uint32_t draw = r.counter++;
    if ((draw >> 2) != r.cached_block) {
        r.cached_block = draw >> 2;
        philox4x32_10(r.pixel, r.sample, r.branch, draw >> 2, (uint32_t)r.seed,
                      (uint32_t)(r.seed >> 32) ^ (r.frame * 0x9E3779B9u), r.o);
    }
    const uint32_t w = draw & 3u;
    const uint32_t v = w == 0 ? r.o[0] : (w == 1 ? r.o[1] : (w == 2 ? r.o[2] : r.o[3]));
    return (double)v / 4294967295.0;
}

// ---- Ray (Ray.cpp:3-8) --------------------------------------------------------------
struct Ray { d3 o, d; };       // d normalised by make_ray; inverseDirection is -d

RTX_DEV Ray make_ray(d3 origin, d3 dir) { Ray r; r.d = normalize3(dir); r.o = origin; return r; }

// ---- slab tests (Shape.cpp:12-34 and :100-116) -----------------------------------------
// inv = 1.0 / ray.direction; the reference recomputes it per box from the same
// operands, so computing it once per ray gives the same bits.
RTX_DEV void slab(d3 bmin, d3 bmax, d3 o, d3 inv, double& tmin, double& tmax)
{
    d3 t0 = (bmin - o) * inv;
    d3 t1 = (bmax - o) * inv;
    d3 ts = mk(gmin(t0.x, t1.x), gmin(t0.y, t1.y), gmin(t0.z, t1.z));
    d3 tb = mk(gmax(t0.x, t1.x), gmax(t0.y, t1.y), gmax(t0.z, t1.z));
    tmin = gmax(-kInf, gmax(ts.x, gmax(ts.y, ts.z)));
    tmax = gmin(kInf, gmin(tb.x, gmin(tb.y, tb.z)));
}

// ---- device scene -------------------------------------------------------------------
struct Branch {                 // 80 bytes, 16-byte aligned
    double bmin[3];
    double bmax[3];
    double value;
    int32_t axis;
    int32_t lesser;
    int32_t greater;
    uint32_t leaf_mid;          // first leaf id of the greater subtree (leaves are numbered in walk order)
    uint32_t leaf_hi;           // one past the last leaf id of this subtree
    int32_t pad;
};

struct PrimBox { double bmin[3]; double bmax[3]; };   // 48 bytes

// split plane and children of a KD branch; lesser_axis = lesser * 4 + axis (lesser may be a
// negative leaf code: decode with an arithmetic shift)
struct KdSplit { double value; int32_t lesser_axis; int32_t greater; };
constexpr uint32_t kKdMaxNodes = 1u << 29;
// prim_lca[shape]: the deepest KD node that covers all leaves of the shape (a branch index or a
// leaf code < 0).  kLcaSure on a branch index: every leaf below it whose cell comes within
// kHoldMargin x scene_scale of the shape's box is known to hold the shape.
constexpr int32_t kLcaSure = 1 << 30;
constexpr double kHoldMargin = 1e-7;
// prim_cert[shape] (scenes created as static): kCertValid | nx | ny << 10 | nz << 20, where n_a counts the
// faces of KD cells perpendicular to axis a -- split planes and box faces of the nodes below the
// shape's deepest covering node, and the faces of that node's own cell -- that come within
// kHoldMargin x scene_scale of the shape's box.  Valid only for a shape whose box lies inside that node's
// cell and that every leaf below the node lists whose cell the box itself meets (cells too thin to
// hold a point by the device's margin aside).  See certify_by_stretch().
constexpr uint32_t kCertValid = 0x80000000u;
constexpr uint32_t kCertMaxFaces = 1023u;
// prim_lca of a shape that no leaf lists: the reference's walk never tests it, so no point of a ray
// can certify it (the value is the code of a leaf no tree has: leaf ids stay below kKdMaxNodes)
constexpr int32_t kLcaNone = INT32_MIN;

// Candidate-finding BVH over the shape boxes (built by the library, see rtx_bvh.hpp):
// a node carries both children's boxes; child >= 0 is a node, < 0 is a leaf whose code
// ~child = first << 3 | (count - 1) names a run of bvh_prims.
// The boxes are kept in single precision, rounded outwards: any box that contains the
// shape boxes below it serves, so the arithmetic on them stays FP64 and exact-result
// culling is unaffected, while a node shrinks to 64 bytes (four 16-byte loads per lane).
struct BvhNode {                // 64 bytes
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad[2];
};
// The same node with its twelve planes in half precision, rounded outwards: 32 bytes, two 16-byte
// loads per lane.  planes[0..5] = left child (min x y z, max x y z), planes[6..11] = right child.
struct BvhNode16 {
    uint16_t planes[12];
    int32_t left, right;
};
constexpr int32_t kBvhEmpty = INT32_MIN;
// Large triangles enter the BVH in pieces (rtx_bvh.hpp: split_large_triangles): several boxes, each
// around a part of the triangle, in place of the one box around all of it.  A hit in front of the
// ray's origin lies in a piece's box that the ray enters, so the walk proposes it as before.  But
// Triangle::rayHit also reports a front-facing triangle BEHIND the origin (no test of the
// parameter's sign), and the reference accepts it whenever the ray passes the triangle's own box,
// i.e. whenever the origin lies inside that box: with one box per triangle the walk arrives at such a
// triangle anyway (its box surrounds the origin), with pieces it need not.  The pieces of the split
// triangles therefore also form a second, small BVH (bvh_behind_root, in the same arrays) that is
// walked with the REVERSED ray after the first: a hit behind the origin lies in a piece's box that
// the reversed ray enters, within the same distance bound.  Shapes are tested with the ray itself in
// both walks, so a triangle met twice answers the same twice.
constexpr uint32_t kBvhMaxShapes = 1u << 24;     // 4 per leaf under the BVH depth cap of 22

// The shapes of the BVH leaves, in leaf order, with what the traversal needs inline so a
// leaf costs one dependent load instead of four (leaf -> shape id -> info -> positions).
struct BvhPrim {                // 80 bytes
    double p[9];                // triangle: its three positions; other shapes: unused
    uint32_t shape;
    uint32_t info;              // prim_info[shape]
};

struct DScene {
    int32_t root;
    int32_t bvh_root;
    int32_t bvh_behind_root;        // the BVH over the pieces of the triangles that were split (kBvhEmpty: none were)
    const BvhNode* bvh_nodes;
    const BvhNode16* bvh_nodes16;   // null: the scene's coordinates do not fit half precision well enough
    const BvhPrim* bvh_prims;
    // inverse leaf index: the KD leaves that reference shape p are
    // prim_leaf_ids[prim_leaf_first[p] .. prim_leaf_first[p+1]) in increasing order, and
    // prim_leaf_pos holds the shape's position in leaf_refs for each of them
    const uint32_t* prim_leaf_first;
    const uint32_t* prim_leaf_ids;
    const uint32_t* prim_leaf_pos;
    // shortcut for the common case of the verification: per leaf the intersection of all its
    // ancestors' boxes and split half-spaces, per shape the deepest KD node (or leaf, < 0)
    // whose subtree holds every leaf that references the shape
    const PrimBox* leaf_cell;
    const int32_t* prim_lca;
    const uint32_t* prim_cert;      // per shape: cell faces near its box, per axis (0: not known)
    const KdSplit* kd_split;        // per branch: what locating a point needs, in one 16-byte load
    uint32_t n_branches, n_leaves, n_leaf_refs, n_shapes, n_prims, n_lights, n_materials, n_textures;
    double scene_scale;             // largest |coordinate| of the scene box
    const Branch* branches;
    const PrimBox* branch_cbox;     // per branch: union of the shape boxes below it
    const PrimBox* leaf_cbox;       // per leaf: union of its shape boxes
    const uint2* leaves;            // (first, count)
    const uint32_t* leaf_refs;
    const PrimBox* prim_box;
    const uint32_t* prim_info;      // type | index << 2
    const uint32_t* prim_material;
    const double* tri_pos;          // 9 per triangle
    const double* tri_nrm;          // 9 per triangle
    const double* tri_uv;           // 6 per triangle
    const double* sph_center;       // 3 per sphere
    const double* sph_radius;
    const double* sph_r2;
    const double* sph_nmat;         // 16 per sphere
    const uint32_t* lens_spheres;   // 2 prim ids per lens
    const rtx_material* materials;
    const rtx_texture* textures;
    const uint8_t* texels;
    const rtx_light* lights;
};

struct Hit {
    d3 position, normal, bary;
    double u, v, depth;
    int32_t prim;
};

// ---- Triangle::rayHit (Triangle.cpp:8-72) ---------------------------------------------
// depth-only form used during traversal; returns false on a miss
RTX_DEV bool triangle_depth(d3 a, d3 b, d3 c, d3 o, d3 d, double& depth);
RTX_DEV bool triangle_depth(const double* __restrict__ p, d3 o, d3 d, double& depth)
{
    return triangle_depth(ld3(p), ld3(p + 3), ld3(p + 6), o, d, depth);
}
RTX_DEV bool triangle_depth(d3 a, d3 b, d3 c, d3 o, d3 d, double& depth)
{
    d3 E1 = b - a, E2 = c - a;
    d3 pvec = cross(d, E2);
    double det = dot(E1, pvec);
    if (det < 0.0) return false;
    double inv = 1.0 / det;
    d3 tvec = o - a;
    double u = dot(tvec, pvec) * inv;
    if (u < 0.0 || u > 1.0) return false;
    d3 qvec = cross(tvec, E1);
    double v = dot(d, qvec) * inv;
    if (v < 0.0 || u + v > 1.0) return false;
    double bx = 1.0 - (u + v);
    d3 pos = ((a * bx) + (b * u)) + (c * v);
    depth = distance3(pos, o);
    return true;
}

// The same test, also saying on which side of the origin the hit lies.  Triangle::rayHit computes
// no ray parameter (Triangle.cpp:8-72): a front-facing triangle behind the origin is a hit like any
// other, with depth = its distance.  The sign of Moeller-Trumbore's parameter, dot(E2, q) / det with
// det > 0, tells the two apart; it steers the KD certificate only (never a result), so it need not
// be exact near zero.
RTX_DEV bool triangle_depth_sided(d3 a, d3 b, d3 c, d3 o, d3 d, double& depth, bool& behind)
{
    d3 E1 = b - a, E2 = c - a;
    d3 pvec = cross(d, E2);
    double det = dot(E1, pvec);
    if (det < 0.0) return false;
    double inv = 1.0 / det;
    d3 tvec = o - a;
    double u = dot(tvec, pvec) * inv;
    if (u < 0.0 || u > 1.0) return false;
    d3 qvec = cross(tvec, E1);
    double v = dot(d, qvec) * inv;
    if (v < 0.0 || u + v > 1.0) return false;
    double bx = 1.0 - (u + v);
    d3 pos = ((a * bx) + (b * u)) + (c * v);
    depth = distance3(pos, o);
    behind = dot(E2, qvec) < 0.0;
    return true;
}

// full record of a triangle hit (the ray is known to hit)
RTX_DEV bool triangle_full(const DScene& s, uint32_t tri, d3 o, d3 d, Hit& h)
{
    const double* p = s.tri_pos + 9 * (size_t)tri;
    d3 a = ld3(p), b = ld3(p + 3), c = ld3(p + 6);
    d3 E1 = b - a, E2 = c - a;
    d3 pvec = cross(d, E2);
    double det = dot(E1, pvec);
    if (det < 0.0) return false;
    double inv = 1.0 / det;
    d3 tvec = o - a;
    double u = dot(tvec, pvec) * inv;
    if (u < 0.0 || u > 1.0) return false;
    d3 qvec = cross(tvec, E1);
    double v = dot(d, qvec) * inv;
    if (v < 0.0 || u + v > 1.0) return false;
    h.bary = mk(1.0 - (u + v), u, v);
    h.position = ((a * h.bary.x) + (b * h.bary.y)) + (c * h.bary.z);
    h.depth = distance3(h.position, o);
    const double* n = s.tri_nrm + 9 * (size_t)tri;
    h.normal = ((ld3(n) * h.bary.x) + (ld3(n + 3) * h.bary.y)) + (ld3(n + 6) * h.bary.z);
    // mat3x2 is a float matrix (Triangle.cpp:64): uv is evaluated in FP32
    const double* t = s.tri_uv + 6 * (size_t)tri;
    float bx = (float)h.bary.x, by = (float)h.bary.y, bz = (float)h.bary.z;
    float fu = (float)t[0] * bx + (float)t[2] * by + (float)t[4] * bz;
    float fv = (float)t[1] * bx + (float)t[3] * by + (float)t[5] * bz;
    h.u = (double)fu; h.v = (double)fv;
    return true;
}

// solveQuadratic (CoordinateHelpers.cpp:22-32); returns false when both roots are NaN
RTX_DEV bool solve_quadratic(double a, double b, double c, double& r1, double& r2)
{
    double disc = (b * b) - (4.0 * a * c);
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    r1 = (-b + sq) / (2.0 * a);
    r2 = (-b - sq) / (2.0 * a);
    return true;
}

// Sphere::rayHit up to the ray parameter (Sphere.cpp:47-70)
RTX_DEV bool sphere_t(d3 center, double r2, d3 o, d3 d, double& t)
{
    d3 L = o - center;
    double a = dot(d, d);
    double b = 2.0 * dot(d, L);
    double c = dot(L, L) - r2;
    double t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return false;
    if (isnan(t0) && isnan(t1)) return false;
    if (t0 > t1) { double x = t0; t0 = t1; t1 = x; }
    if (t0 < 0) { t0 = t1; if (t0 < 0) return false; }
    t = t0;
    return true;
}

// need_uv = false leaves uv at 0: for callers that do not read it (no sampler of the hit's
// material takes uv, or only position and normal are wanted), atan2 being the costliest part
RTX_DEV bool sphere_full(const DScene& s, uint32_t sph, d3 o, d3 d, Hit& h, bool need_uv = true)
{
    d3 c = ld3(s.sph_center + 3 * (size_t)sph);
    double t;
    if (!sphere_t(c, s.sph_r2[sph], o, d, t)) return false;
    h.position = o + (d * t);
    d3 n = normalize3(h.position - c);
    const double* m = s.sph_nmat + 16 * (size_t)sph;
    // normalMat * dvec4(n, 1.0): (m0*v0 + m1*v1) + (m2*v2 + m3*v3), truncated to xyz
    h.normal.x = (m[0] * n.x + m[4] * n.y) + (m[8] * n.z + m[12] * 1.0);
    h.normal.y = (m[1] * n.x + m[5] * n.y) + (m[9] * n.z + m[13] * 1.0);
    h.normal.z = (m[2] * n.x + m[6] * n.y) + (m[10] * n.z + m[14] * 1.0);
    h.u = need_uv ? atan2(h.normal.x, h.normal.z) / (2 * kPi) + 0.5 : 0.0;
    h.v = need_uv ? h.normal.y * 0.5 + 0.5 : 0.0;
    h.bary = mk(0.0, 0.0, 0.0);
    h.depth = t;
    return true;
}

// Biconvex::rayHit (Biconvex.cpp:55-85) up to (depth, winning inner sphere)
RTX_DEV bool biconvex_depth(const DScene& s, uint32_t lens, d3 o, d3 d, double& depth, uint32_t& prim)
{
    uint32_t p1 = s.lens_spheres[2 * lens], p2 = s.lens_spheres[2 * lens + 1];
    uint32_t i1 = s.prim_info[p1] >> 2, i2 = s.prim_info[p2] >> 2;
    d3 c1 = ld3(s.sph_center + 3 * (size_t)i1), c2 = ld3(s.sph_center + 3 * (size_t)i2);
    double t1, t2;
    bool h1 = sphere_t(c1, s.sph_r2[i1], o, d, t1);
    bool h2 = sphere_t(c2, s.sph_r2[i2], o, d, t2);
    // Sphere::pointInSphere (Sphere.cpp:20-27) of the other sphere
    h1 = h1 && (distance3(o + (d * t1), c2) <= s.sph_radius[i2]);
    h2 = h2 && (distance3(o + (d * t2), c1) <= s.sph_radius[i1]);
    double best = kInf;
    bool any = false;
    if (h1 && t1 < best) { best = t1; depth = t1; prim = p1; any = true; }
    if (h2 && t2 < best) { depth = t2; prim = p2; any = true; }
    // the reference returns s1Hit || s2Hit even when neither depth beat +inf; `hit`
    // is then left untouched and its uninitialised depth is compared -- treated as a miss
    return any;
}

// full HitResult of primitive `prim` (a triangle or a sphere id, as reported by traversal)
RTX_DEV bool prim_full(const DScene& s, uint32_t prim, d3 o, d3 d, Hit& h, bool need_uv = true)
{
    uint32_t info = s.prim_info[prim];
    h.prim = (int32_t)prim;
    if ((info & 3u) == RTX_PRIM_TRIANGLE) return triangle_full(s, info >> 2, o, d, h);
    return sphere_full(s, info >> 2, o, d, h, need_uv);
}

// whether any channel of the material is read through a sampler, i.e. whether the hit's uv matters
RTX_DEV bool material_reads_uv(const rtx_material& m)
{
    return (m.color_fn.kind | m.ns_fn.kind | m.ni_fn.kind | m.transparency_fn.kind | m.metalness_fn.kind | m.smoothness_fn.kind) != 0u;
}

// ---- materials (Material.cpp:111-161, Procedural.cpp:4-25, Texture.cpp:39-64) -------------
RTX_DEV bool checker_even(double u, double v, double squares)
{
    int fx = (int)floor(u * squares), fy = (int)floor(v * squares);
    return ((fx + fy) % 2) == 0;
}

RTX_DEV const uint8_t* texel(const DScene& s, uint32_t tex, double u, double v)
{
    rtx_texture t = s.textures[tex];
    u = u - 1.0 * floor(u / 1.0);
    v = v - 1.0 * floor(v / 1.0);
    uint32_t px = (uint32_t)floor(u * (double)t.width);
    uint32_t py = (uint32_t)floor(v * (double)t.height);
    uint64_t idx = (uint64_t)(px + py * t.width);
    uint64_t last = (uint64_t)t.width * t.height - 1;
    if (idx > last) idx = last;         // the reference reads out of bounds here
    return s.texels + t.offset + idx * 3u;
}

RTX_DEV double sample_scalar(const DScene& s, rtx_sampler f, double constant, double u, double v)
{
    if (f.kind == RTX_SAMPLER_DOUBLE_CHECKERS) return checker_even(u, v, 10.0) ? 1.0 : 0.0;
    if (f.kind == RTX_SAMPLER_TEXTURE_DOUBLE) {
        const uint8_t* p = texel(s, f.texture, u, v);
        return ((double)p[0] / 255.0 + (double)p[1] / 255.0 + (double)p[2] / 255.0) / 3.0;
    }
    return constant;
}

RTX_DEV d3 material_color(const DScene& s, const rtx_material& m, double u, double v)
{
    if (m.color_fn.kind == RTX_SAMPLER_RY_CHECKERS)
        return checker_even(u, v, 1000.0) ? mk(1.0, 0.0, 0.0) : mk(1.0, 1.0, 0.0);
    if (m.color_fn.kind == RTX_SAMPLER_TEXTURE_VEC3) {
        // the texture colour sampler is declared function<vec3(dvec2)> (Texture.hpp:29): float
        const uint8_t* p = texel(s, m.color_fn.texture, u, v);
        return mk((double)(float)((double)p[0] / 255.0), (double)(float)((double)p[1] / 255.0),
                  (double)(float)((double)p[2] / 255.0));
    }
    return ld3(m.color);
}

// ---- lights ---------------------------------------------------------------------------
RTX_DEV Ray light_ray(const rtx_light& l, d3 p, double u0, double u1)
{
    if (l.kind == RTX_LIGHT_POINT) return make_ray(p, normalize3(ld3(l.position) - p));
    if (l.kind == RTX_LIGHT_DIRECTIONAL) return make_ray(p, normalize3((ld3(l.direction) * kFar) - p));
    d3 a = (ld3(l.right) * u0) * l.width;
    d3 b = (ld3(l.up) * u1) * l.height;
    d3 target = (ld3(l.bottom_left) + a) + b;
    return make_ray(p, normalize3(target - p));
}

RTX_DEV double light_distance(const rtx_light& l, const Ray& r)
{
    if (l.kind == RTX_LIGHT_POINT) return distance3(ld3(l.position), r.o);
    if (l.kind == RTX_LIGHT_DIRECTIONAL) return kFar;
    d3 n = ld3(l.direction);
    double denom = dot(n, r.d);
    if (fabs(denom) < kEps) return kInf;             // the reference exit(-1)s, SquareLight.cpp:80-83
    d3 dd = ld3(l.position) - r.o;
    double t = dot(dd, n) / denom;
    if (t < kEps) return kInf;                       // SquareLight.cpp:87-90
    d3 hp = r.o + (r.d * t);
    return distance3(r.o, hp);
}

RTX_DEV double light_attenuation(const rtx_light& l, double d, const double* att)
{
    if (l.kind == RTX_LIGHT_DIRECTIONAL) return 1.0;
    return (att[0] + (att[1] * d) + (att[2] * (d * d)));
}

// lightingFunction (Raytracer.cpp:187-197).  `color` and `ns` are the material's colour and
// specular exponent at the hit (material_color, sample_scalar of ns_fn): the same for every light
// sample of a hit, so the caller samples them once.
// pow(x, ns) of lightingFunction (Raytracer.cpp:191): x in [0, 1], ns the material's specular exponent.
// The FP64 pow of the device library is some 230 instructions, per light sample of every lit hit; the
// presets' exponents are whole numbers (2, 50, 100: Material.cpp:9-18), and a whole exponent up to 1 024
// is a handful of multiplications (square and multiply).  The product is within 0.45 * ns units in the
// last place of the correctly rounded power (9e-15 relative at ns = 100, bounded by a known-answer test)
// -- a deviation in a colour, ten orders of magnitude inside the 1e-4 the images are held to, never in
// a decision.  Any other exponent (a sampled one) takes pow.
RTX_DEV double specular_power(double x, double ns)
{
    if (ns >= 0.0 && ns <= 1024.0 && ns == (double)(uint32_t)ns) {
        uint32_t e = (uint32_t)ns;
        double r = 1.0, b = x;
        while (e) {
            if (e & 1u) r = r * b;
            e >>= 1;
            if (e) b = b * b;
        }
        return r;
    }
    return pow(x, ns);
}

RTX_DEV d3 lighting(d3 ray_inv_dir, d3 ldir, d3 normal, double attenuation, d3 color, double ns, d3 lcolor)
{
    d3 refl = normalize3((normal * (dot(ldir, normal) * 2.0)) - ldir);
    double spec = specular_power(gmax(0.0, dot(refl, ray_inv_dir)), ns);
    d3 specular = (lcolor * spec) / attenuation;
    double diff = gmax(0.0, dot(normal, ldir));
    d3 diffuse = ((color * lcolor) * diff) / attenuation;
    return diffuse + specular;
}

// getRefractionRay (Raytracer.cpp:122-182)
RTX_DEV d3 refraction_ray(d3 n, d3 d, double ior, bool entering)
{
    double prev = 1.0, next = ior;
    if (!entering) { n = -n; double t = prev; prev = next; next = t; }
    double cosA1 = dot(d, n);
    double sinA1 = sqrt(1.0 - (cosA1 * cosA1));
    double ratio = prev / next;
    double sinA2 = sinA1 * ratio;
    d3 best = d;
    if (sinA2 <= -1.0 || sinA2 >= 1.0) return best;          // total internal reflection: unchanged
    double max_close = -kInf, k1, k2;
    if (solve_quadratic(1.0, 2.0 * cosA1, 1.0 - (1.0 / (ratio * ratio)), k1, k2)) {
        if (!isnan(k1)) {
            d3 r1 = normalize3(d + (n * k1));
            double c1 = dot(d, r1);
            if (c1 > max_close && c1 >= 0.0) { max_close = c1; best = r1; }
        }
        if (!isnan(k2)) {
            d3 r2 = normalize3(d + (n * k2));
            double c2 = dot(d, r2);
            if (c2 > max_close && c2 >= 0.0) { max_close = c2; best = r2; }
        }
    }
    if (max_close <= 0.0) return d;
    return normalize3(best);
}

// glm::rotate(v, angle, normal) of gtx/rotate_vector = mat3(rotate(mat4(1), angle, normal)) * v,
// with c = cos(angle), s = sin(angle) supplied by the caller
RTX_DEV d3 rotate_vec(d3 v, double c, double s, d3 normal)
{
    d3 axis = normalize3(normal);
    d3 temp = axis * (1.0 - c);
    double R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = temp.x * axis.y + s * axis.z;
    R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;
    R[2][1] = temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    double M[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double e0 = (k == 0 ? 1.0 : 0.0) * R[i][0];
            double e1 = (k == 1 ? 1.0 : 0.0) * R[i][1];
            double e2 = (k == 2 ? 1.0 : 0.0) * R[i][2];
            M[i][k] = (e0 + e1) + e2;
        }
    d3 o;
    o.x = M[0][0] * v.x + M[1][0] * v.y + M[2][0] * v.z;
    o.y = M[0][1] * v.x + M[1][1] * v.y + M[2][1] * v.z;
    o.z = M[0][2] * v.x + M[1][2] * v.y + M[2][2] * v.z;
    return o;
}

// -rotate(d, pi, n) (Raytracer.cpp:284)
RTX_DEV d3 mirror_direction(d3 d, d3 n) { return -rotate_vec(d, kCosPi, kSinPi, n); }

// randomHemisphericalVector of the Whitted variant (Raytracer-Noah-Desktop.cpp:149-166):
// two uniforms, theta first; not confined to the hemisphere; NaN for a normal parallel to
// worldUp (the rotation axis is then the zero vector) -- both as in the reference
RTX_DEV d3 random_hemispherical_desktop(d3 normal, Rng& rng)
{
    normal = normalize3(normal);
    d3 world_up = normal.y > 0.9 ? mk(0.0, 1.0, 0.0) : mk(1.0, 0.0, 0.0);
    double theta = (2.0 * kPi) * rng_next(rng);
    double phi = acos(1.0 - (2.0 * rng_next(rng)));
    d3 nv = mk(sin(theta) * sin(phi), sin(theta) * cos(phi), sin(phi));
    d3 rot_vector = cross(normal, world_up);
    double rot_rads = acos(dot(normal, world_up));
    return rotate_vec(nv, cos(rot_rads), sin(rot_rads), rot_vector);
}

// Xchi / Gp (CookTorrance.cpp:11-15,56-72)
RTX_DEV double xchi(double x) { int i = (x > 0); return (1.0 * (double)i) + (-1.0 * ((double)(1 - i))); }
RTX_DEV double gp(d3 v, d3 n, d3 h, double a)
{
    double vdh = dot(v, h);
    double chi = xchi(vdh / dot(v, n));
    vdh *= vdh;
    double right = (1.0 - vdh) / vdh;
    return (chi * 2.0) / (1.0 + sqrt(1.0 + (a * a * right)));
}

// CookTorance (CookTorrance.cpp:134-190)
RTX_DEV d3 cook_torrance(d3 vi, d3 vo, d3 vn, double smooth, d3 down, d3 F0, d3& kS)
{
    double rough = 1.0 - smooth;
    d3 vh = normalize3(vi + vo);
    double ndi = dot(vn, vi);
    double cosT = dot(vo, vn);
    double sinT = sqrt(1 - cosT * cosT);
    // pow(b, 5.0) (CookTorrance.cpp:41) as four multiplications: within 3 units in the last place
    // of the C library's pow (relative 5.6e-16 at most; tests/test_oracle_kat.py bounds it), the same order
    // as the device pow's own distance from it
    const double b1 = 1.0 - dot(vh, vi), b2 = b1 * b1;
    double p = (b2 * b2) * b1;
    d3 fres = F0 + (mk(1.0 - F0.x, 1.0 - F0.y, 1.0 - F0.z) * p);
    double geo = gp(vi, vn, vh, rough) * gp(vo, vn, vh, rough);
    double denom = 4.0 * (ndi * dot(vh, vn) + 0.05);
    kS = fres;
    return (((down * geo) * fres) * sinT) / denom;
}

// entering = acos(dot(normal, inverseDirection)) < pi / 2.0 (Raytracer.cpp:248-249).  The quotient
// pi / 2.0 is the double P just BELOW the real pi/2 (by 6.1e-17), and acos(x) = pi/2 - x - ...
// rounds to P or above for every x up to half a unit in the last place beyond that gap: the
// comparison is true exactly for x >= kEnteringMin = 1.7225464241988333e-16 and x <= 1 (beyond 1,
// as for NaN, acos gives NaN and the comparison is false; the normal of a triangle is not
// normalised, Triangle.cpp:57-60, so |x| > 1 occurs).  That threshold was found by bisection on the
// C library's acos and checked on every double within two million places of it
// (tests/test_oracle_kat.py sweeps the bracket against the oracle, which calls acos).
constexpr double kEnteringMin = 1.7225464241988333e-16;     // bits 0x3ca8d313198a2e04
RTX_DEV bool is_entering(d3 normal, d3 ray_inv_dir)
{
    const double x = dot(normal, ray_inv_dir);
    return x >= kEnteringMin && x <= 1.0;
}

}  // namespace rtx
