// rtx_kernels.hpp -- the wavefront pipeline's kernels (gfx950, wave64).
//
//   k_generate        per-pixel lambda up to the camera ray (Raytracer.cpp:676-700)
//   k_trace_pool      shootRay for a ray queue: KD traversal + leaf scans
//   k_trace_api       the same for caller-supplied rays, with leaf sequences/counters
//   k_shade_path      pathTrace body (Raytracer.cpp:217-355), one level per launch
//   k_shade_whitted_a rayTrace up to the light rays (Raytracer-Noah-Desktop.cpp:452-490)
//   k_shade_whitted_b rayTrace after the shadow rays returned (:490-553)
//   k_resolve         drawBuffer[pixel] += colorAcum / (n*n)  (Raytracer.cpp:705-710)
//
// The reference recurses; here each recursion level is one pass over a queue of
// rays that carry their accumulated weight, and every queue is compacted with
// wave64 ballot / mbcnt prefix sums, combined per workgroup in LDS so that a
// workgroup issues one atomic on the queue tail.
#pragma once
#include "rtx_device.hpp"

namespace rtx {

constexpr int kBlock = 256;             // threads per workgroup (4 waves)
constexpr int kTraceBlock = 64;         // workgroup of the traversal kernels
constexpr int kStack = 24;              // per-lane stack entries: KD depth is capped at 21 (KDTree.cpp:69), BVH depth at 22

// ---- ray queue (SoA, resident in HBM) ---------------------------------------------------
struct RayQueue {
    double *ox, *oy, *oz, *dx, *dy, *dz;    // Ray (direction already normalised by Ray::Ray)
    double *wx, *wy, *wz;                   // weight this ray's radiance carries into its sample
    // (the reference's currentIOR argument is passed down and never read, Raytracer.cpp:217,
    //  Raytracer-Noah-Desktop.cpp:452; the recursion depth is the same for all rays of a level)
    uint32_t *slot;                         // sample slot the ray contributes to
    uint32_t *branch, *counter;             // random stream (branch code, next draw index)
    int32_t *hit_prim;                      // filled by k_trace_pool
    double *hit_depth;
};

// Light rays of one level.  A hit with compact index h (its place among the level's hits) owns
// the entries m * seg + h, m = 0 .. samples-1, where seg is the number of rays in the shaded
// queue: sample-major over the whole level, so for one m neighbouring lanes touch neighbouring
// entries in all three kernels.  The origin is shared by a hit's samples and stored once, at h.
struct ShadowQueue {
    double *ox, *oy, *oz;                   // per hit
    double *dx, *dy, *dz;                   // per entry
    double *limit;                          // light distance: hits beyond it do not matter
    double *hit_depth;
};

// the gather rays of GLOBAL_LIGHTING (Raytracer-Noah-Desktop.cpp:499-524): closest-hit queries
// whose hit (or miss) and distance are all that is used; laid out like the shadow queue
struct GatherQueue {
    double *ox, *oy, *oz;                   // per hit
    double *dx, *dy, *dz;                   // per entry
    int32_t *hit_prim;
    double *hit_depth;
};

struct Counters { unsigned long long* c; };     // 8 global work counters, or null

// Division of a 32-bit unsigned number by a divisor every thread of a launch shares (Granlund and
// Montgomery, "Division by invariant integers using multiplication", 1994, figure 4.1): the host works
// out a multiplier and two shifts once, the device spends a multiply-high, two adds and two shifts
// where a division by a register takes some twenty instructions.  Exact for every n and every d >= 1.
struct FastDiv {
    uint32_t mul, shift1, shift2, d;
};
inline FastDiv make_fast_div(uint32_t d)
{
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while (l < 32 && (1ull << l) < d) ++l;          // ceil(log2 d)
    f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.shift1 = l < 1 ? l : 1;
    f.shift2 = l > 0 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& f)
{
    const uint32_t t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.shift1)) >> f.shift2;
}

struct RenderParams {
    rtx_render_desc d;
    // derived from d by the host (fill_derived): divisors of the pixel and sub-sample indices, and the
    // camera's sub-sample offsets offx = sx * (cx / (n + 1)) with cx = 1 / (width - 1) (Raytracer.cpp:
    // 676-700) -- the two divisions are the same for every ray of the frame and are done once, in the
    // same FP64 arithmetic
    FastDiv div_band;           // by 8 * tile_w
    FastDiv div_nn;             // by n * n
    FastDiv div_n;              // by n
    FastDiv div_group;          // by tile_row_group (1 when rows are not grouped)
    FastDiv div_stride;         // by tile_row_stride
    double sub_step_x, sub_step_y;
    uint32_t chunk_t0;          // first pixel index (in the order of pixel_of) of this chunk
    uint32_t chunk_pixels;      // pixels in this chunk
    uint32_t pass;
    uint32_t n;                 // pixel_multisample_n
    uint32_t shadow_per_hit;    // sum over lights of samples
    uint32_t queue_cap;         // entries a ray queue can hold; writes beyond are dropped and
                                //   the host turns the overshooting count into RTX_ERR_LIMIT
    uint32_t layer;             // recursion depth of the rays being shaded (the same for a whole level)
    uint32_t seg;               // rays in the queue being shaded = stride between a hit's shadow / gather entries
    // GLOBAL_LIGHTING at the level being shaded (all rays of a level share their layer):
    uint32_t gather_samples;    // globalSamples per light (:500); 0 = off
    uint32_t gather_traced;     // 1 when layer+1 <= MAXBOUNCES, i.e. the nested call shoots its ray
    uint32_t atomic_samples;    // 1: several rays of one launch may add to the same sample slot
    uint32_t overwrite;         // 1: this pass stores into accum instead of adding to it
    uint32_t camera_queue;      // 1: the queue being shaded is level 0 and is not stored: ray i is the camera
                                //   ray of sample slot i, its weight 1, its random stream at (0, 0)
    uint32_t direct_store;      // 0: sample slots are cleared and added to.  1: every slot gets its whole value
                                //   from one thread of one launch (a pathTrace sample is weight x terminal
                                //   radiance, added once; a rayTrace call without children is finished by the
                                //   thread that shades its hit or miss), so slots are stored, neither cleared
                                //   nor read back.  2: as 1 with one sub-sample per pixel: the value goes
                                //   straight into the framebuffer and no resolve launch follows
    uint32_t band_stride;       // 1, or 2 when the tile's two chunks are its alternate bands of 8 rows: the chunk's
    uint32_t band_first;        //   pixel index walks the bands band_first, band_first + band_stride, ... of the tile
    float* accum;               // the framebuffer (direct_store == 2)
    uint32_t light_rays_from_origin;    // 1: the scene's one light takes no random numbers (Point / Directional), so a
                                //   light ray is a function of its origin: the shadow queue holds origins only and
                                //   the traversal and the second shading half recompute direction and light distance
};

// What the traversal launch of such a level needs to make its own rays and to clear the
// chunk's state (it is the chunk's first launch).
struct CameraQueue {
    RenderParams p;             // meaningful when `on`
    double* sample;             // the chunk's accumulator, 3 doubles per sample slot
    uint32_t* counts;           // the ring of queue lengths (8 words)
    uint32_t on;
    uint32_t clear_samples;     // 0 when the slots are stored, not added to (RenderParams::direct_store)
};

// ---- traversal core -------------------------------------------------------------------------
template <bool COUNT>
struct Work {
    uint32_t branch = 0, leaf = 0, aabb = 0, tri = 0, sph = 0, accepted = 0;
    uint32_t content = 0;           // content-box tests of the culled walk / BVH child-box tests
    uint32_t fallback = 0;          // rays re-traced in reference order
    uint32_t leaf_count = 0;
    unsigned long long leaf_hash = 14695981039346656037ull;
    // wave-level tallies of the BVH walk, kept by whichever lane is the first active one at the time:
    // times the wave ran the node step, times it ran a trip of the leaf loop, and the trips it would
    // need if a leaf phase's shape tests were dealt out over all 64 lanes
    uint32_t wave_node = 0, wave_trip = 0, wave_trip_packed = 0;
};
template <>
struct Work<false> {};

// rayHitListOfShapes (KDTree.cpp:206-220) over leaf references (or over all shapes
// when refs == nullptr: the !KDTRACE path, Raytracer.cpp:117)
template <bool COUNT>
RTX_DEV void scan_refs(const DScene& s, const uint32_t* __restrict__ refs, uint32_t count, d3 o, d3 d, d3 inv,
                       int32_t& best_prim, double& best_depth, Work<COUNT>& w)
{
    for (uint32_t k = 0; k < count; ++k) {
        uint32_t prim = refs ? refs[k] : k;
        const PrimBox* pb = s.prim_box + prim;
        double tmin, tmax;
        slab(ld3(pb->bmin), ld3(pb->bmax), o, inv, tmin, tmax);
        if constexpr (COUNT) w.aabb++;
        if (tmin <= tmax && tmax >= 0.0) {
            uint32_t info = s.prim_info[prim];
            uint32_t type = info & 3u, idx = info >> 2;
            double depth = kInf;
            uint32_t rp = prim;
            bool ok;
            if (type == RTX_PRIM_TRIANGLE) {
                if constexpr (COUNT) w.tri++;
                ok = triangle_depth(s.tri_pos + 9 * (size_t)idx, o, d, depth);
            } else if (type == RTX_PRIM_SPHERE) {
                if constexpr (COUNT) w.sph++;
                ok = sphere_t(ld3(s.sph_center + 3 * (size_t)idx), s.sph_r2[idx], o, d, depth);
            } else {
                if constexpr (COUNT) w.sph += 2;
                ok = biconvex_depth(s, idx, o, d, depth, rp);
            }
            if (ok && depth < best_depth) {
                if constexpr (COUNT) w.accepted++;
                best_depth = depth;
                best_prim = (int32_t)rp;
            }
        }
    }
}

// traverseKDTree (KDTree.cpp:158-204) as an iterative walk: lesser subtree first,
// the greater child is deferred on a per-lane stack kept in LDS (stack[level*kTraceBlock]).
template <bool COUNT>
RTX_DEV void trace_lane(const DScene& s, d3 o, d3 d, bool linear, int32_t* stack, int32_t& best_prim,
                        double& best_depth, Work<COUNT>& w, uint32_t* seq, uint32_t seq_cap)
{
    best_prim = -1;
    best_depth = kInf;
    d3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    if (linear) { scan_refs<COUNT>(s, nullptr, s.n_shapes, o, d, inv, best_prim, best_depth, w); return; }
    int sp = 0;
    int32_t node = s.root;
    for (;;) {
        if (node >= 0) {
            const Branch* b = s.branches + node;
            double tmin, tmax;
            slab(ld3(b->bmin), ld3(b->bmax), o, inv, tmin, tmax);
            if constexpr (COUNT) w.branch++;
            if (tmin < tmax && tmax >= 0.0) {
                int axis = b->axis;
                double value = b->value;
                double oa = pick(o, axis), da = pick(d, axis);
                double enter = oa + (da * tmin);
                double exit_ = oa + (da * tmax);
                int32_t lesser = b->lesser, greater = b->greater;
                bool only_lesser, only_greater;
                if (enter <= value) { only_lesser = exit_ < value; only_greater = false; }
                else { only_greater = exit_ > value; only_lesser = false; }
                if (only_greater) { node = greater; }
                else {
                    if (!only_lesser) { stack[sp * kTraceBlock] = greater; ++sp; }
                    node = lesser;
                }
                continue;
            }
        } else {
            uint32_t leaf = (uint32_t)~node;
            if constexpr (COUNT) {
                if (seq && w.leaf_count < seq_cap) seq[w.leaf_count] = leaf;
                w.leaf_count++;
                for (int b = 0; b < 4; ++b) { w.leaf_hash ^= (leaf >> (8 * b)) & 0xffu; w.leaf_hash *= 1099511628211ull; }
                w.leaf++;
            }
            uint2 fc = s.leaves[leaf];
            scan_refs<COUNT>(s, s.leaf_refs + fc.x, fc.y, o, d, inv, best_prim, best_depth, w);
        }
        if (sp == 0) break;
        --sp;
        node = stack[sp * kTraceBlock];
    }
}

// ---- loads through the scalar cache ------------------------------------------------------------
// The walks wait on chains of dependent loads, and a load through the scalar cache returns sooner
// than one through the vector memory pipeline.  Where every active lane of the wave asks for the
// same address -- neighbouring rays at the same node, leaf or shape -- the value is fetched once
// with a scalar load (reads only: the scalar cache is never written through).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
// (the address is the same in every active lane: say so to the compiler)
RTX_DEV const void* uniform_address(const void* p)
{
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    return (const void*)(((uint64_t)hi << 32) | lo);
}
RTX_DEV uint32_t scalar_load_4(const void* p)
{
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_address(p)));
    return v;
}
RTX_DEV u32x2 scalar_load_8(const void* p)
{
    u32x2 v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_address(p)));
    return v;
}
RTX_DEV u32x4 scalar_load_16(const void* p)
{
    u32x4 v;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_address(p)));
    return v;
}
RTX_DEV u32x8 scalar_load_32(const void* p)
{
    u32x8 v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_address(p)));
    return v;
}
RTX_DEV u32x16 scalar_load_64(const void* p)
{
    u32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_address(p)));
    return v;
}
// two loads from one record in flight together (one wait instead of two)
RTX_DEV void scalar_load_64_16(const void* p, u32x16& a, u32x4& b)
{
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b) : "s"(uniform_address(p)));
}
RTX_DEV void scalar_load_32_16(const void* p, u32x8& a, u32x4& b)
{
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b) : "s"(uniform_address(p)));
}
RTX_DEV double dbl(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
// index of the first active lane, and whether all active lanes hold it
RTX_DEV bool wave_uniform(uint32_t index, uint32_t& first)
{
    first = (uint32_t)__builtin_amdgcn_readfirstlane((int)index);
    return __all(index == first);
}

// a shape's box (48 bytes), through the scalar cache when `uniform` (all active lanes ask for shape `first`)
struct Box6 { d3 mn, mx; };
RTX_DEV Box6 load_prim_box(const PrimBox* __restrict__ boxes, uint32_t shape, bool uniform, uint32_t first)
{
    Box6 r;
    if (uniform) {
        const char* p = reinterpret_cast<const char*>(boxes + first);
        u32x8 a; u32x4 b;
        scalar_load_32_16(p, a, b);
        r.mn = mk(dbl(a[0], a[1]), dbl(a[2], a[3]), dbl(a[4], a[5]));
        r.mx = mk(dbl(a[6], a[7]), dbl(b[0], b[1]), dbl(b[2], b[3]));
    } else {
        const PrimBox* b = boxes + shape;
        r.mn = mk(b->bmin[0], b->bmin[1], b->bmin[2]);
        r.mx = mk(b->bmax[0], b->bmax[1], b->bmax[2]);
    }
    return r;
}

// ---- culled traversal: same answer, less work ---------------------------------------------------
// shootRay's result is the lexicographic minimum of (depth, position in the reference's
// visit order) over the candidates the reference's walk generates: it replaces the best
// hit only on a strictly smaller depth (KDTree.cpp:212) and always walks lesser before
// greater, so leaf-reference index order IS its visit order.  That minimum does not depend
// on the order in which candidates are examined, and candidates that provably cannot win
// need not be examined at all.  This walk therefore
//   * applies the reference's own branch predicate (region slab test + enter/exit against
//     the split value), so it never enters a subtree the reference would not enter;
//   * skips a node when the ray misses the node's content box (the union of the shape boxes
//     below it): the slab arithmetic is monotone in the box for finite 1/direction, so every
//     shape-box test below would fail as well (Shape.cpp:100-116);
//   * skips a node when the Euclidean distance from the ray origin to the content box
//     exceeds the best depth so far by a safety margin: every hit point the reference
//     computes lies in its shape's box (a convex combination of the vertices, or a point
//     on the sphere), and depth is its distance from the origin (Triangle.cpp:52-55,
//     Sphere.cpp:73-77);
//   * visits the nearer child first so that the bound tightens early;
//   * in ANYHIT form (shadow rays) stops at the first candidate with depth <= limit,
//     which decides `lightDistance < depth` (Raytracer-Noah-Desktop.cpp:492) identically.
// Rays whose 1/direction is not finite (axis-parallel rays: NaNs can reach the slab tests)
// take the reference-order walk instead.
RTX_DEV void slab_finite(const PrimBox* __restrict__ b, d3 o, d3 inv, double& tmin, double& tmax)
{
    // for non-NaN operands min/max instructions agree with glm's selects up to the sign
    // of zero, which no comparison below can observe
    double t0x = (b->bmin[0] - o.x) * inv.x, t1x = (b->bmax[0] - o.x) * inv.x;
    double t0y = (b->bmin[1] - o.y) * inv.y, t1y = (b->bmax[1] - o.y) * inv.y;
    double t0z = (b->bmin[2] - o.z) * inv.z, t1z = (b->bmax[2] - o.z) * inv.z;
    tmin = __builtin_fmax(__builtin_fmin(t0x, t1x), __builtin_fmax(__builtin_fmin(t0y, t1y), __builtin_fmin(t0z, t1z)));
    tmax = __builtin_fmin(__builtin_fmax(t0x, t1x), __builtin_fmin(__builtin_fmax(t0y, t1y), __builtin_fmax(t0z, t1z)));
}

// the same two tests on a box held in registers
RTX_DEV void slab_finite(d3 bmin, d3 bmax, d3 o, d3 inv, double& tmin, double& tmax)
{
    double t0x = (bmin.x - o.x) * inv.x, t1x = (bmax.x - o.x) * inv.x;
    double t0y = (bmin.y - o.y) * inv.y, t1y = (bmax.y - o.y) * inv.y;
    double t0z = (bmin.z - o.z) * inv.z, t1z = (bmax.z - o.z) * inv.z;
    tmin = __builtin_fmax(__builtin_fmin(t0x, t1x), __builtin_fmax(__builtin_fmin(t0y, t1y), __builtin_fmin(t0z, t1z)));
    tmax = __builtin_fmin(__builtin_fmax(t0x, t1x), __builtin_fmin(__builtin_fmax(t0y, t1y), __builtin_fmax(t0z, t1z)));
}

// Slab test and distance cull of one BVH child box from shared differences.  `gap` is a lower
// bound of how far the box is from the ray origin, to be compared with the caller's bound:
// the squared Euclidean distance (closest-hit form), or, cheaper, the largest per-axis gap
// (LINF form: never above the Euclidean distance, so culling with it is as safe).
template <bool LINF>
RTX_DEV void slab_and_gap(d3 bmin, d3 bmax, d3 o, d3 inv, double& tmin, double& tmax, double& gap)
{
    const double ax = bmin.x - o.x, bx = bmax.x - o.x;
    const double ay = bmin.y - o.y, by = bmax.y - o.y;
    const double az = bmin.z - o.z, bz = bmax.z - o.z;
    const double t0x = ax * inv.x, t1x = bx * inv.x;
    const double t0y = ay * inv.y, t1y = by * inv.y;
    const double t0z = az * inv.z, t1z = bz * inv.z;
    tmin = __builtin_fmax(__builtin_fmin(t0x, t1x), __builtin_fmax(__builtin_fmin(t0y, t1y), __builtin_fmin(t0z, t1z)));
    tmax = __builtin_fmin(__builtin_fmax(t0x, t1x), __builtin_fmin(__builtin_fmax(t0y, t1y), __builtin_fmax(t0z, t1z)));
    // o - bmax == -(bmax - o) exactly
    const double gx = __builtin_fmax(ax, -bx), gy = __builtin_fmax(ay, -by), gz = __builtin_fmax(az, -bz);
    if constexpr (LINF) {
        gap = __builtin_fmax(gx, __builtin_fmax(gy, gz));
    } else {
        const double ex = __builtin_fmax(gx, 0.0), ey = __builtin_fmax(gy, 0.0), ez = __builtin_fmax(gz, 0.0);
        gap = (ex * ex + ey * ey) + ez * ez;
    }
}

RTX_DEV double box_distance2(d3 bmin, d3 bmax, d3 o)
{
    double ex = __builtin_fmax(__builtin_fmax(bmin.x - o.x, o.x - bmax.x), 0.0);
    double ey = __builtin_fmax(__builtin_fmax(bmin.y - o.y, o.y - bmax.y), 0.0);
    double ez = __builtin_fmax(__builtin_fmax(bmin.z - o.z, o.z - bmax.z), 0.0);
    return (ex * ex + ey * ey) + ez * ez;
}

RTX_DEV double box_distance2(const PrimBox* __restrict__ b, d3 o)
{
    double ex = __builtin_fmax(__builtin_fmax(b->bmin[0] - o.x, o.x - b->bmax[0]), 0.0);
    double ey = __builtin_fmax(__builtin_fmax(b->bmin[1] - o.y, o.y - b->bmax[1]), 0.0);
    double ez = __builtin_fmax(__builtin_fmax(b->bmin[2] - o.z, o.z - b->bmax[2]), 0.0);
    return (ex * ex + ey * ey) + ez * ez;
}

template <bool COUNT, bool ANYHIT>
RTX_DEV void trace_lane_culled(const DScene& s, d3 o, d3 d, double limit, int32_t* stack, int32_t& best_prim,
                               double& best_depth, Work<COUNT>& w)
{
    best_prim = -1;
    best_depth = kInf;
    uint32_t best_pos = 0xffffffffu;
    d3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    // margin of the distance bound: far above any rounding in the reference's hit points
    const double abs_margin = 1e-9 * (s.scene_scale + __builtin_fmax(fabs(o.x), __builtin_fmax(fabs(o.y), fabs(o.z))));
    double bound2 = kInf;
    if constexpr (ANYHIT) { double bd = limit * (1.0 + 1e-7) + abs_margin; bound2 = bd * bd; }
    int sp = 0;
    int32_t node = s.root;
    for (;;) {
        const PrimBox* cb = node >= 0 ? s.branch_cbox + node : s.leaf_cbox + (uint32_t)~node;
        double tmin, tmax;
        slab_finite(cb, o, inv, tmin, tmax);
        if constexpr (COUNT) w.content++;
        if (tmin <= tmax && tmax >= 0.0 && !(box_distance2(cb, o) > bound2)) {
            if (node >= 0) {
                const Branch* b = s.branches + node;
                slab(ld3(b->bmin), ld3(b->bmax), o, inv, tmin, tmax);
                if constexpr (COUNT) w.branch++;
                if (tmin < tmax && tmax >= 0.0) {
                    int axis = b->axis;
                    double value = b->value;
                    double oa = pick(o, axis), da = pick(d, axis);
                    double enter = oa + (da * tmin);
                    double exit_ = oa + (da * tmax);
                    int32_t lesser = b->lesser, greater = b->greater;
                    bool only_lesser, only_greater;
                    if (enter <= value) { only_lesser = exit_ < value; only_greater = false; }
                    else { only_greater = exit_ > value; only_lesser = false; }
                    if (only_greater) node = greater;
                    else if (only_lesser) node = lesser;
                    else {
                        bool lesser_first = da >= 0.0;
                        stack[sp * kTraceBlock] = lesser_first ? greater : lesser;
                        ++sp;
                        node = lesser_first ? lesser : greater;
                    }
                    continue;
                }
            } else {
                uint32_t leaf = (uint32_t)~node;
                if constexpr (COUNT) w.leaf++;
                uint2 fc = s.leaves[leaf];
                for (uint32_t k = 0; k < fc.y; ++k) {
                    uint32_t prim = s.leaf_refs[fc.x + k];
                    slab_finite(s.prim_box + prim, o, inv, tmin, tmax);
                    if constexpr (COUNT) w.aabb++;
                    if (tmin <= tmax && tmax >= 0.0) {
                        uint32_t info = s.prim_info[prim];
                        uint32_t type = info & 3u, idx = info >> 2;
                        double depth = kInf;
                        uint32_t rp = prim;
                        bool ok;
                        if (type == RTX_PRIM_TRIANGLE) {
                            if constexpr (COUNT) w.tri++;
                            ok = triangle_depth(s.tri_pos + 9 * (size_t)idx, o, d, depth);
                        } else if (type == RTX_PRIM_SPHERE) {
                            if constexpr (COUNT) w.sph++;
                            ok = sphere_t(ld3(s.sph_center + 3 * (size_t)idx), s.sph_r2[idx], o, d, depth);
                        } else {
                            if constexpr (COUNT) w.sph += 2;
                            ok = biconvex_depth(s, idx, o, d, depth, rp);
                        }
                        if (ok) {
                            if constexpr (ANYHIT) {
                                if (depth <= limit) { best_depth = depth; best_prim = (int32_t)rp; return; }
                            } else {
                                uint32_t pos = fc.x + k;
                                if (depth < best_depth || (depth == best_depth && pos < best_pos)) {
                                    if constexpr (COUNT) w.accepted++;
                                    best_depth = depth; best_prim = (int32_t)rp; best_pos = pos;
                                    double bd = best_depth * (1.0 + 1e-7) + abs_margin;
                                    bound2 = bd * bd;
                                }
                            }
                        }
                    }
                }
            }
        }
        if (sp == 0) break;
        --sp;
        node = stack[sp * kTraceBlock];
    }
    if constexpr (ANYHIT) { best_depth = kInf; best_prim = -1; }     // nothing at or before the light
}

// ---- BVH walk with membership verification: same answer, far less work ---------------------------
// The candidates shootRay chooses from are the shapes S whose box test passes and whose
// rayHit succeeds AND that sit in some leaf traverseKDTree visits for this ray.  The BVH
// enumerates the first set (each shape once, nearest first, with the same box-miss and
// distance culls as above, which are exact for any boxes that contain the shape boxes); the
// second condition is then checked against the reference's own tree:
//   * the walk order of leaves equals their numbering, and the inverse leaf index lists, per
//     shape, the leaves that reference it in increasing order;
//   * branch_predicate() evaluates which children traverseKDTree would enter at a branch
//     (KDTree.cpp:169-201), so descending from the root towards a given leaf tells whether
//     that leaf is visited; a refused child excludes its whole range of leaf ids at once.
// first_visited_position() therefore returns the shape's position in the reference's visit
// order (its leaf_refs index in the first visited leaf holding it), or none.  Exact depth
// ties between different shapes are resolved with these positions, as the strict `<` of
// KDTree.cpp:212 does.
constexpr uint32_t kNoPosition = 0xffffffffu;

RTX_DEV void branch_predicate(const Branch* __restrict__ b, d3 o, d3 d, d3 inv, bool& go_lesser, bool& go_greater)
{
    double tmin, tmax;
    // callers have finite 1/direction, where this equals the select-based slab test
    slab_finite(reinterpret_cast<const PrimBox*>(b->bmin), o, inv, tmin, tmax);
    go_lesser = go_greater = false;
    if (tmin < tmax && tmax >= 0.0) {
        int axis = b->axis;
        double value = b->value;
        double oa = pick(o, axis), da = pick(d, axis);
        double enter = oa + (da * tmin);
        double exit_ = oa + (da * tmax);
        if (enter <= value) { go_lesser = true; go_greater = !(exit_ < value); }
        else { go_greater = true; go_lesser = !(exit_ > value); }
    }
}

// index of the first entry >= key in the sorted range ids[lo, hi)
RTX_DEV uint32_t lower_bound_u32(const uint32_t* __restrict__ ids, uint32_t lo, uint32_t hi, uint32_t key)
{
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (ids[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <bool COUNT>
RTX_DEV uint32_t first_visited_position(const DScene& s, uint32_t shape, d3 o, d3 d, d3 inv, Work<COUNT>& w)
{
    const uint32_t first = s.prim_leaf_first[shape], end = s.prim_leaf_first[shape + 1];
    uint32_t i = first;
    while (i < end) {
        uint32_t target = s.prim_leaf_ids[i];
        int32_t node = s.root;
        uint32_t skip_to;
        for (;;) {
            if (node < 0) return s.prim_leaf_pos[i];           // reached the leaf `target`: it is visited
            const Branch* b = s.branches + node;
            bool gl, gg;
            branch_predicate(b, o, d, inv, gl, gg);
            if constexpr (COUNT) w.branch++;
            if (target < b->leaf_mid) {
                if (gl) node = b->lesser; else { skip_to = b->leaf_mid; break; }
            } else {
                if (gg) node = b->greater; else { skip_to = b->leaf_hi; break; }
            }
        }
        i = lower_bound_u32(s.prim_leaf_ids, i + 1, end, skip_to);
    }
    return kNoPosition;
}

// Cheap existence proof: follow the KD tree towards the leaf that contains the point p
// (normally the hit point) as long as the reference would go that way, then look the shape
// up in that leaf.  A `false` only means "not shown this way".
template <bool COUNT>
RTX_DEV bool member_through_point(const DScene& s, uint32_t shape, d3 p, d3 o, d3 d, d3 inv, Work<COUNT>& w)
{
    int32_t node = s.root;
    while (node >= 0) {
        const Branch* b = s.branches + node;
        bool gl, gg;
        branch_predicate(b, o, d, inv, gl, gg);
        if constexpr (COUNT) w.branch++;
        bool lesser_side = pick(p, b->axis) <= b->value;
        if (lesser_side ? gl : gg) node = lesser_side ? b->lesser : b->greater;
        else if (lesser_side ? gg : gl) node = lesser_side ? b->greater : b->lesser;
        else return false;
    }
    uint32_t leaf = (uint32_t)~node;
    const uint32_t first = s.prim_leaf_first[shape], end = s.prim_leaf_first[shape + 1];
    uint32_t at = lower_bound_u32(s.prim_leaf_ids, first, end, leaf);
    if constexpr (COUNT) w.leaf++;
    return at < end && s.prim_leaf_ids[at] == leaf;
}

// Cheapest existence proof.  When some point p = o + d*t (t > 0) of the ray lies inside the
// cell of a KD leaf -- inside every ancestor's box and on the leaf's side of every ancestor's
// split plane -- by a margin far above the rounding error of the slab arithmetic (margin
// ~1e-9 of the scene size, errors ~1e-15), traverseKDTree provably reaches that leaf: at each
// ancestor the box is hit over a parameter interval of positive length around t, and since
// the coordinate along the split axis is monotone in the parameter, either the entry point or
// the exit point is on p's side of the plane by more than the error, which is exactly what
// KDTree.cpp:171-194 tests.  The leaf is located from the deepest node that covers all leaves
// of the shape, so only a few levels are walked and only with (axis, value).
// The point tried first is the hit point.  A hit point ON a face of its cell (a floor that
// is the bottom of the scene box, geometry lying in a split plane) cannot pass; the second
// try is the point a little earlier on the ray, as far back as the offending faces require.
// A `false` only means "not shown this way".
RTX_DEV int32_t locate_leaf(const DScene& s, int32_t node, d3 p)
{
    while (node >= 0) {
        uint32_t first;
        KdSplit k;
        if (wave_uniform((uint32_t)node, first)) {
            const u32x4 v = scalar_load_16(s.kd_split + first);
            k.value = dbl(v[0], v[1]); k.lesser_axis = (int32_t)v[2]; k.greater = (int32_t)v[3];
        } else {
            k = s.kd_split[node];
        }
        node = pick(p, k.lesser_axis & 3) <= k.value ? (k.lesser_axis >> 2) : k.greater;
    }
    return node;
}

// whether `leaf` is among the leaves of `shape`; shape_uniform: all active lanes ask about shape `first_shape`
RTX_DEV bool leaf_holds(const DScene& s, uint32_t shape, uint32_t leaf, bool shape_uniform = false, uint32_t first_shape = 0u)
{
    uint32_t lo, hi;
    if (shape_uniform) {
        const u32x2 r = scalar_load_8(s.prim_leaf_first + first_shape);
        lo = r[0]; hi = r[1];
    } else {
        lo = s.prim_leaf_first[shape]; hi = s.prim_leaf_first[shape + 1];
    }
    const uint32_t end = hi;
    uint32_t first_leaf;
    if (shape_uniform && wave_uniform(leaf, first_leaf)) {
        // the whole search runs on scalar loads: every lane looks for the same leaf in the same list
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (scalar_load_4(s.prim_leaf_ids + mid) < first_leaf) lo = mid + 1; else hi = mid;
        }
        return lo < end && scalar_load_4(s.prim_leaf_ids + lo) == first_leaf;
    }
    uint32_t at = lower_bound_u32(s.prim_leaf_ids, lo, hi, leaf);
    return at < end && s.prim_leaf_ids[at] == leaf;
}

RTX_DEV Box6 load_leaf_cell(const DScene& s, uint32_t leaf)
{
    uint32_t first;
    const bool uniform = wave_uniform(leaf, first);
    return load_prim_box(s.leaf_cell, leaf, uniform, first);
}

RTX_DEV bool member_by_cell(const DScene& s, uint32_t shape, d3 o, d3 d, d3 inv, double t, double margin)
{
    uint32_t first_shape;
    const bool shape_uniform = wave_uniform(shape, first_shape);
    const int32_t lca_code = shape_uniform ? (int32_t)scalar_load_4(s.prim_lca + first_shape) : s.prim_lca[shape];
    if (lca_code == kLcaNone) return false;     // in no leaf at all: the other proofs find its list empty and refuse it too
    const int32_t lca = lca_code < 0 ? lca_code : (lca_code & ~kLcaSure);
    // The leaf list need not be searched when the shape has one leaf only (the one located), or when
    // every leaf whose cell comes within kHoldMargin of the shape's box is known to hold it (kLcaSure)
    // and the point tried is seen to lie that close to the box: it is in the cell located, so the
    // cell reaches the widened box.
    const bool one_leaf = lca_code < 0, sure = !one_leaf && (lca_code & kLcaSure);
    Box6 sb;
    if (sure) sb = load_prim_box(s.prim_box, shape, shape_uniform, first_shape);
    auto listed = [&](d3 q) {
        if (one_leaf) return true;
        if (!sure) return false;
        const double e = 0.9 * kHoldMargin * s.scene_scale;
        return q.x >= sb.mn.x - e && q.x <= sb.mx.x + e && q.y >= sb.mn.y - e && q.y <= sb.mx.y + e &&
               q.z >= sb.mn.z - e && q.z <= sb.mx.z + e;
    };
    // A shape whose box is flat on an axis (a floor) is hit ON that plane, and where the plane is a face of
    // the KD cells -- the bottom of the scene box -- the hit point can never be inside a cell by the
    // margin: the point tried first is then the one the step-back rule below would arrive at, a little
    // earlier on the ray, which saves the first of two descents.  Any point of the ray serves; if
    // this one fails too, the rule steps back from it.
    if (sure) {
        double early = 0.0;
        if (sb.mn.x == sb.mx.x) early = __builtin_fmax(early, __builtin_fabs(inv.x));
        if (sb.mn.y == sb.mx.y) early = __builtin_fmax(early, __builtin_fabs(inv.y));
        if (sb.mn.z == sb.mx.z) early = __builtin_fmax(early, __builtin_fabs(inv.z));
        const double t_early = t - (2.5 * margin) * early;
        if (early > 0.0 && t_early > 0.0) t = t_early;
    }
    d3 p = o + (d * t);
    uint32_t leaf = (uint32_t)~locate_leaf(s, lca, p);
    Box6 c = load_leaf_cell(s, leaf);
    // how far each coordinate is short of being `margin` inside the cell (<= 0: fine)
    const double lx = (c.mn.x + margin) - p.x, hx = p.x - (c.mx.x - margin);
    const double ly = (c.mn.y + margin) - p.y, hy = p.y - (c.mx.y - margin);
    const double lz = (c.mn.z + margin) - p.z, hz = p.z - (c.mx.z - margin);
    if (lx <= 0.0 && hx <= 0.0 && ly <= 0.0 && hy <= 0.0 && lz <= 0.0 && hz <= 0.0) return listed(p) || leaf_holds(s, shape, leaf, shape_uniform, first_shape);
    // step back along the ray.  A face the ray leaves through is cleared by `margin`; a face it came in
    // through is crossed again, by `margin` into the cell on its other side (geometry lying in a split
    // plane is held by the leaves of both sides).  The point stepped to is checked like the first.
    double back = 0.0;
    if (lx > 0.0) { if (d.x == 0.0) return false; back = __builtin_fmax(back, (d.x < 0.0 ? lx + margin : 3.0 * margin - lx) / __builtin_fabs(d.x)); }
    if (hx > 0.0) { if (d.x == 0.0) return false; back = __builtin_fmax(back, (d.x > 0.0 ? hx + margin : 3.0 * margin - hx) / __builtin_fabs(d.x)); }
    if (ly > 0.0) { if (d.y == 0.0) return false; back = __builtin_fmax(back, (d.y < 0.0 ? ly + margin : 3.0 * margin - ly) / __builtin_fabs(d.y)); }
    if (hy > 0.0) { if (d.y == 0.0) return false; back = __builtin_fmax(back, (d.y > 0.0 ? hy + margin : 3.0 * margin - hy) / __builtin_fabs(d.y)); }
    if (lz > 0.0) { if (d.z == 0.0) return false; back = __builtin_fmax(back, (d.z < 0.0 ? lz + margin : 3.0 * margin - lz) / __builtin_fabs(d.z)); }
    if (hz > 0.0) { if (d.z == 0.0) return false; back = __builtin_fmax(back, (d.z > 0.0 ? hz + margin : 3.0 * margin - hz) / __builtin_fabs(d.z)); }
    const double t2 = t - back;
    if (!(t2 > 0.0)) return false;
    p = o + (d * t2);
    leaf = (uint32_t)~locate_leaf(s, lca, p);
    c = load_leaf_cell(s, leaf);
    bool inside = p.x - margin >= c.mn.x && p.x + margin <= c.mx.x &&
                  p.y - margin >= c.mn.y && p.y + margin <= c.mx.y &&
                  p.z - margin >= c.mn.z && p.z + margin <= c.mx.z;
    return inside && (listed(p) || leaf_holds(s, shape, leaf, shape_uniform, first_shape));
}

// What a walk leaves for a separate verification pass (k_verify), per ray:
constexpr uint32_t kCandSettled = 0xffffffffu;      // nothing to verify: no geometric candidate (the miss is final), or
                                                    //   the ray was walked in reference order already
constexpr uint32_t kCandReference = 0xfffffffeu;    // not walked (non-finite ray): take the reference-order walk
constexpr uint32_t kCandTie = 0x80000000u;          // flag: another shape ties at the winning depth
constexpr uint32_t kCandBehind = 0x40000000u;       // flag: the winning hit lies behind the ray's origin (triangle_depth_sided)
// otherwise: the winning candidate's shape id.

// The certificate of a walk's winner: shape `cand & 0xffffff...` hit at `depth`.  The proofs want a
// point of the ray (parameter > 0) in a KD leaf that holds the shape.  For a hit in front of the origin
// that is the hit point.  A hit BEHIND the origin is not on the ray the reference's walk follows;
// but the shape's box test passed (it reaches in front of the origin) and the box holds the hit
// point, so the ray is inside the box from the origin on, and every KD leaf whose cell the box
// touches holds the shape: the point used is the middle of the ray's stretch inside the box.
// The certificate without a search.  The candidate's own box test passed, so the ray runs inside
// the shape's box B over a stretch [t0, t1] of its parameter, t1 >= 0.  For a shape marked kCertValid
// the host has established that (a) B lies inside the cell of the deepest KD node covering the
// shape's leaves, (b) every leaf below that node whose cell meets B lists the shape (which is what the
// reference's builder does, KDTree.cpp:78-113; cells too thin to hold a point by the margin are not
// asked), and (c) at most n_a faces of KD cells perpendicular to axis a come within kHoldMargin x
// scene_scale of B.  A point of the stretch lies within `margin` of such a face only while the ray's
// coordinate on the axis crosses a band of width 2 * margin, i.e. for 2 * margin / |d_a| of the
// parameter; so all faces together spoil at most 2 * margin * sum(n_a / |d_a|) of the stretch.  If the
// part of the stretch in front of the origin is longer than that, it holds a point x that keeps
// `margin` from every face near B, hence from every face of the cell it is in (the others are
// kHoldMargin x scene_scale from B, far more): x is inside its leaf's cell by the margin, so
// traverseKDTree reaches that leaf (member_by_cell's argument), the leaf is below the covering node by
// (a), its cell holds the point x of B and so lists the shape by (b), and the reference tests the shape
// there.  (x is a point of the real line through the ray's origin and direction; nothing is computed
// from it, and the rounding of the two slab parameters is far below the 64 margins of slack.)  Nothing is located:
// two independent loads (the counts and the box) instead of a chain of a dozen dependent ones.
// A `false` only means "not shown this way" (a flat box, a ray nearly parallel to a face, a
// shape in too many leaves to count).
RTX_DEV bool certify_by_stretch(const DScene& s, uint32_t shape, d3 o, d3 inv, double margin)
{
    uint32_t first_shape;
    const bool shape_uniform = wave_uniform(shape, first_shape);
    const uint32_t cert = shape_uniform ? scalar_load_4(s.prim_cert + first_shape) : s.prim_cert[shape];
    // (an origin so far away that the margin, which grows with it, is no longer small against the
    // distance that separates the counted faces from the others: not shown this way)
    if (!(cert & kCertValid) || !(2.0 * margin <= kHoldMargin * s.scene_scale)) return false;
    const Box6 b = load_prim_box(s.prim_box, shape, shape_uniform, first_shape);
    double tmin, tmax;
    slab_finite(b.mn, b.mx, o, inv, tmin, tmax);
    const double stretch = tmax - __builtin_fmax(tmin, 0.0);
    const double spoilt = (2.0 * margin) * (((double)(cert & 1023u) * __builtin_fabs(inv.x) + (double)((cert >> 10) & 1023u) * __builtin_fabs(inv.y)) +
                                            (double)((cert >> 20) & 1023u) * __builtin_fabs(inv.z));
    // (a factor four and a floor far above the rounding of the slab arithmetic; NaN compares false)
    return stretch > 4.0 * spoilt + 64.0 * margin;
}

template <bool COUNT>
RTX_DEV bool certify_winner(const DScene& s, uint32_t cand, d3 o, d3 d, d3 inv, double depth, double margin, Work<COUNT>& w)
{
    const uint32_t shape = cand & ~(kCandTie | kCandBehind);
    if (s.prim_cert && certify_by_stretch(s, shape, o, inv, margin)) return true;
    double t = depth;
    if (cand & kCandBehind) {
        double tmin, tmax;
        slab_finite(s.prim_box + shape, o, inv, tmin, tmax);
        t = 0.5 * (__builtin_fmax(tmin, 0.0) + tmax);
        if (!(t > 0.0)) t = depth;          // nothing in front of the origin to stand on: the proofs below decide
    }
    if (member_by_cell(s, shape, o, d, inv, t, margin)) return true;
    const d3 at = o + (d * t);
    return member_through_point<COUNT>(s, shape, at, o, d, inv, w) ||
           first_visited_position<COUNT>(s, shape, o, d, inv, w) != kNoPosition;
}

// Optimistic form: the BVH walk treats every geometric candidate as genuine, so its loop
// holds no long divergent section; the winner is verified once, after the loop, when the
// lanes of the wave have reconverged.  Returns false when the optimistic answer cannot be
// certified -- the winner is in no visited leaf, or two different primitives tie exactly
// at the winning depth (their visit order would decide) -- and the caller then runs the
// reference-order walk for that ray.  Both are rare.
// Single-precision form of the child-box test of the BVH walk.  The tree only proposes (every
// candidate is tested against its own FP64 shape box and certified against the KD walk), so a child
// box may be entered needlessly but must never be refused when the FP64 test would enter it.
// Per ray and axis: i = float(1/d), and the parameter of a plane p is t = p * i - o/d, one fused
// multiply-add.  With i and o/d rounded to single precision and the result rounded once, the
// computed t is within 2^-23 * |1/d| * (|p| + |o|) of the exact one; |p| <= scene_scale for every
// box of the tree, so widening the interval by delta = 2^-22 * |1/d| * (scene_scale + |o|) on that
// axis covers it with a factor two to spare (it amounts to growing every box by 2^-22 of the
// scene's size).  delta is folded into the additive constants: the plane the ray meets first on
// the axis (the lower one when d > 0) gets -o/d - delta, the other -o/d + delta; `a` is the constant
// that goes with a box's lower plane and `b` the one that goes with its upper plane, so that
// min / max of the two products are the widened entry and exit on the axis whatever the sign of d.
// An axis with |1/d| > 2^60 or |o/d| > 2^100 refuses nothing, and scenes whose coordinates leave
// the range in which all of this stays finite (scene_scale > 2^100) are not walked this way at all.
struct SlabF32 {
    float ix, iy, iz;
    float ax, ay, az, bx, by, bz;
};
RTX_DEV float float_not_below(double x)
{
    float f = (float)x;
    if ((double)f < x) f = __uint_as_float(__float_as_uint(f) + (f >= 0.0f ? 1u : 0xffffffffu));
    return f;
}
RTX_DEV void slab_f32_axis(double o, double inv, float scale, float& i, float& a, float& b)
{
    const double oi = o * inv;
    const bool positive = inv >= 0.0;
    float cn, cf;
    if (fabs(inv) <= 0x1p60 && fabs(oi) <= 0x1p100) {
        i = (float)inv;
        const float c = (float)(-oi);
        // |1/d| * (scene_scale + |o|) = |1/d| * scene_scale + |o/d|, from the rounded values at hand
        const float delta = 0x1p-22f * (fmaf(fabsf(i), scale, fabsf(c)) * 1.000001f) + 1e-30f;
        cn = c - delta; cf = c + delta;
    } else {
        i = positive ? 1.0f : -1.0f;
        cn = -__builtin_huge_valf(); cf = __builtin_huge_valf();
    }
    a = positive ? cn : cf;
    b = positive ? cf : cn;
}
RTX_DEV SlabF32 make_slab_f32(const DScene& s, d3 o, d3 inv)
{
    SlabF32 f;
    const float scale = float_not_below(s.scene_scale * 1.001);
    slab_f32_axis(o.x, inv.x, scale, f.ix, f.ax, f.bx);
    slab_f32_axis(o.y, inv.y, scale, f.iy, f.ay, f.by);
    slab_f32_axis(o.z, inv.z, scale, f.iz, f.az, f.bz);
    return f;
}
// The same for the ray from the same origin in the opposite direction: 1/d and o/d change sign,
// delta does not, and the plane met first becomes the one met last, so every constant is negated
// (each parameter is minus the forward one, rounded the same, and entry and exit swap).
RTX_DEV void reverse_slab_f32(SlabF32& f)
{
    f.ix = -f.ix; f.iy = -f.iy; f.iz = -f.iz;
    f.ax = -f.ax; f.ay = -f.ay; f.az = -f.az;
    f.bx = -f.bx; f.by = -f.by; f.bz = -f.bz;
}
// entry and exit parameter (widened) of the box [lo, hi]
RTX_DEV void slab_f32(const SlabF32& f, float lox, float loy, float loz, float hix, float hiy, float hiz, float& tn, float& tf)
{
    const float x0 = fmaf(lox, f.ix, f.ax), x1 = fmaf(hix, f.ix, f.bx);
    const float y0 = fmaf(loy, f.iy, f.ay), y1 = fmaf(hiy, f.iy, f.by);
    const float z0 = fmaf(loz, f.iz, f.az), z1 = fmaf(hiz, f.iz, f.bz);
    tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
}

// The same test on a child box kept in half precision (BvhNode16): w0 = (min.x, min.y), w1 = (min.z,
// max.x), w2 = (max.y, max.z) as pairs of halves.  A half converts to single precision exactly and the
// multiply-add takes it as it is (v_fma_mix_f32), so the arithmetic and its error bound are those above;
// a plane rounded outwards to half precision lies at most 2^-10 beyond the scene's box, which the factor
// 1.001 on scene_scale in make_slab_f32 covers.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
RTX_DEV void slab_f16(const SlabF32& f, uint32_t w0, uint32_t w1, uint32_t w2, float& tn, float& tf)
{
    const half2_t a = __builtin_bit_cast(half2_t, w0), b = __builtin_bit_cast(half2_t, w1), c = __builtin_bit_cast(half2_t, w2);
    const float x0 = fmaf((float)a.x, f.ix, f.ax), x1 = fmaf((float)b.y, f.ix, f.bx);
    const float y0 = fmaf((float)a.y, f.iy, f.ay), y1 = fmaf((float)c.x, f.iy, f.by);
    const float z0 = fmaf((float)b.x, f.iz, f.az), z1 = fmaf((float)c.y, f.iz, f.bz);
    tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
}

RTX_DEV bool first_active_lane();

// N16: the nodes are read from DScene::bvh_nodes16 (the caller has seen that the scene has them)
template <bool COUNT, bool ANYHIT, bool N16>
RTX_DEV bool trace_lane_bvh(const DScene& s, d3 o, d3 d, double limit, int32_t* stack, int32_t& best_prim,
                            double& best_depth, Work<COUNT>& w)
{
    best_prim = -1;
    best_depth = kInf;
    if (s.bvh_root == kBvhEmpty) return true;
    if (!(s.scene_scale <= 0x1p100)) return false;     // beyond the single-precision box test: the reference-order walk
    uint32_t best_shape = 0;
    bool tie = false;
    d3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    // (recomputed where it is wanted rather than held in registers through the loop)
    auto margin = [&]() {
        double x = o.x;
        asm volatile("" : "+v"(x));
        return 1e-9 * (s.scene_scale + __builtin_fmax(fabs(x), __builtin_fmax(fabs(o.y), fabs(o.z))));
    };
    // A child is skipped when the ray enters its box beyond `bound`: the best depth so far
    // (closest hit) or the light distance (any hit), with a margin.  Directions are unit
    // vectors, so the ray parameter is the distance from the origin up to rounding, every hit
    // point the reference computes lies on the ray and in its shape's box up to rounding, and
    // the margin (1e-7 relative + 1e-9 of the scene) is far above those errors.
    SlabF32 f32 = make_slab_f32(s, o, inv);
    // the bound as the single-precision test uses it: not below the FP64 one
    float bound32 = __builtin_huge_valf();
    if constexpr (ANYHIT) bound32 = float_not_below(limit * (1.0 + 1e-7) + margin());
    int sp = 0;
    int32_t node = s.bvh_root;
    // the second walk (pieces of split triangles, ray reversed: see kBvhEmpty's note) waits at the
    // bottom of the stack, as an entry that is neither a node nor a leaf code, so that the loop keeps
    // its one way out (sp == 0).  (Written as a second exit test on a per-lane flag, the loop was
    // compiled with lanes that had finished early restarted at the entry's value: a memory fault.)
    if (s.bvh_behind_root != kBvhEmpty) { stack[0] = kBvhEmpty; sp = 1; }
    bool done = false;
    while (!done) {
        if (node >= 0) {
            float lmin, lmax, rmin, rmax;
            int32_t left, right;
            const int32_t first_node = __builtin_amdgcn_readfirstlane(node);
            const bool same_node = __all(node == first_node);
            if constexpr (N16) {
                // a 32-byte node: one scalar load, or two 16-byte loads per lane instead of four
                if (same_node) {
                    const u32x8 v = scalar_load_32(s.bvh_nodes16 + first_node);
                    slab_f16(f32, v[0], v[1], v[2], lmin, lmax);
                    slab_f16(f32, v[3], v[4], v[5], rmin, rmax);
                    left = (int32_t)v[6]; right = (int32_t)v[7];
                } else {
                    const uint4* n = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(s.bvh_nodes16) + ((uint32_t)node << 5));
                    const uint4 n0 = n[0], n1 = n[1];
                    slab_f16(f32, n0.x, n0.y, n0.z, lmin, lmax);
                    slab_f16(f32, n0.w, n1.x, n1.y, rmin, rmax);
                    left = (int32_t)n1.z; right = (int32_t)n1.w;
                }
            } else if (same_node) {
                // every lane of the wave that stands at a node stands at this one: the node comes
                // through the scalar cache in one load, not through the vector memory pipeline, and
                // the slab arithmetic takes its boxes straight from the scalar registers
                const u32x16 v = scalar_load_64(s.bvh_nodes + first_node);
                slab_f32(f32, __uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                         __uint_as_float(v[3]), __uint_as_float(v[4]), __uint_as_float(v[5]), lmin, lmax);
                slab_f32(f32, __uint_as_float(v[6]), __uint_as_float(v[7]), __uint_as_float(v[8]),
                         __uint_as_float(v[9]), __uint_as_float(v[10]), __uint_as_float(v[11]), rmin, rmax);
                left = (int32_t)v[12]; right = (int32_t)v[13];
            } else {
                // one node = four 16-byte loads (at a 32-bit byte offset from the array's base, which stays in
                // scalar registers: the node array is below 2^30 bytes, kBvhMaxShapes)
                const float4* n = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.bvh_nodes) + ((uint32_t)node << 6));
                const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
                slab_f32(f32, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, lmin, lmax);
                slab_f32(f32, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, rmin, rmax);
                left = __float_as_int(n3.x); right = __float_as_int(n3.y);
            }
            if constexpr (COUNT) { w.content += 2; if (first_active_lane()) w.wave_node++; }
            bool hl = lmin <= lmax && lmax >= 0.0f && !(lmin > bound32);
            bool hr = rmin <= rmax && rmax >= 0.0f && !(rmin > bound32);
            // (written without branches except for the deferred child's store)
            const bool both = hl && hr;
            const bool left_first = lmin <= rmin;
            // (lane masks combined with scalar instructions, no selects)
            const bool go_left = (both && left_first) || (!both && hl);
            if (both) {
                stack[sp * kTraceBlock] = left_first ? right : left;
                ++sp;
            }
            // (The compiler turns this `continue` into an inner loop of node steps that a lane leaves at a
            // leaf or at a dead end, and that runs until every lane has left it; leaf code and stack pops
            // follow for all lanes together.  Measured against it on the 16-spp frames, all slower: dead
            // ends popping inside that inner loop, so that lanes stop for one another at leaves only
            // (pathTrace frame 17.2 -> 20.4 ms); one node step per turn of a single loop (26.4 ms); runs
            // of at most 2 / 4 / 8 node steps (24.4 / 22.4 / 20.6 ms).)
            if (hl || hr) {
                node = go_left ? left : right;
                continue;
            }
        } else {
            const uint32_t code = (uint32_t)~node;
            // (records at 32-bit byte offsets from the array's base: at most 2^24 of 80 bytes each)
            const BvhPrim* q = reinterpret_cast<const BvhPrim*>(reinterpret_cast<const char*>(s.bvh_prims) + (code >> 3) * (uint32_t)sizeof(BvhPrim));
            const uint32_t count = (code & 7u) + 1u;
            // all lanes that stand at a leaf stand at this one: its records come through the scalar cache
            const int32_t first_leaf = __builtin_amdgcn_readfirstlane(node);
            const bool same_leaf = __all(node == first_leaf);
            const BvhPrim* uq = s.bvh_prims + ((uint32_t)~first_leaf >> 3);
            if constexpr (COUNT) {
                uint32_t tests = 0;
                for (uint32_t c = 1; c <= 8u; ++c) tests += c * (uint32_t)__popcll(__ballot(count == c));
                if (first_active_lane()) w.wave_trip_packed += (tests + 63u) / 64u;
            }
            for (uint32_t k = 0; k < count; ++k, ++q, ++uq) {
                if constexpr (COUNT) { if (first_active_lane()) w.wave_trip++; }
                // Shape::rayAABB && rayHit (KDTree.cpp:208-211) commute: rayHit first, and the
                // shape's box is only fetched for a hit that could change the result
                // the whole record in one go: ids and positions arrive together instead of the
                // positions waiting for the type to be known
                d3 ta, tb, tc;
                uint32_t shape, info;
                if (same_leaf) {
                    u32x16 a; u32x4 b;
                    scalar_load_64_16(uq, a, b);
                    ta = mk(dbl(a[0], a[1]), dbl(a[2], a[3]), dbl(a[4], a[5]));
                    tb = mk(dbl(a[6], a[7]), dbl(a[8], a[9]), dbl(a[10], a[11]));
                    tc = mk(dbl(a[12], a[13]), dbl(a[14], a[15]), dbl(b[0], b[1]));
                    shape = b[2]; info = b[3];
                } else {
                    ta = ld3(q->p); tb = ld3(q->p + 3); tc = ld3(q->p + 6);
                    shape = q->shape; info = q->info;
                }
                uint32_t type = info & 3u, idx = info >> 2;
                double depth = kInf;
                uint32_t rp = shape;
                bool ok, behind = false;
                if (type == RTX_PRIM_TRIANGLE) {
                    if constexpr (COUNT) w.tri++;
                    ok = triangle_depth_sided(ta, tb, tc, o, d, depth, behind);
                } else if (type == RTX_PRIM_SPHERE) {
                    if constexpr (COUNT) w.sph++;
                    ok = sphere_t(ld3(s.sph_center + 3 * (size_t)idx), s.sph_r2[idx], o, d, depth);
                } else {
                    if constexpr (COUNT) w.sph += 2;
                    ok = biconvex_depth(s, idx, o, d, depth, rp);
                }
                if (!ok) continue;
                if (ANYHIT ? !(depth <= limit) : !(depth <= best_depth)) continue;
                double tmin, tmax;
                const Box6 sb = load_prim_box(s.prim_box, shape, same_leaf, same_leaf ? (uint32_t)__builtin_amdgcn_readfirstlane((int)shape) : 0u);
                slab_finite(sb.mn, sb.mx, o, inv, tmin, tmax);
                if constexpr (COUNT) w.aabb++;
                if (!(tmin <= tmax && tmax >= 0.0)) continue;
                if constexpr (ANYHIT) {
                    best_depth = depth; best_prim = (int32_t)rp; best_shape = shape | (behind ? kCandBehind : 0u); done = true; break;
                } else if (depth < best_depth) {
                    if constexpr (COUNT) w.accepted++;
                    best_depth = depth; best_prim = (int32_t)rp; best_shape = shape | (behind ? kCandBehind : 0u); tie = false;
                    bound32 = float_not_below(best_depth * (1.0 + 1e-7) + margin());
                } else if ((int32_t)rp != best_prim) {
                    tie = true;
                }
            }
        }
        if (done || sp == 0) break;
        --sp;
        node = stack[sp * kTraceBlock];
        if (node == kBvhEmpty) {
            // the first walk is over: now the tree of the pieces, boxes tested against the reversed ray
            // (shapes against the ray itself)
            reverse_slab_f32(f32);
            node = s.bvh_behind_root;
        }
    }
    if (best_prim < 0) return true;                 // no geometric candidate at all: the reference finds none either
    if (tie) return false;
    return certify_winner<COUNT>(s, best_shape, o, d, inv, best_depth, margin(), w);
}

RTX_DEV bool culling_is_exact(d3 o, d3 d)
{
    d3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    return isfinite(inv.x) && isfinite(inv.y) && isfinite(inv.z) && isfinite(o.x) && isfinite(o.y) && isfinite(o.z);
}

RTX_DEV bool first_active_lane()
{
    const unsigned long long m = __ballot(1);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u;
}

RTX_DEV void add_counters(Counters g, const Work<true>& w)
{
    if (!g.c) return;
    if (w.wave_node) atomicAdd(g.c + 13, (unsigned long long)w.wave_node);
    if (w.wave_trip) atomicAdd(g.c + 14, (unsigned long long)w.wave_trip);
    if (w.wave_trip_packed) atomicAdd(g.c + 15, (unsigned long long)w.wave_trip_packed);
    atomicAdd(g.c + 0, (unsigned long long)w.branch); atomicAdd(g.c + 1, (unsigned long long)w.leaf);
    atomicAdd(g.c + 2, (unsigned long long)w.aabb); atomicAdd(g.c + 3, (unsigned long long)w.tri);
    atomicAdd(g.c + 4, (unsigned long long)w.sph); atomicAdd(g.c + 5, (unsigned long long)w.accepted);
    atomicAdd(g.c + 6, (unsigned long long)w.content);
    atomicAdd(g.c + 7, (unsigned long long)w.fallback);
}
RTX_DEV void add_counters(Counters, const Work<false>&) {}

// SIMD efficiency of a walk that keeps one ray per lane for the life of the wave: the wave
// runs as long as its busiest lane.  Steps = node visits + shape tests; c[10] sums 64 * (the
// wave's largest lane count), c[11] all lanes' counts.
RTX_DEV void add_static_efficiency(Counters g, const Work<true>& w)
{
    if (!g.c) return;
    uint32_t steps = w.content / 2u + w.branch + w.tri + w.sph;
    uint32_t mx = steps;
    const unsigned long long active = __ballot(1);      // the last wave of a launch may be partial
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t other = (uint32_t)__shfl_xor((int)mx, off, 64);
        if ((active >> (lane ^ (uint32_t)off)) & 1ull) mx = max(mx, other);
    }
    if (lane == 0) atomicAdd(g.c + 10, 64ull * mx);
    atomicAdd(g.c + 11, (unsigned long long)steps);
}
RTX_DEV void add_static_efficiency(Counters, const Work<false>&) {}

// ---- pixel <-> slot mapping ------------------------------------------------------------------
// Pixel index t in [0, tile_w*tile_h) walks the tile in bands of 8 rows and, inside a band,
// in blocks of 8 columns (the last band / block may be narrower), so that the 64 lanes of a
// wave are neighbouring pixels.  Every t is a pixel: nothing needs compacting away.
RTX_DEV void pixel_of(const RenderParams& p, uint32_t t, uint32_t& x, uint32_t& y)
{
    const uint32_t W = p.d.tile_w, H = p.d.tile_h;
    const uint32_t own_band = fast_div(t, p.div_band);  // among the chunk's own bands (all full but possibly the tile's last)
    uint32_t r = t - own_band * 8u * W;
    const uint32_t band = own_band * p.band_stride + p.band_first;
    uint32_t hb = min(8u, H - 8u * band);
    // (blocks are 8 x 8 except in the tile's last band and last column: shifts, not divisions, for the rest)
    uint32_t bx = hb == 8u ? r >> 6 : r / (8u * hb);
    uint32_t r2 = r - bx * 8u * hb;
    uint32_t wb = min(8u, W - 8u * bx);
    uint32_t ly = wb == 8u ? r2 >> 3 : r2 / wb, lx = r2 - ly * wb;
    x = p.d.tile_x0 + bx * 8u + lx;
    const uint32_t row = band * 8u + ly;            // row of the tile
    uint32_t gy = row;
    if (p.d.tile_row_group) {
        const uint32_t group = fast_div(row, p.div_group);
        gy = group * p.d.tile_row_stride + (row - group * p.d.tile_row_group);
    }
    y = p.d.tile_y0 + gy;
}
// the tile row of frame row y (the inverse of the mapping above)
RTX_DEV uint32_t tile_row_of(const RenderParams& p, uint32_t y)
{
    const uint32_t dy = y - p.d.tile_y0;
    if (!p.d.tile_row_group) return dy;
    const uint32_t group = fast_div(dy, p.div_stride);
    return group * p.d.tile_row_group + (dy - group * p.d.tile_row_stride);
}

// The camera ray of sample slot `slot` of the chunk (Raytracer.cpp:676-700).
RTX_DEV Ray camera_ray(const RenderParams& p, uint32_t slot)
{
    uint32_t nn = p.n * p.n;
    const uint32_t pixel = fast_div(slot, p.div_nn), sub = slot - pixel * nn;
    uint32_t t = p.chunk_t0 + pixel;
    uint32_t x, y;
    pixel_of(p, t, x, y);
    const uint32_t sub_x = fast_div(sub, p.div_n);
    uint32_t sx = sub_x + 1, sy = sub - sub_x * p.n + 1;
    const rtx_camera& c = p.d.camera;
    double nx = ((double)x / (double)p.d.width) - 0.5;
    double ny = ((double)y / (double)p.d.height) - 0.5;
    d3 eye = ld3(c.eye);
    d3 on_screen = (((ld3(c.right) * nx) + (ld3(c.up) * ny)) + eye) + (ld3(c.forward) * c.focal);
    double offx = (double)sx * p.sub_step_x;
    double offy = (double)sy * p.sub_step_y;
    d3 v = normalize3((on_screen + mk(offx, offy, 0.0)) - eye);
    return make_ray(eye, v);
}

// The queue of level 0 holds nothing that cannot be recomputed: ray i is sample slot i, its
// weight is 1, its random stream starts at (branch 0, counter 0) and origin and direction are
// the camera's.  With p.camera_queue set the kernels of level 0 therefore compute those
// instead of reading them, and nothing writes them: 84 bytes per camera ray that are neither
// stored by a generate kernel nor fetched by the traversal and by each shading kernel.
RTX_DEV Ray queue_ray(const RenderParams& p, const RayQueue& q, uint32_t i)
{
    if (p.camera_queue) return camera_ray(p, i);
    Ray r;
    r.o = mk(q.ox[i], q.oy[i], q.oz[i]);
    r.d = mk(q.dx[i], q.dy[i], q.dz[i]);
    return r;
}
RTX_DEV uint32_t queue_slot(const RenderParams& p, const RayQueue& q, uint32_t i) { return p.camera_queue ? i : q.slot[i]; }
RTX_DEV d3 queue_weight(const RenderParams& p, const RayQueue& q, uint32_t i)
{
    return p.camera_queue ? mk(1.0, 1.0, 1.0) : mk(q.wx[i], q.wy[i], q.wz[i]);
}

// Which ray a thread (or a refilled lane) works on.  Flat queues (seg == 0): ray i of the first
// min(n, *n_device) entries, origin and direction at i.  Light-ray queues (seg != 0, n = samples
// * seg): the level's *n_device hits times n / seg samples.  They are enumerated in tiles of
// kRayTile hits, sample-major within a tile -- ray i is hit h = t * kRayTile + j, sample m, with
// i = (t * samples + m) * kRayTile + j -- so that a wavefront's lanes are neighbouring hits of one
// sample (neighbouring entries m * seg + h, origins at h) while the samples of a tile are walked
// soon after one another, when the parts of the BVH they need are still in cache.  The last tile
// may be partial: ray_place() says whether i names a ray at all.
constexpr uint32_t kRayTile = 256;
RTX_DEV uint32_t ray_total(uint32_t n, const uint32_t* __restrict__ n_device, uint32_t seg)
{
    if (seg == 0) return n_device ? min(n, *n_device) : n;
    const uint32_t tiles = (*n_device + kRayTile - 1) / kRayTile;
    return tiles * kRayTile * (n / seg);        // <= n + kRayTile * samples: the host keeps that below 2^32
}
RTX_DEV bool ray_place(uint32_t i, uint32_t n, const uint32_t* __restrict__ n_device, uint32_t seg, uint32_t& entry,
                       uint32_t& origin)
{
    if (seg == 0) { entry = origin = i; return true; }
    const uint32_t samples = n / seg;
    const uint32_t group = i / kRayTile, j = i % kRayTile;
    const uint32_t t = group / samples, m = group - t * samples;
    origin = t * kRayTile + j;
    entry = m * seg + origin;
    return origin < *n_device;
}

// shootRay for caller-supplied rays (rtx_trace_closest).  MODE 0 walks in the reference's
// order and reports leaf sequences; MODE 1 (culled KD walk) and 3 (BVH) are the render
// path's walks, for which the leaf outputs describe nothing the reference does.
template <int MODE>
__global__ void __launch_bounds__(kTraceBlock)
k_trace_api(DScene s, const double* __restrict__ origin, const double* __restrict__ direction, uint32_t n,
            uint32_t linear, int32_t* prim, double* depth, double* position, double* normal, double* uv, double* bary,
            uint32_t* leaf_count, unsigned long long* leaf_hash, uint32_t* leaf_seq, uint32_t leaf_seq_cap,
            uint32_t* counters)
{
    __shared__ int32_t stack[kStack * kTraceBlock];
    uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x;
    if (i >= n) return;
    Ray r = make_ray(ld3(origin + 3 * (size_t)i), ld3(direction + 3 * (size_t)i));
    Work<true> w;
    int32_t bp = -1; double bd = kInf;
    bool settled = false;
    if (MODE != 0 && linear == 0 && culling_is_exact(r.o, r.d)) {
        if (MODE == 1) { trace_lane_culled<true, false>(s, r.o, r.d, kInf, stack + threadIdx.x, bp, bd, w); settled = true; }
        else settled = s.bvh_nodes16 ? trace_lane_bvh<true, false, true>(s, r.o, r.d, kInf, stack + threadIdx.x, bp, bd, w)
                                     : trace_lane_bvh<true, false, false>(s, r.o, r.d, kInf, stack + threadIdx.x, bp, bd, w);
    }
    if (!settled)
        trace_lane<true>(s, r.o, r.d, linear != 0, stack + threadIdx.x, bp, bd, w,
                         leaf_seq ? leaf_seq + (size_t)i * leaf_seq_cap : nullptr, leaf_seq_cap);
    prim[i] = bp;
    depth[i] = bd;
    Hit h;
    h.position = mk(0, 0, 0); h.normal = mk(0, 0, 0); h.bary = mk(0, 0, 0); h.u = 0; h.v = 0;
    if (bp >= 0) prim_full(s, (uint32_t)bp, r.o, r.d, h);
    if (position) { position[3 * (size_t)i] = h.position.x; position[3 * (size_t)i + 1] = h.position.y; position[3 * (size_t)i + 2] = h.position.z; }
    if (normal) { normal[3 * (size_t)i] = h.normal.x; normal[3 * (size_t)i + 1] = h.normal.y; normal[3 * (size_t)i + 2] = h.normal.z; }
    if (uv) { uv[2 * (size_t)i] = h.u; uv[2 * (size_t)i + 1] = h.v; }
    if (bary) { bary[3 * (size_t)i] = h.bary.x; bary[3 * (size_t)i + 1] = h.bary.y; bary[3 * (size_t)i + 2] = h.bary.z; }
    if (leaf_count) leaf_count[i] = w.leaf_count;
    if (leaf_hash) leaf_hash[i] = w.leaf_hash;
    if (counters) {
        uint32_t* c = counters + 6 * (size_t)i;
        // (for the render walks the first counter includes the BVH child-box tests: their node work)
        c[0] = w.branch + (MODE != 0 ? w.content : 0u); c[1] = w.leaf; c[2] = w.aabb; c[3] = w.tri; c[4] = w.sph; c[5] = w.accepted;
    }
}

// shootRay for a queue: reads (o, d), writes (hit_prim, hit_depth).
// MODE 0: the reference's walk order; 1 / 2: culled KD walk, closest hit / any hit with
// depth <= limit[i] (depth = +inf when there is none); 3 / 4: the same through the BVH.
// `perm` (flat queues only, optional) lists the queue entries in the order they are to be
// walked: thread j takes ray perm[j] (rays binned by direction, see k_tile_count).
// Origin, direction and (any-hit walks) light distance of ray i of a pool launch: stored in the
// queue, or the camera's for a level-0 queue that is not stored, or the one light's for a
// shadow queue that holds origins only.
RTX_DEV void pool_ray(const DScene& s, const CameraQueue& cam, uint32_t light_rays_from_origin,
                      const double* __restrict__ ox, const double* __restrict__ oy, const double* __restrict__ oz,
                      const double* __restrict__ dx, const double* __restrict__ dy, const double* __restrict__ dz,
                      const double* __restrict__ limit, uint32_t i, uint32_t at, d3& o, d3& d, double& lim)
{
    if (cam.on) {
        const Ray r = camera_ray(cam.p, i);
        o = r.o; d = r.d;
    } else if (light_rays_from_origin) {
        // Light::getRay / getDistance (PointLight.cpp:15-27, DirectionalLight.cpp:4-31) of the one light
        o = mk(ox[at], oy[at], oz[at]);
        const Ray lr = light_ray(s.lights[0], o, 0.0, 0.0);
        d = lr.d;
        lim = light_distance(s.lights[0], lr);
    } else {
        o = mk(ox[at], oy[at], oz[at]); d = mk(dx[i], dy[i], dz[i]);
        if (limit) lim = limit[i];
    }
}

template <bool COUNT, int MODE>
__global__ void __launch_bounds__(kTraceBlock) __attribute__((amdgpu_waves_per_eu(5, 5)))
k_trace_pool(DScene s, const double* __restrict__ ox, const double* __restrict__ oy, const double* __restrict__ oz,
             const double* __restrict__ dx, const double* __restrict__ dy, const double* __restrict__ dz,
             const double* __restrict__ limit, uint32_t n, const uint32_t* __restrict__ n_device, uint32_t seg,
             uint32_t linear, int32_t* hit_prim, double* hit_depth, Counters g, CameraQueue cam, uint32_t light_rays_from_origin,
             const uint32_t* __restrict__ perm)
{
    __shared__ int32_t stack[kStack * kTraceBlock];
    const uint32_t ray = blockIdx.x * kTraceBlock + threadIdx.x;
    // Level 0 of a chunk is the chunk's first launch and clears the ring of queue lengths: before
    // any thread leaves, since a chunk may hold fewer rays than the ring has words.
    if (cam.on && ray < 8u) cam.counts[ray] = 0u;
    // the grid is sized for n; when the real queue length is only known on the device
    // (n_device), the surplus workgroups leave here
    if (ray >= ray_total(n, n_device, seg)) return;
    uint32_t i, at;
    if (!ray_place(ray, n, n_device, seg, i, at)) return;
    if (perm) i = at = perm[ray];
    d3 o, d;
    double lim = kInf;          // any-hit walks: the light distance
    if (cam.on) {
        // ray i is the camera ray of sample slot i; the slot's accumulator is part of the chunk's
        // state and cleared here as well
        if (cam.clear_samples) {
            double* acc = cam.sample + 3 * (size_t)i;
            acc[0] = 0.0; acc[1] = 0.0; acc[2] = 0.0;
        }
    }
    pool_ray(s, cam, light_rays_from_origin, ox, oy, oz, dx, dy, dz, (MODE == 2 || MODE == 4) ? limit : nullptr, i, at, o, d, lim);
    Work<COUNT> w;
    int32_t bp = -1; double bd = kInf;
    bool settled = false;
    if (MODE != 0 && linear == 0 && culling_is_exact(o, d)) {
        if (MODE == 1) { trace_lane_culled<COUNT, false>(s, o, d, kInf, stack + threadIdx.x, bp, bd, w); settled = true; }
        else if (MODE == 2) { trace_lane_culled<COUNT, true>(s, o, d, lim, stack + threadIdx.x, bp, bd, w); settled = true; }
        else if (MODE == 3) {
            settled = s.bvh_nodes16 ? trace_lane_bvh<COUNT, false, true>(s, o, d, kInf, stack + threadIdx.x, bp, bd, w)
                                    : trace_lane_bvh<COUNT, false, false>(s, o, d, kInf, stack + threadIdx.x, bp, bd, w);
        } else {
            settled = s.bvh_nodes16 ? trace_lane_bvh<COUNT, true, true>(s, o, d, lim, stack + threadIdx.x, bp, bd, w)
                                    : trace_lane_bvh<COUNT, true, false>(s, o, d, lim, stack + threadIdx.x, bp, bd, w);
        }
    }
    if (!settled) {
        if constexpr (COUNT) w.fallback++;
        trace_lane<COUNT>(s, o, d, linear != 0, stack + threadIdx.x, bp, bd, w, nullptr, 0);
    }
    if (hit_prim) hit_prim[i] = bp;
    if (hit_depth) hit_depth[i] = bd;
    add_counters(g, w);
    add_static_efficiency(g, w);
}

// ---- the BVH walk with lanes refilled from the queue ---------------------------------------------
// Rays of one wave need very different numbers of steps (bounce rays: a third of the lanes
// busy on average with one ray per lane), and a wave runs as long as its busiest lane.  Here a
// wave owns `rays_per_wave` consecutive rays of the queue: a lane whose ray is finished parks
// it, and once `refill_at` lanes of the wave are parked they all store their candidates
// together and take the next rays of the wave's range.  Storing and fetching in groups keeps
// those sections, which run with only the parked lanes active, a small part of the
// instruction stream.  Ranges are handed out by the hardware's own workgroup dispatch, which
// balances the load; a shared ray counter was tried and is slower (an atomic on one address
// completes at under 10^8 a second across the eight XCDs, below the rate at which waves ask).
// The membership verification is a chain of dependent loads; inside this loop it would stall
// the whole wave once per group, so it is left to k_verify, which runs afterwards over the
// queue with every lane busy.  `cand` tells it what to do per ray:
// (the candidate codes are defined above trace_lane_bvh)
// The walk and its culls are those of trace_lane_bvh; with k_verify the results are identical.
template <bool COUNT, bool ANYHIT>
__global__ void __launch_bounds__(kTraceBlock)
k_trace_refill(DScene s, const double* __restrict__ ox, const double* __restrict__ oy, const double* __restrict__ oz,
               const double* __restrict__ dx, const double* __restrict__ dy, const double* __restrict__ dz,
               const double* __restrict__ limit, uint32_t n, const uint32_t* __restrict__ n_device, uint32_t seg,
               uint32_t refill_at, uint32_t rays_per_wave, int32_t* hit_prim, double* hit_depth, uint32_t* cand, Counters g)
{
    __shared__ int32_t stack_mem[kStack * kTraceBlock];
    int32_t* const stack = stack_mem + threadIdx.x;
    const uint32_t lane = threadIdx.x;              // one wave per workgroup
    const uint32_t n_entries = n;
    n = ray_total(n, n_device, seg);
    if ((unsigned long long)blockIdx.x * rays_per_wave >= n) return;
    bool have = false;          // this lane is walking a ray
    bool pending = false;       // this lane holds a ray whose result is not stored yet
    bool exact = true;          // the ray may take the culled walk (finite origin and 1/direction)
    // the unassigned rest of this wave's range (wave-uniform)
    const unsigned long long first = (unsigned long long)blockIdx.x * rays_per_wave;
    uint32_t cur = (uint32_t)min(first, (unsigned long long)n);
    const uint32_t end = (uint32_t)min(first + rays_per_wave, (unsigned long long)n);
    bool exhausted = cur == end;
    uint32_t ray = 0;
    d3 o = mk(0, 0, 0), d = mk(0, 0, 1), inv = mk(0, 0, 0);
    double lim = kInf, abs_margin = 0.0, best_depth = kInf;
    SlabF32 f32 = {};
    float bound32 = __builtin_huge_valf();
    int32_t best_prim = -1, node = 0;
    (void)lim;
    uint32_t best_shape = 0;
    bool tie = false;
    int sp = 0;
    uint32_t leaf_k = 0;        // next shape of the leaf being scanned
    Work<COUNT> w;
    unsigned long long wave_iters = 0, lane_iters = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!have);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle == 64u || (!exhausted && n_idle >= refill_at)) {
            if (pending && !have) {
                if (hit_prim) hit_prim[ray] = best_prim;
                hit_depth[ray] = best_depth;
                cand[ray] = !exact ? kCandReference : (best_prim < 0 ? kCandSettled : (best_shape | (tie ? kCandTie : 0u)));
                pending = false;
            }
            if (exhausted) break;                   // here every lane is parked and has retired
            if (!have) {
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t idx = cur + below;
                uint32_t at = 0;
                if (idx < end && ray_place(idx, n_entries, n_device, seg, ray, at)) {
                    o = mk(ox[at], oy[at], oz[at]); d = mk(dx[ray], dy[ray], dz[ray]);
                    if constexpr (ANYHIT) lim = limit[ray];
                    inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                    exact = isfinite(inv.x) && isfinite(inv.y) && isfinite(inv.z) && isfinite(o.x) && isfinite(o.y) && isfinite(o.z) &&
                            s.scene_scale <= 0x1p100;
                    abs_margin = 1e-9 * (s.scene_scale + __builtin_fmax(fabs(o.x), __builtin_fmax(fabs(o.y), fabs(o.z))));
                    f32 = make_slab_f32(s, o, inv);
                    bound32 = ANYHIT ? float_not_below(lim * (1.0 + 1e-7) + abs_margin) : __builtin_huge_valf();
                    best_prim = -1; best_depth = kInf; best_shape = 0; tie = false;
                    sp = 0; node = s.bvh_root; leaf_k = 0;
                    if (s.bvh_behind_root != kBvhEmpty) { stack[0] = kBvhEmpty; sp = 1; }      // (as in trace_lane_bvh)
                    pending = true;
                    have = exact && s.bvh_root != kBvhEmpty;
                }
            }
            cur = min(end, cur + n_idle);
            exhausted = cur == end;
            continue;
        }
        if constexpr (COUNT) { wave_iters++; lane_iters += have ? 1u : 0u; }
        if (have) {
            bool pop = true;
            if (node >= 0) {
                float lmin, lmax, rmin, rmax;
                int32_t left, right;
                if (s.bvh_nodes16) {
                    // (the node format trace_lane_bvh uses for this scene: the same boxes, the same walk)
                    const uint4* nd = reinterpret_cast<const uint4*>(s.bvh_nodes16 + node);
                    const uint4 n0 = nd[0], n1 = nd[1];
                    slab_f16(f32, n0.x, n0.y, n0.z, lmin, lmax);
                    slab_f16(f32, n0.w, n1.x, n1.y, rmin, rmax);
                    left = (int32_t)n1.z; right = (int32_t)n1.w;
                } else {
                    // one node = four 16-byte loads
                    const float4* nd = reinterpret_cast<const float4*>(s.bvh_nodes + node);
                    const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
                    slab_f32(f32, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, lmin, lmax);
                    slab_f32(f32, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, rmin, rmax);
                    left = __float_as_int(n3.x); right = __float_as_int(n3.y);
                }
                if constexpr (COUNT) w.content += 2;
                const bool hl = lmin <= lmax && lmax >= 0.0f && !(lmin > bound32);
                const bool hr = rmin <= rmax && rmax >= 0.0f && !(rmin > bound32);
                if (hl && hr) {
                    const bool left_first = lmin <= rmin;
                    stack[sp * kTraceBlock] = left_first ? right : left;
                    ++sp;
                    node = left_first ? left : right;
                    pop = false;
                } else if (hl) { node = left; pop = false; }
                else if (hr) { node = right; pop = false; }
            } else {
                // one shape of the leaf per iteration: lanes at nodes and lanes in leaves share
                // every iteration, so a leaf must not hold the others up for its whole run
                const uint32_t code = (uint32_t)~node;
                const BvhPrim* q = s.bvh_prims + (code >> 3) + leaf_k;
                const uint32_t count = (code & 7u) + 1u;
                ++leaf_k;
                pop = leaf_k >= count;
                do {
                    // Shape::rayAABB && rayHit (KDTree.cpp:208-211) commute: rayHit first, and the
                    // shape's box is only fetched for a hit that could change the result
                    const d3 ta = ld3(q->p), tb = ld3(q->p + 3), tc = ld3(q->p + 6);
                    const uint32_t shape = q->shape, info = q->info;
                    const uint32_t type = info & 3u, idx = info >> 2;
                    double depth = kInf;
                    uint32_t rp = shape;
                    bool hit, behind = false;
                    if (type == RTX_PRIM_TRIANGLE) {
                        if constexpr (COUNT) w.tri++;
                        hit = triangle_depth_sided(ta, tb, tc, o, d, depth, behind);
                    } else if (type == RTX_PRIM_SPHERE) {
                        if constexpr (COUNT) w.sph++;
                        hit = sphere_t(ld3(s.sph_center + 3 * (size_t)idx), s.sph_r2[idx], o, d, depth);
                    } else {
                        if constexpr (COUNT) w.sph += 2;
                        hit = biconvex_depth(s, idx, o, d, depth, rp);
                    }
                    if (!hit) break;
                    if (ANYHIT ? !(depth <= lim) : !(depth <= best_depth)) break;
                    double tmin, tmax;
                    slab_finite(s.prim_box + shape, o, inv, tmin, tmax);
                    if constexpr (COUNT) w.aabb++;
                    if (!(tmin <= tmax && tmax >= 0.0)) break;
                    if constexpr (ANYHIT) {
                        best_depth = depth; best_prim = (int32_t)rp; best_shape = shape | (behind ? kCandBehind : 0u); sp = 0; pop = true;
                    } else if (depth < best_depth) {
                        if constexpr (COUNT) w.accepted++;
                        best_depth = depth; best_prim = (int32_t)rp; best_shape = shape | (behind ? kCandBehind : 0u); tie = false;
                        bound32 = float_not_below(best_depth * (1.0 + 1e-7) + abs_margin);
                    } else if ((int32_t)rp != best_prim) {
                        tie = true;
                    }
                } while (false);
                if (pop) leaf_k = 0;
            }
            if (pop) {
                if (sp == 0) have = false;
                else {
                    --sp;
                    node = stack[sp * kTraceBlock];
                    if (node == kBvhEmpty) { reverse_slab_f32(f32); node = s.bvh_behind_root; }
                }
            }
        }
    }
    add_counters(g, w);
    if constexpr (COUNT) {
        if (g.c) {
            if (lane == 0) atomicAdd(g.c + 10, 64ull * wave_iters);
            atomicAdd(g.c + 11, lane_iters);
        }
    }
}

// Second half of the refilled walk: certify each ray's optimistic answer against the reference's
// KD tree, or take the reference-order walk for it (ties, failed certificates, non-finite rays).
template <bool COUNT>
__global__ void __launch_bounds__(kTraceBlock)
k_verify(DScene s, const double* __restrict__ ox, const double* __restrict__ oy, const double* __restrict__ oz,
         const double* __restrict__ dx, const double* __restrict__ dy, const double* __restrict__ dz,
         uint32_t n, const uint32_t* __restrict__ n_device, uint32_t seg, const uint32_t* __restrict__ cand,
         int32_t* hit_prim, double* hit_depth, Counters g)
{
    __shared__ int32_t stack[kStack * kTraceBlock];
    const uint32_t ray = blockIdx.x * kTraceBlock + threadIdx.x;
    if (ray >= ray_total(n, n_device, seg)) return;
    uint32_t i, at;
    if (!ray_place(ray, n, n_device, seg, i, at)) return;
    const uint32_t c = cand[i];
    if (c == kCandSettled) return;
    const d3 o = mk(ox[at], oy[at], oz[at]), d = mk(dx[i], dy[i], dz[i]);
    Work<COUNT> w;
    bool ok = false;
    if (c != kCandReference && !(c & kCandTie)) {
        const double depth = hit_depth[i];
        const d3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
        const double abs_margin = 1e-9 * (s.scene_scale + __builtin_fmax(fabs(o.x), __builtin_fmax(fabs(o.y), fabs(o.z))));
        ok = certify_winner<COUNT>(s, c, o, d, inv, depth, abs_margin, w);
    }
    if (!ok) {
        if constexpr (COUNT) w.fallback++;
        int32_t bp = -1; double bd = kInf;
        trace_lane<COUNT>(s, o, d, false, stack + threadIdx.x, bp, bd, w, nullptr, 0);
        if (hit_prim) hit_prim[i] = bp;
        hit_depth[i] = bd;
    }
    add_counters(g, w);
}

// Block-wide reservation of queue space: one atomic per workgroup.  Every thread of the
// workgroup must call it.  `lds` is kBlock/64 + 2 words of scratch.
RTX_DEV uint32_t block_reserve(bool pred, uint32_t* counter, uint32_t per_item, uint32_t* lds, uint32_t* block_total = nullptr,
                               uint32_t* block_base = nullptr, uint32_t* rank = nullptr)
{
    unsigned long long mask = __ballot(pred);
    uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    uint32_t wave = threadIdx.x >> 6;
    __syncthreads();                                    // lds may still be read from a previous call
    if ((threadIdx.x & 63u) == 0) lds[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (uint32_t k = 0; k < kBlock / 64; ++k) { uint32_t c = lds[k]; lds[k] = sum; sum += c; }
        lds[kBlock / 64] = sum ? atomicAdd(counter, sum * per_item) : 0u;
        lds[kBlock / 64 + 1] = sum;
    }
    __syncthreads();
    if (block_total) *block_total = lds[kBlock / 64 + 1];
    if (block_base) *block_base = lds[kBlock / 64];
    if (rank) *rank = lds[wave] + prefix;
    return lds[kBlock / 64] + (lds[wave] + prefix) * per_item;
}

RTX_DEV void push_ray(const RayQueue& q, uint32_t j, const Ray& r, d3 w, uint32_t slot, uint32_t branch, uint32_t counter)
{
    q.ox[j] = r.o.x; q.oy[j] = r.o.y; q.oz[j] = r.o.z;
    q.dx[j] = r.d.x; q.dy[j] = r.d.y; q.dz[j] = r.d.z;
    q.wx[j] = w.x; q.wy[j] = w.y; q.wz[j] = w.z;
    q.slot[j] = slot; q.branch[j] = branch; q.counter[j] = counter;
}

// camera rays of one chunk (Raytracer.cpp:676-700); one thread per sample slot.  Also clears
// the chunk's state: its slot of the sample accumulator and the queue-length counters.
__global__ void __launch_bounds__(kBlock)
k_generate(RenderParams p, RayQueue q, double* sample, uint32_t* counts)
{
    uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot < 8u) counts[slot] = 0u;
    uint32_t nn = p.n * p.n;
    uint32_t total = p.chunk_pixels * nn;
    if (slot >= total || slot >= p.queue_cap) return;
    if (!p.direct_store) { sample[3 * (size_t)slot + 0] = 0.0; sample[3 * (size_t)slot + 1] = 0.0; sample[3 * (size_t)slot + 2] = 0.0; }
    // ray i of the first queue is sample slot i
    push_ray(q, slot, camera_ray(p, slot), mk(1.0, 1.0, 1.0), slot, 0u, 0u);
}

RTX_DEV void sample_add(const RenderParams& p, double* sample, uint32_t slot, d3 v)
{
    // A zero term (black clear colour, an opaque surface's transparent part) leaves the sum as
    // it is and is not sent at all.  Every sample slot has one live ray per launch, so a plain
    // read-modify-write is enough -- except with the mirror extension of the Whitted
    // integrator, whose two children per hit can reach a slot in the same launch: atomics then.
    if (v.x == 0.0 && v.y == 0.0 && v.z == 0.0) return;
    double* at = sample + 3 * (size_t)slot;
    if (p.atomic_samples) {
        unsafeAtomicAdd(at + 0, v.x);
        unsafeAtomicAdd(at + 1, v.y);
        unsafeAtomicAdd(at + 2, v.z);
    } else {
        at[0] += v.x; at[1] += v.y; at[2] += v.z;
    }
}

// drawBuffer[x + y*frameX] += colorAcum / double(n*n) (Raytracer.cpp:710) for pixel k of the
// chunk: the quotient is rounded to float and added in FP32 (or stored: 0.0f + it is what
// adding to a cleared buffer gives).
// v / (double)count.  When the count is a power of two its reciprocal is exact and the product
// is the same correctly rounded value as the quotient, for a multiplication in place of an FP64 division
// sequence per component (sample counts are 1, 4, 16, 64 in the reference's configurations); any other
// count is divided by.
RTX_DEV d3 div_by_count(d3 v, uint32_t count)
{
    if ((count & (count - 1u)) == 0u && count != 0u) return v * __hiloint2double((int)((1023u - (uint32_t)__ffs((int)count) + 1u) << 20), 0);
    return v / (double)count;
}

RTX_DEV void resolve_pixel(const RenderParams& p, uint32_t k, d3 acum, float* accum)
{
    uint32_t x, y;
    pixel_of(p, p.chunk_t0 + k, x, y);
    d3 px = div_by_count(acum, p.n * p.n);
    size_t at = 3 * ((size_t)(x - p.d.tile_x0) + (size_t)tile_row_of(p, y) * p.d.tile_w);
    if (p.overwrite) {
        accum[at + 0] = 0.0f + (float)px.x;
        accum[at + 1] = 0.0f + (float)px.y;
        accum[at + 2] = 0.0f + (float)px.z;
    } else {
        accum[at + 0] += (float)px.x;
        accum[at + 1] += (float)px.y;
        accum[at + 2] += (float)px.z;
    }
}

// The whole value of a sample slot (p.direct_store != 0): what the cleared slot would hold after
// its additions, 0.0 + v, is stored -- or, with one sub-sample per pixel, resolved at once
// (colorAcum = 0 + that value, Raytracer.cpp:690-705).
RTX_DEV void sample_store(const RenderParams& p, double* sample, uint32_t slot, d3 v)
{
    v = mk(0.0 + v.x, 0.0 + v.y, 0.0 + v.z);
    if (p.direct_store == 2u) {
        resolve_pixel(p, slot, mk(0.0, 0.0, 0.0) + v, p.accum);
    } else {
        double* at = sample + 3 * (size_t)slot;
        at[0] = v.x; at[1] = v.y; at[2] = v.z;
    }
}

// A term of a slot that gets exactly one term (pathTrace): stored or added, as the mode says.
RTX_DEV void sample_term(const RenderParams& p, double* sample, uint32_t slot, d3 v)
{
    if (p.direct_store) sample_store(p, sample, slot, v);
    else sample_add(p, sample, slot, v);
}

// A term of a slot whose terms all come from this thread (rayTrace without children): kept in
// `sum`, which starts at zero and is stored by the caller, or added to the slot at once.
RTX_DEV void sample_term(const RenderParams& p, double* sample, uint32_t slot, d3 v, d3& sum)
{
    if (p.direct_store) {
        if (!(v.x == 0.0 && v.y == 0.0 && v.z == 0.0)) sum = sum + v;      // as sample_add: zero terms change nothing
    } else {
        sample_add(p, sample, slot, v);
    }
}

RTX_DEV Rng ray_rng(const RenderParams& p, const RayQueue& q, uint32_t i, uint32_t slot)
{
    uint32_t nn = p.n * p.n;
    const uint32_t pixel = fast_div(slot, p.div_nn), sub = slot - pixel * nn;
    uint32_t t = p.chunk_t0 + pixel;
    uint32_t x, y;
    pixel_of(p, t, x, y);
    Rng r;
    r.seed = p.d.seed; r.frame = p.d.frame; r.pixel = x + y * p.d.width;
    r.sample = p.pass * nn + sub;
    r.branch = p.camera_queue ? 0u : q.branch[i];
    r.counter = p.camera_queue ? 0u : q.counter[i];
    return r;
}

// ---- orders of a queue: rays binned by a small key -----------------------------------------------
// A counting sort over kBins keys: count, scan, scatter into a permutation.  Two keys:
//   KEY 0, the material a ray hit (0 = miss, else 1 + material mod 63), for the shading kernels:
//     the branches of pathTrace (refract / mirror / diffuse / emissive, textured or not) are chosen
//     by the material, so a wave whose rays all hit one material runs one of them;
//   KEY 1, the ray's direction (a 4 x 4 grid over the octahedral map of the unit sphere), for the
//     traversal of bounce rays.  The queue of a level below the camera's is in the order of the
//     rays' parents -- neighbouring origins, directions all over the hemisphere -- and a wavefront
//     walks as long as its longest ray while its lanes fetch nodes from all over the tree.  Rays
//     of one direction bin out of neighbouring origins are nearly parallel neighbours: they take
//     similar paths through the BVH (fewer distinct lines per fetch, fewer idle lanes), and the
//     many rays that leave the scene at once (half of all bounce rays off an upward surface) end
//     up in wavefronts of their own.  Measured on 4 M bounce rays off the backpack scene: 2.91 ms
//     in queue order, 2.06 ms binned (1.98 ms fully sorted by direction and origin cell).
constexpr uint32_t kDirSide = 4;             // direction bins per side of the octahedral square
constexpr uint32_t kBins = kDirSide * kDirSide;

RTX_DEV uint32_t shade_bin(const DScene& s, int32_t prim)
{
    return prim < 0 ? 0u : 1u + s.prim_material[prim] % (kBins - 1u);
}

RTX_DEV uint32_t direction_bin(double x, double y, double z)
{
    // octahedral map: project on |x|+|y|+|z| = 1, fold the lower hemisphere over the upper one's
    // corners; single precision is plenty for a bin number
    float fx = (float)x, fy = (float)y, fz = (float)z;
    float a = fabsf(fx) + fabsf(fy) + fabsf(fz);
    float px = fx / a, py = fy / a;
    if (fz < 0.0f) {
        float qx = (1.0f - fabsf(py)) * (px < 0.0f ? -1.0f : 1.0f);
        float qy = (1.0f - fabsf(px)) * (py < 0.0f ? -1.0f : 1.0f);
        px = qx; py = qy;
    }
    const int side = (int)kDirSide;
    int bx = (int)((px * 0.5f + 0.5f) * (float)side), by = (int)((py * 0.5f + 0.5f) * (float)side);
    bx = min(max(bx, 0), side - 1); by = min(max(by, 0), side - 1);       // also what a NaN direction gets: (0, 0)
    return (uint32_t)(by * side + bx);
}

template <int KEY>
RTX_DEV uint32_t bin_key(const DScene& s, const RayQueue& q, uint32_t i)
{
    if (KEY == 0) return shade_bin(s, q.hit_prim[i]);
    return direction_bin(q.dx[i], q.dy[i], q.dz[i]);
}

// The sort is exact across tiles of kTileRays consecutive rays and free within a tile: inside a bin
// the tiles follow each other in queue order.  That matters: with the bins' contents in the order in
// which workgroups happen to reach an atomic counter (a window of half a million rays on this GPU),
// binned bounce rays walk SLOWER than in queue order (3.3 ms against 2.9 ms for the 4 M rays above);
// exact down to 2 048 rays they keep the gain (2.15 ms, 2.06 ms with the whole queue order kept).
constexpr uint32_t kTileRays = 2048;        // rays per workgroup of the three kernels: kBlock x 8

// hist[bin * n_tiles + tile] = rays of the tile in the bin; bins[bin] += the same (bin totals)
template <int KEY>
__global__ void __launch_bounds__(kBlock)
k_tile_count(DScene s, RayQueue q, uint32_t n, const uint32_t* __restrict__ n_device, uint32_t n_tiles, uint32_t* hist, uint32_t* bins)
{
    __shared__ uint32_t h[kBins];
    if (n_device) n = min(n, *n_device);
    if (threadIdx.x < kBins) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * kTileRays;
#pragma unroll
    for (uint32_t j = 0; j < kTileRays / kBlock; ++j) {
        const uint32_t i = first + j * kBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[bin_key<KEY>(s, q, i)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kBins) {
        const uint32_t c = h[threadIdx.x];
        hist[threadIdx.x * n_tiles + blockIdx.x] = c;
        if (c) atomicAdd(bins + threadIdx.x, c);
    }
}

// workgroup b: hist[b * n_tiles + t] becomes the first place of tile t's rays of bin b in the order
// (exclusive prefix sum over the bin's tiles, on top of the totals of the bins before it)
__global__ void __launch_bounds__(kBlock)
k_tile_scan(uint32_t n_tiles, uint32_t* hist, const uint32_t* __restrict__ bins)
{
    __shared__ uint32_t part[kBlock];
    __shared__ uint32_t base;
    const uint32_t bin = blockIdx.x, t = threadIdx.x;
    if (t == 0) {
        uint32_t sum = 0;
        for (uint32_t b = 0; b < bin; ++b) sum += bins[b];
        base = sum;
    }
    uint32_t* row = hist + (size_t)bin * n_tiles;
    const uint32_t per = (n_tiles + kBlock - 1) / kBlock;
    const uint32_t lo = min(t * per, n_tiles), hi = min(lo + per, n_tiles);
    uint32_t mine = 0;
    for (uint32_t k = lo; k < hi; ++k) mine += row[k];
    part[t] = mine;
    __syncthreads();
    if (t == 0) {
        uint32_t sum = base;
        for (uint32_t k = 0; k < kBlock; ++k) { uint32_t c = part[k]; part[k] = sum; sum += c; }
    }
    __syncthreads();
    uint32_t at = part[t];
    for (uint32_t k = lo; k < hi; ++k) { uint32_t c = row[k]; row[k] = at; at += c; }
}

// perm[place] = ray, places of a tile's rays of one bin handed out from hist; the first workgroup
// also clears the bin totals for the next sort
template <int KEY>
__global__ void __launch_bounds__(kBlock)
k_tile_scatter(DScene s, RayQueue q, uint32_t n, const uint32_t* __restrict__ n_device, uint32_t n_tiles,
               const uint32_t* __restrict__ hist, uint32_t* bins, uint32_t* perm)
{
    __shared__ uint32_t next[kBins];
    if (n_device) n = min(n, *n_device);
    if (threadIdx.x < kBins) {
        next[threadIdx.x] = hist[threadIdx.x * n_tiles + blockIdx.x];
        if (blockIdx.x == 0) bins[threadIdx.x] = 0u;
    }
    __syncthreads();
    const uint32_t first = blockIdx.x * kTileRays;
#pragma unroll
    for (uint32_t j = 0; j < kTileRays / kBlock; ++j) {
        const uint32_t i = first + j * kBlock + threadIdx.x;
        if (i < n) perm[atomicAdd(&next[bin_key<KEY>(s, q, i)], 1u)] = i;
    }
}

// ---- pathTrace, one recursion level (Raytracer.cpp:217-355) -------------------------------------
// The reference's result is `weight-product * terminal radiance`: refraction returns the
// child's radiance unchanged (:272) and a bounce returns ctSpecular + kD*diffuse, both
// linear in the child's radiance (:302-348).  Each ray therefore carries the product of
// the per-level factors and the terminal value is added once to its sample.
// `perm` (optional) lists the queue entries in the order they are to be shaded: thread j
// shades ray perm[j].
// (five waves per SIMD: left to itself the register allocation wavers between 96 and 100, i.e. between
// five waves and four, with changes elsewhere in the file; four cost the pathTrace frame 1.5 %)
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5)))
k_shade_path(DScene s, RenderParams p, RayQueue q, uint32_t n, const uint32_t* __restrict__ n_device,
             const uint32_t* __restrict__ perm, RayQueue next, uint32_t* next_count, double* sample)
{
    if (n_device) n = min(n, *n_device);        // the grid is sized for an upper bound of the queue length
    const uint32_t j0 = blockIdx.x * kBlock + threadIdx.x;
    const bool live = j0 < n;
    const uint32_t i = live ? (perm ? perm[j0] : j0) : 0u;
    bool spawn = false;
    Ray child; d3 cw = mk(0, 0, 0); uint32_t slot = 0, cbranch = 0, ccounter = 0;
    const uint32_t layer = p.layer;
    if (live) {
        slot = queue_slot(p, q, i);
        d3 w = queue_weight(p, q, i);
        d3 clear = ld3(p.d.clear_color);
        int32_t prim = q.hit_prim[i];
        if (prim < 0) {
            sample_term(p, sample, slot, w * (clear * 0.1));
        } else {
            const Ray qr = queue_ray(p, q, i);
            d3 o = qr.o, d = qr.d;
            d3 inv_dir = -d;
            Hit h;
            const rtx_material& m = s.materials[s.prim_material[prim]];
            prim_full(s, (uint32_t)prim, o, d, h, material_reads_uv(m));
            if (p.d.flags & RTX_FLAG_BASIC) {
                sample_term(p, sample, slot, w * material_color(s, m, h.u, h.v));
            } else {
                double trans = sample_scalar(s, m.transparency_fn, m.transparency, h.u, h.v);
                double smooth = sample_scalar(s, m.smoothness_fn, m.smoothness, h.u, h.v);
                double metal = sample_scalar(s, m.metalness_fn, m.metalness, h.u, h.v);
                Rng rng = ray_rng(p, q, i, slot);
                double transparency_decider = rng_next(rng);
                double reflectance_decider = rng_next(rng);
                (void)rng_next(rng);                                // specularDecider (:241)
                bool entering = is_entering(h.normal, inv_dir);
                d3 emission = ld3(m.emission);
                if (!(emission.x == 0.0 && emission.y == 0.0 && emission.z == 0.0)) {
                    sample_term(p, sample, slot, w * emission);
                } else {
                    double ni = sample_scalar(s, m.ni_fn, m.ni, h.u, h.v);
                    d3 dir, pos;
                    if (transparency_decider < trans) {
                        dir = normalize3(refraction_ray(normalize3(h.normal), normalize3(d), ni, entering));
                        pos = h.position + ((h.normal * (entering ? -1.0 : 1.0)) * p.d.bias);
                        child = make_ray(pos, dir);
                        cw = w;
                    } else {
                        if (reflectance_decider < smooth) {
                            dir = mirror_direction(d, h.normal);
                        } else {
                            // randomHemisphericalVector (:73-80)
                            d3 nn = normalize3(h.normal);
                            d3 out;
                            do {
                                double a = (rng_next(rng) * 2.0) - 1.0;
                                double b = (rng_next(rng) * 2.0) - 1.0;
                                double c = (rng_next(rng) * 2.0) - 1.0;
                                out = normalize3(mk(a, b, c));
                            } while (dot(nn, out) < 0.0);
                            dir = normalize3(out);
                        }
                        pos = h.position + (h.normal * p.d.bias);
                        child = make_ray(pos, dir);
                        double f = fabs((1.0 - ni) / (1.0 + ni));
                        d3 F0a = mk(f, f, f);
                        F0a = F0a * F0a;
                        d3 matC = material_color(s, m, h.u, h.v);
                        d3 F0 = (F0a * (1.0 - metal)) + (matC * metal);
                        d3 kS;
                        d3 one = mk(1.0, 1.0, 1.0);
                        d3 spec = cook_torrance(inv_dir, child.d, h.normal, smooth, one, F0, kS);
                        double diff = dot(dir, h.normal);
                        d3 kD = mk(1.0 - kS.x, 1.0 - kS.y, 1.0 - kS.z) * (1.0 - metal);
                        d3 diffuse = (one * diff) * matC;
                        cw = w * (spec + (kD * diffuse));
                    }
                    cbranch = rng.branch; ccounter = rng.counter;
                    if (layer + 1 > p.d.max_depth) sample_term(p, sample, slot, cw * (clear * 0.1));  // :220-222
                    else spawn = true;
                }
            }
        }
    }
    __shared__ uint32_t reserve_lds[kBlock / 64 + 2];
    uint32_t j = block_reserve(spawn, next_count, 1, reserve_lds);
    if (spawn && j < p.queue_cap) push_ray(next, j, child, cw, slot, cbranch, ccounter);
}

// ---- rayTrace (Raytracer-Noah-Desktop.cpp:452-564), first half: emit the light rays -----------------
// A hit gets the compact index h = hit_index[i] and with it the shadow-queue entries m * seg + h
// (m over the lights' samples) and, with GLOBAL_LIGHTING, the gather-queue entries of the same
// form (m over lights x gather samples).  Random numbers are drawn in the reference's order: per
// light first the SquareLight samples, then the gather directions.  GATHER = false compiles the
// gather out (it costs registers, hence resident waves, even when it does not run).
template <bool GATHER>
__global__ void __launch_bounds__(kBlock)
k_shade_whitted_a(DScene s, RenderParams p, RayQueue q, uint32_t n, const uint32_t* __restrict__ n_device, ShadowQueue sq,
                  uint32_t* hit_count, uint32_t* hit_index, GatherQueue gq, double* sample)
{
    if (n_device) n = min(n, *n_device);
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    bool hit = false;
    if (i < n) {
        int32_t prim = q.hit_prim[i];
        if (prim < 0) {
            sample_term(p, sample, queue_slot(p, q, i), queue_weight(p, q, i) * ld3(p.d.clear_color));
        } else {
            hit = true;
        }
    }
    const bool lit = !(p.d.flags & RTX_FLAG_NO_LIGHTING) && s.n_lights > 0;
    const uint32_t gather_samples = GATHER ? p.gather_samples : 0u;
    const bool gather_traced = GATHER && lit && p.gather_traced && gather_samples > 0;
    bool emit = hit && lit && (p.shadow_per_hit > 0 || gather_samples > 0);
    __shared__ uint32_t reserve_lds[kBlock / 64 + 2];
    const uint32_t h_index = block_reserve(emit, hit_count, 1, reserve_lds);     // < seg: at most one per ray
    if (i < n) hit_index[i] = h_index;
    if (emit) {
        const Ray qr = queue_ray(p, q, i);
        d3 o = qr.o, d = qr.d;
        Hit h;
        prim_full(s, (uint32_t)q.hit_prim[i], o, d, h, false);      // position and normal only
        uint32_t slot = queue_slot(p, q, i);
        Rng rng = ray_rng(p, q, i, slot);
        d3 from = h.position + (h.normal * p.d.bias);
        // Light::getRay starts every light ray of this hit at `from` (Ray.cpp:3-8 keeps the origin)
        sq.ox[h_index] = from.x; sq.oy[h_index] = from.y; sq.oz[h_index] = from.z;
        if constexpr (GATHER) {
            if (gather_traced) { gq.ox[h_index] = h.position.x; gq.oy[h_index] = h.position.y; gq.oz[h_index] = h.position.z; }
        }
        uint32_t k = h_index, g = h_index;
        for (uint32_t li = 0; li < s.n_lights; ++li) {
            const rtx_light& l = s.lights[li];
            uint32_t samples = l.multisample ? p.d.light_multisample_n * p.d.light_multisample_n : 1u;
            for (uint32_t sm = 0; sm < samples; ++sm, k += p.seg) {
                double u0 = 0.0, u1 = 0.0;
                if (l.kind == RTX_LIGHT_SQUARE) { u0 = rng_next(rng); u1 = rng_next(rng); }
                if (p.light_rays_from_origin) continue;     // recomputed from the origin where it is needed
                Ray lr = light_ray(l, from, u0, u1);
                sq.dx[k] = lr.d.x; sq.dy[k] = lr.d.y; sq.dz[k] = lr.d.z;
                sq.limit[k] = light_distance(l, lr);
            }
            if constexpr (GATHER) {
                if (gather_traced) {
                    for (uint32_t gm = 0; gm < gather_samples; ++gm, g += p.seg) {
                        // Ray lightRay(minRayResult.position, randomHemisphericalVector(normal)) (:503-505)
                        Ray gr = make_ray(h.position, random_hemispherical_desktop(h.normal, rng));
                        gq.dx[g] = gr.d.x; gq.dy[g] = gr.d.y; gq.dz[g] = gr.d.z;
                    }
                } else {
                    rng.counter += 2u * gather_samples;     // directions whose nested call ends at the layer limit
                }
            }
        }
        q.counter[i] = rng.counter;       // the refraction child continues this stream
    } else if (hit && p.camera_queue) {
        q.counter[i] = 0u;                // nothing has written this ray's counter yet
    }
}

// second half: direct lighting from the shadow results, the gather's sum, blend, spawn children
template <bool GATHER>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_shade_whitted_b(DScene s, RenderParams p, RayQueue q, uint32_t n, const uint32_t* __restrict__ n_device, ShadowQueue sq,
                  const uint32_t* hit_index, GatherQueue gq, RayQueue next, uint32_t* next_count, double* sample)
{
    if (n_device) n = min(n, *n_device);
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    bool spawn_t = false, spawn_r = false;
    Ray child_t, child_r; d3 wt = mk(0, 0, 0), wr = mk(0, 0, 0);
    double ni = 1.0; uint32_t slot = 0, branch = 0, counter = 0;
    const uint32_t layer = p.layer;
    if (i < n && q.hit_prim[i] >= 0) {
        d3 sum = mk(0.0, 0.0, 0.0);
        slot = queue_slot(p, q, i); branch = p.camera_queue ? 0u : q.branch[i]; counter = q.counter[i];
        d3 w = queue_weight(p, q, i);
        d3 clear = ld3(p.d.clear_color);
        const Ray qr = queue_ray(p, q, i);
        d3 o = qr.o, d = qr.d;
        d3 inv_dir = -d;
        Hit h;
        const rtx_material& m = s.materials[s.prim_material[q.hit_prim[i]]];
        prim_full(s, (uint32_t)q.hit_prim[i], o, d, h, material_reads_uv(m));
        bool entering = is_entering(h.normal, inv_dir);
        d3 local;
        if (p.d.flags & RTX_FLAG_NO_LIGHTING) {
            local = material_color(s, m, h.u, h.v);
        } else {
            d3 direct = mk(0, 0, 0), global = mk(0, 0, 0);
            const d3 mcolor = material_color(s, m, h.u, h.v);
            const double mns = sample_scalar(s, m.ns_fn, m.ns, h.u, h.v);
            const uint32_t h_index = hit_index[i];
            const uint32_t stride = p.seg;
            uint32_t k = h_index, g = h_index;
            for (uint32_t li = 0; li < s.n_lights; ++li) {
                const rtx_light& l = s.lights[li];
                uint32_t samples = l.multisample ? p.d.light_multisample_n * p.d.light_multisample_n : 1u;
                for (uint32_t sm = 0; sm < samples; ++sm, k += stride) {
                    // light->getDistance(lightRay) (:490) was evaluated when the ray was made
                    double ldist;
                    d3 ld;
                    if (p.light_rays_from_origin) {
                        const Ray lr = light_ray(l, h.position + (h.normal * p.d.bias), 0.0, 0.0);
                        ld = lr.d;
                        ldist = light_distance(l, lr);
                    } else {
                        ldist = sq.limit[k];
                    }
                    if (ldist < sq.hit_depth[k]) {
                        if (!p.light_rays_from_origin) ld = mk(sq.dx[k], sq.dy[k], sq.dz[k]);
                        double att = light_attenuation(l, ldist, p.d.attenuation);
                        direct = direct + lighting(inv_dir, ld, h.normal, att, mcolor, mns, ld3(l.color));
                    }
                }
                direct = div_by_count(direct, samples);      // inside the light loop (:498)
                if (GATHER && p.gather_samples) {           // GLOBAL_LIGHTING (:499-524)
                    for (uint32_t gm = 0; gm < p.gather_samples; ++gm, g += stride) {
                        int32_t gp = p.gather_traced ? gq.hit_prim[g] : -1;
                        if (gp >= 0) {
                            // distance(origin, lightRayResult.position): a triangle's depth is that
                            // very distance (Triangle.cpp:52-55), a sphere's position is o + d*t
                            double t = gq.hit_depth[g];
                            d3 gd = mk(gq.dx[g], gq.dy[g], gq.dz[g]);
                            double dist = t;
                            if ((s.prim_info[gp] & 3u) != RTX_PRIM_TRIANGLE) {
                                d3 go = mk(gq.ox[h_index], gq.oy[h_index], gq.oz[h_index]);
                                dist = distance3(go, go + (gd * t));
                            }
                            double att = light_attenuation(l, dist, p.d.attenuation);
                            global = global + lighting(inv_dir, gd, h.normal, att, mcolor, mns, ld3(l.color));
                        } else {
                            global = global + (clear * 0.04);
                        }
                    }
                    global = div_by_count(global, p.d.global_samples);  // :523
                }
            }
            local = direct + global;
        }
        double trans = sample_scalar(s, m.transparency_fn, m.transparency, h.u, h.v);
        ni = sample_scalar(s, m.ni_fn, m.ni, h.u, h.v);
        double wtrans = 1.0 - (1.0 - trans);
        bool refract = fabs(trans - 0.0) >= kEps;           // epsilonNotEqual (:535)
        if (refract) {
            d3 dir = refraction_ray(normalize3(h.normal), normalize3(d), ni, entering);
            d3 pos = h.position + ((h.normal * (entering ? -1.0 : 1.0)) * p.d.bias);
            child_t = make_ray(pos, dir);
            wt = w * wtrans;
            if (layer + 1 > p.d.max_depth) sample_term(p, sample, slot, wt * clear, sum);  // :456-460
            else spawn_t = true;
        } else {
            sample_term(p, sample, slot, w * (clear * wtrans), sum);    // transparentRayResult stays clearColor (:532)
        }
        double wl = 1.0 - trans;
        if (p.d.flags & RTX_FLAG_WHITTED_REFLECT) {
            double smooth = sample_scalar(s, m.smoothness_fn, m.smoothness, h.u, h.v);
            if (smooth != 0.0) {
                child_r = make_ray(h.position + (h.normal * p.d.bias), mirror_direction(d, h.normal));
                wr = w * (smooth * wl);
                if (layer + 1 > p.d.max_depth) sample_term(p, sample, slot, wr * clear, sum);
                else spawn_r = true;
                wl = wl * (1.0 - smooth);
            }
        }
        sample_term(p, sample, slot, w * (local * wl), sum);
        if (p.direct_store) sample_store(p, sample, slot, sum);
    }
    __shared__ uint32_t reserve_lds[kBlock / 64 + 2];
    uint32_t jt = block_reserve(spawn_t, next_count, 1, reserve_lds);
    if (spawn_t && jt < p.queue_cap) push_ray(next, jt, child_t, wt, slot, branch, counter);
    uint32_t jr = block_reserve(spawn_r, next_count, 1, reserve_lds);
    if (spawn_r && jr < p.queue_cap) push_ray(next, jr, child_r, wr, slot, branch | (1u << (layer & 31u)), 0u);
}

// drawBuffer[x + y*frameX] += colorAcum / double(n*n)  (Raytracer.cpp:705-710): the
// sub-samples are summed in the reference's order (sx outer, sy inner), the quotient
// is rounded to float and added in FP32.
__global__ void __launch_bounds__(kBlock)
k_resolve(RenderParams p, const double* __restrict__ sample, float* accum)
{
    uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= p.chunk_pixels) return;
    uint32_t nn = p.n * p.n;
    d3 acum = mk(0, 0, 0);
    for (uint32_t sub = 0; sub < nn; ++sub) {
        const double* sp = sample + 3 * ((size_t)k * nn + sub);
        acum = acum + mk(sp[0], sp[1], sp[2]);
    }
    resolve_pixel(p, k, acum, accum);
}

// Ray totals of one level (one thread), so that the host need not read the per-level counters:
// work[12] += rays spawned for the next level, work[8] += shadow rays, work[9] += gather rays.
// It also clears the counter pair of the level after next, which the next level's kernels fill.
__global__ void k_level_totals(const uint32_t* next_len, const uint32_t* hits, unsigned long long* work,
                               uint32_t shadow_per_hit, uint32_t gather_per_hit, uint32_t* pair_after_next)
{
    work[12] += *next_len;
    work[8] += (unsigned long long)*hits * shadow_per_hit;
    work[9] += (unsigned long long)*hits * gather_per_hit;
    pair_after_next[0] = 0u; pair_after_next[1] = 0u;
}

// ---- present (Raytracer.cpp:720-725, :422, :90-105) ------------------------------------------------
// linear float -> 8-bit sRGB as an sRGB framebuffer stores it: clamp, transfer function, round.
// The conversion is monotone, so it is a search in the 255 thresholds at which the byte steps
// up (srgb_thresholds() on the host evaluates the formula with the C library's powf, the one
// a CPU restatement uses): thresholds[b] is the smallest float that converts to b or more,
// thresholds[0] is unused.  NaN and values <= 0 give 0, values >= 1 give 255.
__global__ void __launch_bounds__(kBlock)
k_present(const float* __restrict__ accum, uint32_t width, uint32_t height, float passes, uint32_t tonemap, uint32_t flip,
          const float* __restrict__ thresholds, uint8_t* __restrict__ rgb8)
{
    __shared__ float t[256];
    t[threadIdx.x] = thresholds[threadIdx.x];           // kBlock is 256
    __syncthreads();
    const uint32_t row_values = width * 3u;
    const uint32_t in_row = blockIdx.x * kBlock + threadIdx.x;     // one thread per channel value
    const uint32_t row = blockIdx.y;
    if (in_row >= row_values) return;
    float v = accum[(size_t)row * row_values + in_row];
    if (passes != 1.0f) v = v / passes;
    if (tonemap == RTX_TONEMAP_CBRT) v = cbrtf(v);
    uint32_t lo = 0u, hi = 255u;
#pragma unroll
    for (int step = 0; step < 8; ++step) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        const bool up = v >= t[mid];
        lo = up ? mid : lo;
        hi = up ? hi : mid - 1u;
    }
    const uint32_t out_row = flip ? height - 1u - row : row;
    rgb8[(size_t)out_row * row_values + in_row] = (uint8_t)lo;
}

}  // namespace rtx
