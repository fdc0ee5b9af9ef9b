/*
 * rt_oracle.c -- CPU restatement (plain C, FP64) of the reference's per-pixel
 * trace loop.  TEST INFRASTRUCTURE ONLY: see rt_oracle.h for who may load it
 * and for the pinning status (PARITY UNPINNED beyond the survey's known-answer
 * values; the reference has no tests and does not build here).
 *
 * Each function cites the reference text it follows (paths relative to
 * Raytracer/ in noahwhygodwhy/Raytracer).  Vector arithmetic follows the
 * published definitions of the reference's third-party dependency glm
 * 0.9.9.8 (packages.config:5), which is absent from this image:
 *   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   cross(x,y)    = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
 *   length(v)     = sqrt(dot(v,v));  distance(a,b) = length(b-a)
 *   normalize(v)  = v * (1/sqrt(dot(v,v)))
 *   min(x,y)      = (y<x)?y:x;  max(x,y) = (x<y)?y:x
 *   mix(x,y,a)    = x*(1-a) + y*a;  mod(x,y) = x - y*floor(x/y)
 *   mat4*vec4     = (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
 *   mat3*vec3     = m0*v0 + m1*v1 + m2*v2 (left to right)
 *   rotate(angle, axis) = the axis-angle matrix of gtc/matrix_transform.inl
 * Build with -ffp-contract=off so no multiply-add is fused.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const double* p) { return V(p[0], p[1], p[2]); }
static inline void st3(double* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 divs(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline double dot(v3 a, v3 b) { double tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
static inline v3 cross(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline double length3(v3 a) { return sqrt(dot(a, a)); }
static inline double distance3(v3 a, v3 b) { return length3(sub(b, a)); }
static inline v3 normalize3(v3 a) { return muls(a, 1.0 / sqrt(dot(a, a))); }
static inline double gmin(double x, double y) { return (y < x) ? y : x; }
static inline double gmax(double x, double y) { return (x < y) ? y : x; }
static inline double comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

static const double PI = 3.14159265358979323846264338327950288;

/* ------------------------------------------------------------------ RNG --- */
/* Philox-4x32-10 (Salmon et al., SC'11), replacing rand() at the reference's
 * call sites Raytracer.cpp:65-71 and SquareLight.cpp:48-51. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uniform in [0,1] inclusive, like rand()/RAND_MAX (Raytracer.cpp:65-67) */
/* frameBuffer[j] = drawBuffer[j] / float(i+1); cbrt per channel (Raytracer.cpp:720-725); the
 * quad is drawn into an sRGB framebuffer (:422), read back as bytes and flipped (:102-103) */
void orc_present(const float* accum, uint32_t width, uint32_t height, uint32_t passes, uint32_t tonemap,
                 int flip, uint8_t* rgb8)
{
    for (uint32_t y = 0; y < height; ++y) {
        uint32_t oy = flip ? height - 1u - y : y;
        for (uint32_t k = 0; k < width * 3u; ++k) {
            float v = accum[(size_t)y * width * 3u + k] / (float)passes;
            if (tonemap) v = cbrtf(v);
            if (!(v > 0.0f)) v = 0.0f;
            if (v > 1.0f) v = 1.0f;
            float e = v <= 0.0031308f ? 12.92f * v : 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
            rgb8[(size_t)oy * width * 3u + k] = (uint8_t)(int)(e * 255.0f + 0.5f);
        }
    }
}

double orc_uniform(uint64_t seed, uint32_t frame, uint32_t pixel, uint32_t sample,
                   uint32_t branch, uint32_t draw_index)
{
    uint32_t ctr[4] = { pixel, sample, branch, draw_index >> 2 };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) ^ (frame * 0x9E3779B9u) };
    uint32_t out[4];
    orc_philox4x32_10(ctr, key, out);
    return (double)out[draw_index & 3u] / 4294967295.0;
}

typedef struct {
    uint64_t seed; uint32_t frame, pixel, sample, branch, counter;
} rng_t;

static double rng_next(rng_t* r)
{
    return orc_uniform(r->seed, r->frame, r->pixel, r->sample, r->branch, r->counter++);
}

/* ------------------------------------------------------------------ Ray --- */
typedef struct { v3 origin, direction, inverse_direction; } ray_t;

/* Ray::Ray, Ray.cpp:3-8: re-normalises, inverseDirection is the NEGATED direction */
static ray_t make_ray(v3 origin, v3 dir)
{
    ray_t r;
    r.direction = normalize3(dir);
    r.inverse_direction = neg(r.direction);
    r.origin = origin;
    return r;
}

typedef struct {
    v3 position, normal, bary;
    double u, v;
    double depth;
    int32_t prim;
} hit_t;

typedef struct {
    uint64_t branch, leaf, aabb, tri, sph, accepted;
    uint32_t* seq; uint32_t seq_cap; uint32_t leaf_count; uint64_t leaf_hash;
} trace_ctx;

/* ---------------------------------------------------------- slab tests ---- */
/* shared arithmetic of AABB::rayAABB, Shape.cpp:14-24 and :100-110 */
static void slab(const double* bmin, const double* bmax, const ray_t* r, double* tmin, double* tmax)
{
    v3 invD = V(1.0 / r->direction.x, 1.0 / r->direction.y, 1.0 / r->direction.z);
    v3 t0 = mulv(sub(ld3(bmin), r->origin), invD);
    v3 t1 = mulv(sub(ld3(bmax), r->origin), invD);
    v3 ts = V(gmin(t0.x, t1.x), gmin(t0.y, t1.y), gmin(t0.z, t1.z));
    v3 tb = V(gmax(t0.x, t1.x), gmax(t0.y, t1.y), gmax(t0.z, t1.z));
    double lo = -INFINITY, hi = INFINITY;
    lo = gmax(lo, gmax(ts.x, gmax(ts.y, ts.z)));
    hi = gmin(hi, gmin(tb.x, gmin(tb.y, tb.z)));
    *tmin = lo; *tmax = hi;
}

/* AABB::rayAABB(ray, enter, exit), Shape.cpp:12-34 (strict tMin < tMax) */
static int aabb_enter_exit(const double* bmin, const double* bmax, const ray_t* r, v3* enter, v3* exit_)
{
    double tmin, tmax;
    slab(bmin, bmax, r, &tmin, &tmax);
    int hit = (tmin < tmax) && (tmax >= 0.0);
    *enter = add(r->origin, muls(r->direction, tmin));
    *exit_ = add(r->origin, muls(r->direction, tmax));
    return hit;
}

/* AABB::rayAABB(ray), Shape.cpp:100-116 (tMin <= tMax) */
static int aabb_hit(const double* bmin, const double* bmax, const ray_t* r)
{
    double tmin, tmax;
    slab(bmin, bmax, r, &tmin, &tmax);
    return (tmin <= tmax) && (tmax >= 0.0);
}

/* ------------------------------------------------------------ shapes ------ */
/* Triangle::rayHit, Triangle.cpp:8-72 */
static int triangle_hit(const double* pos, const double* nrm, const double* uvs, const ray_t* r, hit_t* h)
{
    v3 a = ld3(pos), b = ld3(pos + 3), c = ld3(pos + 6);
    v3 E1 = sub(b, a), E2 = sub(c, a);
    v3 pvec = cross(r->direction, E2);
    double det = dot(E1, pvec);
    if (det < 0.0) return 0;
    double inv = 1.0 / det;
    v3 tvec = sub(r->origin, a);
    double u = dot(tvec, pvec) * inv;
    if (u < 0.0 || u > 1.0) return 0;
    v3 qvec = cross(tvec, E1);
    double v = dot(r->direction, qvec) * inv;
    if (v < 0.0 || u + v > 1.0) return 0;
    h->bary = V(1.0 - (u + v), u, v);
    h->position = add(add(muls(a, h->bary.x), muls(b, h->bary.y)), muls(c, h->bary.z));
    h->depth = distance3(h->position, r->origin);
    h->normal = add(add(muls(ld3(nrm), h->bary.x), muls(ld3(nrm + 3), h->bary.y)),
                    muls(ld3(nrm + 6), h->bary.z));
    /* mat3x2 is a FLOAT matrix (Triangle.cpp:64): uv is evaluated in FP32 */
    float bx = (float)h->bary.x, by = (float)h->bary.y, bz = (float)h->bary.z;
    float m00 = (float)uvs[0], m01 = (float)uvs[1];
    float m10 = (float)uvs[2], m11 = (float)uvs[3];
    float m20 = (float)uvs[4], m21 = (float)uvs[5];
    float fu = m00 * bx + m10 * by + m20 * bz;
    float fv = m01 * bx + m11 * by + m21 * bz;
    h->u = (double)fu; h->v = (double)fv;
    return 1;
}

/* solveQuadratic, CoordinateHelpers.cpp:22-32 */
void orc_solve_quadratic(double a, double b, double c, double* A1, double* A2)
{
    double disc = (b * b) - (4.0 * a * c);
    if (disc < 0.0) { *A1 = NAN; *A2 = NAN; return; }
    *A1 = (-b + sqrt(disc)) / (2.0 * a);
    *A2 = (-b - sqrt(disc)) / (2.0 * a);
}

/* Reference-cost mode (orc_set_reference_costs): the restatement above hoists work the reference
 * repeats -- it copies the leaf's std::vector<Shape*> by value on every leaf visit
 * (rayHitListOfShapes takes it by value, KDTree.cpp:206), evaluates getModel(t) and
 * transformPos per sphere test (Sphere.cpp:47-48) and inverts the 4x4 model matrix per accepted
 * sphere hit (transformNormal, Sphere.cpp:74, CoordinateHelpers.cpp:12-16).  With the mode on, the
 * same work is done again (its results discarded: they equal the hoisted ones), so that a CPU
 * baseline timed this way is not flattered by the clean-ups.  Results are unchanged. */
static int g_reference_costs = 0;
void orc_set_reference_costs(int on) { g_reference_costs = on; }
static volatile double g_cost_sink;

static void cost_mat4_times_point(const double* m, v3 p)
{
    double o[4];
    for (int i = 0; i < 4; ++i) o[i] = (m[0 + i] * p.x + m[4 + i] * p.y) + (m[8 + i] * p.z + m[12 + i] * 1.0);
    g_cost_sink = o[0] / o[3] + o[1] / o[3] + o[2] / o[3];
}

/* glm::inverse(mat4): cofactor expansion (what transformNormal runs per hit) */
static void cost_mat4_inverse(const double* m)
{
    double c00 = m[10] * m[15] - m[14] * m[11], c02 = m[6] * m[15] - m[14] * m[7], c03 = m[6] * m[11] - m[10] * m[7];
    double c04 = m[9] * m[15] - m[13] * m[11], c06 = m[5] * m[15] - m[13] * m[7], c07 = m[5] * m[11] - m[9] * m[7];
    double c08 = m[9] * m[14] - m[13] * m[10], c10 = m[5] * m[14] - m[13] * m[6], c11 = m[5] * m[10] - m[9] * m[6];
    double c12 = m[8] * m[15] - m[12] * m[11], c14 = m[4] * m[15] - m[12] * m[7], c15 = m[4] * m[11] - m[8] * m[7];
    double c16 = m[8] * m[14] - m[12] * m[10], c18 = m[4] * m[14] - m[12] * m[6], c19 = m[4] * m[10] - m[8] * m[6];
    double c20 = m[8] * m[13] - m[12] * m[9], c22 = m[4] * m[13] - m[12] * m[5], c23 = m[4] * m[9] - m[8] * m[5];
    double f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    double f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    double v0[4] = {m[4], m[0], m[0], m[0]}, v1[4] = {m[5], m[1], m[1], m[1]};
    double v2[4] = {m[6], m[2], m[2], m[2]}, v3_[4] = {m[7], m[3], m[3], m[3]};
    double inv[16], sign = 1.0;
    for (int k = 0; k < 4; ++k, sign = -sign) {
        inv[0 + k] = sign * ((v1[k] * f0[k] - v2[k] * f1[k]) + v3_[k] * f2[k]);
        inv[4 + k] = -sign * ((v0[k] * f0[k] - v2[k] * f3[k]) + v3_[k] * f4[k]);
        inv[8 + k] = sign * ((v0[k] * f1[k] - v1[k] * f3[k]) + v3_[k] * f5[k]);
        inv[12 + k] = -sign * ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]);
    }
    double det = (m[0] * inv[0] + m[1] * inv[4]) + (m[2] * inv[8] + m[3] * inv[12]);
    double one_over = 1.0 / det, acc = 0.0;
    for (int k = 0; k < 16; ++k) acc += inv[k] * one_over;
    g_cost_sink = acc;
}

/* Sphere::rayHit, Sphere.cpp:41-81, with transOrigin = transformPos(origin, model)
 * and normalMat = inverse(transpose(model)) (CoordinateHelpers.cpp:12-20) evaluated
 * once per frame by the host */
static int sphere_hit(const double* center, double r2, const double* nm, const ray_t* r, hit_t* h)
{
    v3 c0 = ld3(center);
    if (g_reference_costs) cost_mat4_times_point(nm, c0);       /* getModel + transformPos per test */
    v3 L = sub(r->origin, c0);
    double a = dot(r->direction, r->direction);
    double b = 2.0 * dot(r->direction, L);
    double c = dot(L, L) - r2;
    double t0, t1;
    orc_solve_quadratic(a, b, c, &t0, &t1);
    if (isnan(t0) && isnan(t1)) return 0;
    if (t0 > t1) { double t = t0; t0 = t1; t1 = t; }
    if (t0 < 0) { t0 = t1; if (t0 < 0) return 0; }
    h->position = add(r->origin, muls(r->direction, t0));
    v3 n = normalize3(sub(h->position, c0));
    if (g_reference_costs) cost_mat4_inverse(nm);               /* inverse(transpose(model)) per hit */
    /* normalMat * dvec4(n, 1.0), truncated to xyz */
    double o[3];
    for (int i = 0; i < 3; ++i) {
        double mul0 = nm[0 + i] * n.x, mul1 = nm[4 + i] * n.y;
        double mul2 = nm[8 + i] * n.z, mul3 = nm[12 + i] * 1.0;
        o[i] = (mul0 + mul1) + (mul2 + mul3);
    }
    h->normal = V(o[0], o[1], o[2]);
    h->u = atan2(h->normal.x, h->normal.z) / (2 * PI) + 0.5;
    h->v = h->normal.y * 0.5 + 0.5;
    h->bary = V(0.0, 0.0, 0.0);
    h->depth = t0;
    return 1;
}

/* Sphere::pointInSphere, Sphere.cpp:20-27 */
static int point_in_sphere(const rtx_scene_desc* s, uint32_t sph, v3 p)
{
    return distance3(p, ld3(s->sphere_center + 3 * sph)) <= s->sphere_radius[sph];
}

static int prim_sphere_hit(const rtx_scene_desc* s, uint32_t prim, const ray_t* r, hit_t* h, trace_ctx* ctx)
{
    uint32_t i = s->prim_index[prim];
    if (ctx) ctx->sph++;
    int ok = sphere_hit(s->sphere_center + 3 * i, s->sphere_r2[i], s->sphere_normal_matrix + 16 * i, r, h);
    if (ok) h->prim = (int32_t)prim;
    return ok;
}

/* Biconvex::rayHit, Biconvex.cpp:55-85 */
static int biconvex_hit(const rtx_scene_desc* s, uint32_t lens, const ray_t* r, hit_t* h, trace_ctx* ctx)
{
    uint32_t p1 = s->lens_spheres[2 * lens], p2 = s->lens_spheres[2 * lens + 1];
    hit_t h1, h2;
    int s1 = prim_sphere_hit(s, p1, r, &h1, ctx);
    int s2 = prim_sphere_hit(s, p2, r, &h2, ctx);
    s1 = s1 && point_in_sphere(s, s->prim_index[p2], h1.position);
    s2 = s2 && point_in_sphere(s, s->prim_index[p1], h2.position);
    double min_distance = INFINITY;
    if (s1 && h1.depth < min_distance) { min_distance = h1.depth; *h = h1; }
    if (s2 && h2.depth < min_distance) { *h = h2; }
    return s1 || s2;
}

static int prim_hit(const rtx_scene_desc* s, uint32_t prim, const ray_t* r, hit_t* h, trace_ctx* ctx)
{
    uint32_t i = s->prim_index[prim];
    switch (s->prim_type[prim]) {
    case RTX_PRIM_TRIANGLE:
        if (ctx) ctx->tri++;
        if (triangle_hit(s->tri_position + 9 * i, s->tri_normal + 9 * i, s->tri_uv + 6 * i, r, h)) {
            h->prim = (int32_t)prim; return 1;
        }
        return 0;
    case RTX_PRIM_SPHERE:
        return prim_sphere_hit(s, prim, r, h, ctx);
    default:
        return biconvex_hit(s, i, r, h, ctx);
    }
}

/* rayHitListOfShapes, KDTree.cpp:206-220 */
static void hit_list(const rtx_scene_desc* s, const uint32_t* refs, uint32_t count, const ray_t* r,
                     hit_t* best, trace_ctx* ctx)
{
    for (uint32_t k = 0; k < count; ++k) {
        uint32_t prim = refs ? refs[k] : k;
        hit_t h;
        ctx->aabb++;
        if (aabb_hit(s->prim_aabb_min + 3 * prim, s->prim_aabb_max + 3 * prim, r)) {
            if (prim_hit(s, prim, r, &h, ctx)) {
                if (h.depth < best->depth) { ctx->accepted++; *best = h; }
            }
        }
    }
}

/* traverseKDTree, KDTree.cpp:158-204.  A missed branch box returns without
 * descending (the reference falls off the end of the function there). */
static void traverse(const rtx_scene_desc* s, int32_t node, const ray_t* r, hit_t* best, trace_ctx* ctx)
{
    if (node < 0) {
        uint32_t leaf = (uint32_t)~node;
        if (ctx->seq && ctx->leaf_count < ctx->seq_cap) ctx->seq[ctx->leaf_count] = leaf;
        ctx->leaf_count++;
        for (int b = 0; b < 4; ++b) {
            ctx->leaf_hash ^= (leaf >> (8 * b)) & 0xffu;
            ctx->leaf_hash *= 1099511628211ull;
        }
        ctx->leaf++;
        if (g_reference_costs) {
            /* the by-value std::vector<Shape*> argument: allocate, copy 8-byte pointers, free */
            uint32_t cnt = s->leaf_count[leaf];
            const uint32_t* refs = s->leaf_refs + s->leaf_first[leaf];
            uint64_t* copy = (uint64_t*)malloc((size_t)(cnt ? cnt : 1) * sizeof(uint64_t));
            uint32_t* narrow = (uint32_t*)malloc((size_t)(cnt ? cnt : 1) * sizeof(uint32_t));
            for (uint32_t k = 0; k < cnt; ++k) copy[k] = refs[k];
            for (uint32_t k = 0; k < cnt; ++k) narrow[k] = (uint32_t)copy[k];
            hit_list(s, narrow, cnt, r, best, ctx);
            free(narrow);
            free(copy);
            return;
        }
        hit_list(s, s->leaf_refs + s->leaf_first[leaf], s->leaf_count[leaf], r, best, ctx);
        return;
    }
    ctx->branch++;
    v3 enter, exit_;
    int axis = s->branch_axis[node];
    double value = s->branch_value[node];
    if (aabb_enter_exit(s->branch_min + 3 * node, s->branch_max + 3 * node, r, &enter, &exit_)) {
        if (comp(enter, axis) <= value) {
            if (comp(exit_, axis) < value) {
                traverse(s, s->branch_lesser[node], r, best, ctx);
            } else {
                traverse(s, s->branch_lesser[node], r, best, ctx);
                traverse(s, s->branch_greater[node], r, best, ctx);
            }
        } else {
            if (comp(exit_, axis) > value) {
                traverse(s, s->branch_greater[node], r, best, ctx);
            } else {
                traverse(s, s->branch_lesser[node], r, best, ctx);
                traverse(s, s->branch_greater[node], r, best, ctx);
            }
        }
    }
}

/* shootRay, Raytracer.cpp:108-120 */
static hit_t shoot_ray(const rtx_scene_desc* s, const ray_t* r, uint32_t flags, trace_ctx* ctx)
{
    hit_t best;
    memset(&best, 0, sizeof best);
    best.prim = -1;
    best.depth = INFINITY;
    if (flags & RTX_FLAG_LINEAR_SCAN) hit_list(s, NULL, s->n_shapes, r, &best, ctx);
    else traverse(s, s->kd_root, r, &best, ctx);
    return best;
}

int orc_trace_closest(const rtx_scene_desc* s, const double* origin, const double* direction,
                      size_t n, uint32_t flags, int32_t* prim, double* depth, double* position,
                      double* normal, double* uv, double* bary, uint32_t* leaf_count,
                      uint64_t* leaf_hash, uint32_t* leaf_seq, uint32_t leaf_seq_cap,
                      uint32_t* counters)
{
    if (!s || !origin || !direction || !prim || !depth) return RTX_ERR_INVALID;
#pragma omp parallel for schedule(dynamic, 256)
    for (long long i = 0; i < (long long)n; ++i) {
        ray_t r = make_ray(ld3(origin + 3 * i), ld3(direction + 3 * i));
        trace_ctx ctx;
        memset(&ctx, 0, sizeof ctx);
        ctx.leaf_hash = 14695981039346656037ull;
        ctx.seq = leaf_seq ? leaf_seq + (size_t)i * leaf_seq_cap : NULL;
        ctx.seq_cap = leaf_seq_cap;
        hit_t h = shoot_ray(s, &r, flags, &ctx);
        prim[i] = h.prim;
        depth[i] = h.depth;
        if (position) st3(position + 3 * i, h.position);
        if (normal) st3(normal + 3 * i, h.normal);
        if (uv) { uv[2 * i] = h.u; uv[2 * i + 1] = h.v; }
        if (bary) st3(bary + 3 * i, h.bary);
        if (leaf_count) leaf_count[i] = ctx.leaf_count;
        if (leaf_hash) leaf_hash[i] = ctx.leaf_hash;
        if (counters) {
            uint32_t* c = counters + 6 * i;
            c[0] = (uint32_t)ctx.branch; c[1] = (uint32_t)ctx.leaf; c[2] = (uint32_t)ctx.aabb;
            c[3] = (uint32_t)ctx.tri; c[4] = (uint32_t)ctx.sph; c[5] = (uint32_t)ctx.accepted;
        }
    }
    return RTX_OK;
}

/* --------------------------------------------------------- materials ------ */
/* ryCheckers10x10 / doubleCheckers10x10, Procedural.cpp:4-25 */
static int checker_even(double u, double v, double squares)
{
    int fx = (int)floor(u * squares), fy = (int)floor(v * squares);
    return ((fx + fy) % 2) == 0;
}

/* texel fetch shared by Texture::vec3Sample / doubleSample, Texture.cpp:39-64.
 * The reference indexes past the image when uv*size rounds up to size; here the
 * index is clamped to the last texel instead of reading out of bounds. */
static const uint8_t* texel(const rtx_scene_desc* s, uint32_t tex, double u, double v)
{
    const rtx_texture* t = &s->textures[tex];
    u = u - 1.0 * floor(u / 1.0);
    v = v - 1.0 * floor(v / 1.0);
    uint32_t px = (uint32_t)floor(u * (double)t->width);
    uint32_t py = (uint32_t)floor(v * (double)t->height);
    uint64_t idx = (uint64_t)(px + py * t->width);
    uint64_t last = (uint64_t)t->width * t->height - 1;
    if (idx > last) idx = last;
    return s->texels + t->offset + idx * 3u;
}

static double sample_scalar(const rtx_scene_desc* s, rtx_sampler f, double constant, double u, double v)
{
    switch (f.kind) {
    case RTX_SAMPLER_DOUBLE_CHECKERS: return checker_even(u, v, 10.0) ? 1.0 : 0.0;
    case RTX_SAMPLER_TEXTURE_DOUBLE: {
        const uint8_t* p = texel(s, f.texture, u, v);
        return ((double)p[0] / 255.0 + (double)p[1] / 255.0 + (double)p[2] / 255.0) / 3.0;
    }
    default: return constant;
    }
}

/* Material::getColor, Material.cpp:111-120 */
static v3 material_color(const rtx_scene_desc* s, const rtx_material* m, double u, double v)
{
    switch (m->color_fn.kind) {
    case RTX_SAMPLER_RY_CHECKERS:
        return checker_even(u, v, 1000.0) ? V(1.0, 0.0, 0.0) : V(1.0, 1.0, 0.0);
    case RTX_SAMPLER_TEXTURE_VEC3: {
        /* getVec3Sampler returns function<vec3(dvec2)> (Texture.hpp:29): rounded to float */
        const uint8_t* p = texel(s, m->color_fn.texture, u, v);
        return V((double)(float)((double)p[0] / 255.0), (double)(float)((double)p[1] / 255.0),
                 (double)(float)((double)p[2] / 255.0));
    }
    default: return ld3(m->color);
    }
}
static double material_ns(const rtx_scene_desc* s, const rtx_material* m, double u, double v) { return sample_scalar(s, m->ns_fn, m->ns, u, v); }
static double material_ni(const rtx_scene_desc* s, const rtx_material* m, double u, double v) { return sample_scalar(s, m->ni_fn, m->ni, u, v); }
static double material_transparency(const rtx_scene_desc* s, const rtx_material* m, double u, double v) { return sample_scalar(s, m->transparency_fn, m->transparency, u, v); }
static double material_metalness(const rtx_scene_desc* s, const rtx_material* m, double u, double v) { return sample_scalar(s, m->metalness_fn, m->metalness, u, v); }
static double material_smoothness(const rtx_scene_desc* s, const rtx_material* m, double u, double v) { return sample_scalar(s, m->smoothness_fn, m->smoothness, u, v); }

/* ------------------------------------------------------------ lights ------ */
/* PointLight::getRay PointLight.cpp:20-23; DirectionalLight::getRay
 * DirectionalLight.cpp:18-28; SquareLight::getRay SquareLight.cpp:53-64
 * (u0,u1 = the two randDub() draws) */
static ray_t light_ray(const rtx_light* l, v3 p, double u0, double u1)
{
    switch (l->kind) {
    case RTX_LIGHT_POINT:
        return make_ray(p, normalize3(sub(ld3(l->position), p)));
    case RTX_LIGHT_DIRECTIONAL: {
        v3 far_point = muls(ld3(l->direction), 10000000000000.0);
        return make_ray(p, normalize3(sub(far_point, p)));
    }
    default: {
        v3 a = muls(muls(ld3(l->right), u0), l->width);
        v3 b = muls(muls(ld3(l->up), u1), l->height);
        v3 target = add(add(ld3(l->bottom_left), a), b);
        return make_ray(p, normalize3(sub(target, p)));
    }
    }
}

/* getDistance: PointLight.cpp:15-18, DirectionalLight.cpp:14-16,
 * SquareLight.cpp:70-94 (where the reference exit(-1)s this returns +inf,
 * which fails the shadow test `distance < depth`) */
static double light_distance(const rtx_light* l, const ray_t* r)
{
    switch (l->kind) {
    case RTX_LIGHT_POINT: return distance3(ld3(l->position), r->origin);
    case RTX_LIGHT_DIRECTIONAL: return 10000000000000.0;
    default: {
        v3 n = ld3(l->direction);
        double denom = dot(n, r->direction);
        if (fabs(denom) < 2.220446049250313e-16) return INFINITY;
        v3 d = sub(ld3(l->position), r->origin);
        double t = dot(d, n) / denom;
        if (t < 2.220446049250313e-16) return INFINITY;
        v3 hp = add(r->origin, muls(r->direction, t));
        return distance3(r->origin, hp);
    }
    }
}

/* getAttenuation: PointLight.cpp:25-27, SquareLight.cpp:66-68, DirectionalLight.cpp:29-31 */
static double light_attenuation(const rtx_light* l, double d, const double* att)
{
    if (l->kind == RTX_LIGHT_DIRECTIONAL) return 1.0;
    return (att[0] + (att[1] * d) + (att[2] * (d * d)));
}

/* lightingFunction, Raytracer.cpp:187-197 (= Raytracer-Noah-Desktop.cpp:309-320) */
static v3 lighting(const rtx_scene_desc* s, const ray_t* ray, const ray_t* lray, v3 normal, double u, double v,
                   double attenuation, const rtx_material* m, v3 lcolor)
{
    v3 refl = normalize3(sub(muls(normal, dot(lray->direction, normal) * 2.0), lray->direction));
    double spec = pow(gmax(0.0, dot(refl, ray->inverse_direction)), material_ns(s, m, u, v));
    v3 specular = divs(muls(lcolor, spec), attenuation);
    double diff = gmax(0.0, dot(normal, lray->direction));
    v3 diffuse = divs(muls(mulv(material_color(s, m, u, v), lcolor), diff), attenuation);
    return add(diffuse, specular);
}

/* getRefractionRay, Raytracer.cpp:122-182 */
static v3 refraction_ray(v3 n, v3 d, double ior, int entering, int* internal_only)
{
    double prev = 1.0, next = ior;
    if (!entering) { n = neg(n); double t = prev; prev = next; next = t; }
    double cosA1 = dot(d, n);
    double sinA1 = sqrt(1.0 - (cosA1 * cosA1));
    double ratio = prev / next;
    double sinA2 = sinA1 * ratio;
    v3 best = d;
    if (sinA2 <= -1.0 || sinA2 >= 1.0) { *internal_only = 1; return best; }
    double max_close = -INFINITY, k1 = NAN, k2 = NAN;
    orc_solve_quadratic(1.0, 2.0 * cosA1, 1.0 - (1.0 / (ratio * ratio)), &k1, &k2);
    if (!isnan(k1)) {
        v3 r1 = normalize3(add(d, muls(n, k1)));
        double c1 = dot(d, r1);
        if (c1 > max_close && c1 >= 0.0) { max_close = c1; best = r1; }
    }
    if (!isnan(k2)) {
        v3 r2 = normalize3(add(d, muls(n, k2)));
        double c2 = dot(d, r2);
        if (c2 > max_close && c2 >= 0.0) { max_close = c2; best = r2; }
    }
    if (max_close <= 0.0) return d;     /* the reference prints and continues, :169-172 */
    return normalize3(best);
}

/* glm::rotate(v, angle, normal) of gtx/rotate_vector = mat3(rotate(mat4(1), angle, normal)) * v,
 * with c = cos(angle), s = sin(angle) supplied by the caller */
static v3 rotate_vec(v3 v, double c, double s, v3 normal)
{
    v3 axis = normalize3(normal);
    v3 temp = muls(axis, 1.0 - c);
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
    /* Result[i] = m[0]*R[i][0] + m[1]*R[i][1] + m[2]*R[i][2] with m = identity */
    double M[3][3];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            double e0 = (k == 0 ? 1.0 : 0.0) * R[i][0];
            double e1 = (k == 1 ? 1.0 : 0.0) * R[i][1];
            double e2 = (k == 2 ? 1.0 : 0.0) * R[i][2];
            M[i][k] = (e0 + e1) + e2;
        }
    v3 o;
    o.x = M[0][0] * v.x + M[1][0] * v.y + M[2][0] * v.z;
    o.y = M[0][1] * v.x + M[1][1] * v.y + M[2][1] * v.z;
    o.z = M[0][2] * v.x + M[1][2] * v.y + M[2][2] * v.z;
    return o;
}

/* -rotate(d, pi, n), Raytracer.cpp:284 */
static v3 mirror_direction(v3 d, v3 n)
{
    return neg(rotate_vec(d, cos(PI), sin(PI), n));
}

/* Xchi / Gp / G, CookTorrance.cpp:11-15,56-76 */
static double xchi(double x) { int i = (x > 0); return (1.0 * (double)i) + (-1.0 * ((double)(1 - i))); }
static double gp(v3 v, v3 n, v3 h, double a)
{
    double vdh = dot(v, h);
    double chi = xchi(vdh / dot(v, n));
    vdh *= vdh;
    double right = (1.0 - vdh) / vdh;
    return (chi * 2.0) / (1.0 + sqrt(1.0 + (a * a * right)));
}

/* CookTorance, CookTorrance.cpp:134-190 (fresnelSchlick :39-42) */
static v3 cook_torrance(const rtx_scene_desc* s, const ray_t* in, const ray_t* out, v3 normal, double u, double v,
                        v3 down, const rtx_material* m, v3 F0, v3* kS)
{
    double rough = 1.0 - material_smoothness(s, m, u, v);
    v3 vi = in->inverse_direction, vo = out->direction, vn = normal;
    v3 vh = normalize3(add(vi, vo));
    double ndi = dot(vn, vi);
    double cosT = dot(vo, vn);
    double sinT = sqrt(1 - cosT * cosT);
    double p = pow(1.0 - dot(vh, vi), 5);
    v3 fres = add(F0, muls(V(1.0 - F0.x, 1.0 - F0.y, 1.0 - F0.z), p));
    double geo = gp(vi, vn, vh, rough) * gp(vo, vn, vh, rough);
    double denom = 4.0 * (ndi * dot(vh, vn) + 0.05);
    *kS = fres;
    return divs(muls(mulv(muls(down, geo), fres), sinT), denom);
}

/* randomHemisphericalVector, Raytracer.cpp:73-80 (randDubThree :69-71) */
static v3 random_hemispherical(v3 normal, rng_t* rng)
{
    v3 out;
    normal = normalize3(normal);
    do {
        double a = (rng_next(rng) * 2.0) - 1.0;
        double b = (rng_next(rng) * 2.0) - 1.0;
        double c = (rng_next(rng) * 2.0) - 1.0;
        out = normalize3(V(a, b, c));
    } while (dot(normal, out) < 0.0);
    return out;
}

/* randomHemisphericalVector of the Whitted variant, Raytracer-Noah-Desktop.cpp:149-166:
 * two uniforms (theta first, then phi); b1/b2 are computed there and never used.  The
 * vector is NOT confined to the hemisphere, and for a normal parallel to worldUp the
 * rotation axis is the zero vector, whose normalisation makes every component NaN --
 * both kept as they are. */
static v3 random_hemispherical_desktop(v3 normal, rng_t* rng)
{
    normal = normalize3(normal);
    v3 world_up = normal.y > 0.9 ? V(0.0, 1.0, 0.0) : V(1.0, 0.0, 0.0);
    double theta = (2.0 * PI) * rng_next(rng);
    double phi = acos(1.0 - (2.0 * rng_next(rng)));
    v3 nv = V(sin(theta) * sin(phi), sin(theta) * cos(phi), sin(phi));
    v3 rot_vector = cross(normal, world_up);
    double rot_rads = acos(dot(normal, world_up));
    return rotate_vec(nv, cos(rot_rads), sin(rot_rads), rot_vector);
}

/* ---------------------------------------------------------- integrators --- */
typedef struct {
    const rtx_scene_desc* s;
    const rtx_render_desc* d;
    uint64_t rays_primary, rays_secondary, rays_shadow;
    trace_ctx ctx;
} render_ctx;

static int is_entering(v3 normal, const ray_t* ray)
{
    double angle = acos(dot(normal, ray->inverse_direction));
    return angle < (PI / 2.0);
}

static void count_ray(render_ctx* rc, uint32_t layer)
{
    if (layer == 0) rc->rays_primary++; else rc->rays_secondary++;
}

/* pathTrace, Raytracer.cpp:217-355 */
static v3 path_trace(render_ctx* rc, const ray_t* ray, rng_t* rng, double current_ior, uint32_t layer)
{
    const rtx_scene_desc* s = rc->s;
    const rtx_render_desc* d = rc->d;
    v3 clear = ld3(d->clear_color);
    (void)current_ior;
    if (layer > d->max_depth) return muls(clear, 0.1);
    count_ray(rc, layer);
    hit_t h = shoot_ray(s, ray, d->flags, &rc->ctx);
    if (h.prim < 0) return muls(clear, 0.1);
    const rtx_material* m = &s->materials[s->prim_material[h.prim]];
    if (d->flags & RTX_FLAG_BASIC) return material_color(s, m, h.u, h.v);

    double trans = material_transparency(s, m, h.u, h.v);
    double smooth = material_smoothness(s, m, h.u, h.v);
    double metal = material_metalness(s, m, h.u, h.v);
    double transparency_decider = rng_next(rng);
    double reflectance_decider = rng_next(rng);
    (void)rng_next(rng);                             /* specularDecider, drawn and unused */
    int entering = is_entering(h.normal, ray);
    int internal_only = 0;
    v3 radiance = ld3(m->emission);
    if (radiance.x == 0.0 && radiance.y == 0.0 && radiance.z == 0.0) {
        double ni = material_ni(s, m, h.u, h.v);
        if (transparency_decider < trans) {
            v3 dir = normalize3(refraction_ray(normalize3(h.normal), normalize3(ray->direction), ni,
                                               entering, &internal_only));
            v3 pos = add(h.position, muls(muls(h.normal, entering ? -1.0 : 1.0), d->bias));
            ray_t child = make_ray(pos, dir);
            radiance = path_trace(rc, &child, rng, ni, layer + 1);
        } else {
            v3 dir, pos;
            if (reflectance_decider < smooth) {
                dir = mirror_direction(ray->direction, h.normal);
                pos = add(h.position, muls(h.normal, d->bias));
            } else {
                dir = normalize3(random_hemispherical(h.normal, rng));
                pos = add(h.position, muls(h.normal, d->bias));
            }
            ray_t child = make_ray(pos, dir);
            v3 down = path_trace(rc, &child, rng, ni, layer + 1);
            double f = fabs((1.0 - ni) / (1.0 + ni));
            v3 F0a = V(f, f, f);
            F0a = mulv(F0a, F0a);
            v3 matC = material_color(s, m, h.u, h.v);
            v3 F0 = add(muls(F0a, 1.0 - metal), muls(matC, metal));
            v3 kS = V(0, 0, 0);
            v3 spec = cook_torrance(s, ray, &child, h.normal, h.u, h.v, down, m, F0, &kS);
            double diff = dot(dir, h.normal);
            v3 kD = muls(V(1.0 - kS.x, 1.0 - kS.y, 1.0 - kS.z), 1.0 - metal);
            v3 diffuse = mulv(muls(down, diff), material_color(s, m, h.u, h.v));
            radiance = add(radiance, add(spec, mulv(kD, diffuse)));
        }
    }
    return radiance;
}

/* globalSamples of Raytracer-Noah-Desktop.cpp:500:
 *   int(MONTE_CARLO_SAMPLES * ((double(MAXBOUNCES)/double(layer+1))/double(MAXBOUNCES)))
 * MAXBOUNCES = 0 makes the factor NaN, whose conversion is undefined: no samples then. */
static int global_samples_at(const rtx_render_desc* d, uint32_t layer)
{
    if (d->max_depth == 0) return 0;
    double f = ((double)d->max_depth / (double)(layer + 1)) / (double)d->max_depth;
    return (int)((double)d->global_samples * f);
}

/* The gather's nested rayTrace call (Desktop :510) feeds only two things back: whether it
 * hit, and where (its colour lands in a shadowing local, :509).  With this switch off the
 * oracle evaluates just those -- one shootRay when layer+1 <= MAXBOUNCES; with it on it runs
 * the whole nested call as the reference does, on a random stream of its own, and throws
 * the colour away.  Both give the same image (tests/test_oracle_kat.py); the nested work
 * grows as the product of the per-layer sample counts. */
static int g_global_full_recursion = 0;
void orc_set_global_full_recursion(int on) { g_global_full_recursion = on; }

/* rayTrace, Raytracer-Noah-Desktop.cpp:452-564, LIGHTING on; GLOBAL_LIGHTING
 * (:499-524) with RTX_FLAG_GLOBAL_LIGHTING.  RTX_FLAG_WHITTED_REFLECT adds a
 * mirror ray that the reference does
 * not have (its rayTrace only recurses for refraction):
 *   local = (1-smooth)*local + smooth*rayTrace(mirror ray), smooth != 0,
 * with the mirror ray of pathTrace case 2 (Raytracer.cpp:284-285).  The mirror
 * child draws from its own random stream (branch | 1<<layer, counter 0) so the
 * result does not depend on evaluation order.  `hit_out`, when not NULL, receives
 * minRayResult (untouched when the layer limit ends the call before shootRay). */
static int ray_trace(render_ctx* rc, const ray_t* ray, rng_t* rng, v3* color, double current_ior, uint32_t layer,
                     hit_t* hit_out)
{
    const rtx_scene_desc* s = rc->s;
    const rtx_render_desc* d = rc->d;
    v3 clear = ld3(d->clear_color);
    (void)current_ior;
    if (layer > d->max_depth) { *color = clear; return 0; }
    *color = V(0, 0, 0);
    count_ray(rc, layer);
    hit_t h = shoot_ray(s, ray, d->flags, &rc->ctx);
    if (hit_out) *hit_out = h;
    if (h.prim < 0) { *color = clear; return 0; }
    const rtx_material* m = &s->materials[s->prim_material[h.prim]];
    int entering = is_entering(h.normal, ray);
    v3 direct = V(0, 0, 0);
    v3 global = V(0, 0, 0);
    if (!(d->flags & RTX_FLAG_NO_LIGHTING)) {
        for (uint32_t li = 0; li < s->n_lights; ++li) {
            const rtx_light* l = &s->lights[li];
            int samples = l->multisample ? (int)(d->light_multisample_n * d->light_multisample_n) : 1;
            for (int i = 0; i < samples; ++i) {
                double u0 = 0.0, u1 = 0.0;
                if (l->kind == RTX_LIGHT_SQUARE) { u0 = rng_next(rng); u1 = rng_next(rng); }
                ray_t lray = light_ray(l, add(h.position, muls(h.normal, d->bias)), u0, u1);
                rc->rays_shadow++;
                hit_t lh = shoot_ray(s, &lray, d->flags, &rc->ctx);
                double ldist = light_distance(l, &lray);
                if (ldist < lh.depth) {
                    double att = light_attenuation(l, ldist, d->attenuation);
                    direct = add(direct, lighting(s, ray, &lray, h.normal, h.u, h.v, att, m, ld3(l->color)));
                }
            }
            direct = divs(direct, (double)samples);    /* inside the light loop, :498 */
            if (d->flags & RTX_FLAG_GLOBAL_LIGHTING) {  /* :499-524, inside the light loop as well */
                int gs = global_samples_at(d, layer);
                for (int i = 0; i < gs; ++i) {
                    v3 origin = h.position;             /* no bias here */
                    ray_t lray = make_ray(origin, random_hemispherical_desktop(h.normal, rng));
                    hit_t lh;
                    memset(&lh, 0, sizeof lh);
                    int lhit;
                    if (g_global_full_recursion) {
                        rng_t fork = *rng;
                        fork.branch = (rng->branch * 0x9E3779B1u) ^ ((layer + 1u) << 24) ^ (li << 16) ^ (uint32_t)i ^ 0x80000000u;
                        fork.counter = 0;
                        v3 unused;
                        lhit = ray_trace(rc, &lray, &fork, &unused, material_ni(s, m, h.u, h.v), layer + 1, &lh);
                    } else if (layer + 1 > d->max_depth) {
                        lhit = 0;                       /* the nested call returns before shootRay, :456-460 */
                    } else {
                        count_ray(rc, layer + 1);
                        lh = shoot_ray(s, &lray, d->flags, &rc->ctx);
                        lhit = lh.prim >= 0;
                    }
                    if (lhit) {
                        double ray_distance = distance3(origin, lh.position);
                        double att = light_attenuation(l, ray_distance, d->attenuation);
                        global = add(global, lighting(s, ray, &lray, h.normal, h.u, h.v, att, m, ld3(l->color)));
                    } else {
                        global = add(global, muls(clear, 0.04));
                    }
                }
                global = divs(global, (double)d->global_samples);   /* :523 */
            }
        }
    }
    v3 transparent = clear;
    double trans = material_transparency(s, m, h.u, h.v);
    double ni = material_ni(s, m, h.u, h.v);
    if (fabs(trans - 0.0) >= 2.220446049250313e-16) {   /* epsilonNotEqual, :535 */
        int internal_only = 0;
        v3 dir = refraction_ray(normalize3(h.normal), normalize3(ray->direction), ni, entering, &internal_only);
        v3 pos = add(h.position, muls(muls(h.normal, entering ? -1.0 : 1.0), d->bias));
        ray_t child = make_ray(pos, dir);
        ray_trace(rc, &child, rng, &transparent, ni, layer + 1, NULL);
    }
    v3 local;
    if (d->flags & RTX_FLAG_NO_LIGHTING) local = material_color(s, m, h.u, h.v);
    else local = add(direct, global);                   /* :553 */
    if (d->flags & RTX_FLAG_WHITTED_REFLECT) {
        double smooth = material_smoothness(s, m, h.u, h.v);
        if (smooth != 0.0) {
            v3 mdir = mirror_direction(ray->direction, h.normal);
            ray_t child = make_ray(add(h.position, muls(h.normal, d->bias)), mdir);
            rng_t fork = *rng;
            fork.branch = rng->branch | (1u << (layer & 31u));
            fork.counter = 0;
            v3 reflected;
            ray_trace(rc, &child, &fork, &reflected, ni, layer + 1, NULL);
            local = add(muls(local, 1.0 - smooth), muls(reflected, smooth));
        }
    }
    *color = add(muls(local, 1.0 - trans), muls(transparent, 1.0 - (1.0 - trans)));
    return 1;
}

/* per-pixel lambda, Raytracer.cpp:676-700 (no aspect ratio; sub-pixel offsets
 * along world x/y; y = 0 is the bottom row) */
void orc_camera_ray(const rtx_render_desc* d, uint32_t px, uint32_t py, uint32_t sx, uint32_t sy,
                    double origin[3], double direction[3])
{
    const rtx_camera* c = &d->camera;
    double nx = ((double)px / (double)d->width) - 0.5;
    double ny = ((double)py / (double)d->height) - 0.5;
    v3 eye = ld3(c->eye);
    v3 on_screen = add(add(add(muls(ld3(c->right), nx), muls(ld3(c->up), ny)), eye),
                       muls(ld3(c->forward), c->focal));
    double cx = 1.0 / ((double)d->width - 1.0), cy = 1.0 / ((double)d->height - 1.0);
    uint32_t n = d->pixel_multisample_n;
    double ox = (double)sx * (cx / (double)(n + 1));
    double oy = (double)sy * (cy / (double)(n + 1));
    v3 ray_vec = normalize3(sub(add(on_screen, V(ox, oy, 0.0)), eye));
    st3(origin, eye);
    st3(direction, ray_vec);
}

int orc_render(const rtx_scene_desc* s, const rtx_render_desc* d, float* accum, double* rgb64,
               rtx_stats* stats, int threads, int* threads_used)
{
    if (!s || !d || !accum) return RTX_ERR_INVALID;
    uint32_t n = d->pixel_multisample_n ? d->pixel_multisample_n : 1;
    long long npix = (long long)d->tile_w * d->tile_h;
    uint64_t t_prim = 0, t_sec = 0, t_shadow = 0;
    uint64_t c_branch = 0, c_leaf = 0, c_aabb = 0, c_tri = 0, c_sph = 0, c_acc = 0;
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);     /* otherwise OMP_NUM_THREADS / the OpenMP default */
#endif
    for (uint32_t pass = d->first_pass; pass < d->first_pass + d->passes; ++pass) {
#pragma omp parallel reduction(+ : t_prim, t_sec, t_shadow, c_branch, c_leaf, c_aabb, c_tri, c_sph, c_acc)
        {
#ifdef _OPENMP
#pragma omp single
            used = omp_get_num_threads();
#endif
            render_ctx rc;
            memset(&rc, 0, sizeof rc);
            rc.s = s; rc.d = d;
#pragma omp for schedule(dynamic, 64)
            for (long long t = 0; t < npix; ++t) {
                uint32_t x = d->tile_x0 + (uint32_t)(t % d->tile_w);
                uint32_t row = (uint32_t)(t / d->tile_w);       /* tile rows may be interleaved groups of frame rows (rtx.h) */
                uint32_t y = d->tile_y0 + (d->tile_row_group ? (row / d->tile_row_group) * d->tile_row_stride + row % d->tile_row_group : row);
                v3 acum = V(0, 0, 0);
                for (uint32_t sx = 1; sx <= n; ++sx)
                    for (uint32_t sy = 1; sy <= n; ++sy) {
                        double o[3], dir[3];
                        rtx_render_desc dd = *d;
                        dd.pixel_multisample_n = n;
                        orc_camera_ray(&dd, x, y, sx, sy, o, dir);
                        ray_t ray = make_ray(ld3(o), ld3(dir));
                        rng_t rng = { d->seed, d->frame, x + y * d->width,
                                      pass * n * n + (sx - 1) * n + (sy - 1), 0, 0 };
                        v3 c;
                        if (d->integrator == RTX_INTEGRATOR_PATH) c = path_trace(&rc, &ray, &rng, 1.0, 0);
                        else ray_trace(&rc, &ray, &rng, &c, 1.0, 0, NULL);
                        acum = add(acum, c);
                    }
                v3 px = divs(acum, (double)(n * n));
                float* a = accum + 3 * t;
                a[0] += (float)px.x; a[1] += (float)px.y; a[2] += (float)px.z;
                if (rgb64) st3(rgb64 + 3 * t, px);
            }
            t_prim += rc.rays_primary; t_sec += rc.rays_secondary; t_shadow += rc.rays_shadow;
            c_branch += rc.ctx.branch; c_leaf += rc.ctx.leaf; c_aabb += rc.ctx.aabb;
            c_tri += rc.ctx.tri; c_sph += rc.ctx.sph; c_acc += rc.ctx.accepted;
        }
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->rays_primary = t_prim; stats->rays_secondary = t_sec; stats->rays_shadow = t_shadow;
        stats->branch_visits = c_branch; stats->leaf_visits = c_leaf; stats->aabb_tests = c_aabb;
        stats->triangle_tests = c_tri; stats->sphere_tests = c_sph; stats->hits_accepted = c_acc;
    }
    if (threads_used) *threads_used = used;
    return RTX_OK;
}

/* ------------------------------------------- known-answer entry points ---- */
void orc_ray_init(const double origin[3], const double dir[3], double out_dir[3], double out_inv[3])
{
    ray_t r = make_ray(ld3(origin), ld3(dir));
    st3(out_dir, r.direction);
    if (out_inv) st3(out_inv, r.inverse_direction);
}

int orc_aabb_enter_exit(const double bmin[3], const double bmax[3], const double origin[3],
                        const double dir[3], double enter[3], double exit_[3])
{
    ray_t r = make_ray(ld3(origin), ld3(dir));
    v3 e, x;
    int hit = aabb_enter_exit(bmin, bmax, &r, &e, &x);
    st3(enter, e); st3(exit_, x);
    return hit;
}

int orc_aabb_hit(const double bmin[3], const double bmax[3], const double origin[3], const double dir[3])
{
    ray_t r = make_ray(ld3(origin), ld3(dir));
    return aabb_hit(bmin, bmax, &r);
}

int orc_triangle_hit(const double pos[9], const double nrm[9], const double uvs[6],
                     const double origin[3], const double dir[3], double position[3],
                     double normal[3], double uv[2], double bary[3], double* depth)
{
    ray_t r = make_ray(ld3(origin), ld3(dir));
    hit_t h;
    memset(&h, 0, sizeof h);
    int ok = triangle_hit(pos, nrm, uvs, &r, &h);
    if (ok) { st3(position, h.position); st3(normal, h.normal); uv[0] = h.u; uv[1] = h.v; st3(bary, h.bary); *depth = h.depth; }
    return ok;
}

int orc_sphere_hit(const double center[3], double r2, const double nmat[16], const double origin[3],
                   const double dir[3], double position[3], double normal[3], double uv[2], double* depth)
{
    ray_t r = make_ray(ld3(origin), ld3(dir));
    hit_t h;
    memset(&h, 0, sizeof h);
    int ok = sphere_hit(center, r2, nmat, &r, &h);
    if (ok) { st3(position, h.position); st3(normal, h.normal); uv[0] = h.u; uv[1] = h.v; *depth = h.depth; }
    return ok;
}

/* the `entering` test of pathTrace / rayTrace for a given dot(normal, inverseDirection):
   normal = (x, 0, 0), inverseDirection = (1, 0, 0) */
int orc_entering_for_dot(double x)
{
    ray_t r;
    memset(&r, 0, sizeof r);
    r.inverse_direction = V(1.0, 0.0, 0.0);
    return is_entering(V(x, 0.0, 0.0), &r);
}

void orc_refraction_ray(const double n[3], const double d[3], double ior, int entering, double out[3], int* internal_only)
{
    int io = 0;
    st3(out, refraction_ray(ld3(n), ld3(d), ior, entering, &io));
    if (internal_only) *internal_only = io;
}

void orc_mirror_direction(const double d[3], const double n[3], double out[3])
{
    st3(out, mirror_direction(ld3(d), ld3(n)));
}

void orc_lighting(const double ray_dir[3], const double light_dir[3], const double normal[3],
                  const double uv[2], double attenuation, const rtx_material* m,
                  const rtx_scene_desc* s, const double light_color[3], double out[3])
{
    ray_t ray = make_ray(V(0, 0, 0), ld3(ray_dir));
    ray_t lray = make_ray(V(0, 0, 0), ld3(light_dir));
    st3(out, lighting(s, &ray, &lray, ld3(normal), uv[0], uv[1], attenuation, m, ld3(light_color)));
}

void orc_cook_torrance(const double in_dir[3], const double out_dir[3], const double normal[3],
                       const double uv[2], const double down[3], const rtx_material* m,
                       const rtx_scene_desc* s, const double f0[3], double out[3], double ks[3])
{
    ray_t in = make_ray(V(0, 0, 0), ld3(in_dir));
    ray_t o = make_ray(V(0, 0, 0), ld3(out_dir));
    v3 k;
    st3(out, cook_torrance(s, &in, &o, ld3(normal), uv[0], uv[1], ld3(down), m, ld3(f0), &k));
    st3(ks, k);
}

void orc_light_ray(const rtx_light* l, const double p[3], double u0, double u1, double dir[3])
{
    ray_t r = light_ray(l, ld3(p), u0, u1);
    st3(dir, r.direction);
}

double orc_light_distance(const rtx_light* l, const double origin[3], const double dir[3])
{
    ray_t r = make_ray(ld3(origin), ld3(dir));
    return light_distance(l, &r);
}

double orc_light_attenuation(const rtx_light* l, double distance, const double att[3])
{
    return light_attenuation(l, distance, att);
}

void orc_material_color(const rtx_material* m, const rtx_scene_desc* s, const double uv[2], double out[3])
{
    st3(out, material_color(s, m, uv[0], uv[1]));
}
