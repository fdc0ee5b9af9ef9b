/*
 * rt_oracle.h -- CPU restatement of the reference's per-pixel trace loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Pinning status: the reference ships no tests, golden images or fixtures for
 * this path, and it does not build in this image (glm, assimp, GLFW and ppl.h
 * are absent), so the oracle is pinned only by the known-answer values the
 * survey recorded from the reference's own functions (tests/golden/
 * reference_kat.json).  Anything those values do not cover is PARITY UNPINNED.
 *
 * The scene/render descriptions are the POD structs of include/rtx.h so that
 * the oracle and the HIP library consume byte-identical inputs.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* batched shootRay; same contract as rtx_trace_closest */
int orc_trace_closest(const rtx_scene_desc* s, const double* origin, const double* direction,
                      size_t n, uint32_t flags, int32_t* prim, double* depth, double* position,
                      double* normal, double* uv, double* bary, uint32_t* leaf_count,
                      uint64_t* leaf_hash, uint32_t* leaf_seq, uint32_t leaf_seq_cap,
                      uint32_t* counters);

/* per-pixel lambda + integrator; same contract as rtx_render.  threads <= 0
 * keeps OpenMP's default (OMP_NUM_THREADS); the number used is returned in *threads_used
 * (may be NULL).  rgb64, when not NULL, receives the FP64 colorAcum/(n*n) of the
 * LAST pass per pixel (tile_w*tile_h*3) so tests can compare before FP32 rounding. */
int orc_render(const rtx_scene_desc* s, const rtx_render_desc* d, float* accum, double* rgb64,
               rtx_stats* stats, int threads, int* threads_used);

/* GLOBAL_LIGHTING gather (Raytracer-Noah-Desktop.cpp:499-524): on != 0 makes orc_render run
 * the gather's nested rayTrace calls in full, as the reference does, although only their
 * hit/position can reach the image; the default evaluates just the hit.  Process-wide. */
void orc_set_global_full_recursion(int on);

/* camera ray of one sub-sample: Raytracer.cpp:676-700 */
void orc_camera_ray(const rtx_render_desc* d, uint32_t px, uint32_t py, uint32_t sx, uint32_t sy,
                    double origin[3], double direction[3]);

double orc_uniform(uint64_t seed, uint32_t frame, uint32_t pixel, uint32_t sample,
                   uint32_t branch, uint32_t draw_index);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* present step (Raytracer.cpp:720-725 cbrt in FP32 when tonemap != 0; :422 sRGB framebuffer;
 * :103 vertical flip): accum is width*height*3 floats, row 0 = bottom; rgb8 gets the bytes */
void orc_present(const float* accum, uint32_t width, uint32_t height, uint32_t passes, uint32_t tonemap,
                 int flip, uint8_t* rgb8);

/* ---- function-level entry points for known-answer tests ------------------ */
void orc_ray_init(const double origin[3], const double dir[3], double out_dir[3], double out_inv[3]);
int  orc_aabb_enter_exit(const double bmin[3], const double bmax[3], const double origin[3],
                         const double dir[3], double enter[3], double exit_[3]);
int  orc_aabb_hit(const double bmin[3], const double bmax[3], const double origin[3],
                  const double dir[3]);
int  orc_triangle_hit(const double pos[9], const double nrm[9], const double uvs[6],
                      const double origin[3], const double dir[3], double position[3],
                      double normal[3], double uv[2], double bary[3], double* depth);
int  orc_sphere_hit(const double center[3], double r2, const double nmat[16],
                    const double origin[3], const double dir[3], double position[3],
                    double normal[3], double uv[2], double* depth);
void orc_solve_quadratic(double a, double b, double c, double* r1, double* r2);
/* 1: repeat the work the reference repeats (leaf vector copied per visit, KDTree.cpp:206; model
   matrix applied per sphere test and inverted per sphere hit, Sphere.cpp:47-48,74) so that a timed
   CPU baseline carries the reference's own overheads.  Results are unchanged. */
void orc_set_reference_costs(int on);
int orc_entering_for_dot(double x);     /* acos(x) < pi/2.0, as Raytracer.cpp:248-249 evaluates it */
void orc_refraction_ray(const double n[3], const double d[3], double ior, int entering,
                        double out[3], int* internal_only);
void orc_mirror_direction(const double d[3], const double n[3], double out[3]);
void orc_lighting(const double ray_dir[3], const double light_dir[3], const double normal[3],
                  const double uv[2], double attenuation, const rtx_material* m,
                  const rtx_scene_desc* s, const double light_color[3], double out[3]);
void orc_cook_torrance(const double in_dir[3], const double out_dir[3], const double normal[3],
                       const double uv[2], const double down[3], const rtx_material* m,
                       const rtx_scene_desc* s, const double f0[3], double out[3], double ks[3]);
void orc_light_ray(const rtx_light* l, const double p[3], double u0, double u1, double dir[3]);
double orc_light_distance(const rtx_light* l, const double origin[3], const double dir[3]);
double orc_light_attenuation(const rtx_light* l, double distance, const double att[3]);
void orc_material_color(const rtx_material* m, const rtx_scene_desc* s, const double uv[2],
                        double out[3]);

#ifdef __cplusplus
}
#endif
#endif
