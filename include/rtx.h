/*
 * rtx.h -- C ABI of the MI355X trace-loop library (librtx_hip.so).
 *
 * The reference renderer (noahwhygodwhy/Raytracer) has no FFI: its per-pixel
 * trace loop is a set of C++ call sites inside one executable.  This header is
 * the seam introduced between the reference's frame driver (main(),
 * Raytracer/Raytracer.cpp:595-752) and everything below it.  Each entry point
 * names the reference call site it replaces.
 *
 * All pointers are plain host pointers unless the name says `_device`.
 * All arrays are tightly packed, little endian.  Vectors are xyz triples.
 * Every function returns RTX_OK (0) or a negative rtx_status; the message for
 * the calling thread's last failure is available from rtx_last_error().
 * Nothing here exits the process or throws (the reference printf+exit(-1)s,
 * e.g. Raytracer/SquareLight.cpp:80-90).
 *
 * Current device: every entry point that takes a scene makes the scene's
 * device the calling thread's current HIP device (hipSetDevice) and leaves it
 * so; rtx_init and rtx_scene_create* do the same with the device rtx_init
 * chose.  A caller that drives several GPUs from one thread sets its own
 * device again after a call.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 1

typedef enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_INVALID = -1,   /* bad argument / malformed scene description   */
    RTX_ERR_DEVICE = -2,    /* HIP runtime failure                           */
    RTX_ERR_LIMIT = -3,     /* a documented size limit was exceeded          */
    RTX_ERR_NO_GPU = -4     /* no usable gfx950 device                       */
} rtx_status;

/* ---- shape tags (reference: virtual Shape::rayHit, Raytracer/Shape.hpp:52-65) */
enum { RTX_PRIM_TRIANGLE = 0, RTX_PRIM_SPHERE = 1, RTX_PRIM_BICONVEX = 2 };

/* ---- per-channel material sampler (reference: std::function members,
 *      Raytracer/Material.hpp:70-76; callables cannot cross a C ABI so the
 *      host classes record which factory produced them) */
enum {
    RTX_SAMPLER_CONST = 0,          /* no callback: use the constant field            */
    RTX_SAMPLER_RY_CHECKERS = 1,    /* ryCheckers10x10, Raytracer/Procedural.cpp:4-14 */
    RTX_SAMPLER_DOUBLE_CHECKERS = 2,/* doubleCheckers10x10, Procedural.cpp:16-25      */
    RTX_SAMPLER_TEXTURE_VEC3 = 3,   /* Texture::vec3Sample via getVec3Sampler (float) */
    RTX_SAMPLER_TEXTURE_DOUBLE = 4  /* Texture::doubleSample, Texture.cpp:56-64       */
};

typedef struct rtx_sampler {
    uint32_t kind;      /* RTX_SAMPLER_*                       */
    uint32_t texture;   /* index into rtx_scene_desc.textures  */
} rtx_sampler;

/* reference: class Material, Raytracer/Material.hpp:25-85 */
typedef struct rtx_material {
    double color[3];
    double ns;
    double ni;
    double transparency;
    double metalness;
    double smoothness;
    double emission[3];
    rtx_sampler color_fn;
    rtx_sampler ns_fn;
    rtx_sampler ni_fn;
    rtx_sampler transparency_fn;
    rtx_sampler metalness_fn;
    rtx_sampler smoothness_fn;
} rtx_material;

/* reference: class Texture, Raytracer/Texture.hpp:17-33 (RGB8, 3 channels forced) */
typedef struct rtx_texture {
    uint32_t width;
    uint32_t height;
    uint64_t offset;    /* byte offset of texel (0,0) in rtx_scene_desc.texels */
} rtx_texture;

enum { RTX_LIGHT_POINT = 0, RTX_LIGHT_DIRECTIONAL = 1, RTX_LIGHT_SQUARE = 2 };

/* reference: Light / PointLight / DirectionalLight / SquareLight,
 * Raytracer/Light.hpp:13-26, PointLight.cpp, DirectionalLight.cpp, SquareLight.cpp */
typedef struct rtx_light {
    uint32_t kind;          /* RTX_LIGHT_*                                          */
    uint32_t multisample;   /* Light::multisample (true only for SquareLight)       */
    double color[3];
    double position[3];     /* point / square                                       */
    double direction[3];    /* directional: reverseDirection (unit, pointing to the
                               light); square: unit normal                          */
    double right[3];        /* square frame, SquareLight.cpp:16-23                  */
    double up[3];
    double bottom_left[3];  /* SquareLight.cpp:34                                   */
    double width;
    double height;
} rtx_light;

/*
 * One frame's flattened scene = reference FrameInfo (Raytracer/Raytracer.hpp:60-67,
 * plus `lights` from Raytracer-Noah-Desktop.cpp:104-111) after redoAABBs +
 * buildKDTree (Raytracer.cpp:634-636) with Animatable::getModel(t) evaluated.
 *
 * KD child encoding: c >= 0 is branch index c; c < 0 is leaf index ~c.
 * Leaves are numbered in the walk order of printKDTree (KDTree.cpp:137-155:
 * lesser subtree first), so a ray's leaf-visit sequence is strictly increasing.
 * Primitive ids 0..n_shapes-1 are positions in FrameInfo::shapes; ids
 * n_shapes..n_prims-1 are the private inner spheres of Biconvex lenses
 * (Biconvex.cpp:43-44), which the reference reports as HitResult::shape.
 */
typedef struct rtx_scene_desc {
    uint32_t abi_version;       /* RTX_ABI_VERSION */

    /* KD tree (KDBranch / KDLeaf, Raytracer/KDTree.hpp:31-55) */
    int32_t  kd_root;
    uint32_t n_branches;
    uint32_t n_leaves;
    uint32_t n_leaf_refs;
    const double*   branch_min;     /* [n_branches*3]  KDBranch::area.min        */
    const double*   branch_max;     /* [n_branches*3]  KDBranch::area.max        */
    const double*   branch_value;   /* [n_branches]                              */
    const int32_t*  branch_axis;    /* [n_branches]    0,1,2                     */
    const int32_t*  branch_lesser;  /* [n_branches]    child encoding            */
    const int32_t*  branch_greater; /* [n_branches]                              */
    const uint32_t* leaf_first;     /* [n_leaves]      into leaf_refs            */
    const uint32_t* leaf_count;     /* [n_leaves]                                */
    const uint32_t* leaf_refs;      /* [n_leaf_refs]   primitive ids             */

    /* primitives */
    uint32_t n_shapes;              /* entries of FrameInfo::shapes              */
    uint32_t n_prims;               /* n_shapes + hidden lens spheres            */
    const uint8_t*  prim_type;      /* [n_prims] RTX_PRIM_*                      */
    const uint32_t* prim_index;     /* [n_prims] index into the per-type arrays  */
    const uint32_t* prim_material;  /* [n_prims] (ignored for BICONVEX)          */
    const double*   prim_aabb_min;  /* [n_prims*3] Shape::boundingBox            */
    const double*   prim_aabb_max;  /* [n_prims*3]                               */

    uint32_t n_triangles;
    const double* tri_position;     /* [n_triangles*9]  a,b,c                    */
    const double* tri_normal;       /* [n_triangles*9]  per-vertex normals       */
    const double* tri_uv;           /* [n_triangles*6]  per-vertex texCoords     */

    uint32_t n_spheres;
    const double* sphere_center;    /* [n_spheres*3] transformPos(origin, model) */
    const double* sphere_radius;    /* [n_spheres]                               */
    const double* sphere_r2;        /* [n_spheres]                               */
    const double* sphere_normal_matrix; /* [n_spheres*16] inverse(transpose(model)),
                                       column major (glm layout)                 */

    uint32_t n_lenses;
    const uint32_t* lens_spheres;   /* [n_lenses*2] primitive ids of s1, s2      */

    uint32_t n_materials;
    const rtx_material* materials;

    uint32_t n_textures;
    const rtx_texture* textures;
    uint64_t n_texel_bytes;
    const uint8_t* texels;

    uint32_t n_lights;
    const rtx_light* lights;
} rtx_scene_desc;

/* camera basis computed by the frame driver, Raytracer/Raytracer.cpp:642-660 */
typedef struct rtx_camera {
    double eye[3];
    double right[3];
    double up[3];
    double forward[3];
    double focal;
} rtx_camera;

enum { RTX_INTEGRATOR_PATH = 0,     /* pathTrace, Raytracer/Raytracer.cpp:217-355       */
       RTX_INTEGRATOR_WHITTED = 1   /* rayTrace, Raytracer-Noah-Desktop.cpp:452-564     */ };

enum {
    RTX_FLAG_LINEAR_SCAN     = 1u << 0, /* !KDTRACE: rayHitListOfShapes over all shapes
                                           (Raytracer.cpp:114-118)                         */
    RTX_FLAG_BASIC           = 1u << 1, /* BASIC_BITCH: colour of the first hit
                                           (Raytracer.cpp:228-230)                         */
    RTX_FLAG_NO_LIGHTING     = 1u << 2, /* Whitted without LIGHTING
                                           (Raytracer-Noah-Desktop.cpp:555)                */
    RTX_FLAG_WHITTED_REFLECT = 1u << 3, /* extension, not in the reference: a mirror ray
                                           as in pathTrace case 2 (Raytracer.cpp:284-285)
                                           blended by smoothness                           */
    RTX_FLAG_COUNTERS        = 1u << 4, /* fill the traversal work counters of rtx_stats   */
    RTX_FLAG_REFERENCE_ORDER = 1u << 5, /* walk the KD tree node for node as traverseKDTree
                                           does (KDTree.cpp:158-204).  Without it the
                                           render path walks the same tree but skips
                                           nodes that provably cannot change the closest
                                           hit and stops shadow rays at the first occluder;
                                           hit ids, depths and images are identical, only
                                           the work (and the counters) shrink             */
    RTX_FLAG_KD_CULLED       = 1u << 6, /* find candidates by walking the reference's KD
                                           tree with culling instead of through the
                                           library's own BVH (same results; for comparison) */
    RTX_FLAG_STAGE_TIMES     = 1u << 7, /* fill the per-stage ms_* fields of rtx_stats     */
    RTX_FLAG_RENDER_WALK     = 1u << 8, /* rtx_trace_closest only: use the render path's
                                           walk instead of the reference-order one; the
                                           leaf outputs and counters then describe that
                                           walk's own work, not the reference's            */
    RTX_FLAG_SORT_MATERIAL   = 1u << 9, /* pathTrace levels are shaded in material order
                                           (rays binned by the material they hit) instead
                                           of queue order; same image                      */
    RTX_FLAG_GLOBAL_LIGHTING = 1u << 10,/* Whitted with GLOBAL_LIGHTING: the hemisphere gather
                                           inside the light loop
                                           (Raytracer-Noah-Desktop.cpp:69,499-524), with
                                           rtx_render_desc.global_samples = MONTE_CARLO_SAMPLES */
    RTX_FLAG_REFILL_LANES    = 1u << 11,/* the BVH walk refills finished lanes from the queue
                                           instead of keeping one ray per lane for the life of
                                           a wavefront (same results; fewer idle lanes but
                                           measured slower, for comparison)                   */
    RTX_FLAG_TRACE_TIMES     = 1u << 12,/* fill only ms_traverse and ms_shadow of the statistics: the
                                           traversal launches; fewer events on the stream than
                                           RTX_FLAG_STAGE_TIMES                               */
    RTX_FLAG_SERIAL_CHUNKS   = 1u << 13,/* render the tile's chunks one after the other on the
                                           scene's stream instead of two at a time on the
                                           library's two streams (same image; launches then do
                                           not overlap, so their durations can be measured)   */
    RTX_FLAG_OVERWRITE       = 1u << 14,/* the call's first pass stores its result in accum
                                           instead of adding to it (as if the tile had been
                                           cleared first: clearBuffers, Raytracer.cpp:53-58),
                                           which saves the caller a fill of the buffer       */
    RTX_FLAG_STORE_CAMERA_RAYS = 1u << 15,/* a generate kernel writes the camera rays into the
                                           first ray queue and the kernels of level 0 read them
                                           back, as every later level does.  Without it those
                                           kernels compute the camera ray from the sample slot
                                           (same arithmetic, same image; 84 bytes per ray less
                                           to write and to read three times).  Implied by
                                           RTX_FLAG_REFILL_LANES                              */
    RTX_FLAG_ACCUMULATE_SAMPLES = 1u << 16,/* every term of a sub-sample is added to a cleared
                                           accumulator slot and a resolve launch sums the slots
                                           of a pixel.  Without it a slot whose whole value comes
                                           from one thread (pathTrace: always; rayTrace: when no
                                           material refracts) is stored once, and with one
                                           sub-sample per pixel written to the framebuffer at
                                           once.  Same image                                  */
    RTX_FLAG_QUEUE_ORDER     = 1u << 18,/* the rays of the levels below the camera's are walked in queue
                                           order.  Without it a level of 65 536 rays or more is walked in
                                           the order of the rays' direction bins (neighbouring lanes then
                                           follow similar paths; same hits, same image)            */
    RTX_FLAG_STORE_LIGHT_RAYS = 1u << 17  /* the first shading half stores every light ray's
                                           direction and light distance in the shadow queue.
                                           Without it, in a scene with one Point or Directional
                                           light (no random numbers: the ray is a function of its
                                           origin), only origins are stored and the traversal and
                                           the second shading half recompute the rest.  Same
                                           image                                               */
};

/* The reference's compile-time configuration (Raytracer.cpp:9-19,27-40;
 * Raytracer-Noah-Desktop.cpp:61-80,101) as run-time fields. */
typedef struct rtx_render_desc {
    uint32_t integrator;            /* RTX_INTEGRATOR_*                                 */
    uint32_t flags;                 /* RTX_FLAG_*                                       */
    uint32_t width;                 /* frameX                                           */
    uint32_t height;                /* frameY                                           */
    uint32_t tile_x0, tile_y0;      /* sub-rectangle rendered by this call              */
    uint32_t tile_w, tile_h;        /*   (0,0,width,height for the full frame)          */
    uint32_t pixel_multisample_n;   /* PIXEL_MULTISAMPLE_N (n*n sub-samples)            */
    uint32_t light_multisample_n;   /* LIGHT_MULTISAMPLES_N                             */
    uint32_t max_depth;             /* MAX_PATH (path) / MAXBOUNCES (Whitted)           */
    uint32_t first_pass;            /* index of the first Monte-Carlo pass of this call */
    uint32_t passes;                /* number of passes accumulated by this call        */
    uint32_t frame;                 /* RNG key component                                */
    uint64_t seed;                  /* RNG key component (reference: srand(0u))         */
    double bias;
    double clear_color[3];
    double attenuation[3];          /* LIGHT_CONTSTANT, LIGHT_LINEAR, LIGHT_QUADRATIC   */
    rtx_camera camera;
    uint32_t global_samples;        /* MONTE_CARLO_SAMPLES of the Whitted variant
                                       (Raytracer-Noah-Desktop.cpp:73): gather rays per light
                                       and hit at layer 0, divided by layer+1 below (:500);
                                       read only with RTX_FLAG_GLOBAL_LIGHTING             */
    uint32_t tile_row_group;        /* 0: the tile's rows are consecutive frame rows.  Otherwise the
                                       tile is every tile_row_stride-th group of tile_row_group rows:
                                       tile row r is frame row
                                         tile_y0 + (r / tile_row_group) * tile_row_stride + r % tile_row_group
                                       (one rank's interleaved share of a frame in one call; accum
                                       holds the tile's rows packed, tile_w * tile_h * 3 floats)  */
    uint32_t tile_row_stride;       /* >= tile_row_group when that is not 0                   */
    uint32_t reserved;              /* 0 */
} rtx_render_desc;

typedef struct rtx_stats {
    /* closest-hit queries (= shootRay calls) by kind; always filled */
    uint64_t rays_primary;
    uint64_t rays_secondary;
    uint64_t rays_shadow;
    /* per-ray work counters summed over all queries; filled with RTX_FLAG_COUNTERS.
       With RTX_FLAG_REFERENCE_ORDER they are the reference's own work. */
    uint64_t branch_visits;         /* KDBranch slab tests                       */
    uint64_t leaf_visits;
    uint64_t aabb_tests;            /* Shape::rayAABB calls                      */
    uint64_t triangle_tests;        /* Triangle::rayHit calls                    */
    uint64_t sphere_tests;          /* Sphere::rayHit calls (2 per lens)         */
    uint64_t hits_accepted;         /* closer-hit replacements                   */
    uint64_t content_tests;         /* content-box / BVH child-box tests         */
    uint64_t fallback_rays;         /* rays the fast walk could not certify and
                                       re-traced in reference order               */
    /* device time per stage in milliseconds (HIP events on the library stream);
       ms_total always, the per-stage sums with RTX_FLAG_STAGE_TIMES (ms_traverse and
       ms_shadow also with RTX_FLAG_TRACE_TIMES) */
    double ms_generate;
    double ms_traverse;
    double ms_shade;
    double ms_shadow;
    double ms_resolve;
    double ms_total;
    uint32_t traverse_launches;
    uint32_t max_queue;
    /* SIMD efficiency of the traversal launches, with RTX_FLAG_COUNTERS: a step is a node visit
       or a shape test; lane_steps sums them over all rays, wave_steps sums 64 * (the largest
       lane count of each wavefront).  lane_steps / wave_steps = fraction of lanes doing work. */
    uint64_t wave_steps;
    uint64_t lane_steps;
} rtx_stats;

typedef struct rtx_scene rtx_scene;

/* Select the device and create library state.  device < 0 keeps the current one. */
int rtx_init(int device);

/* Number of visible HIP devices (0 when there is none). */
int rtx_device_count(void);

/* Upload one frame's flattened scene.  Replaces: FrameInfo handed to the
 * per-pixel lambda (Raytracer.cpp:553-557,634-636).  The library copies
 * everything; the caller keeps ownership of the host arrays. */
int rtx_scene_create(const rtx_scene_desc* desc, rtx_scene** out);

/* The same for a scene that will be rendered many times (a camera moving through fixed
 * geometry).  No counterpart in the reference, which rebuilds its FrameInfo for every frame: this
 * variant spends more host time once (about 1 ms per thousand shapes) to find out for which
 * shapes the KD tree lists every leaf their box reaches, and the traversal's membership check of a
 * hit then skips the search of that shape's leaf list.  Results are identical to rtx_scene_create's. */
int rtx_scene_create_static(const rtx_scene_desc* desc, rtx_scene** out);

/* Run later work for this scene on `hip_stream` (a hipStream_t; NULL = default). */
int rtx_scene_set_stream(rtx_scene* scene, void* hip_stream);

/*
 * Asynchronous calls for frame sequences.  The reference's frame loop (Raytracer.cpp:595-752)
 * finishes one frame before it starts the next; a caller that keeps its buffers on the device
 * does not need the host to wait in between.  With enabled != 0, rtx_render_device and
 * rtx_present_device return once their work is queued on the scene's stream (a render call can
 * still wait inside for a queue length it needs) and take no `stats`; rtx_scene_wait waits for
 * everything queued so far and returns the statistics summed over the render calls since the
 * last hand-out (ms_total stays 0).  The host-buffer calls rtx_render / rtx_present are refused on an
 * asynchronous scene.  Switching back (enabled = 0) waits first.
 */
int rtx_scene_set_async(rtx_scene* scene, int enabled);
int rtx_scene_wait(rtx_scene* scene, rtx_stats* totals);

/*
 * Another handle on the same device copy of the scene, with a stream, queues and statistics of
 * its own (no counterpart in the reference, which has one frame in flight).  Frames rendered
 * through different handles on different streams run concurrently: the tail of one frame's
 * launches is filled by the next frame's.  The device arrays are freed with the last handle.
 */
int rtx_scene_clone(const rtx_scene* scene, rtx_scene** out);

void rtx_scene_destroy(rtx_scene* scene);

/*
 * Batched shootRay (Raytracer.cpp:108-120): for each i builds
 * Ray(origin_i, direction_i) (Ray.cpp:3-8, normalising the direction) and
 * returns the closest hit.  Output arrays may be NULL except prim and depth.
 *   prim[i]        primitive id or -1           depth[i]   HitResult::depth (+inf on miss)
 *   position/normal [n*3], uv [n*2], bary [n*3] HitResult fields of the winner
 *   leaf_count[i]  number of KDLeaf visits      leaf_hash[i]  FNV-1a-64 over the leaf ids
 *   leaf_seq       [n*leaf_seq_cap] first leaf ids visited, rest untouched
 *   counters       [n*6] branch, leaf, aabb, triangle, sphere, accepted per ray
 * flags: RTX_FLAG_LINEAR_SCAN; RTX_FLAG_RENDER_WALK (optionally with RTX_FLAG_KD_CULLED).
 * Without RTX_FLAG_RENDER_WALK the walk is the reference's, node for node.
 */
int rtx_trace_closest(rtx_scene* scene, const double* origin, const double* direction, size_t n,
                      uint32_t flags, int32_t* prim, double* depth, double* position,
                      double* normal, double* uv, double* bary, uint32_t* leaf_count,
                      uint64_t* leaf_hash, uint32_t* leaf_seq, uint32_t leaf_seq_cap,
                      uint32_t* counters);

/*
 * The per-pixel lambda plus integrator (Raytracer.cpp:672-716 calling pathTrace,
 * or rayTrace of Raytracer-Noah-Desktop.cpp:452-564), for `passes` passes of the
 * Monte-Carlo loop (Raytracer.cpp:666).  `accum` is the tile's slice of the
 * reference's drawBuffer: float RGB, tile_w*tile_h*3, row 0 = bottom row; each
 * pass performs accum[p] += float(colorAcum / (n*n)) in pass order (FP32, as
 * Raytracer.hpp:77-78).  The caller initialises it (clearBuffers, :53-58).
 * rtx_render takes a host buffer; rtx_render_device takes device memory and is
 * what a caller with resident buffers (a multi-GPU gather) uses.  Both return
 * after the work has completed on the scene's stream.
 */
int rtx_render(rtx_scene* scene, const rtx_render_desc* desc, float* accum, rtx_stats* stats);
int rtx_render_device(rtx_scene* scene, const rtx_render_desc* desc, float* accum_device,
                      rtx_stats* stats);

/*
 * rtx_render_device for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the
 * reference.  The ray queues, light-ray queues, sample slots and counters of the call live in a
 * static device array of RTX_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until
 * the process ends; the call allocates, creates and frees nothing, reads nothing back and does
 * not wait: it returns once the work is queued on the scene's stream (synchronise that stream
 * before reading accum_device).  Same image, bit for bit, as rtx_render_device.
 * Limits: the tile is rendered as one chunk, so
 *     768 + 272.2 * S + 40 * S * L + 36 * S * G   (each array rounded up to 256 bytes)
 * must not exceed RTX_NOALLOC_SCRATCH_BYTES, with S = tile_w * tile_h * n * n sample slots, L light
 * samples per hit (Whitted) and G gather rays per hit (GLOBAL_LIGHTING): e.g. 256 x 256 pixels at
 * 1 spp with four light samples, or 128 x 96 at 4 spp path traced.  Larger tiles get
 * RTX_ERR_LIMIT: split them.  Flags that need a value read back are refused with
 * RTX_ERR_INVALID: RTX_FLAG_COUNTERS, _STAGE_TIMES, _TRACE_TIMES, _WHITTED_REFLECT,
 * _SORT_MATERIAL, _REFILL_LANES.  No statistics are kept.  A pathTrace frame launches all
 * max_depth levels (no early stop), so pass the depth you need.
 * All calls on one GPU share that GPU's array: calls that overlap there (other streams, other
 * threads, other scenes) race.  Serialise them.
 */
#define RTX_NOALLOC_SCRATCH_BYTES (64u << 20)
int rtx_render_device_noalloc(rtx_scene* scene, const rtx_render_desc* desc, float* accum_device);

/*
 * The present step: what the reference does to drawBuffer between a pass and the PNG.
 *   frameBuffer = drawBuffer / float(passes); with RTX_TONEMAP_CBRT each channel goes
 *   through cbrt in FP32 (Raytracer.cpp:720-725; the Whitted variant shows the buffer as
 *   is); the pass-through shaders draw it into an sRGB framebuffer (GL_FRAMEBUFFER_SRGB,
 *   Raytracer.cpp:422: clamp to [0,1], sRGB transfer function, 8 bits); saveImage reads
 *   the pixels back and writes them flipped (Raytracer.cpp:90-105).
 * `accum` is a width*height*3 float buffer, row 0 = bottom row; `rgb8` receives
 * width*height*3 bytes, with flip != 0 top row first (the PNG's order).  The _device form
 * takes device pointers and runs on the scene's stream (returns after completion).
 */
enum { RTX_TONEMAP_NONE = 0, RTX_TONEMAP_CBRT = 1 };
int rtx_present(rtx_scene* scene, const float* accum, uint32_t width, uint32_t height, uint32_t passes,
                uint32_t tonemap, int flip, uint8_t* rgb8);
int rtx_present_device(rtx_scene* scene, const float* accum_device, uint32_t width, uint32_t height,
                       uint32_t passes, uint32_t tonemap, int flip, uint8_t* rgb8_device);

/* The draw-index'th uniform of one sample's random stream, in [0,1]
 * (replaces rand()/RAND_MAX, Raytracer.cpp:65-67).  Exposed so a host-side
 * reference can be driven with identical samples. */
double rtx_uniform(uint64_t seed, uint32_t frame, uint32_t pixel, uint32_t sample,
                   uint32_t branch, uint32_t draw_index);

const char* rtx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
