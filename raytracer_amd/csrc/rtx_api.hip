// rtx_api.hip -- the C ABI of include/rtx.h over the kernels of rtx_kernels.hpp.
//
// Host responsibilities: validate the flattened scene (a malformed tree would make
// a kernel read out of bounds), lay it out in device memory, own the ray queues,
// and drive the per-level loop of the wavefront pipeline.  There is no CPU
// implementation of any stage here: without a working HIP device every entry
// point fails with RTX_ERR_NO_GPU / RTX_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <memory>
#include <utility>
#include <string>
#include <mutex>
#include <vector>

#include "rtx_kernels.hpp"
#include <future>

#include "rtx_bvh.hpp"

using namespace rtx;

namespace {

thread_local std::string g_error;
// The device rtx_init chose.  HIP's current device is a per-thread setting that starts at 0 on
// every new thread, so each entry point selects its device itself: rtx_scene_create this one,
// everything that takes a scene the scene's.  -1 (no rtx_init yet): the calling thread's current.
std::atomic<int> g_device{-1};

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(RTX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define USE_DEVICE(dev)                                                                       \
    do {                                                                                      \
        if ((dev) >= 0) HIP_TRY(hipSetDevice(dev));                                           \
    } while (0)

struct DeviceBuffers {
    std::vector<void*> ptrs;
    // Copies go through a non-blocking stream of their own: a copy on the default stream would
    // wait for every kernel the process has in flight, and a scene is typically uploaded while
    // the previous frame still renders.  finish() waits for the copies and gives the stream back.
    hipStream_t copy_stream = nullptr;
    ~DeviceBuffers()
    {
        (void)finish();
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t upload(const T* host, size_t count, const T** out)
    {
        *out = nullptr;
        if (count == 0) return hipSuccess;
        hipError_t e = hipSuccess;
        if (!copy_stream) e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        void* p = nullptr;
        e = hipMalloc(&p, count * sizeof(T));
        if (e != hipSuccess) return e;
        ptrs.push_back(p);
        e = hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, copy_stream);
        *out = static_cast<const T*>(p);
        return e;
    }
    hipError_t finish()
    {
        if (!copy_stream) return hipSuccess;
        hipError_t e = hipStreamSynchronize(copy_stream);
        (void)hipStreamDestroy(copy_stream);
        copy_stream = nullptr;
        return e;
    }
};

// Growable device workspace for the ray queues.
struct Workspace {
    size_t capacity = 0;            // rays per queue
    size_t shadow_capacity = 0;
    size_t gather_capacity = 0;
    char* base = nullptr;
    char* shadow_base_mem = nullptr;
    char* gather_base_mem = nullptr;
    RayQueue q[2];
    ShadowQueue sq;
    GatherQueue gq;
    uint32_t* cand = nullptr;               // per traced ray: candidate left by k_trace_refill for k_verify
    size_t cand_capacity = 0;
    uint32_t* hit_index = nullptr;          // per ray: compact index among the level's hits
    uint32_t* perm = nullptr;               // shading order (material sort)
    uint32_t* bins = nullptr;               // kBins bin totals of the sort in progress (2 * kBins words kept)
    uint32_t* tile_hist = nullptr;          // kBins x tiles: per bin and tile of kTileRays rays, count / first place
    double* sample = nullptr;               // 3 per slot
    uint32_t* counts = nullptr;             // ring of four (queue length, lit hits) pairs, indexed by level % 4
    unsigned long long* work = nullptr;     // 8 work counters, [8] shadow rays, [9] gather rays, [10..11] SIMD efficiency, [12] secondary rays
    uint32_t* host_counts = nullptr;        // pinned
    unsigned long long* host_work = nullptr;
    bool borrowed = false;                  // laid out in memory this object does not own: nothing to free
    ~Workspace()
    {
        if (borrowed) return;
        if (base) (void)hipFree(base);
        if (shadow_base_mem) (void)hipFree(shadow_base_mem);
        if (gather_base_mem) (void)hipFree(gather_base_mem);
        if (cand) (void)hipFree(cand);
        if (counts) (void)hipFree(counts);
        if (bins) (void)hipFree(bins);
        if (work) (void)hipFree(work);
        if (host_counts) (void)hipHostFree(host_counts);
        if (host_work) (void)hipHostFree(host_work);
    }
};

template <class T>
T* carve(char*& p, size_t count)
{
    T* r = reinterpret_cast<T*>(p);
    p += ((count * sizeof(T) + 255) / 256) * 256;
    return r;
}

size_t queue_bytes(size_t cap)
{
    auto al = [](size_t b) { return ((b + 255) / 256) * 256; };     // as carve() rounds
    size_t per_queue = 9 * al(cap * 8) + 3 * al(cap * 4) + al(cap * 4) + al(cap * 8);
    return 2 * per_queue + 2 * al(cap * 4) + al(cap * 3 * 8) + 6 * al(cap * 8) + al((size_t)kBins * ((cap + kTileRays - 1) / kTileRays) * 4);
}

// Device time per stage: event pairs are recorded on the library stream without
// waiting and read back once the stream has drained.
struct StageTimer {
    bool on;
    hipStream_t stream;
    struct Span { hipEvent_t a, b; double* acc; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    hipEvent_t cur = nullptr;
    StageTimer(bool on = false, hipStream_t s = nullptr) : on(on), stream(s) {}
    ~StageTimer()
    {
        for (Span& sp : spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
        if (cur) (void)hipEventDestroy(cur);
    }
    void begin()
    {
        if (!on) return;
        (void)hipEventCreate(&cur);
        (void)hipEventRecord(cur, stream);
    }
    void end(double* acc)
    {
        if (!on || !cur) return;
        hipEvent_t b = nullptr;
        (void)hipEventCreate(&b);
        (void)hipEventRecord(b, stream);
        spans.push_back(Span{cur, b, acc});
        cur = nullptr;
    }
    // call after the stream has been synchronised
    void collect()
    {
        for (Span& sp : spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) *sp.acc += ms;
            (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b);
        }
        spans.clear();
    }
};

}  // namespace

struct rtx_scene {
    DScene d;
    std::shared_ptr<DeviceBuffers> buffers;     // the scene arrays in HBM; shared by the handles rtx_scene_clone makes
    // Two lanes: a frame's chunks are rendered two at a time on the library's own streams, so that
    // while one chunk's launch drains (a traversal launch ends in a tail of a few long rays) the
    // other chunk's kernels fill the machine.  A lane owns its queues and counters.
    Workspace ws[2];
    hipStream_t lane_stream[2] = {nullptr, nullptr};
    hipEvent_t fork_event = nullptr, join_event[2] = {nullptr, nullptr};
    // Statistics of the render calls since they were last handed out: with asynchronous calls
    // (rtx_scene_set_async) that is several frames.  Ray totals by kind and the work counters
    // accumulate on the device (Workspace::work), the rest here.
    bool async = false;
    rtx_stats pending;
    StageTimer lane_timer[2], lane_trace_timer[2];
    rtx_scene() { std::memset(&pending, 0, sizeof pending); }
    ~rtx_scene()
    {
        for (int k = 0; k < 2; ++k) if (lane_stream[k]) (void)hipStreamSynchronize(lane_stream[k]);
        for (int k = 0; k < 2; ++k) {
            if (lane_stream[k]) (void)hipStreamDestroy(lane_stream[k]);
            if (join_event[k]) (void)hipEventDestroy(join_event[k]);
        }
        if (fork_event) (void)hipEventDestroy(fork_event);
    }
    hipStream_t stream = nullptr;
    int device = 0;
    uint32_t kd_depth = 0;
    uint32_t bvh_depth = 0;
    const float* srgb_thresholds = nullptr;     // device copy of srgb_thresholds(), for k_present
    bool light_rays_plain = false;  // one light, Point or Directional, one sample per hit
    bool may_refract = false;       // some material can have transparency != 0 (rayTrace spawns a child then)
    std::vector<uint8_t> light_multisample;     // Light::multisample per light, kept on the host
    int compute_units = 0;
    uint32_t refill_range = 256;    // most rays a wave of the refilled walk owns
    uint32_t refill_at = 16;        // parked lanes that trigger a store + fetch
};

namespace {

// Where the 8-bit sRGB value of a framebuffer write steps up (GL_FRAMEBUFFER_SRGB, Raytracer.cpp:422):
// table[b] = the smallest float in [0, 1] whose conversion clamp -> transfer function ->
// round gives b or more.  Positive floats order like their bit patterns, so each threshold is
// found by bisection on the bits.
const float* srgb_thresholds()
{
    static float table[256];
    static const bool ready = [] {
        auto convert = [](float c) -> int {
            float e = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
            return (int)(e * 255.0f + 0.5f);
        };
        auto as_float = [](uint32_t bits) { float f; std::memcpy(&f, &bits, 4); return f; };
        const uint32_t one = 0x3f800000u;
        table[0] = -std::numeric_limits<float>::infinity();
        for (int b = 1; b < 256; ++b) {
            uint32_t lo = 0u, hi = one;         // convert(lo) < b <= convert(hi)
            while (hi - lo > 1u) {
                uint32_t mid = lo + (hi - lo) / 2u;
                if (convert(as_float(mid)) >= b) hi = mid; else lo = mid;
            }
            table[b] = as_float(hi);
        }
        return true;
    }();
    (void)ready;
    return table;
}

inline size_t al256(size_t b) { return ((b + 255) / 256) * 256; }
size_t shadow_bytes(size_t shadow_cap) { return 5 * al256(shadow_cap * 8); }
size_t gather_bytes(size_t gather_cap) { return 4 * al256(gather_cap * 8) + al256(gather_cap * 4); }

// the three parts of a workspace laid out from p on (queue_bytes / shadow_bytes / gather_bytes long)
void lay_out_queues(Workspace& w, char* p, size_t cap)
{
    for (int k = 0; k < 2; ++k) {
        RayQueue& q = w.q[k];
        q.ox = carve<double>(p, cap); q.oy = carve<double>(p, cap); q.oz = carve<double>(p, cap);
        q.dx = carve<double>(p, cap); q.dy = carve<double>(p, cap); q.dz = carve<double>(p, cap);
        q.wx = carve<double>(p, cap); q.wy = carve<double>(p, cap); q.wz = carve<double>(p, cap);
        q.slot = carve<uint32_t>(p, cap);
        q.branch = carve<uint32_t>(p, cap); q.counter = carve<uint32_t>(p, cap);
        q.hit_prim = carve<int32_t>(p, cap); q.hit_depth = carve<double>(p, cap);
    }
    w.hit_index = carve<uint32_t>(p, cap);
    // origins of the light rays, one per hit
    w.sq.ox = carve<double>(p, cap); w.sq.oy = carve<double>(p, cap); w.sq.oz = carve<double>(p, cap);
    w.gq.ox = carve<double>(p, cap); w.gq.oy = carve<double>(p, cap); w.gq.oz = carve<double>(p, cap);
    w.perm = carve<uint32_t>(p, cap);
    w.sample = carve<double>(p, cap * 3);
    w.tile_hist = carve<uint32_t>(p, (size_t)kBins * ((cap + kTileRays - 1) / kTileRays));
    w.capacity = cap;
}
void lay_out_shadow(Workspace& w, char* p, size_t shadow_cap)
{
    ShadowQueue& s = w.sq;
    s.dx = carve<double>(p, shadow_cap); s.dy = carve<double>(p, shadow_cap); s.dz = carve<double>(p, shadow_cap);
    s.limit = carve<double>(p, shadow_cap);
    s.hit_depth = carve<double>(p, shadow_cap);
    w.shadow_capacity = shadow_cap;
}
void lay_out_gather(Workspace& w, char* p, size_t gather_cap)
{
    GatherQueue& g = w.gq;
    g.dx = carve<double>(p, gather_cap); g.dy = carve<double>(p, gather_cap); g.dz = carve<double>(p, gather_cap);
    g.hit_depth = carve<double>(p, gather_cap);
    g.hit_prim = carve<int32_t>(p, gather_cap);
    w.gather_capacity = gather_cap;
}

int ensure_workspace(Workspace& w, size_t cap, size_t shadow_cap, size_t gather_cap, bool with_cand)
{
    if (!w.counts) {
        HIP_TRY(hipMalloc(&w.counts, 8 * sizeof(uint32_t)));
        HIP_TRY(hipMemset(w.counts, 0, 8 * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&w.work, 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(w.work, 0, 16 * sizeof(unsigned long long)));     // zero between collections
        HIP_TRY(hipMalloc(&w.bins, 2 * kBins * sizeof(uint32_t)));
        HIP_TRY(hipMemset(w.bins, 0, 2 * kBins * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc(&w.host_counts, 8 * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc(&w.host_work, 16 * sizeof(unsigned long long)));
        std::memset(w.host_work, 0, 16 * sizeof(unsigned long long));
    }
    if (cap > w.capacity) {
        if (w.base) { HIP_TRY(hipFree(w.base)); w.base = nullptr; w.capacity = 0; }
        HIP_TRY(hipMalloc(&w.base, queue_bytes(cap)));
        lay_out_queues(w, w.base, cap);
    }
    if (shadow_cap > w.shadow_capacity) {
        if (w.shadow_base_mem) { HIP_TRY(hipFree(w.shadow_base_mem)); w.shadow_base_mem = nullptr; w.shadow_capacity = 0; }
        HIP_TRY(hipMalloc(&w.shadow_base_mem, shadow_bytes(shadow_cap)));
        lay_out_shadow(w, w.shadow_base_mem, shadow_cap);
    }
    if (gather_cap > w.gather_capacity) {
        if (w.gather_base_mem) { HIP_TRY(hipFree(w.gather_base_mem)); w.gather_base_mem = nullptr; w.gather_capacity = 0; }
        HIP_TRY(hipMalloc(&w.gather_base_mem, gather_bytes(gather_cap)));
        lay_out_gather(w, w.gather_base_mem, gather_cap);
    }
    const size_t cand_cap = with_cand ? std::max(cap, std::max(shadow_cap, gather_cap)) : 0;
    if (cand_cap > w.cand_capacity) {
        if (w.cand) { HIP_TRY(hipFree(w.cand)); w.cand = nullptr; w.cand_capacity = 0; }
        HIP_TRY(hipMalloc(&w.cand, cand_cap * sizeof(uint32_t)));
        w.cand_capacity = cand_cap;
    }
    return RTX_OK;
}

// depth of the subtree under `node`, checking every index on the way; -1 = malformed
int validate_tree(const rtx_scene_desc* s, int32_t node, int depth, std::vector<uint8_t>& seen_branch)
{
    if (depth > 64) return -1;
    if (node < 0) {
        uint32_t leaf = (uint32_t)~node;
        if (leaf >= s->n_leaves) return -1;
        uint64_t first = s->leaf_first[leaf], count = s->leaf_count[leaf];
        if (first + count > s->n_leaf_refs) return -1;
        return depth;
    }
    if ((uint32_t)node >= s->n_branches || seen_branch[node]) return -1;
    seen_branch[node] = 1;
    if (s->branch_axis[node] < 0 || s->branch_axis[node] > 2) return -1;
    int a = validate_tree(s, s->branch_lesser[node], depth + 1, seen_branch);
    if (a < 0) return -1;
    int b = validate_tree(s, s->branch_greater[node], depth + 1, seen_branch);
    if (b < 0) return -1;
    return std::max(a, b);
}

int validate_scene(const rtx_scene_desc* s, uint32_t* kd_depth)
{
    if (!s) return fail(RTX_ERR_INVALID, "null scene description");
    if (s->abi_version != RTX_ABI_VERSION) return fail(RTX_ERR_INVALID, "abi_version %u, library speaks %u", s->abi_version, RTX_ABI_VERSION);
    if (s->n_shapes > s->n_prims) return fail(RTX_ERR_INVALID, "n_shapes > n_prims");
    if (s->n_prims >= (1u << 30)) return fail(RTX_ERR_LIMIT, "too many primitives");
    if (s->n_branches >= kKdMaxNodes || s->n_leaves >= kKdMaxNodes) return fail(RTX_ERR_LIMIT, "too many KD nodes");
    for (uint32_t i = 0; i < s->n_prims; ++i) {
        uint32_t idx = s->prim_index[i];
        switch (s->prim_type[i]) {
        case RTX_PRIM_TRIANGLE: if (idx >= s->n_triangles) return fail(RTX_ERR_INVALID, "prim %u: triangle index out of range", i); break;
        case RTX_PRIM_SPHERE: if (idx >= s->n_spheres) return fail(RTX_ERR_INVALID, "prim %u: sphere index out of range", i); break;
        case RTX_PRIM_BICONVEX: if (idx >= s->n_lenses) return fail(RTX_ERR_INVALID, "prim %u: lens index out of range", i); break;
        default: return fail(RTX_ERR_INVALID, "prim %u: unknown type %u", i, s->prim_type[i]);
        }
        if (s->prim_type[i] != RTX_PRIM_BICONVEX && s->prim_material[i] >= s->n_materials)
            return fail(RTX_ERR_INVALID, "prim %u: material out of range", i);
    }
    // the host-side builders sort on these: a NaN breaks their comparators' ordering
    for (uint32_t i = 0; i < s->n_prims; ++i)
        for (int k = 0; k < 3; ++k) {
            const double lo = s->prim_aabb_min[3 * (size_t)i + k], hi = s->prim_aabb_max[3 * (size_t)i + k];
            if (!std::isfinite(lo) || !std::isfinite(hi) || lo > hi)
                return fail(RTX_ERR_INVALID, "prim %u: bounding box is not finite or is inverted", i);
        }
    for (uint32_t i = 0; i < s->n_branches; ++i)
        for (int k = 0; k < 3; ++k) {
            const double lo = s->branch_min[3 * (size_t)i + k], hi = s->branch_max[3 * (size_t)i + k];
            // (a child region clipped at a split plane outside its parent's box is inverted: legal)
            if (std::isnan(lo) || std::isnan(hi) || std::isnan(s->branch_value[i]))
                return fail(RTX_ERR_INVALID, "KD branch %u: box or split value is NaN", i);
        }
    for (uint32_t i = 0; i < 2 * s->n_lenses; ++i) {
        uint32_t p = s->lens_spheres[i];
        if (p >= s->n_prims || s->prim_type[p] != RTX_PRIM_SPHERE) return fail(RTX_ERR_INVALID, "lens %u: inner sphere id invalid", i / 2);
    }
    for (uint32_t i = 0; i < s->n_leaf_refs; ++i)
        if (s->leaf_refs[i] >= s->n_shapes) return fail(RTX_ERR_INVALID, "leaf ref %u out of range", i);
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        const rtx_texture& t = s->textures[i];
        if (t.width == 0 || t.height == 0 || t.offset + (uint64_t)t.width * t.height * 3 > s->n_texel_bytes)
            return fail(RTX_ERR_INVALID, "texture %u outside the texel buffer", i);
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const rtx_material& m = s->materials[i];
        const rtx_sampler* f[6] = {&m.color_fn, &m.ns_fn, &m.ni_fn, &m.transparency_fn, &m.metalness_fn, &m.smoothness_fn};
        for (int k = 0; k < 6; ++k) {
            if (f[k]->kind > RTX_SAMPLER_TEXTURE_DOUBLE) return fail(RTX_ERR_INVALID, "material %u: unknown sampler", i);
            if ((f[k]->kind == RTX_SAMPLER_TEXTURE_VEC3 || f[k]->kind == RTX_SAMPLER_TEXTURE_DOUBLE) && f[k]->texture >= s->n_textures)
                return fail(RTX_ERR_INVALID, "material %u: texture out of range", i);
        }
    }
    for (uint32_t i = 0; i < s->n_lights; ++i)
        if (s->lights[i].kind > RTX_LIGHT_SQUARE) return fail(RTX_ERR_INVALID, "light %u: unknown kind", i);
    std::vector<uint8_t> seen(s->n_branches, 0);
    int depth = validate_tree(s, s->kd_root, 0, seen);
    if (depth < 0) return fail(RTX_ERR_INVALID, "KD tree is malformed (index out of range, cycle, or deeper than 64)");
    if (depth > kStack) return fail(RTX_ERR_LIMIT, "KD tree depth %d exceeds the traversal stack (%d)", depth, kStack);
    *kd_depth = (uint32_t)depth;
    return RTX_OK;
}

inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }
inline dim3 trace_grid_for(size_t n) { return dim3((unsigned)((n + kTraceBlock - 1) / kTraceBlock)); }

}  // namespace

// Scratch memory for rtx_render_device_noalloc: one chunk's ray queues, light-ray queues, sample
// slots and counters live here instead of in memory the call would have to allocate.  Every
// device the code object is loaded on holds its own copy at its own address; the name below is
// a host-side stand-in, never a device address: host code uses the pointer that
// hipGetSymbolAddress returned for the device (static_scratch()).
static __device__ __align__(256) unsigned char scratch[RTX_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::atomic<unsigned char*> g_scratch_ptr[kMaxDevices];

// the device's copy of `scratch`; the device must be current
int static_scratch(int device, unsigned char** out)
{
    if (device < 0 || device >= kMaxDevices) return fail(RTX_ERR_LIMIT, "device %d: at most %d devices", device, kMaxDevices);
    unsigned char* p = g_scratch_ptr[device].load();
    if (!p) {
        HIP_TRY(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RTX_ERR_DEVICE, "hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_ptr[device].store(p);
    }
    *out = p;
    return RTX_OK;
}
constexpr size_t kScratchHeader = 3 * 256;      // queue-length ring, work counters, histogram bins
}  // namespace

// fn(lo, hi) over [0, n) in a few contiguous pieces, each on a thread of its own
template <class F>
static void in_pieces(uint32_t n, F fn, uint32_t want_pieces = 0)
{
    const uint32_t pieces = want_pieces ? std::max(1u, std::min(want_pieces, n)) : (n < (1u << 16) ? 1u : 4u);
    std::vector<std::future<void>> rest;
    for (uint32_t k = 1; k < pieces; ++k)
        rest.push_back(std::async(std::launch::async, fn, (uint32_t)((uint64_t)n * k / pieces), (uint32_t)((uint64_t)n * (k + 1) / pieces)));
    fn(0u, n / pieces);
    for (auto& f : rest) f.get();
}

extern "C" {

const char* rtx_last_error(void) { return g_error.c_str(); }

int rtx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rtx_init(int device)
{
    int n = rtx_device_count();
    if (n <= 0) return fail(RTX_ERR_NO_GPU, "no HIP device is visible");
    if (device >= n) return fail(RTX_ERR_INVALID, "device %d of %d", device, n);
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    HIP_TRY(hipGetDeviceProperties(&prop, cur));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RTX_ERR_NO_GPU, "device %d is %s; this library is built for gfx950 only", cur, prop.gcnArchName);
    g_device.store(cur);
    // the device's copy of the allocation-free entry point's scratch array, looked up once
    unsigned char* scratch_ptr = nullptr;
    return static_scratch(cur, &scratch_ptr);
}

}  // extern "C"

// rtx_scene_create and rtx_scene_create_static; settled: also work out which shapes' leaf lists are
// complete (kLcaSure), which costs host time once and saves the device a search per hit
static int scene_create(const rtx_scene_desc* s, rtx_scene** out, bool settled)
{
    if (!out) return fail(RTX_ERR_INVALID, "null output handle");
    *out = nullptr;
    uint32_t kd_depth = 0;
    int rc = validate_scene(s, &kd_depth);
    if (rc != RTX_OK) return rc;
    if (rtx_device_count() <= 0) return fail(RTX_ERR_NO_GPU, "no HIP device is visible");
    USE_DEVICE(g_device.load());        // a worker thread that prepares the next frame starts on device 0

    // RTX_TIMINGS=1: the host stages of this call, in milliseconds, on stderr
    const bool timings = std::getenv("RTX_TIMINGS") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timings) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "rtx_scene_create: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    lap("validate");
    rtx_scene* sc = new rtx_scene();
    sc->kd_depth = kd_depth;
    (void)hipGetDevice(&sc->device);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, sc->device) == hipSuccess) sc->compute_units = prop.multiProcessorCount;
        if (sc->compute_units < 1) sc->compute_units = 256;
        if (const char* env = std::getenv("RTX_REFILL_AT")) { long v = std::atol(env); if (v >= 1 && v <= 64) sc->refill_at = (uint32_t)v; }
        if (const char* env = std::getenv("RTX_REFILL_RANGE")) { long v = std::atol(env); if (v >= 64 && v <= (1 << 20)) sc->refill_range = (uint32_t)v / 64 * 64; }
    }
    DScene& d = sc->d;
    std::memset(&d, 0, sizeof d);
    d.root = s->kd_root;
    d.n_branches = s->n_branches; d.n_leaves = s->n_leaves; d.n_leaf_refs = s->n_leaf_refs;
    d.n_shapes = s->n_shapes; d.n_prims = s->n_prims; d.n_lights = s->n_lights;
    d.n_materials = s->n_materials; d.n_textures = s->n_textures;

    std::vector<Branch> branches(s->n_branches);
    for (uint32_t i = 0; i < s->n_branches; ++i) {
        Branch& b = branches[i];
        std::memset(&b, 0, sizeof b);
        for (int k = 0; k < 3; ++k) { b.bmin[k] = s->branch_min[3 * i + k]; b.bmax[k] = s->branch_max[3 * i + k]; }
        b.value = s->branch_value[i]; b.axis = s->branch_axis[i];
        b.lesser = s->branch_lesser[i]; b.greater = s->branch_greater[i];
    }
    std::vector<uint2> leaves(s->n_leaves);
    for (uint32_t i = 0; i < s->n_leaves; ++i) leaves[i] = make_uint2(s->leaf_first[i], s->leaf_count[i]);
    std::vector<PrimBox> boxes(s->n_prims);
    std::vector<uint32_t> info(s->n_prims);
    for (uint32_t i = 0; i < s->n_prims; ++i) {
        for (int k = 0; k < 3; ++k) { boxes[i].bmin[k] = s->prim_aabb_min[3 * i + k]; boxes[i].bmax[k] = s->prim_aabb_max[3 * i + k]; }
        info[i] = (uint32_t)s->prim_type[i] | (s->prim_index[i] << 2);
    }

    // the candidate BVH depends on the shape boxes alone: built on another thread while this
    // one derives the tables that hang off the KD tree
    if (s->n_shapes > kBvhMaxShapes) {
        delete sc;
        return fail(RTX_ERR_LIMIT, "%u shapes: the candidate BVH addresses at most %u", s->n_shapes, kBvhMaxShapes);
    }
    for (uint32_t i = 0; i < s->n_materials; ++i)
        if (s->materials[i].transparency != 0.0 || s->materials[i].transparency_fn.kind != RTX_SAMPLER_CONST) sc->may_refract = true;
    for (uint32_t i = 0; i < s->n_lights; ++i) sc->light_multisample.push_back(s->lights[i].multisample ? 1 : 0);
    sc->light_rays_plain = s->n_lights == 1 && !s->lights[0].multisample &&
                           (s->lights[0].kind == RTX_LIGHT_POINT || s->lights[0].kind == RTX_LIGHT_DIRECTIONAL);
    lap("copy branches, boxes");
    BvhBuild bvh;
    int32_t bvh_behind_root = kBvhEmpty;
    std::vector<BvhPrim> bvh_prims;
    std::future<void> bvh_done = std::async(std::launch::async, [&]() {
        // (RTX_BVH_SPLIT_AREA: the box area, in medians of the scene, above which a triangle enters the
        // BVH in pieces; 0 = every shape whole)
        double split_factor = kBvhSplitArea;
        if (const char* env = std::getenv("RTX_BVH_SPLIT_AREA")) split_factor = std::atof(env);
        std::vector<PrimBox> ref_boxes, part_boxes;
        std::vector<uint32_t> ref_shape, part_shape;
        if (split_factor > 0.0) split_large_triangles(boxes, s->n_shapes, info.data(), s->tri_position, split_factor, ref_boxes, ref_shape, part_boxes, part_shape);
        if (!part_boxes.empty() && ref_boxes.size() <= kBvhMaxShapes) {
            build_bvh(ref_boxes, (uint32_t)ref_boxes.size(), bvh, &ref_shape);
            BvhBuild behind;
            build_bvh(part_boxes, (uint32_t)part_boxes.size(), behind, &part_shape);
            bvh_behind_root = append_bvh(bvh, behind);
            if (timings) std::fprintf(stderr, "rtx_scene_create: %u shapes as %zu BVH references, %zu of them pieces of split triangles\n",
                                      s->n_shapes, ref_boxes.size(), part_boxes.size());
        } else {
            build_bvh(boxes, s->n_shapes, bvh);
        }
        bvh_prims.resize(bvh.prims.size());
        for (size_t i = 0; i < bvh.prims.size(); ++i) {
            BvhPrim& q = bvh_prims[i];
            std::memset(&q, 0, sizeof q);
            q.shape = bvh.prims[i];
            q.info = info[q.shape];
            if ((q.info & 3u) == RTX_PRIM_TRIANGLE) std::memcpy(q.p, s->tri_position + 9 * (size_t)(q.info >> 2), sizeof q.p);
        }
    });
    struct Join { std::future<void>& f; ~Join() { if (f.valid()) f.wait(); } } join_bvh{bvh_done};

    const double big = std::numeric_limits<double>::max();
    // Three derivations that do not depend on one another run side by side: the inverse leaf
    // index and the leaf cells on threads of their own, the content boxes and leaf-id ranges here.
    // (The cells read the branches' boxes and planes only, the ranges are other members.)
    // inverse leaf index: for every shape the leaves that reference it, in increasing order
    std::vector<uint32_t> prim_leaf_first(s->n_shapes + 1, 0), prim_leaf_ids(s->n_leaf_refs), prim_leaf_pos(s->n_leaf_refs);
    std::future<void> inverse_done = std::async(std::launch::async, [&]() {
        for (uint32_t r = 0; r < s->n_leaf_refs; ++r) prim_leaf_first[s->leaf_refs[r] + 1]++;
        for (uint32_t p = 0; p < s->n_shapes; ++p) prim_leaf_first[p + 1] += prim_leaf_first[p];
        std::vector<uint32_t> cursor(prim_leaf_first.begin(), prim_leaf_first.end() - 1);
        for (uint32_t leaf = 0; leaf < s->n_leaves; ++leaf)
            for (uint32_t r = 0; r < s->leaf_count[leaf]; ++r) {
                uint32_t pos = s->leaf_first[leaf] + r, p = s->leaf_refs[pos];
                // a shape listed twice in one leaf keeps its first position
                if (cursor[p] > prim_leaf_first[p] && prim_leaf_ids[cursor[p] - 1] == leaf) continue;
                prim_leaf_ids[cursor[p]] = leaf; prim_leaf_pos[cursor[p]] = pos; cursor[p]++;
            }
        // compact away slots left by duplicates
        bool dup = false;
        for (uint32_t p = 0; p < s->n_shapes; ++p) if (cursor[p] != prim_leaf_first[p + 1]) dup = true;
        if (dup) {
            std::vector<uint32_t> nf(s->n_shapes + 1, 0), ni, np;
            for (uint32_t p = 0; p < s->n_shapes; ++p) {
                for (uint32_t k = prim_leaf_first[p]; k < cursor[p]; ++k) { ni.push_back(prim_leaf_ids[k]); np.push_back(prim_leaf_pos[k]); }
                nf[p + 1] = (uint32_t)ni.size();
            }
            prim_leaf_first.swap(nf); prim_leaf_ids.swap(ni); prim_leaf_pos.swap(np);
        }
    });
    struct Join2 { std::future<void>& f; ~Join2() { if (f.valid()) f.wait(); } } join_inverse{inverse_done};
    // leaf cells for member_by_cell()
    std::vector<PrimBox> leaf_cell(s->n_leaves);
    std::future<void> cells_done = std::async(std::launch::async, [&]() {
        PrimBox all;
        for (int k = 0; k < 3; ++k) { all.bmin[k] = -big; all.bmax[k] = big; }
        std::vector<std::pair<int32_t, PrimBox>> st;
        st.push_back({s->kd_root, all});
        while (!st.empty()) {
            auto [node, cell] = st.back();
            st.pop_back();
            if (node < 0) { leaf_cell[(uint32_t)~node] = cell; continue; }
            const Branch& b = branches[node];
            for (int k = 0; k < 3; ++k) { cell.bmin[k] = std::max(cell.bmin[k], b.bmin[k]); cell.bmax[k] = std::min(cell.bmax[k], b.bmax[k]); }
            PrimBox lo = cell, hi = cell;
            lo.bmax[b.axis] = std::min(lo.bmax[b.axis], b.value);
            hi.bmin[b.axis] = std::max(hi.bmin[b.axis], b.value);
            st.push_back({b.lesser, lo});
            st.push_back({b.greater, hi});
        }
    });
    Join2 join_cells{cells_done};

    // content boxes for the culled walk: per leaf the union of its shape boxes, per branch
    // the union of its children's (an empty leaf gets an inverted box that no ray hits)
    std::vector<PrimBox> leaf_cbox(s->n_leaves), branch_cbox(s->n_branches);
    auto grow = [](PrimBox& a, const PrimBox& b) {
        for (int k = 0; k < 3; ++k) { a.bmin[k] = std::min(a.bmin[k], b.bmin[k]); a.bmax[k] = std::max(a.bmax[k], b.bmax[k]); }
    };
    in_pieces(s->n_leaves, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; ++i) {
            PrimBox c;
            for (int k = 0; k < 3; ++k) { c.bmin[k] = big; c.bmax[k] = -big; }
            for (uint32_t r = 0; r < s->leaf_count[i]; ++r) grow(c, boxes[s->leaf_refs[s->leaf_first[i] + r]]);
            leaf_cbox[i] = c;
        }
    });
    lap("leaf content boxes");
    std::vector<uint32_t> leaf_lo(s->n_branches, 0);
    {
        // children before parents: explicit post-order walk (the tree was validated above)
        std::vector<std::pair<int32_t, int>> st;
        if (s->kd_root >= 0) st.push_back({s->kd_root, 0});
        while (!st.empty()) {
            auto [node, phase] = st.back();
            if (phase == 0) {
                st.back().second = 1;
                int32_t l = s->branch_lesser[node], g = s->branch_greater[node];
                if (l >= 0) st.push_back({l, 0});
                if (g >= 0) st.push_back({g, 0});
            } else {
                st.pop_back();
                PrimBox c;
                for (int k = 0; k < 3; ++k) { c.bmin[k] = big; c.bmax[k] = -big; }
                for (int32_t ch : {s->branch_lesser[node], s->branch_greater[node]})
                    grow(c, ch >= 0 ? branch_cbox[ch] : leaf_cbox[(uint32_t)~ch]);
                branch_cbox[node] = c;
                // leaf-id range of the subtree (leaves are numbered in walk order, so it is contiguous)
                int32_t g = s->branch_greater[node];
                branches[node].leaf_mid = g >= 0 ? leaf_lo[g] : (uint32_t)~g;
                branches[node].leaf_hi = g >= 0 ? branches[g].leaf_hi : (uint32_t)~g + 1;
                int32_t l = s->branch_lesser[node];
                leaf_lo[node] = l >= 0 ? leaf_lo[l] : (uint32_t)~l;
            }
        }
    }
    // the numbering the verification relies on: every subtree's leaves are consecutive
    for (uint32_t i = 0; i < s->n_branches; ++i) {
        int32_t l = s->branch_lesser[i];
        uint32_t l_hi = l >= 0 ? branches[l].leaf_hi : (uint32_t)~l + 1;
        if (l_hi != branches[i].leaf_mid || branches[i].leaf_hi <= branches[i].leaf_mid) {
            delete sc;
            return fail(RTX_ERR_INVALID, "KD leaves are not numbered in walk order (branch %u)", i);
        }
    }
    lap("branch content boxes, ranges");
    inverse_done.get();
    cells_done.get();
    lap("wait for index and cells");
    std::vector<KdSplit> kd_split(s->n_branches);
    for (uint32_t i = 0; i < s->n_branches; ++i) {
        kd_split[i].value = branches[i].value;
        kd_split[i].lesser_axis = branches[i].lesser * 4 + branches[i].axis;
        kd_split[i].greater = branches[i].greater;
    }
    std::vector<int32_t> prim_lca(s->n_shapes, s->kd_root);
    std::vector<uint32_t> prim_cert;        // filled for static scenes (see certify_by_stretch)
    // A shape whose leaf list provably holds EVERY leaf below that node whose cell comes within
    // hold_margin of the shape's box gets kLcaSure: a point of that box inside such a cell is then
    // in a leaf of the shape, and the device need not search the list (member_by_cell).  The
    // reference's builder hands a shape to every child its box reaches (KDTree.cpp:78-113), so its
    // trees earn the mark throughout; a caller's own tree is taken at its word only where it checks out.
    double shape_scale = 0.0;
    for (const PrimBox& b : boxes)
        for (int k = 0; k < 3; ++k) shape_scale = std::max(shape_scale, std::max(std::fabs(b.bmin[k]), std::fabs(b.bmax[k])));
    const double hold_margin = kHoldMargin * shape_scale;
    // (the check walks the tree once per shape, memory-bound on the host: 55 ms for 68 k shapes, 70 ms on
    // top of a 0.12 s scene for 960 k -- more than a scene that is rebuilt for every frame earns back,
    // so it is done for scenes the caller creates as static only)
    const bool mark_sure = settled && hold_margin > 0.0 && std::isfinite(hold_margin);
    if (mark_sure) prim_cert.assign(s->n_shapes, 0u);
    struct Sub { int32_t node; PrimBox cell; };          // a subtree and the cell its root was handed
    auto enter = [&](PrimBox& c, const Branch& b) {
        for (int k = 0; k < 3; ++k) { c.bmin[k] = std::max(c.bmin[k], b.bmin[k]); c.bmax[k] = std::min(c.bmax[k], b.bmax[k]); }
    };
    // one step of the check of shape p: false when the subtree is a leaf the widened box touches
    // that is missing from the shape's list; children that need a look go to `out`
    auto look = [&](uint32_t p, const Sub& it, std::vector<Sub>& out) {
        const PrimBox& sb = boxes[p];
        for (int k = 0; k < 3; ++k)
            if (!(sb.bmin[k] - hold_margin <= it.cell.bmax[k] && sb.bmax[k] + hold_margin >= it.cell.bmin[k])) return true;
        const uint32_t* ids = prim_leaf_ids.data() + prim_leaf_first[p];
        const uint32_t* end = prim_leaf_ids.data() + prim_leaf_first[p + 1];
        if (it.node < 0) return std::binary_search(ids, end, (uint32_t)~it.node);
        const Branch& b = branches[it.node];
        // a subtree whose leaves are all in the list needs no descent
        const uint32_t* in_lo = std::lower_bound(ids, end, leaf_lo[it.node]);
        const uint32_t* in_hi = std::lower_bound(in_lo, end, b.leaf_hi);
        if ((uint32_t)(in_hi - in_lo) == b.leaf_hi - leaf_lo[it.node]) return true;
        Sub l{b.lesser, it.cell}, g{b.greater, it.cell};
        enter(l.cell, b); enter(g.cell, b);
        l.cell.bmax[b.axis] = std::min(l.cell.bmax[b.axis], b.value);
        g.cell.bmin[b.axis] = std::max(g.cell.bmin[b.axis], b.value);
        out.push_back(l); out.push_back(g);
        return true;
    };
    auto look_through = [&](uint32_t p, std::vector<Sub>& st) {       // depth first, until st is empty
        while (!st.empty()) {
            const Sub it = st.back();
            st.pop_back();
            if (!look(p, it, st)) return false;
        }
        return true;
    };
    // The faces of KD cells that come within hold_margin of shape p's box, per axis, for the device's
    // certificate by stretch: the faces of the covering node's cell, and below it every face a node adds
    // to the cell it is handed -- a side of its own box that cuts into the cell, and its split plane.
    // 0 when the box is not inside the covering node's cell, when a leaf below that node whose cell meets
    // the box (and is thick enough to hold a point by the margin) does not list the shape, or when a
    // count passes kCertMaxFaces.
    // A cell thinner than this on some axis never holds a point by the device's margin (at least 1e-9 of
    // the scene's size from every face), and neither does an inverted one: such leaves are never stood on.
    const double thin = 2e-9 * shape_scale;
    auto count_faces = [&](uint32_t p, const Sub& root, std::vector<Sub>& st) -> uint32_t {
        const PrimBox& sb = boxes[p];
        const uint32_t* ids = prim_leaf_ids.data() + prim_leaf_first[p];
        const uint32_t* ids_end = prim_leaf_ids.data() + prim_leaf_first[p + 1];
        uint32_t cnt[3] = {0, 0, 0};
        auto near_box = [&](int k, double v) { return v >= sb.bmin[k] - hold_margin && v <= sb.bmax[k] + hold_margin; };
        PrimBox top = root.cell;
        if (root.node >= 0) enter(top, branches[root.node]);
        for (int k = 0; k < 3; ++k) {
            if (!(sb.bmin[k] >= top.bmin[k] && sb.bmax[k] <= top.bmax[k])) return 0u;
            if (near_box(k, top.bmin[k])) cnt[k]++;
            if (near_box(k, top.bmax[k])) cnt[k]++;
        }
        st.clear();
        st.push_back(Sub{root.node, top});
        while (!st.empty()) {
            const Sub it = st.back();
            st.pop_back();
            bool apart = false;
            for (int k = 0; k < 3; ++k)
                if (!(sb.bmin[k] - hold_margin <= it.cell.bmax[k] && sb.bmax[k] + hold_margin >= it.cell.bmin[k])) apart = true;
            if (apart) continue;
            if (it.node < 0) {
                // a leaf that a point of the box can be inside of, by the margin, must list the shape
                bool stands = true;
                for (int k = 0; k < 3; ++k)
                    if (!(it.cell.bmax[k] - it.cell.bmin[k] >= thin && sb.bmin[k] <= it.cell.bmax[k] && sb.bmax[k] >= it.cell.bmin[k])) stands = false;
                if (stands && !std::binary_search(ids, ids_end, (uint32_t)~it.node)) return 0u;
                continue;
            }
            const Branch& b = branches[it.node];
            for (int k = 0; k < 3; ++k) {
                if (b.bmin[k] > it.cell.bmin[k] && near_box(k, b.bmin[k])) cnt[k]++;
                if (b.bmax[k] < it.cell.bmax[k] && near_box(k, b.bmax[k])) cnt[k]++;
            }
            if (near_box(b.axis, b.value)) cnt[b.axis]++;
            if (cnt[0] > kCertMaxFaces || cnt[1] > kCertMaxFaces || cnt[2] > kCertMaxFaces) return 0u;
            Sub l{b.lesser, it.cell}, g{b.greater, it.cell};
            enter(l.cell, b); enter(g.cell, b);
            l.cell.bmax[b.axis] = std::min(l.cell.bmax[b.axis], b.value);
            g.cell.bmin[b.axis] = std::max(g.cell.bmin[b.axis], b.value);
            st.push_back(l); st.push_back(g);
        }
        return kCertValid | cnt[0] | (cnt[1] << 10) | (cnt[2] << 20);
    };
    // shapes in very many leaves (a floor: a hundred thousand) are checked afterwards, each by all threads
    constexpr uint32_t kLongList = 4096;
    std::vector<std::pair<uint32_t, Sub>> long_lists;
    std::mutex long_lists_lock;
    in_pieces(s->n_shapes, [&](uint32_t first, uint32_t last) {
        std::vector<Sub> st;
        for (uint32_t p = first; p < last; ++p) {
            // in no leaf: the reference never tests the shape, and a candidate the BVH proposes on it
            // must never be certified -- not even in a tree that is one leaf, where the root's code
            // would read "the shape's one leaf"
            if (prim_leaf_first[p] == prim_leaf_first[p + 1]) { prim_lca[p] = kLcaNone; continue; }
            const uint32_t* ids = prim_leaf_ids.data() + prim_leaf_first[p];
            const uint32_t n_ids = prim_leaf_first[p + 1] - prim_leaf_first[p];
            const uint32_t lo = ids[0], hi = ids[n_ids - 1];
            int32_t node = s->kd_root;
            PrimBox cell;
            for (int k = 0; k < 3; ++k) { cell.bmin[k] = -big; cell.bmax[k] = big; }
            while (node >= 0) {
                const Branch& b = branches[node];
                if (hi < b.leaf_mid) { enter(cell, b); cell.bmax[b.axis] = std::min(cell.bmax[b.axis], b.value); node = b.lesser; }
                else if (lo >= b.leaf_mid) { enter(cell, b); cell.bmin[b.axis] = std::max(cell.bmin[b.axis], b.value); node = b.greater; }
                else break;
            }
            prim_lca[p] = node;
            if (node < 0 || !mark_sure) continue;
            if (n_ids > kLongList) {
                std::lock_guard<std::mutex> hold(long_lists_lock);
                long_lists.push_back({p, Sub{node, cell}});
                continue;
            }
            st.clear();
            st.push_back(Sub{node, cell});
            if (look_through(p, st)) prim_lca[p] = node | kLcaSure;
            prim_cert[p] = count_faces(p, Sub{node, cell}, st);
        }
    }, mark_sure ? 12u : 0u);
    for (const auto& [p, root] : long_lists) {
        // breadth first until there are subtrees enough for every thread, then those side by side
        std::vector<Sub> front{root}, next;
        bool sure = true;
        while (sure && !front.empty() && front.size() < 4096) {
            next.clear();
            for (const Sub& it : front) if (!look(p, it, next)) { sure = false; break; }
            front.swap(next);
        }
        std::atomic<bool> ok{sure};
        if (sure && !front.empty()) {
            const uint32_t shape = p;
            in_pieces((uint32_t)front.size(), [&](uint32_t first, uint32_t last) {
                std::vector<Sub> st;
                for (uint32_t k = first; k < last && ok.load(std::memory_order_relaxed); ++k) {
                    st.clear();
                    st.push_back(front[k]);
                    if (!look_through(shape, st)) ok.store(false, std::memory_order_relaxed);
                }
            }, 8);
        }
        if (ok.load()) prim_lca[p] |= kLcaSure;
    }
    lap("kd_split, prim_lca");
    if (timings && mark_sure) {
        size_t sure = 0, valid = 0, sum[3] = {0, 0, 0};
        for (uint32_t p = 0; p < s->n_shapes; ++p) {
            if (prim_lca[p] >= 0 && (prim_lca[p] & kLcaSure)) ++sure;
            if (prim_cert[p] & kCertValid) { ++valid; for (int k = 0; k < 3; ++k) sum[k] += (prim_cert[p] >> (10 * k)) & 1023u; }
        }
        std::fprintf(stderr, "rtx_scene_create: %u shapes, %zu with complete leaf lists, %zu with face counts (mean %.1f / %.1f / %.1f per axis)\n",
                     s->n_shapes, sure, valid, valid ? (double)sum[0] / valid : 0.0, valid ? (double)sum[1] / valid : 0.0,
                     valid ? (double)sum[2] / valid : 0.0);
    }
    bvh_done.get();
    lap("wait for the BVH");
    d.bvh_root = bvh.root;
    d.bvh_behind_root = bvh_behind_root;
    sc->bvh_depth = (uint32_t)bvh.depth;
    d.scene_scale = shape_scale;

    lap("bvh_prims, scale");
    sc->buffers = std::make_shared<DeviceBuffers>();
    DeviceBuffers& B = *sc->buffers;
    hipError_t e = hipSuccess;
    auto up = [&](auto host, size_t count, auto out_ptr) { if (e == hipSuccess) e = B.upload(host, count, out_ptr); };
    up(branches.data(), branches.size(), &d.branches);
    up(branch_cbox.data(), branch_cbox.size(), &d.branch_cbox);
    up(leaf_cbox.data(), leaf_cbox.size(), &d.leaf_cbox);
    up(bvh.nodes.data(), bvh.nodes.size(), &d.bvh_nodes);
    // (RTX_BVH_NODES=32 keeps the walk on the single-precision nodes)
    std::vector<BvhNode16> nodes16;
    const char* nodes_env = std::getenv("RTX_BVH_NODES");
    const bool half_nodes = !(nodes_env && std::atoi(nodes_env) == 32) && nodes_to_half(bvh.nodes, nodes16);
    if (timings) std::fprintf(stderr, "rtx_scene_create: %zu BVH nodes, walked in %s precision\n", bvh.nodes.size(), half_nodes ? "half" : "single");
    if (half_nodes) up(nodes16.data(), nodes16.size(), &d.bvh_nodes16);
    up(srgb_thresholds(), 256, &sc->srgb_thresholds);
    up(bvh_prims.data(), bvh_prims.size(), &d.bvh_prims);
    up(prim_leaf_first.data(), prim_leaf_first.size(), &d.prim_leaf_first);
    up(prim_leaf_ids.data(), prim_leaf_ids.size(), &d.prim_leaf_ids);
    up(prim_leaf_pos.data(), prim_leaf_pos.size(), &d.prim_leaf_pos);
    up(leaf_cell.data(), leaf_cell.size(), &d.leaf_cell);
    up(prim_lca.data(), prim_lca.size(), &d.prim_lca);
    up(prim_cert.data(), prim_cert.size(), &d.prim_cert);
    up(kd_split.data(), kd_split.size(), &d.kd_split);
    up(leaves.data(), leaves.size(), &d.leaves);
    up(s->leaf_refs, (size_t)s->n_leaf_refs, &d.leaf_refs);
    up(boxes.data(), boxes.size(), &d.prim_box);
    up(info.data(), info.size(), &d.prim_info);
    up(s->prim_material, (size_t)s->n_prims, &d.prim_material);
    up(s->tri_position, (size_t)s->n_triangles * 9, &d.tri_pos);
    up(s->tri_normal, (size_t)s->n_triangles * 9, &d.tri_nrm);
    up(s->tri_uv, (size_t)s->n_triangles * 6, &d.tri_uv);
    up(s->sphere_center, (size_t)s->n_spheres * 3, &d.sph_center);
    up(s->sphere_radius, (size_t)s->n_spheres, &d.sph_radius);
    up(s->sphere_r2, (size_t)s->n_spheres, &d.sph_r2);
    up(s->sphere_normal_matrix, (size_t)s->n_spheres * 16, &d.sph_nmat);
    up(s->lens_spheres, (size_t)s->n_lenses * 2, &d.lens_spheres);
    up(s->materials, (size_t)s->n_materials, &d.materials);
    up(s->textures, (size_t)s->n_textures, &d.textures);
    up(s->texels, (size_t)s->n_texel_bytes, &d.texels);
    up(s->lights, (size_t)s->n_lights, &d.lights);
    // Ordering: the uploads ran on the copy stream; finish() waits for them on the host, so every
    // later call on any stream and through any handle (clones share these arrays, read-only) sees
    // them complete.  The host arrays (the caller's and the tables above) are read until here.
    { hipError_t done = B.finish(); if (e == hipSuccess) e = done; }
    if (e != hipSuccess) {
        delete sc;
        return fail(RTX_ERR_DEVICE, "scene upload: %s", hipGetErrorString(e));
    }
    lap("upload");
    *out = sc;
    return RTX_OK;
}

extern "C" {

int rtx_scene_create(const rtx_scene_desc* desc, rtx_scene** out) { return scene_create(desc, out, false); }
int rtx_scene_create_static(const rtx_scene_desc* desc, rtx_scene** out) { return scene_create(desc, out, true); }

int rtx_scene_set_stream(rtx_scene* sc, void* hip_stream)
{
    if (!sc) return fail(RTX_ERR_INVALID, "null scene");
    hipStream_t next = static_cast<hipStream_t>(hip_stream);
    if (next != sc->stream) {
        // Ordering: everything a handle queues -- its lanes are forked from and joined to the scene's
        // stream -- is ordered by that stream, and the handle's queues and counters (or, for
        // rtx_render_device_noalloc, the device's static scratch array) are reused by the next call.
        // Work still running on the old stream would race with calls queued on the new one, so the old
        // stream is drained before the switch, whether or not this handle owns a workspace yet.
        USE_DEVICE(sc->device);
        HIP_TRY(hipStreamSynchronize(sc->stream));
    }
    sc->stream = next;
    return RTX_OK;
}

int rtx_scene_clone(const rtx_scene* sc, rtx_scene** out)
{
    if (!sc || !out) return fail(RTX_ERR_INVALID, "null argument");
    USE_DEVICE(sc->device);
    rtx_scene* c = new rtx_scene();
    c->d = sc->d;
    c->buffers = sc->buffers;
    c->device = sc->device;
    c->kd_depth = sc->kd_depth;
    c->bvh_depth = sc->bvh_depth;
    c->may_refract = sc->may_refract;
    c->light_rays_plain = sc->light_rays_plain;
    c->srgb_thresholds = sc->srgb_thresholds;
    c->light_multisample = sc->light_multisample;
    c->compute_units = sc->compute_units;
    c->refill_range = sc->refill_range;
    c->refill_at = sc->refill_at;
    *out = c;
    return RTX_OK;
}

void rtx_scene_destroy(rtx_scene* sc)
{
    if (sc) (void)hipSetDevice(sc->device);     // its streams and buffers are released on their own device
    delete sc;
}

int rtx_trace_closest(rtx_scene* sc, const double* origin, const double* direction, size_t n, uint32_t flags,
                      int32_t* prim, double* depth, double* position, double* normal, double* uv, double* bary,
                      uint32_t* leaf_count, uint64_t* leaf_hash, uint32_t* leaf_seq, uint32_t leaf_seq_cap,
                      uint32_t* counters)
{
    if (!sc || !origin || !direction || !prim || !depth) return fail(RTX_ERR_INVALID, "null argument");
    if (n == 0) return RTX_OK;
    if (n > 0x7fffffffu) return fail(RTX_ERR_LIMIT, "at most 2^31-1 rays per call");
    USE_DEVICE(sc->device);
    DeviceBuffers tmp;
    const double *d_o = nullptr, *d_d = nullptr;
    HIP_TRY(tmp.upload(origin, n * 3, &d_o));
    HIP_TRY(tmp.upload(direction, n * 3, &d_d));
    // Ordering: the ray uploads run on the buffers' own non-blocking copy stream, which is ordered
    // with nothing else; finish() waits for them on the host, so the launch on the scene's stream
    // below cannot overtake them (the blocking hipMemcpy calls further down are complete when they return).
    HIP_TRY(tmp.finish());
    auto dalloc = [&](size_t bytes, void** p) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) tmp.ptrs.push_back(*p);
        return e;
    };
    int32_t* d_prim = nullptr; double* d_depth = nullptr;
    double *d_pos = nullptr, *d_nrm = nullptr, *d_uv = nullptr, *d_bary = nullptr;
    uint32_t *d_lc = nullptr, *d_seq = nullptr, *d_ctr = nullptr;
    unsigned long long* d_lh = nullptr;
    HIP_TRY(dalloc(n * 4, (void**)&d_prim));
    HIP_TRY(dalloc(n * 8, (void**)&d_depth));
    if (position) HIP_TRY(dalloc(n * 24, (void**)&d_pos));
    if (normal) HIP_TRY(dalloc(n * 24, (void**)&d_nrm));
    if (uv) HIP_TRY(dalloc(n * 16, (void**)&d_uv));
    if (bary) HIP_TRY(dalloc(n * 24, (void**)&d_bary));
    if (leaf_count) HIP_TRY(dalloc(n * 4, (void**)&d_lc));
    if (leaf_hash) HIP_TRY(dalloc(n * 8, (void**)&d_lh));
    if (leaf_seq && leaf_seq_cap) {
        HIP_TRY(dalloc(n * (size_t)leaf_seq_cap * 4, (void**)&d_seq));
        HIP_TRY(hipMemcpy(d_seq, leaf_seq, n * (size_t)leaf_seq_cap * 4, hipMemcpyHostToDevice));
    }
    if (counters) HIP_TRY(dalloc(n * 24, (void**)&d_ctr));
    const uint32_t linear = (flags & RTX_FLAG_LINEAR_SCAN) ? 1u : 0u;
#define RTX_LAUNCH_API(MODE)                                                                                      \
    hipLaunchKernelGGL(k_trace_api<MODE>, trace_grid_for(n), dim3(kTraceBlock), 0, sc->stream, sc->d, d_o, d_d, (uint32_t)n, \
                       linear, d_prim, d_depth, d_pos, d_nrm, d_uv, d_bary, d_lc, d_lh, d_seq, leaf_seq_cap, d_ctr)
    if (!(flags & RTX_FLAG_RENDER_WALK) || linear) RTX_LAUNCH_API(0);
    else if (flags & RTX_FLAG_KD_CULLED) RTX_LAUNCH_API(1);
    else RTX_LAUNCH_API(3);
#undef RTX_LAUNCH_API
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(sc->stream));
    HIP_TRY(hipMemcpy(prim, d_prim, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(depth, d_depth, n * 8, hipMemcpyDeviceToHost));
    if (position) HIP_TRY(hipMemcpy(position, d_pos, n * 24, hipMemcpyDeviceToHost));
    if (normal) HIP_TRY(hipMemcpy(normal, d_nrm, n * 24, hipMemcpyDeviceToHost));
    if (uv) HIP_TRY(hipMemcpy(uv, d_uv, n * 16, hipMemcpyDeviceToHost));
    if (bary) HIP_TRY(hipMemcpy(bary, d_bary, n * 24, hipMemcpyDeviceToHost));
    if (leaf_count) HIP_TRY(hipMemcpy(leaf_count, d_lc, n * 4, hipMemcpyDeviceToHost));
    if (leaf_hash) HIP_TRY(hipMemcpy(leaf_hash, d_lh, n * 8, hipMemcpyDeviceToHost));
    if (d_seq) HIP_TRY(hipMemcpy(leaf_seq, d_seq, n * (size_t)leaf_seq_cap * 4, hipMemcpyDeviceToHost));
    if (counters) HIP_TRY(hipMemcpy(counters, d_ctr, n * 24, hipMemcpyDeviceToHost));
    return RTX_OK;
}

// limit == nullptr: closest hit; otherwise any hit with depth <= limit[i] (shadow rays).
// seg == 0: the first min(n, *n_device) rays of a flat queue; seg != 0: a light-ray queue of
// n = samples * seg entries, of which *n_device hits x samples are in use (see ray_place).
// copy a lane's device counters to its pinned mirror and zero them for the next period
static hipError_t enqueue_work_readback(Workspace& w, hipStream_t stream)
{
    hipError_t e = hipMemcpyAsync(w.host_work, w.work, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemsetAsync(w.work, 0, 16 * sizeof(unsigned long long), stream);
    return e;
}

// after the streams have drained: everything since the statistics were last handed out
static void hand_out_stats(rtx_scene* sc, rtx_stats* out)
{
    rtx_stats& t = sc->pending;
    unsigned long long work[16] = {0};
    for (int k = 0; k < 2; ++k) {
        if (!sc->ws[k].host_work) continue;
        for (int c = 0; c < 16; ++c) { work[c] += sc->ws[k].host_work[c]; sc->ws[k].host_work[c] = 0; }
        sc->lane_timer[k].collect();
        sc->lane_trace_timer[k].collect();
    }
    t.rays_shadow += work[8];
    t.rays_secondary += work[9] + work[12];
    t.branch_visits = work[0]; t.leaf_visits = work[1]; t.aabb_tests = work[2];
    t.triangle_tests = work[3]; t.sphere_tests = work[4]; t.hits_accepted = work[5];
    t.content_tests = work[6]; t.fallback_rays = work[7];
    t.wave_steps = work[10]; t.lane_steps = work[11];
    // RTX_PROBE=1 (with RTX_FLAG_COUNTERS): wave-level tallies of the BVH walk on stderr
    if (std::getenv("RTX_PROBE") && (work[13] | work[14]))
        std::fprintf(stderr, "rtx probe: wave_node_steps %llu wave_leaf_trips %llu wave_leaf_trips_packed %llu\n", work[13], work[14], work[15]);
    if (out) *out = t;
    std::memset(&t, 0, sizeof t);
}

static int launch_trace(rtx_scene* sc, Workspace& ws, hipStream_t stream, const double* ox, const double* oy, const double* oz,
                        const double* dx, const double* dy, const double* dz, const double* limit, uint32_t n,
                        const uint32_t* n_device, uint32_t seg, uint32_t flags, int32_t* hit_prim, double* hit_depth, bool count,
                        const CameraQueue* camera = nullptr, bool light_rays_from_origin = false, const uint32_t* perm = nullptr)
{
    CameraQueue cam;
    if (camera) cam = *camera; else std::memset(&cam, 0, sizeof cam);
    uint32_t linear = (flags & RTX_FLAG_LINEAR_SCAN) ? 1u : 0u;
    int mode;
    if (flags & (RTX_FLAG_REFERENCE_ORDER | RTX_FLAG_LINEAR_SCAN)) mode = 0;
    else if (flags & RTX_FLAG_KD_CULLED) mode = limit ? 2 : 1;
    else mode = limit ? 4 : 3;
    Counters g;
    g.c = count ? ws.work : nullptr;
    // threads to launch: light-ray queues are enumerated in whole tiles of hits (ray_total)
    uint64_t threads = n;
    if (seg) threads = ((uint64_t)seg + kRayTile - 1) / kRayTile * kRayTile * (n / seg);
    if (threads > 0xffffffffull) return fail(RTX_ERR_LIMIT, "more than 2^32 light rays in one launch: lower RTX_MAX_PATHS");
    dim3 grid = trace_grid_for(threads), block(kTraceBlock);
    if (mode >= 3 && (flags & RTX_FLAG_REFILL_LANES) && n <= ws.cand_capacity) {
        if (cam.on) return fail(RTX_ERR_INVALID, "the refilled walk reads stored camera rays");
        // lanes refilled from the wave's own range of the queue; ranges shrink for small launches
        // so that every resident wave slot gets several
        const int anyhit = limit ? 1 : 0;
        const uint64_t slots = (uint64_t)sc->compute_units * 16u;
        const uint32_t rays_per_wave = (uint32_t)std::min<uint64_t>(sc->refill_range, std::max<uint64_t>(64, (threads / (slots * 4)) / 64 * 64));
        dim3 pgrid((unsigned)((threads + rays_per_wave - 1) / rays_per_wave));
#define RTX_LAUNCH_REFILL(COUNT, ANYHIT)                                                                              \
    hipLaunchKernelGGL((k_trace_refill<COUNT, ANYHIT>), pgrid, block, 0, stream, sc->d, ox, oy, oz, dx, dy, dz, limit, n, \
                       n_device, seg, sc->refill_at, rays_per_wave, hit_prim, hit_depth, ws.cand, g)
        if (count) { if (anyhit) RTX_LAUNCH_REFILL(true, true); else RTX_LAUNCH_REFILL(true, false); }
        else { if (anyhit) RTX_LAUNCH_REFILL(false, true); else RTX_LAUNCH_REFILL(false, false); }
#undef RTX_LAUNCH_REFILL
        HIP_TRY(hipGetLastError());
        if (count) hipLaunchKernelGGL(k_verify<true>, grid, block, 0, stream, sc->d, ox, oy, oz, dx, dy, dz, n, n_device,
                                      seg, ws.cand, hit_prim, hit_depth, g);
        else hipLaunchKernelGGL(k_verify<false>, grid, block, 0, stream, sc->d, ox, oy, oz, dx, dy, dz, n, n_device,
                                seg, ws.cand, hit_prim, hit_depth, g);
        HIP_TRY(hipGetLastError());
        return RTX_OK;
    }
#define RTX_LAUNCH_TRACE(COUNT, MODE)                                                                                  \
    hipLaunchKernelGGL((k_trace_pool<COUNT, MODE>), grid, block, 0, stream, sc->d, ox, oy, oz, dx, dy, dz, limit, n, \
                       n_device, seg, linear, hit_prim, hit_depth, g, cam, light_rays_from_origin ? 1u : 0u, perm)
#define RTX_LAUNCH_MODES(COUNT)                                                                   \
    switch (mode) {                                                                               \
    case 0: RTX_LAUNCH_TRACE(COUNT, 0); break;                                                    \
    case 1: RTX_LAUNCH_TRACE(COUNT, 1); break;                                                    \
    case 2: RTX_LAUNCH_TRACE(COUNT, 2); break;                                                    \
    case 3: RTX_LAUNCH_TRACE(COUNT, 3); break;                                                    \
    default: RTX_LAUNCH_TRACE(COUNT, 4); break;                                                   \
    }
    if (count) { RTX_LAUNCH_MODES(true) } else { RTX_LAUNCH_MODES(false) }
#undef RTX_LAUNCH_MODES
#undef RTX_LAUNCH_TRACE
    HIP_TRY(hipGetLastError());
    return RTX_OK;
}

// fixed_scratch == nullptr: rtx_render_device.  Otherwise rtx_render_device_noalloc: the tile is one
// chunk whose workspace is laid out in that memory, everything is queued on the scene's stream and
// nothing is allocated, created, read back or waited for.
static int render_impl(rtx_scene* sc, const rtx_render_desc* rd, float* accum, rtx_stats* stats, unsigned char* fixed_scratch)
{
    const bool fixed = fixed_scratch != nullptr;
    if (!sc || !rd || !accum) return fail(RTX_ERR_INVALID, "null argument");
    if (rd->integrator > RTX_INTEGRATOR_WHITTED) return fail(RTX_ERR_INVALID, "unknown integrator");
    USE_DEVICE(sc->device);
    if (rd->width < 2 || rd->height < 2) return fail(RTX_ERR_INVALID, "frame must be at least 2x2");
    if (rd->tile_w == 0 || rd->tile_h == 0 || rd->tile_x0 + (uint64_t)rd->tile_w > rd->width)
        return fail(RTX_ERR_INVALID, "tile outside the frame");
    {
        // the frame row of the tile's last row (rows may be interleaved groups, see rtx.h)
        const uint32_t g = rd->tile_row_group, last = rd->tile_h - 1;
        if (g && rd->tile_row_stride < g) return fail(RTX_ERR_INVALID, "tile_row_stride below tile_row_group");
        const uint64_t last_y = rd->tile_y0 + (g ? (uint64_t)(last / g) * rd->tile_row_stride + last % g : (uint64_t)last);
        if (last_y >= rd->height) return fail(RTX_ERR_INVALID, "tile outside the frame");
    }
    uint32_t n = rd->pixel_multisample_n ? rd->pixel_multisample_n : 1;
    if (n > 64) return fail(RTX_ERR_LIMIT, "pixel_multisample_n > 64");
    // (a band of 8 rows is indexed with 32 bits)
    if (rd->tile_w > (1u << 28)) return fail(RTX_ERR_LIMIT, "tile_w %u: tiles are at most 2^28 pixels wide", rd->tile_w);
    const bool whitted = rd->integrator == RTX_INTEGRATOR_WHITTED;
    const bool count_work = (rd->flags & RTX_FLAG_COUNTERS) != 0;

    RenderParams p;
    std::memset(&p, 0, sizeof p);
    p.d = *rd;
    p.d.pixel_multisample_n = n;
    p.n = n;
    p.band_stride = 1;
    // what every ray of the frame would otherwise work out for itself (RenderParams)
    p.div_band = make_fast_div(8u * rd->tile_w);
    p.div_nn = make_fast_div(n * n);
    p.div_n = make_fast_div(n);
    p.div_group = make_fast_div(rd->tile_row_group ? rd->tile_row_group : 1u);
    p.div_stride = make_fast_div(rd->tile_row_stride ? rd->tile_row_stride : 1u);
    {
        const double cx = 1.0 / ((double)rd->width - 1.0), cy = 1.0 / ((double)rd->height - 1.0);
        p.sub_step_x = cx / (double)(n + 1);
        p.sub_step_y = cy / (double)(n + 1);
    }
    const uint64_t padded_pixels = (uint64_t)rd->tile_w * rd->tile_h;     // every index is a pixel
    if (padded_pixels * n * n > 0xffffffffull) return fail(RTX_ERR_LIMIT, "more than 2^32 samples per pass in one tile");
    uint32_t nn = n * n;
    uint32_t shadow_per_hit = 0;
    if (whitted && !(rd->flags & RTX_FLAG_NO_LIGHTING)) {
        for (uint8_t multisample : sc->light_multisample)
            shadow_per_hit += multisample ? rd->light_multisample_n * rd->light_multisample_n : 1u;
    }
    p.shadow_per_hit = shadow_per_hit;
    p.atomic_samples = (whitted && (rd->flags & RTX_FLAG_WHITTED_REFLECT)) ? 1u : 0u;
    // GLOBAL_LIGHTING: globalSamples per light at a level (Raytracer-Noah-Desktop.cpp:500); MAXBOUNCES = 0
    // makes the reference's factor NaN, taken as no samples
    const bool gather = whitted && (rd->flags & RTX_FLAG_GLOBAL_LIGHTING) && !(rd->flags & RTX_FLAG_NO_LIGHTING) &&
                        rd->global_samples > 0 && rd->max_depth > 0 && sc->d.n_lights > 0;
    auto gather_samples_at = [&](uint32_t layer) -> uint32_t {
        if (!gather) return 0u;
        double f = ((double)rd->max_depth / (double)(layer + 1)) / (double)rd->max_depth;
        return (uint32_t)(int)((double)rd->global_samples * f);
    };
    const uint64_t gather_per_hit0 = (uint64_t)gather_samples_at(0) * sc->d.n_lights;   // the most any level emits

    // Chunk size: whole 8x8 pixel blocks, bounded by budgets on the queue sizes.  Large chunks
    // matter: the deeper levels of a path-traced chunk hold few rays, and small launches end in
    // long tails (a 1080p 16-spp path-traced frame takes 27.9 ms in 8 Mi-path chunks, 22.6 ms in
    // one chunk).  A path costs about 270 bytes of queue space, a light-ray entry 40.
    size_t max_paths = 32u << 20;
    if (const char* env = std::getenv("RTX_MAX_PATHS")) { long v = std::atol(env); if (v >= 64) max_paths = (size_t)v; }
    size_t max_light_rays = 256u << 20;     // shadow-queue entries of one chunk (also keeps indices in 32 bits)
    size_t max_gather = 64u << 20;          // gather-queue entries of one chunk
    if (const char* env = std::getenv("RTX_MAX_GATHER")) { long v = std::atol(env); if (v >= 64) max_gather = (size_t)v; }
    // the mirror extension can double the queue at every level
    size_t growth = 1;
    if (whitted && (rd->flags & RTX_FLAG_WHITTED_REFLECT))
        for (uint32_t l = 0; l < std::min<uint32_t>(rd->max_depth, 4); ++l) growth *= 2;
    auto pixels_for = [&](size_t budget, uint64_t per_path) {
        return std::max<uint64_t>(64, (budget / (per_path * growth * nn)) / 64 * 64);
    };
    uint64_t pixels_per_chunk = pixels_for(max_paths, 1);
    if (whitted && shadow_per_hit) pixels_per_chunk = std::min(pixels_per_chunk, pixels_for(max_light_rays, shadow_per_hit));
    if (gather_per_hit0) pixels_per_chunk = std::min(pixels_per_chunk, pixels_for(max_gather, gather_per_hit0));
    pixels_per_chunk = std::min<uint64_t>(pixels_per_chunk, padded_pixels);
    // Chunks are rendered two at a time, each on a stream of its own: the launch tails and the
    // one-workgroup kernels of one chunk overlap with the other's launches.  The budgets are for
    // both together; a tile that would fit one chunk is cut in two, and the number of chunks is
    // even with equal sizes.  Small tiles stay in one piece: halves would only make the tails longer.
    uint64_t two_lane_min = 1u << 20;       // sample slots in the tile
    if (const char* env = std::getenv("RTX_TWO_LANE_MIN")) { long v = std::atol(env); if (v >= 128) two_lane_min = (uint64_t)v; }
    if (fixed) pixels_per_chunk = padded_pixels;         // one chunk, or RTX_ERR_LIMIT below
    const bool serial = fixed || (rd->flags & RTX_FLAG_SERIAL_CHUNKS) != 0 || padded_pixels * nn < two_lane_min;
    if (!serial) {
        const uint64_t half = std::max<uint64_t>(64, pixels_per_chunk / 2 / 64 * 64);
        uint64_t n_chunks = (padded_pixels + half - 1) / half;
        n_chunks += n_chunks & 1;
        pixels_per_chunk = std::max<uint64_t>(64, std::min(half, ((padded_pixels + n_chunks - 1) / n_chunks + 63) / 64 * 64));
    }
    // A tile that is exactly two chunks is dealt out in alternate bands of 8 rows (chunk k: bands k, k + 2,
    // ...) instead of a lower and an upper half: the halves of a picture hold different amounts of
    // geometry, and the lane with more of it then runs on alone at the end.  Pixels and their random
    // streams are the same either way.  RTX_CHUNK_SPLIT=halves keeps contiguous halves.
    bool alternate_bands = false;
    uint64_t lane_pixels[2] = {0, 0};
    if (!serial && (padded_pixels + pixels_per_chunk - 1) / pixels_per_chunk == 2) {
        const uint32_t bands = (rd->tile_h + 7) / 8;
        const char* env = std::getenv("RTX_CHUNK_SPLIT");
        if (bands >= 2 && !(env && env[0] == 'h')) {
            alternate_bands = true;
            for (uint32_t b = 0; b < bands; ++b)
                lane_pixels[b & 1] += (uint64_t)std::min<uint32_t>(8u, rd->tile_h - 8u * b) * rd->tile_w;
            pixels_per_chunk = std::max(lane_pixels[0], lane_pixels[1]);
        }
    }
    const int n_lanes = serial ? 1 : 2;
    const size_t cap = (size_t)pixels_per_chunk * nn;
    const size_t queue_cap = cap * growth;
    const size_t fan_out = growth > 1 ? 2 : 1;
    if (queue_cap * std::max<uint64_t>(gather_per_hit0, 1) > 0xf0000000ull || queue_cap * (uint64_t)std::max<uint32_t>(shadow_per_hit, 1) > 0xf0000000ull)
        return fail(RTX_ERR_LIMIT, "light-ray queue of the smallest chunk exceeds 2^32 entries (%u shadow rays per hit)", shadow_per_hit);
    int rc = RTX_OK;
    const size_t shadow_cap = whitted ? queue_cap * std::max<uint32_t>(shadow_per_hit, 1) : 0;
    const size_t gather_cap = (size_t)(queue_cap * gather_per_hit0);
    Workspace fixed_ws;
    if (fixed) {
        if (rd->flags & (RTX_FLAG_WHITTED_REFLECT | RTX_FLAG_SORT_MATERIAL | RTX_FLAG_REFILL_LANES | RTX_FLAG_COUNTERS |
                         RTX_FLAG_STAGE_TIMES | RTX_FLAG_TRACE_TIMES))
            return fail(RTX_ERR_INVALID, "rtx_render_device_noalloc: flags 0x%x need queue lengths, counters or times read back",
                        rd->flags);
        const size_t need = kScratchHeader + queue_bytes(queue_cap) + shadow_bytes(shadow_cap) + gather_bytes(gather_cap);
        if (need > (size_t)RTX_NOALLOC_SCRATCH_BYTES)
            return fail(RTX_ERR_LIMIT, "rtx_render_device_noalloc: %zu sample slots with %u light rays per hit need %zu bytes of "
                        "scratch, the static array holds %zu: render smaller tiles", queue_cap, shadow_per_hit, need,
                        (size_t)RTX_NOALLOC_SCRATCH_BYTES);
        fixed_ws.borrowed = true;
        char* p = reinterpret_cast<char*>(fixed_scratch);
        fixed_ws.counts = carve<uint32_t>(p, 8);
        fixed_ws.work = carve<unsigned long long>(p, 16);
        fixed_ws.bins = carve<uint32_t>(p, 2 * kBins);
        lay_out_queues(fixed_ws, p, queue_cap);
        p += queue_bytes(queue_cap);
        lay_out_shadow(fixed_ws, p, shadow_cap);
        p += shadow_bytes(shadow_cap);
        lay_out_gather(fixed_ws, p, gather_cap);
        // the array keeps the last call's contents: the counters start at zero (the ring of queue
        // lengths and the sample slots are cleared by the chunk's first launch, as always)
        HIP_TRY(hipMemsetAsync(fixed_scratch, 0, kScratchHeader, sc->stream));
    }
    for (int k = 0; k < n_lanes && !fixed; ++k) {
        rc = ensure_workspace(sc->ws[k], queue_cap, shadow_cap, gather_cap, (rd->flags & RTX_FLAG_REFILL_LANES) != 0);
        if (rc != RTX_OK) return rc;
    }
    if (n_lanes > 1 && !sc->lane_stream[0]) {
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipStreamCreateWithFlags(&sc->lane_stream[k], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&sc->join_event[k], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&sc->fork_event, hipEventDisableTiming));
    }
    hipStream_t user = sc->stream;

    if (sc->async && stats) return fail(RTX_ERR_INVALID, "asynchronous scene: statistics come from rtx_scene_wait");
    rtx_stats unkept;
    std::memset(&unkept, 0, sizeof unkept);
    rtx_stats& local = fixed ? unkept : sc->pending;
    // neither an asynchronous scene nor the allocation-free call ever waits for a queue length
    const bool never_wait = sc->async || fixed;
    struct Events {
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    hipEvent_t& ev0 = ev.a;
    hipEvent_t& ev1 = ev.b;
    if (stats) { HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventRecord(ev0, user)); }
    const bool keep_times = !fixed && (stats != nullptr || sc->async);
    // Ordering: the lanes wait for this event, i.e. for everything the caller queued on the scene's
    // stream before this call (the previous frame's render and present through this handle, which
    // used the same queues, included); at the end the scene's stream waits for both lanes' join
    // events, so to the caller the call is ordered as if all of it ran on the scene's stream.
    if (n_lanes > 1) HIP_TRY(hipEventRecord(sc->fork_event, user));

    // rayTrace spawns nothing when no material refracts and the mirror extension is off
    const bool single_level = whitted && !sc->may_refract && !(rd->flags & RTX_FLAG_WHITTED_REFLECT);
    // Queue lengths live on the device: counts[2k] is the length of the queue of level L (k = L % 4),
    // counts[2k+1] the number of its lit hits.  The host reads a length back only to stop early and
    // to size the next launches tightly, which pays when launches are long; short levels are
    // launched for an upper bound of their length without waiting (surplus workgroups leave at
    // once), with a readback every fourth level.
    const bool sync_every_level = fan_out > 1 || (rd->flags & RTX_FLAG_SORT_MATERIAL);
    // Level 0 needs no stored queue: its kernels compute the camera ray of their sample slot, and
    // the level's traversal launch, the first of a chunk, clears the chunk's state.  The refilled
    // walk's kernels read rays from memory only.
    const bool camera_queue = !(rd->flags & (RTX_FLAG_STORE_CAMERA_RAYS | RTX_FLAG_REFILL_LANES));
    // One light that takes no random numbers: its ray is a function of the origin alone, so the
    // shadow queue carries origins only (40 bytes per light ray less to write and to read twice).
    const bool light_rays_from_origin = whitted && sc->d.n_lights == 1 && sc->light_rays_plain &&
                                        !(rd->flags & (RTX_FLAG_STORE_LIGHT_RAYS | RTX_FLAG_REFILL_LANES));
    // Sample slots that get their whole value from one thread of one launch are stored instead of
    // cleared, added to and read back: always so for pathTrace (weight x terminal radiance, added
    // once per path), for rayTrace when it spawns no children.  With one sub-sample per pixel the
    // value goes straight into the framebuffer and the resolve launch is left out.
    const uint32_t direct_store = (rd->flags & RTX_FLAG_ACCUMULATE_SAMPLES) ? 0u
                                  : (!whitted || (single_level && !p.atomic_samples)) ? (nn == 1 ? 2u : 1u) : 0u;
    const uint32_t kLongLevel = 4u << 20;
    // Rays of the levels below the camera's are walked in the order of their direction bin
    // (k_tile_count<1>): worth three more small launches once a level holds this many rays
    uint32_t bin_rays_min = 1u << 16;
    uint32_t bin_shapes_min = 1024;
    if (const char* env = std::getenv("RTX_BIN_SHAPES_MIN")) { long v = std::atol(env); if (v >= 0) bin_shapes_min = (uint32_t)std::min<long>(v, 0x7fffffffL); }
    if (const char* env = std::getenv("RTX_BIN_RAYS_MIN")) { long v = std::atol(env); if (v >= 0) bin_rays_min = (uint32_t)std::min<long>(v, 0x7fffffffL); }
    // (and only for scenes whose walk is long enough to gain: in a scene of a few shapes the three
    // launches cost more than the whole traversal -- lens scene, 1080p depth 8: 1.17 -> 1.39 ms)
    const bool bin_rays = !(rd->flags & (RTX_FLAG_QUEUE_ORDER | RTX_FLAG_REFILL_LANES)) && sc->d.n_shapes >= bin_shapes_min;

    struct Lane {
        Workspace* w = nullptr;
        hipStream_t st = nullptr;
        RenderParams p;
        StageTimer* timer_p = nullptr;
        StageTimer* trace_timer_p = nullptr;
        uint32_t bound = 0, level = 0;
        bool exact = true, alive = false, want_sync = false;
        int cur = 0;
    };
    const bool time_all = keep_times && (rd->flags & RTX_FLAG_STAGE_TIMES);
    // the traversal launches alone (ms_traverse, ms_shadow): a third of the events of all stages
    const bool time_trace = keep_times && (rd->flags & (RTX_FLAG_STAGE_TIMES | RTX_FLAG_TRACE_TIMES));
    Lane lanes[2];
    for (int k = 0; k < n_lanes; ++k) {
        Lane& L = lanes[k];
        L.w = fixed ? &fixed_ws : &sc->ws[k];
        L.st = n_lanes > 1 ? sc->lane_stream[k] : user;
        L.timer_p = &sc->lane_timer[k];
        L.trace_timer_p = &sc->lane_trace_timer[k];
        L.timer_p->on = time_all; L.trace_timer_p->on = time_trace;
        L.timer_p->stream = L.trace_timer_p->stream = L.st;
        L.p = p;
        L.p.direct_store = direct_store;
        L.p.light_rays_from_origin = light_rays_from_origin ? 1u : 0u;
        L.p.accum = accum;
        // this call's own capacity, the one the host checks lengths against below -- not the
        // workspace's, which an earlier, larger call may have grown
        L.p.queue_cap = (uint32_t)std::min<size_t>(queue_cap, 0xffffffffu);
        if (n_lanes > 1) HIP_TRY(hipStreamWaitEvent(L.st, sc->fork_event, 0));
    }

    // an error return leaves nothing of this frame running on the library's streams
    struct Drain {
        Lane* lanes; int n; bool armed = true;
        ~Drain() { if (armed) for (int k = 0; k < n; ++k) if (lanes[k].st) (void)hipStreamSynchronize(lanes[k].st); }
    } drain{lanes, n_lanes > 1 ? n_lanes : 0};

    // one level of a lane's chunk: traverse, shade, totals; afterwards the lane either is done with
    // the chunk (resolve enqueued), wants its next queue length read back, or just goes on
    auto enqueue_level = [&](Lane& L) -> int {
        Workspace& w = *L.w;
        hipStream_t st = L.st;
        RenderParams& lp = L.p;
        const uint32_t level = L.level, bound = L.bound;
        RayQueue& q = w.q[L.cur];
        RayQueue& nq = w.q[L.cur ^ 1];
        uint32_t* len = w.counts + 2 * (level % 4);
        uint32_t* hits = len + 1;
        uint32_t* next_len = w.counts + 2 * ((level + 1) % 4);
        const uint32_t* n_dev = L.exact ? nullptr : len;
        L.trace_timer_p->begin();
        // the shading kernels recompute the hit from its id: the depth is only kept for
        // the refilled walk's verification pass
        lp.layer = level;
        lp.camera_queue = (camera_queue && level == 0) ? 1u : 0u;
        CameraQueue cq;
        cq.p = lp; cq.sample = w.sample; cq.counts = w.counts; cq.on = lp.camera_queue;
        cq.clear_samples = direct_store ? 0u : 1u;
        const uint32_t* walk_order = nullptr;
        if (bin_rays && level > 0 && bound >= bin_rays_min) {
            const uint32_t n_tiles = (bound + kTileRays - 1) / kTileRays;
            hipLaunchKernelGGL(k_tile_count<1>, dim3(n_tiles), dim3(kBlock), 0, st, sc->d, q, bound, n_dev, n_tiles, w.tile_hist, w.bins);
            hipLaunchKernelGGL(k_tile_scan, dim3(kBins), dim3(kBlock), 0, st, n_tiles, w.tile_hist, w.bins);
            hipLaunchKernelGGL(k_tile_scatter<1>, dim3(n_tiles), dim3(kBlock), 0, st, sc->d, q, bound, n_dev, n_tiles, w.tile_hist, w.bins, w.perm);
            HIP_TRY(hipGetLastError());
            walk_order = w.perm;
        }
        int r = launch_trace(sc, w, st, q.ox, q.oy, q.oz, q.dx, q.dy, q.dz, nullptr, bound, n_dev, 0, rd->flags, q.hit_prim,
                             (rd->flags & RTX_FLAG_REFILL_LANES) ? q.hit_depth : nullptr, count_work, &cq, false, walk_order);
        if (r != RTX_OK) return r;
        L.trace_timer_p->end(&local.ms_traverse);
        local.traverse_launches++;
        lp.layer = level;
        uint32_t shadow_mult = 0, gather_mult = 0;
        if (!whitted) {
            L.timer_p->begin();
            const uint32_t* perm = nullptr;
            if (rd->flags & RTX_FLAG_SORT_MATERIAL) {       // exact lengths here (sync_every_level)
                const uint32_t n_tiles = (bound + kTileRays - 1) / kTileRays;
                hipLaunchKernelGGL(k_tile_count<0>, dim3(n_tiles), dim3(kBlock), 0, st, sc->d, q, bound, nullptr, n_tiles, w.tile_hist, w.bins);
                hipLaunchKernelGGL(k_tile_scan, dim3(kBins), dim3(kBlock), 0, st, n_tiles, w.tile_hist, w.bins);
                hipLaunchKernelGGL(k_tile_scatter<0>, dim3(n_tiles), dim3(kBlock), 0, st, sc->d, q, bound, nullptr, n_tiles, w.tile_hist, w.bins, w.perm);
                perm = w.perm;
            }
            hipLaunchKernelGGL(k_shade_path, grid_for(bound), dim3(kBlock), 0, st, sc->d, lp, q, bound, n_dev, perm, nq, next_len, w.sample);
            HIP_TRY(hipGetLastError());
            L.timer_p->end(&local.ms_shade);
        } else {
            lp.gather_samples = gather_samples_at(level);
            lp.gather_traced = level + 1 <= rd->max_depth ? 1u : 0u;
            lp.seg = bound;
            const uint32_t gather_per_hit = lp.gather_traced ? lp.gather_samples * sc->d.n_lights : 0u;
            const uint64_t gather_max = (uint64_t)bound * gather_per_hit;
            const uint64_t shadow_max = (uint64_t)bound * shadow_per_hit;
            if (gather_max > w.gather_capacity) return fail(RTX_ERR_LIMIT, "gather queue overflow");
            if (shadow_max > w.shadow_capacity) return fail(RTX_ERR_LIMIT, "shadow queue overflow");
            L.timer_p->begin();
            if (lp.gather_samples)
                hipLaunchKernelGGL(k_shade_whitted_a<true>, grid_for(bound), dim3(kBlock), 0, st, sc->d, lp, q, bound, n_dev, w.sq,
                                   hits, w.hit_index, w.gq, w.sample);
            else
                hipLaunchKernelGGL(k_shade_whitted_a<false>, grid_for(bound), dim3(kBlock), 0, st, sc->d, lp, q, bound, n_dev, w.sq,
                                   hits, w.hit_index, w.gq, w.sample);
            HIP_TRY(hipGetLastError());
            L.timer_p->end(&local.ms_shade);
            // the number of hits stays on the device: the launches are sized for the most
            // there can be and read the real number themselves
            if (shadow_max > 0) {
                L.trace_timer_p->begin();
                r = launch_trace(sc, w, st, w.sq.ox, w.sq.oy, w.sq.oz, w.sq.dx, w.sq.dy, w.sq.dz, w.sq.limit, (uint32_t)shadow_max,
                                 hits, bound, rd->flags, nullptr, w.sq.hit_depth, count_work, nullptr, light_rays_from_origin);
                if (r != RTX_OK) return r;
                L.trace_timer_p->end(&local.ms_shadow);
                local.traverse_launches++;
            }
            if (gather_max > 0) {
                // the gather's nested rayTrace calls, reduced to what reaches the image: one
                // closest-hit query each (their colour is discarded, :509-510)
                L.trace_timer_p->begin();
                r = launch_trace(sc, w, st, w.gq.ox, w.gq.oy, w.gq.oz, w.gq.dx, w.gq.dy, w.gq.dz, nullptr, (uint32_t)gather_max,
                                 hits, bound, rd->flags, w.gq.hit_prim, w.gq.hit_depth, count_work);
                if (r != RTX_OK) return r;
                L.trace_timer_p->end(&local.ms_traverse);
                local.traverse_launches++;
            }
            L.timer_p->begin();
            if (lp.gather_samples)
                hipLaunchKernelGGL(k_shade_whitted_b<true>, grid_for(bound), dim3(kBlock), 0, st, sc->d, lp, q, bound, n_dev, w.sq,
                                   w.hit_index, w.gq, nq, next_len, w.sample);
            else
                hipLaunchKernelGGL(k_shade_whitted_b<false>, grid_for(bound), dim3(kBlock), 0, st, sc->d, lp, q, bound, n_dev, w.sq,
                                   w.hit_index, w.gq, nq, next_len, w.sample);
            HIP_TRY(hipGetLastError());
            L.timer_p->end(&local.ms_shade);
            shadow_mult = shadow_per_hit;
            gather_mult = gather_per_hit;       // gather rays are layer+1 shootRay calls
        }
        hipLaunchKernelGGL(k_level_totals, dim3(1), dim3(1), 0, st, next_len, hits, w.work, shadow_mult, gather_mult,
                           w.counts + 2 * ((level + 2) % 4));
        HIP_TRY(hipGetLastError());
        // rays are only spawned for layers up to max_depth (Raytracer.cpp:220, Desktop :456)
        if (single_level || level >= rd->max_depth) { L.bound = 0; return RTX_OK; }
        L.cur ^= 1;
        L.level = level + 1;
        // an asynchronous scene never waits for a short level: other frames are in flight and the
        // launches of a level that turns out empty cost less than holding up the host
        if (!fixed && (sync_every_level || bound >= kLongLevel || (level % 4 == 3 && !never_wait))) {
            HIP_TRY(hipMemcpyAsync(w.host_counts, next_len, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            L.want_sync = true;
        } else {
            L.exact = false;        // without the mirror extension a level is never longer than the one before
        }
        return RTX_OK;
    };
    auto finish_chunk = [&](Lane& L) -> int {
        if (direct_store == 2u) { L.alive = false; return RTX_OK; }      // the shading kernels wrote the pixels
        L.timer_p->begin();
        hipLaunchKernelGGL(k_resolve, grid_for(L.p.chunk_pixels), dim3(kBlock), 0, L.st, L.p, L.w->sample, accum);
        HIP_TRY(hipGetLastError());
        L.timer_p->end(&local.ms_resolve);
        L.alive = false;
        return RTX_OK;
    };

    for (uint32_t pass = rd->first_pass; pass < rd->first_pass + rd->passes; ++pass) {
        uint64_t next_t0 = 0;
        for (;;) {
            // hand the next chunks to idle lanes
            for (int k = 0; k < n_lanes; ++k) {
                Lane& L = lanes[k];
                if (L.alive || next_t0 >= padded_pixels) continue;
                L.p.pass = pass;
                L.p.overwrite = ((rd->flags & RTX_FLAG_OVERWRITE) && pass == rd->first_pass) ? 1u : 0u;
                if (alternate_bands) {
                    // lane k's one chunk of this pass: the tile's bands k, k + 2, ...
                    L.p.band_stride = 2; L.p.band_first = (uint32_t)k;
                    L.p.chunk_t0 = 0;
                    L.p.chunk_pixels = (uint32_t)lane_pixels[k];
                    next_t0 += k == 0 ? lane_pixels[0] : lane_pixels[1];
                } else {
                    L.p.chunk_t0 = (uint32_t)next_t0;
                    L.p.chunk_pixels = (uint32_t)std::min<uint64_t>(pixels_per_chunk, padded_pixels - next_t0);
                    next_t0 += L.p.chunk_pixels;
                }
                const size_t slots = (size_t)L.p.chunk_pixels * nn;
                if (!camera_queue) {
                    L.timer_p->begin();
                    hipLaunchKernelGGL(k_generate, grid_for(slots), dim3(kBlock), 0, L.st, L.p, L.w->q[0], L.w->sample, L.w->counts);
                    HIP_TRY(hipGetLastError());
                    L.timer_p->end(&local.ms_generate);
                }
                local.rays_primary += slots;
                local.max_queue = std::max<uint32_t>(local.max_queue, (uint32_t)slots);
                L.bound = (uint32_t)slots;      // one camera ray per sample slot
                L.exact = true;                 // `bound` is the queue's length, not just a bound of it
                L.cur = 0; L.level = 0; L.alive = true; L.want_sync = false;
            }
            bool any = false;
            for (int k = 0; k < n_lanes; ++k) any = any || lanes[k].alive;
            if (!any) break;
            // a lane that waits for its queue length gets its next level enqueued before the
            // other lane is waited for
            for (int k = 0; k < n_lanes; ++k) {
                Lane& L = lanes[k];
                if (!L.alive) continue;
                if (L.want_sync) {
                    // (the pinned word was written by a copy queued on this lane's stream after the level's kernels)
                    HIP_TRY(hipStreamSynchronize(L.st));
                    L.want_sync = false;
                    L.bound = L.w->host_counts[0];
                    L.exact = true;
                    if (L.bound > queue_cap) return fail(RTX_ERR_LIMIT, "ray queue overflow (%u > %zu)", L.bound, queue_cap);
                    local.max_queue = std::max(local.max_queue, L.bound);
                    if (L.bound == 0) { rc = finish_chunk(L); if (rc != RTX_OK) return rc; continue; }
                }
                rc = enqueue_level(L);
                if (rc != RTX_OK) return rc;
                if (L.bound == 0) { rc = finish_chunk(L); if (rc != RTX_OK) return rc; }
            }
        }
    }
    for (int k = 0; k < n_lanes; ++k) {
        Lane& L = lanes[k];
        if (!never_wait) HIP_TRY(enqueue_work_readback(*L.w, L.st));
        if (n_lanes > 1) {
            HIP_TRY(hipEventRecord(sc->join_event[k], L.st));
            HIP_TRY(hipStreamWaitEvent(user, sc->join_event[k], 0));
        }
    }
    drain.armed = false;        // from here the caller's stream waits for the lanes
    if (never_wait) return RTX_OK;
    if (stats) HIP_TRY(hipEventRecord(ev1, user));
    HIP_TRY(hipStreamSynchronize(user));
    if (stats) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        local.ms_total = ms;
    }
    hand_out_stats(sc, stats);
    return RTX_OK;
}

int rtx_render_device(rtx_scene* sc, const rtx_render_desc* rd, float* accum, rtx_stats* stats)
{
    return render_impl(sc, rd, accum, stats, nullptr);
}

int rtx_render_device_noalloc(rtx_scene* sc, const rtx_render_desc* rd, float* accum)
{
    if (!sc) return fail(RTX_ERR_INVALID, "null argument");
    USE_DEVICE(sc->device);
    unsigned char* ptr = nullptr;
    int rc = static_scratch(sc->device, &ptr);
    if (rc != RTX_OK) return rc;
    return render_impl(sc, rd, accum, nullptr, ptr);
}

int rtx_scene_set_async(rtx_scene* sc, int enabled)
{
    if (!sc) return fail(RTX_ERR_INVALID, "null scene");
    if (sc->async && !enabled) {
        int rc = rtx_scene_wait(sc, nullptr);
        if (rc != RTX_OK) return rc;
    }
    sc->async = enabled != 0;
    return RTX_OK;
}

int rtx_scene_wait(rtx_scene* sc, rtx_stats* totals)
{
    if (!sc) return fail(RTX_ERR_INVALID, "null scene");
    USE_DEVICE(sc->device);
    for (int k = 0; k < 2; ++k)
        if (sc->ws[k].work) HIP_TRY(enqueue_work_readback(sc->ws[k], sc->stream));    // the lanes have been joined to it
    HIP_TRY(hipStreamSynchronize(sc->stream));
    hand_out_stats(sc, totals);
    return RTX_OK;
}

int rtx_render(rtx_scene* sc, const rtx_render_desc* rd, float* accum, rtx_stats* stats)
{
    if (!sc || !rd || !accum) return fail(RTX_ERR_INVALID, "null argument");
    if (sc->async) return fail(RTX_ERR_INVALID, "asynchronous scene: use rtx_render_device");
    USE_DEVICE(sc->device);
    size_t bytes = (size_t)rd->tile_w * rd->tile_h * 3 * sizeof(float);
    if (bytes == 0) return fail(RTX_ERR_INVALID, "empty tile");
    float* d_accum = nullptr;
    HIP_TRY(hipMalloc(&d_accum, bytes));
    int rc = RTX_OK;
    hipError_t e = hipMemcpy(d_accum, accum, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(RTX_ERR_DEVICE, "upload of accum: %s", hipGetErrorString(e));
    if (rc == RTX_OK) rc = rtx_render_device(sc, rd, d_accum, stats);
    if (rc == RTX_OK) {
        e = hipMemcpy(accum, d_accum, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RTX_ERR_DEVICE, "download of accum: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_accum);
    return rc;
}

int rtx_present_device(rtx_scene* sc, const float* accum, uint32_t width, uint32_t height, uint32_t passes,
                       uint32_t tonemap, int flip, uint8_t* rgb8)
{
    if (!sc || !accum || !rgb8) return fail(RTX_ERR_INVALID, "null argument");
    if (width == 0 || height == 0 || (uint64_t)width * height * 3 > 0xffffffffull) return fail(RTX_ERR_INVALID, "bad frame size");
    if (passes == 0 || tonemap > RTX_TONEMAP_CBRT) return fail(RTX_ERR_INVALID, "bad passes / tonemap");
    if (height > 65535u) return fail(RTX_ERR_LIMIT, "more than 65535 rows");
    USE_DEVICE(sc->device);
    // Ordering: queued on the scene's stream behind the render calls that filled `accum` through this
    // handle; a caller that renders through one handle and presents through another orders the two
    // streams itself.
    hipLaunchKernelGGL(k_present, dim3(grid_for((size_t)width * 3).x, height), dim3(kBlock), 0, sc->stream, accum, width, height,
                       (float)passes, tonemap, flip ? 1u : 0u, sc->srgb_thresholds, rgb8);
    HIP_TRY(hipGetLastError());
    if (!sc->async) HIP_TRY(hipStreamSynchronize(sc->stream));
    return RTX_OK;
}

int rtx_present(rtx_scene* sc, const float* accum, uint32_t width, uint32_t height, uint32_t passes, uint32_t tonemap,
                int flip, uint8_t* rgb8)
{
    if (!sc || !accum || !rgb8) return fail(RTX_ERR_INVALID, "null argument");
    if (sc->async) return fail(RTX_ERR_INVALID, "asynchronous scene: use rtx_present_device");
    USE_DEVICE(sc->device);
    size_t total = (size_t)width * height * 3;
    if (total == 0) return fail(RTX_ERR_INVALID, "bad frame size");
    float* d_accum = nullptr;
    uint8_t* d_rgb = nullptr;
    HIP_TRY(hipMalloc(&d_accum, total * sizeof(float)));
    hipError_t e = hipMalloc(&d_rgb, total);
    int rc = RTX_OK;
    if (e == hipSuccess) e = hipMemcpy(d_accum, accum, total * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(RTX_ERR_DEVICE, "present upload: %s", hipGetErrorString(e));
    if (rc == RTX_OK) rc = rtx_present_device(sc, d_accum, width, height, passes, tonemap, flip, d_rgb);
    if (rc == RTX_OK) {
        e = hipMemcpy(rgb8, d_rgb, total, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RTX_ERR_DEVICE, "present download: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_accum);
    if (d_rgb) (void)hipFree(d_rgb);
    return rc;
}

double rtx_uniform(uint64_t seed, uint32_t frame, uint32_t pixel, uint32_t sample, uint32_t branch, uint32_t draw_index)
{
    // host evaluation of the device generator (same integer arithmetic)
    uint32_t c0 = pixel, c1 = sample, c2 = branch, c3 = draw_index >> 2;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (frame * 0x9E3779B9u);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    uint32_t o[4] = {c0, c1, c2, c3};
    return (double)o[draw_index & 3u] / 4294967295.0;
}

}  // extern "C"
