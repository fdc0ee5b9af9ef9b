"""ctypes mirrors of the POD structs and constants in include/rtx.h.

Field order and types must match the header exactly; tests/test_abi.py checks
the sizes against the values the C compiler reports.
"""
import ctypes as C

RTX_ABI_VERSION = 1

RTX_OK = 0
RTX_ERR_INVALID = -1
RTX_ERR_DEVICE = -2
RTX_ERR_LIMIT = -3
RTX_ERR_NO_GPU = -4

PRIM_TRIANGLE, PRIM_SPHERE, PRIM_BICONVEX = 0, 1, 2

SAMPLER_CONST = 0
SAMPLER_RY_CHECKERS = 1
SAMPLER_DOUBLE_CHECKERS = 2
SAMPLER_TEXTURE_VEC3 = 3
SAMPLER_TEXTURE_DOUBLE = 4

LIGHT_POINT, LIGHT_DIRECTIONAL, LIGHT_SQUARE = 0, 1, 2

INTEGRATOR_PATH, INTEGRATOR_WHITTED = 0, 1

FLAG_LINEAR_SCAN = 1 << 0
FLAG_BASIC = 1 << 1
FLAG_NO_LIGHTING = 1 << 2
FLAG_WHITTED_REFLECT = 1 << 3
FLAG_COUNTERS = 1 << 4
FLAG_REFERENCE_ORDER = 1 << 5
FLAG_KD_CULLED = 1 << 6
FLAG_STAGE_TIMES = 1 << 7
FLAG_RENDER_WALK = 1 << 8
FLAG_SORT_MATERIAL = 1 << 9
FLAG_GLOBAL_LIGHTING = 1 << 10
FLAG_REFILL_LANES = 1 << 11
FLAG_TRACE_TIMES = 1 << 12
FLAG_SERIAL_CHUNKS = 1 << 13
FLAG_OVERWRITE = 1 << 14
FLAG_STORE_CAMERA_RAYS = 1 << 15
FLAG_ACCUMULATE_SAMPLES = 1 << 16
FLAG_STORE_LIGHT_RAYS = 1 << 17
FLAG_QUEUE_ORDER = 1 << 18

TONEMAP_NONE, TONEMAP_CBRT = 0, 1

# motion functions of the host API (Animatable.cpp:5-35), by rth_add_sphere code
MOTION = {"noMovement": 0, "oscilateX": 1, "oscilateY": 2, "rotateY": 3,
          "circle0": 4, "circle1": 5, "circle2": 6}

D3 = C.c_double * 3


class Sampler(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [
        ("color", D3), ("ns", C.c_double), ("ni", C.c_double), ("transparency", C.c_double),
        ("metalness", C.c_double), ("smoothness", C.c_double), ("emission", D3),
        ("color_fn", Sampler), ("ns_fn", Sampler), ("ni_fn", Sampler),
        ("transparency_fn", Sampler), ("metalness_fn", Sampler), ("smoothness_fn", Sampler),
    ]


class Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint64)]


class Light(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32), ("multisample", C.c_uint32), ("color", D3), ("position", D3),
        ("direction", D3), ("right", D3), ("up", D3), ("bottom_left", D3),
        ("width", C.c_double), ("height", C.c_double),
    ]


_PD = C.POINTER(C.c_double)
_PI32 = C.POINTER(C.c_int32)
_PU32 = C.POINTER(C.c_uint32)
_PU8 = C.POINTER(C.c_uint8)


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("kd_root", C.c_int32), ("n_branches", C.c_uint32), ("n_leaves", C.c_uint32),
        ("n_leaf_refs", C.c_uint32),
        ("branch_min", _PD), ("branch_max", _PD), ("branch_value", _PD),
        ("branch_axis", _PI32), ("branch_lesser", _PI32), ("branch_greater", _PI32),
        ("leaf_first", _PU32), ("leaf_count", _PU32), ("leaf_refs", _PU32),
        ("n_shapes", C.c_uint32), ("n_prims", C.c_uint32),
        ("prim_type", _PU8), ("prim_index", _PU32), ("prim_material", _PU32),
        ("prim_aabb_min", _PD), ("prim_aabb_max", _PD),
        ("n_triangles", C.c_uint32),
        ("tri_position", _PD), ("tri_normal", _PD), ("tri_uv", _PD),
        ("n_spheres", C.c_uint32),
        ("sphere_center", _PD), ("sphere_radius", _PD), ("sphere_r2", _PD),
        ("sphere_normal_matrix", _PD),
        ("n_lenses", C.c_uint32), ("lens_spheres", _PU32),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
        ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
        ("n_texel_bytes", C.c_uint64), ("texels", _PU8),
        ("n_lights", C.c_uint32), ("lights", C.POINTER(Light)),
    ]


class Camera(C.Structure):
    _fields_ = [("eye", D3), ("right", D3), ("up", D3), ("forward", D3), ("focal", C.c_double)]


class RenderDesc(C.Structure):
    _fields_ = [
        ("integrator", C.c_uint32), ("flags", C.c_uint32),
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("tile_x0", C.c_uint32), ("tile_y0", C.c_uint32),
        ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
        ("pixel_multisample_n", C.c_uint32), ("light_multisample_n", C.c_uint32),
        ("max_depth", C.c_uint32), ("first_pass", C.c_uint32), ("passes", C.c_uint32),
        ("frame", C.c_uint32), ("seed", C.c_uint64),
        ("bias", C.c_double), ("clear_color", D3), ("attenuation", D3),
        ("camera", Camera),
        ("global_samples", C.c_uint32), ("tile_row_group", C.c_uint32), ("tile_row_stride", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_secondary", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("branch_visits", C.c_uint64), ("leaf_visits", C.c_uint64), ("aabb_tests", C.c_uint64),
        ("triangle_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("hits_accepted", C.c_uint64),
        ("content_tests", C.c_uint64), ("fallback_rays", C.c_uint64),
        ("ms_generate", C.c_double), ("ms_traverse", C.c_double), ("ms_shade", C.c_double),
        ("ms_shadow", C.c_double), ("ms_resolve", C.c_double), ("ms_total", C.c_double),
        ("traverse_launches", C.c_uint32), ("max_queue", C.c_uint32),
        ("wave_steps", C.c_uint64), ("lane_steps", C.c_uint64),
    ]

    @property
    def rays(self):
        return self.rays_primary + self.rays_secondary + self.rays_shadow

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


def algorithmic_bytes(stats, walk="reference"):
    """Bytes the counted work has to read and write, record by record.

    Reference layout (walk "reference" / "kd-culled"): 72 B per KD branch visit
    (box, split value, axis, two children), 8 B per leaf visit (first, count),
    52 B per shape-box test (box + 4 B leaf reference), 72 B per triangle test
    (three positions), 32 B per sphere test (centre + r^2), 124 B per accepted
    hit (normals, uvs, material id) and 88 B of ray in / hit out; the culled KD
    walk additionally reads one 48 B content box per node it considers.

    Walk "bvh": a 64 B node per two child-box tests (32 B per test), 80 B per
    triangle test (positions + shape id + type, inline in the leaf), 48 B per
    shape-box test (only made for hits that could change the result); the KD
    branch visits counted are those of the membership verification (72 B).
    """
    rays = stats.rays_primary + stats.rays_secondary + stats.rays_shadow
    if walk == "bvh":
        return (32 * stats.content_tests + 72 * stats.branch_visits + 8 * stats.leaf_visits + 48 * stats.aabb_tests
                + 80 * stats.triangle_tests + 32 * stats.sphere_tests + 124 * stats.hits_accepted
                + 88 * rays)
    return (48 * stats.content_tests + 72 * stats.branch_visits + 8 * stats.leaf_visits + 52 * stats.aabb_tests
            + 72 * stats.triangle_tests + 32 * stats.sphere_tests + 124 * stats.hits_accepted
            + 88 * rays)
