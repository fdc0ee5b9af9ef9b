"""Scene recipes, built through the host scene API.

`live_scene` is the scene hard-coded in the reference's main()
(Raytracer/Raytracer.cpp:458-527).  The others are the benchmark
configurations: the reference ships no scene files, so they are assembled from
the pieces its three main() variants show (Raytracer-Noah-Desktop.cpp:665-754,
Raytracer-noah-laptop.cpp:584-611) with seeded procedural geometry standing in
for the FBX meshes (which need the absent assimp to import).
"""
import os

import numpy as np

from . import abi
from .host import HostScene, make_camera


def render_desc(width, height, integrator=abi.INTEGRATOR_WHITTED, camera=None, n=1, light_n=1, max_depth=3,
                bias=1e-3, clear=(0.0, 0.0, 0.0), flags=0, passes=1, first_pass=0, seed=0, frame=0, tile=None,
                global_samples=0, row_group=0, row_stride=0):
    rd = abi.RenderDesc()
    rd.integrator = integrator
    rd.flags = flags
    rd.width, rd.height = width, height
    x0, y0, w, h = tile if tile is not None else (0, 0, width, height)
    rd.tile_x0, rd.tile_y0, rd.tile_w, rd.tile_h = x0, y0, w, h
    rd.tile_row_group, rd.tile_row_stride = row_group, row_stride      # interleaved rows: see rtx.h
    rd.pixel_multisample_n = n
    rd.light_multisample_n = light_n
    rd.max_depth = max_depth
    rd.first_pass, rd.passes = first_pass, passes
    rd.frame, rd.seed = frame, seed
    rd.global_samples = global_samples          # MONTE_CARLO_SAMPLES of the Whitted variant, Desktop :73
    rd.bias = bias
    rd.clear_color[:] = clear
    rd.attenuation[:] = (1.0, 0.09, 0.032)      # LIGHT_CONTSTANT/LINEAR/QUADRATIC, Desktop :78-80
    if camera is not None:
        rd.camera = camera
    return rd


def _floor(scene, material, half=1000.0, y=0.0, order=("acb", "adc")):
    """Two floor triangles with uv 0..1 over the quad (Raytracer.cpp:475-500)."""
    v = {"a": ((-half, y, -half), (0.0, 0.0)), "b": ((half, y, -half), (1.0, 0.0)),
         "c": ((half, y, half), (1.0, 1.0)), "d": ((-half, y, half), (0.0, 1.0))}
    pos, nrm, uv = [], [], []
    for tri in order:
        for k in tri:
            pos.append(v[k][0])
            nrm.append((0.0, 1.0, 0.0))
            uv.append(v[k][1])
    scene.triangles(np.array(pos), np.array(nrm), np.array(uv), material)


def live_scene():
    """The scene of Raytracer.cpp:458-527: 2 floor triangles, mirror sphere, 4 glass
    spheres, 3 emissive spheres on circle0/1/2.  Camera of :642-660."""
    s = HostScene()
    white = s.material("PlainWhiteTees", color=(1.0, 1.0, 1.0))
    glass = s.material("Glass")
    mirror = s.material("Mirror")
    _floor(s, white)
    s.sphere((0.0, 2.5, 0.0), 5.0, mirror)
    for c in [(-7.0, 1.5, 7.0), (-7.0, 1.5, -7.0), (7.0, 1.5, 7.0), (7.0, 1.5, -7.0)]:
        s.sphere(c, 1.5, glass)
    for k, e in enumerate([(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]):
        s.sphere((0.0, 0.0, 0.0), 1.0, s.material("PlainWhiteTees", emission=e), motion=f"circle{k}")
    cam = make_camera((0.0, 35.0, 50.0), (0.0, 0.0, 0.0))
    return s, cam


def whitted_scene():
    """Config 1: glass + mirror sphere over a red/yellow checker floor, one PointLight."""
    s = HostScene()
    checkers = s.material("PlainWhiteTees", color_sampler=abi.SAMPLER_RY_CHECKERS)
    glass = s.material("Glass")
    mirror = s.material("Mirror")
    _floor(s, checkers)
    s.sphere((-2.2, 2.0, 0.0), 2.0, glass)
    s.sphere((2.4, 1.6, -2.5), 1.6, mirror)
    s.point_light((5.0, 5.0, 5.0), (1.0, 1.0, 1.0))
    cam = make_camera((0.0, 3.0, 9.0), (0.0, 1.2, 0.0))
    return s, cam


def bumpy_sphere(slices=246, stacks=123, radius=3.0, center=(0.0, 0.5, 0.0), bump=0.08, seed=12345):
    """Seeded procedural triangle soup: a UV sphere whose radius is modulated by a few
    random sinusoids.  slices*(stacks-1)*2 triangles (60 024 by default), smooth
    per-vertex normals, uv = (longitude, latitude)."""
    rng = np.random.default_rng(seed)
    k = rng.integers(2, 9, size=(6, 2)).astype(np.float64)
    ph = rng.uniform(0.0, 2 * np.pi, size=6)
    amp = rng.uniform(0.3, 1.0, size=6)
    amp /= amp.sum()
    i = np.arange(slices + 1)
    j = np.arange(stacks + 1)
    theta = 2 * np.pi * (i % slices) / slices                # longitude
    phi = np.pi * j / stacks                                 # colatitude
    T, P = np.meshgrid(theta, phi, indexing="ij")
    r = np.ones_like(T)
    for q in range(6):
        r += bump * amp[q] * np.sin(k[q, 0] * T + ph[q]) * np.sin(k[q, 1] * P)
    d = np.stack([np.sin(P) * np.cos(T), np.cos(P), np.sin(P) * np.sin(T)], axis=-1)
    pts = np.asarray(center) + radius * r[..., None] * d
    nrm = d.copy()
    U, Vv = np.meshgrid(i / slices, j / stacks, indexing="ij")
    uvs = np.stack([U, 1.0 - Vv], axis=-1)
    pos, nn, uu = [], [], []

    def quad_tris(i0, j0):
        a, b, c, dd = (i0, j0), (i0 + 1, j0), (i0 + 1, j0 + 1), (i0, j0 + 1)
        out = []
        if j0 > 0:
            out.append((a, b, c))
        if j0 < stacks - 1:
            out.append((a, c, dd))
        return out

    idx = []
    for i0 in range(slices):
        for j0 in range(stacks):
            idx.extend(quad_tris(i0, j0))
    idx = np.array(idx)                                       # [n,3,2]
    ii, jj = idx[..., 0], idx[..., 1]
    pos = pts[ii, jj]                                         # [n,3,3]
    nn = nrm[ii, jj]
    uu = uvs[ii, jj]
    # outward-facing winding: the reference culls triangles whose determinant is negative
    e1 = pos[:, 1] - pos[:, 0]
    e2 = pos[:, 2] - pos[:, 0]
    face = np.cross(e1, e2)
    flip = np.einsum("ij,ij->i", face, nn[:, 0] + nn[:, 1] + nn[:, 2]) < 0
    pos[flip] = pos[flip][:, [0, 2, 1]]
    nn[flip] = nn[flip][:, [0, 2, 1]]
    uu[flip] = uu[flip][:, [0, 2, 1]]
    return pos.reshape(-1, 9), nn.reshape(-1, 9), uu.reshape(-1, 6)


def checker_texture(size=256, cells=8, seed=7):
    """Seeded RGB8 albedo texture (stand-in for the FBX models' albedo maps)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size]
    cell = ((x * cells // size) + (y * cells // size)) % 2
    base = np.where(cell[..., None] == 0, np.array([200, 120, 60]), np.array([60, 120, 200]))
    noise = rng.integers(-20, 21, size=(size, size, 3))
    return np.clip(base + noise, 0, 255).astype(np.uint8)


# the size of the backpack model's diffuse map (Raytracer/Models/backpack/1001_albedo.jpg: 4096 x 4096,
# forced to three channels by loadTexture, Texture.cpp:9-27): 48 MB of texels
ALBEDO_SIZE = 4096
_albedo_cache = {}


def albedo_texture(size=ALBEDO_SIZE, seed=7):
    """Seeded RGB8 albedo map of the reference model's size: patches of a few hundred texels in
    different base colours, a fine weave and per-texel noise on top, so that neighbouring texels
    differ and a wrong texel shows.  The model's own map is a JPEG the image cannot ship as a
    fixture at this size; what the sampler does (nearest texel, wrap, Texture.cpp:39-55) and
    what it costs (one 3-byte read somewhere in 48 MB per hit) depend on the size and on the
    model's uvs, not on the picture.  Cached per process."""
    key = (size, seed)
    if key not in _albedo_cache:
        rng = np.random.default_rng(seed)
        cells = max(1, size // 256)
        palette = rng.integers(40, 216, size=(cells, cells, 3)).astype(np.int16)
        img = np.repeat(np.repeat(palette, size // cells, axis=0), size // cells, axis=1)
        if img.shape[0] != size:            # sizes that are no multiple of the patch count
            img = np.pad(img, ((0, size - img.shape[0]), (0, size - img.shape[1]), (0, 0)), mode="edge")
        i = np.arange(size, dtype=np.int16)
        weave = ((i[:, None] // 4 + i[None, :] // 4) % 2) * 16 - 8
        img += weave[..., None]
        img += rng.integers(-12, 13, size=(size, size, 1), dtype=np.int16)
        tex = np.clip(img, 0, 255).astype(np.uint8)
        tex.setflags(write=False)
        _albedo_cache[key] = tex
    return _albedo_cache[key]


_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def reference_mesh(name):
    """Triangle soup of one of the reference's FBX models (Raytracer/Models/<name>/<name>.fbx as
    addModel places it, Raytracer.cpp:362-366), from the fixture tools/make_mesh_fixture.py
    wrote: per-corner position [n,9], normal [n,9], uv [n,6], bit for bit what the FBX reader
    hands to the Triangle constructor.  'backpack' (67 908 triangles) is the model the
    reference's mesh scene loads (Raytracer-noah-laptop.cpp:597); 'teapot' has 15 704."""
    z = np.load(os.path.join(_GOLDEN, f"mesh_{name}.npz"))
    c = z["records"][z["index"]]                # [n, 3, 8]
    n = len(c)
    return (np.ascontiguousarray(c[:, :, 0:3]).reshape(n, 9), np.ascontiguousarray(c[:, :, 3:6]).reshape(n, 9),
            np.ascontiguousarray(c[:, :, 6:8]).reshape(n, 6))


def mesh_scene(slices=246, stacks=123, t=0.0, ggx=False, square_light=False, mesh=None, texture_size=256):
    """Configs 2 and 4: a triangle mesh + floor.  Config 2: constant white material,
    DirectionalLight((-1,-1,-1)) (Raytracer-noah-laptop.cpp:608-611), orbit camera
    eye = (10 sin t, 0.1, 10 cos t) (:672).  Config 4 (ggx=True): smoothness/metalness
    > 0, albedo texture on the mesh, checker floor and (square_light) a 1x1 SquareLight
    at (0,10,0) facing down (Raytracer-Noah-Desktop.cpp:754).  `mesh` names a reference
    model (reference_mesh) instead of the seeded procedural soup.  `texture_size`: side of the
    albedo map; ALBEDO_SIZE (4096) is the size of the map the reference binds to the backpack
    (Model.cpp:91-94), addressed by the model's own uvs; the default is a small checker."""
    s = HostScene()
    if ggx:
        tex = s.texture(checker_texture() if texture_size == 256 else albedo_texture(texture_size))
        mesh_mat = s.material_custom((1.0, 1.0, 1.0), ns=100.0, ni=1.0, transparency=0.0, metalness=0.5,
                                     smoothness=0.5, samplers={"color": (abi.SAMPLER_TEXTURE_VEC3, tex)})
        floor_mat = s.material("PlainWhiteTees", color_sampler=abi.SAMPLER_RY_CHECKERS)
    else:
        mesh_mat = s.material("PlainWhiteTees")
        floor_mat = s.material("PlainWhiteTees", color=(0.8, 0.8, 0.8))
    pos, nrm, uv = reference_mesh(mesh) if mesh else bumpy_sphere(slices, stacks)
    s.triangles(pos, nrm, uv, mesh_mat)
    _floor(s, floor_mat, half=20.0, y=-3.0)
    if square_light:
        s.square_light((0.0, 10.0, 0.0), (0.0, -1.0, 0.0), 1.0, 1.0)
    else:
        s.directional_light((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))
    cam = make_camera((10.0 * np.sin(t), 0.1, 10.0 * np.cos(t)), (0.0, 0.0, 0.0))
    return s, cam


def frame16_scene(kind, mesh="backpack", texture_size=ALBEDO_SIZE):
    """The two 1080p 16-spp workloads of BASELINE.json configs[3], as bench.py times them and the
    tests check them.  kind 'whitted': the mesh with GGX material parameters and its albedo map,
    checker floor, one 1x1 SquareLight (rayTrace, soft shadows).  kind 'path': the same under
    pathTrace, lit by an emissive sphere where the square light is (pathTrace knows no Light
    objects, Raytracer.cpp:217-355)."""
    scene, _ = mesh_scene(ggx=True, square_light=True, mesh=mesh, texture_size=texture_size)
    if kind == "path":
        scene.sphere((0.0, 10.0, 0.0), 2.0, scene.material("PlainWhiteTees", emission=(8.0, 8.0, 8.0)))
    return scene


def frame16_render_desc(kind, width, height, frame, flags=0, tile=None, row_group=0, row_stride=0, n=4, light_n=2):
    """Render description of frame `frame` of those workloads: orbit camera at t = frame/30,
    n x n sub-samples; rayTrace with light_n x light_n SquareLight samples per hit, or pathTrace
    to depth 8."""
    cam = orbit_camera(frame / 30.0)
    if kind == "whitted":
        return render_desc(width, height, integrator=abi.INTEGRATOR_WHITTED, camera=cam, n=n, light_n=light_n, max_depth=3,
                           bias=1e-2, frame=frame, tile=tile, row_group=row_group, row_stride=row_stride, flags=flags)
    return render_desc(width, height, integrator=abi.INTEGRATOR_PATH, camera=cam, n=n, max_depth=8, bias=1e-3, frame=frame,
                       tile=tile, row_group=row_group, row_stride=row_stride, flags=flags)


def lens_scene():
    """Config 3: the Whitted scene plus a Biconvex lens, a mirror sphere and an emissive
    sphere as the light (path-tracing semantics, Raytracer.cpp:525-527)."""
    s = HostScene()
    checkers = s.material("PlainWhiteTees", color_sampler=abi.SAMPLER_RY_CHECKERS)
    glass = s.material("Glass")
    mirror = s.material("Mirror")
    copper = s.material("Copper")
    lamp = s.material("PlainWhiteTees", emission=(6.0, 6.0, 6.0))
    _floor(s, checkers)
    s.sphere((-2.2, 2.0, 0.0), 2.0, glass)
    s.sphere((2.4, 1.6, -2.5), 1.6, mirror)
    s.sphere((4.5, 1.0, 2.0), 1.0, copper)
    s.sphere((0.0, 12.0, 3.0), 4.0, lamp)
    s.biconvex((0.0, 2.0, 4.0), (0.0, 0.0, 1.0), 3.0, 50.0)
    cam = make_camera((0.0, 3.0, 9.0), (0.0, 1.2, 0.0))
    return s, cam


def animated_scene(copies=1, slices=246, stacks=123):
    """Config 5: `copies` bumpy spheres (about 60k triangles each) + floor + three emissive
    spheres on circle0/1/2 (Animatable.cpp:22-35); the caller picks t = frame/30."""
    s = HostScene()
    floor_mat = s.material("PlainWhiteTees", color_sampler=abi.SAMPLER_RY_CHECKERS)
    _floor(s, floor_mat, half=1000.0, y=-3.0)
    side = int(np.ceil(np.sqrt(copies)))
    for c in range(copies):
        gx, gz = c % side, c // side
        mat = s.material_custom((0.9, 0.9, 0.9), metalness=0.3 + 0.4 * ((c * 7) % 5) / 5.0, smoothness=0.4)
        center = ((gx - (side - 1) / 2) * 7.0, 0.5, (gz - (side - 1) / 2) * 7.0)
        rng_seed = 12345 + c
        pos, nrm, uv = bumpy_sphere(slices, stacks, center=center, seed=rng_seed)
        s.triangles(pos, nrm, uv, mat)
    for k, e in enumerate([(8.0, 2.0, 2.0), (2.0, 8.0, 2.0), (2.0, 2.0, 8.0)]):
        s.sphere((0.0, 0.0, 0.0), 1.0, s.material("PlainWhiteTees", emission=e), motion=f"circle{k}")
    return s


def scalar_sampler_scene():
    """Every scalar material channel driven by a sampler (Material.cpp:111-161): ns, ni,
    transparency, metalness and smoothness come from doubleCheckers10x10 (Procedural.cpp:16-25)
    or from Texture::doubleSample (Texture.cpp:56-64) somewhere in the scene, under a point light
    (rayTrace) and an emissive sphere (pathTrace).  Texture values stay inside (0.2, 0.9) so that
    sampled indices of refraction and exponents are ordinary numbers."""
    s = HostScene()
    rng = np.random.default_rng(21)
    grey = rng.integers(60, 230, size=(32, 48, 3)).astype(np.uint8)
    tex = s.texture(grey)
    ck, td = abi.SAMPLER_DOUBLE_CHECKERS, abi.SAMPLER_TEXTURE_DOUBLE
    floor = s.material_custom((1.0, 1.0, 1.0), ns=20.0, ni=1.0, transparency=0.0, metalness=0.3, smoothness=0.5,
                              samplers={"color": (abi.SAMPLER_RY_CHECKERS, 0), "smoothness": (ck, 0), "metalness": (td, tex)})
    _floor(s, floor, half=12.0)
    glassy = s.material_custom((0.9, 0.95, 1.0), ns=30.0, ni=1.5, transparency=0.5, metalness=0.1, smoothness=0.6,
                               samplers={"transparency": (td, tex), "ns": (ck, 0)})
    s.sphere((-2.2, 2.0, 0.0), 2.0, glassy)
    patchy = s.material_custom((1.0, 0.8, 0.6), ns=10.0, ni=1.3, transparency=0.0, metalness=0.5, smoothness=0.5,
                               samplers={"transparency": (ck, 0), "ni": (td, tex), "ns": (td, tex)})
    s.sphere((2.4, 1.6, -1.5), 1.6, patchy)
    pos, nrm, uv = bumpy_sphere(40, 20, radius=1.2, center=(0.3, 1.2, 2.8))
    mesh = s.material_custom((0.7, 0.9, 0.7), ns=50.0, ni=1.0, transparency=0.0, metalness=0.5, smoothness=0.5,
                             samplers={"smoothness": (td, tex), "metalness": (ck, 0)})
    s.triangles(pos, nrm, uv, mesh)
    s.sphere((0.0, 14.0, 3.0), 4.0, s.material("PlainWhiteTees", emission=(5.0, 5.0, 5.0)))
    s.point_light((5.0, 6.0, 6.0), (1.0, 1.0, 1.0))
    cam = make_camera((0.0, 3.5, 9.5), (0.0, 1.2, 0.0))
    return s, cam


def orbit_camera(t, radius=10.0, height=0.1):
    return make_camera((radius * np.sin(t), height, radius * np.cos(t)), (0.0, 0.0, 0.0))
