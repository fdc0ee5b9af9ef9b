// capi.cpp -- C entry points over the host scene classes, so a scene can be
// assembled from Python (ctypes) exactly as the reference's main() assembles it
// in C++ (Raytracer/Raytracer.cpp:448-557): create materials, push shapes and
// lights, then per frame redoAABBs + buildKDTree + flatten.
#include <chrono>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene.hpp"

using namespace rth;

struct rth_scene {
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Shape>> shapes;
    std::vector<std::unique_ptr<Light>> lights;
    std::vector<std::unique_ptr<Model>> models;
    FrameInfo fi;
    FlatScene flat;
    AABB sceneBox;
    std::string error;
    double timings[4] = {0, 0, 0, 0};      // seconds: redoAABBs, buildKDTree, flatten, freeing the old tree
    int builder = 0;                        // 0: buildFlatKD; 1: buildKDTree's node objects + flattening walk
    int threads = 0;
};

static MovementFunction motionOf(int kind) {
    switch (kind) {
    case 0: return noMovement;
    case 1: return oscilateX;
    case 2: return oscilateY;
    case 3: return rotateY;
    case 4: return circle0;
    case 5: return circle1;
    case 6: return circle2;
    default: throw std::invalid_argument("unknown motion function");
    }
}

static Sampler samplerOf(rth_scene* s, uint32_t kind, int texture) {
    switch (kind) {
    case RTX_SAMPLER_CONST: return Sampler();
    case RTX_SAMPLER_RY_CHECKERS: return ryCheckers10x10;
    case RTX_SAMPLER_DOUBLE_CHECKERS: return doubleCheckers10x10;
    case RTX_SAMPLER_TEXTURE_VEC3: return s->textures.at((size_t)texture)->getVec3Sampler();
    case RTX_SAMPLER_TEXTURE_DOUBLE: return s->textures.at((size_t)texture)->getDoubleSampler();
    default: throw std::invalid_argument("unknown sampler kind");
    }
}

static dvec3 v3(const double* p) { return dvec3(p[0], p[1], p[2]); }

#define RTH_TRY(scene, body)                                   \
    try { body }                                               \
    catch (const std::exception& e) { (scene)->error = e.what(); return -1; }

extern "C" {

rth_scene* rth_scene_new(void) { return new rth_scene(); }
void rth_scene_free(rth_scene* s) {
    if (!s) return;
    delete s->fi.kdTree;
    delete s;
}
const char* rth_scene_error(const rth_scene* s) { return s->error.c_str(); }

int rth_texture_add(rth_scene* s, const uint8_t* rgb, int width, int height) {
    RTH_TRY(s, {
        if (!rgb || width <= 0 || height <= 0) throw std::invalid_argument("bad texture");
        s->textures.emplace_back(new Texture(rgb, width, height, 3));
        return (int)s->textures.size() - 1;
    })
}

// Material(premade, emission, colorFn): Material.cpp:81-107
int rth_material_preset(rth_scene* s, const char* premade, const double* emission,
                        uint32_t color_sampler, int color_texture) {
    RTH_TRY(s, {
        dvec3 e = emission ? v3(emission) : dvec3(0.0);
        s->materials.emplace_back(new Material(std::string(premade), e, samplerOf(s, color_sampler, color_texture)));
        return (int)s->materials.size() - 1;
    })
}

// Material(color, ns, ni, transparency, metalness, smoothness, emission, ...Fn): Material.cpp:38-69
// samplers: 6 (kind, texture) pairs in the order color, ns, ni, transparency, metalness, smoothness
int rth_material_custom(rth_scene* s, const double* color, double ns, double ni, double transparency,
                        double metalness, double smoothness, const double* emission,
                        const uint32_t* sampler_kinds, const int* sampler_textures) {
    RTH_TRY(s, {
        Sampler f[6];
        for (int i = 0; i < 6; ++i)
            f[i] = sampler_kinds ? samplerOf(s, sampler_kinds[i], sampler_textures ? sampler_textures[i] : 0) : Sampler();
        s->materials.emplace_back(new Material(v3(color), ns, ni, transparency, metalness, smoothness,
                                               emission ? v3(emission) : dvec3(0.0), f[0], nullptr, f[1], f[2],
                                               f[3], f[4], f[5]));
        return (int)s->materials.size() - 1;
    })
}

int rth_material_set_color(rth_scene* s, int material, const double* color) {
    RTH_TRY(s, { s->materials.at((size_t)material)->setColor(v3(color)); return 0; })
}

// n triangles: position/normal [n*9], uv [n*6]; Triangle(Vertex, Vertex, Vertex, Material*)
int rth_add_triangles(rth_scene* s, const double* position, const double* normal, const double* uv,
                      size_t n, int material) {
    RTH_TRY(s, {
        Material* m = s->materials.at((size_t)material).get();
        int first = (int)s->shapes.size();
        s->shapes.reserve(s->shapes.size() + n);
        for (size_t i = 0; i < n; ++i) {
            Vertex v[3];
            for (int k = 0; k < 3; ++k) {
                const double* p = position + 9 * i + 3 * k;
                dvec3 nn = normal ? v3(normal + 9 * i + 3 * k) : dvec3(0.0, 1.0, 0.0);
                dvec2 t = uv ? dvec2(uv[6 * i + 2 * k], uv[6 * i + 2 * k + 1]) : dvec2(0.0);
                v[k] = Vertex(v3(p), nn, t);
            }
            s->shapes.emplace_back(new Triangle(v[0], v[1], v[2], m));
        }
        return first;
    })
}

// addModel (Raytracer.cpp:362-366): Model(modelname, pos, rot), its triangles appended to the
// shapes.  material >= 0: every triangle gets that material; material < 0: the model's own
// materials, where slot k samples scene texture texture_ids[k] when that is >= 0 (slot order
// = rth_fbx_texture_names).  Returns the index of the first triangle added.
int rth_add_model(rth_scene* s, const char* models_root, const char* modelname, const double* pos, const double* rot,
                  int material, const int* texture_ids, int n_texture_ids) {
    RTH_TRY(s, {
        int next = 0;
        auto loader = [&](const std::string&) -> Texture* {
            int slot = next++;
            if (!texture_ids || slot >= n_texture_ids || texture_ids[slot] < 0) return nullptr;
            return s->textures.at((size_t)texture_ids[slot]).get();
        };
        std::unique_ptr<Model> m(new Model(models_root, modelname, pos ? v3(pos) : dvec3(0.0), rot ? v3(rot) : dvec3(0.0),
                                           loader));
        int first = (int)s->shapes.size();
        Material* forced = material >= 0 ? s->materials.at((size_t)material).get() : nullptr;
        for (size_t i = 0; i < m->owned.size(); ++i) {
            if (forced) {
                Triangle* t = m->owned[i].get();
                s->shapes.emplace_back(new Triangle((*t)[0], (*t)[1], (*t)[2], forced));
            } else {
                s->shapes.emplace_back(m->owned[i].release());
            }
        }
        m->owned.clear();
        m->children.clear();
        s->models.push_back(std::move(m));
        return first;
    })
}

// Diffuse texture file names of an FBX file's materials, one per line in slot order (an empty
// line: no texture), into `out`; returns the number of triangles in the file, -1 on error
// (message in `out`).
long rth_fbx_probe(const char* path, char* out, size_t out_len) {
    try {
        FbxMesh mesh;
        loadFbx(path, dmat4(1.0), mesh);
        std::string names;
        for (const std::string& f : mesh.materialTextures) { names += f; names += '\n'; }
        if (out && out_len) { std::strncpy(out, names.c_str(), out_len - 1); out[out_len - 1] = 0; }
        return (long)mesh.material.size();
    } catch (const std::exception& e) {
        if (out && out_len) { std::strncpy(out, e.what(), out_len - 1); out[out_len - 1] = 0; }
        return -1;
    }
}

int rth_add_sphere(rth_scene* s, const double* origin, double radius, int material, int motion) {
    RTH_TRY(s, {
        s->shapes.emplace_back(new Sphere(v3(origin), radius, s->materials.at((size_t)material).get(), motionOf(motion)));
        return (int)s->shapes.size() - 1;
    })
}

int rth_add_biconvex(rth_scene* s, const double* origin, const double* forward, double lens_radius,
                     double focal_length) {
    RTH_TRY(s, {
        s->shapes.emplace_back(new Biconvex(v3(origin), v3(forward), lens_radius, focal_length, noMovement));
        return (int)s->shapes.size() - 1;
    })
}

int rth_add_point_light(rth_scene* s, const double* position, const double* color) {
    RTH_TRY(s, { s->lights.emplace_back(new PointLight(v3(position), v3(color))); return (int)s->lights.size() - 1; })
}
int rth_add_directional_light(rth_scene* s, const double* direction, const double* color) {
    RTH_TRY(s, { s->lights.emplace_back(new DirectionalLight(v3(direction), v3(color))); return (int)s->lights.size() - 1; })
}
int rth_add_square_light(rth_scene* s, const double* position, const double* normal, double width,
                         double height, const double* color) {
    RTH_TRY(s, {
        s->lights.emplace_back(new SquareLight(v3(position), v3(normal), width, height, v3(color)));
        return (int)s->lights.size() - 1;
    })
}

// One frame of the reference's loop head (Raytracer.cpp:618,634-636): set the
// time, redoAABBs, buildKDTree, then flatten.  The returned description stays
// valid until the next call on this scene.
const rtx_scene_desc* rth_frame(rth_scene* s, double current_time) {
    try {
        s->fi.shapes.clear();
        for (auto& p : s->shapes) s->fi.shapes.push_back(p.get());
        s->fi.lights.clear();
        for (auto& p : s->lights) s->fi.lights.push_back(p.get());
        s->fi.currentTime = current_time;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t0 = now();
        s->sceneBox = redoAABBs(s->fi);
        double t1 = now();
        delete s->fi.kdTree;
        s->fi.kdTree = nullptr;
        double t2 = now();
        double t3;
        if (s->builder == 1) {
            // the reference's own construction: KDNode objects, then the flattening walk
            s->fi.kdTree = buildKDTree(s->fi.shapes, s->sceneBox);
            t3 = now();
            s->flat.build(s->fi);
        } else {
            // the same tree built straight into the flat arrays (timings[1] stays 0)
            t3 = t2;
            s->flat.build(s->fi, s->sceneBox, s->threads);
        }
        double t4 = now();
        s->timings[0] = t1 - t0; s->timings[1] = t3 - t2; s->timings[2] = t4 - t3; s->timings[3] = t2 - t1;
        return &s->flat.desc();
    } catch (const std::exception& e) {
        s->error = e.what();
        return nullptr;
    }
}

// builder 0: buildFlatKD (default), 1: buildKDTree's node objects + flattening walk; threads <= 0: automatic
void rth_set_kd_builder(rth_scene* s, int builder, int threads) { s->builder = builder; s->threads = threads; }

// branches, leaves, leaf refs, depth, largest leaf of the last rth_frame
void rth_kd_stats(const rth_scene* s, uint32_t out[5]) {
    out[0] = s->flat.stats.branches; out[1] = s->flat.stats.leaves; out[2] = s->flat.stats.leafRefs;
    out[3] = s->flat.stats.depth; out[4] = s->flat.stats.maxLeaf;
}

// seconds spent by the last rth_frame in redoAABBs, buildKDTree, flatten, freeing the previous tree
void rth_frame_timings(const rth_scene* s, double out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = s->timings[i];
}

void rth_scene_box(const rth_scene* s, double out[6]) {
    for (int i = 0; i < 3; ++i) { out[i] = s->sceneBox.min[i]; out[3 + i] = s->sceneBox.max[i]; }
}

// Headless stand-in for saveImage (Raytracer.cpp:90-105, stbi_write_png): 8-bit RGB rows, top
// row first, into a PNG file (one zlib stream, filter 0 on every scanline).  0 on success.
int rth_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    return writePng(path, rgb8, width, height) ? 0 : -1;
}

// camera basis of Raytracer.cpp:642-660
void rth_make_camera(const double* eye, const double* lookat, double fov_degrees, rtx_camera* out) {
    *out = makeCamera(v3(eye), v3(lookat), fov_degrees);
}

}  // extern "C"
