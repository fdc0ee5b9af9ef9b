// scene.cpp -- host scene classes, KD construction and the flattener that turns
// one frame's FrameInfo into the POD arrays of rtx_scene_desc.
#include "scene.hpp"

#include <algorithm>
#include <cstring>
#include <future>
#include <stdexcept>
#include <unordered_map>

namespace rth {

const Sampler ryCheckers10x10(RTX_SAMPLER_RY_CHECKERS);
const Sampler doubleCheckers10x10(RTX_SAMPLER_DOUBLE_CHECKERS);

Texture::Texture(const unsigned char* src, int width, int height, int nrComponents)
    : data(src, src + (size_t)width * height * 3), width(width), height(height), nrComponents(nrComponents) {}

// preset table, Material.cpp:10-19
static const std::unordered_map<std::string, materialStats>& presets() {
    static const std::unordered_map<std::string, materialStats> m = {
        {"Glass", {dvec3(1.0, 1.0, 1.0), 50.0, 1.54, 0.8, 0.2, 1.0}},
        {"PlainWhiteTees", {dvec3(1.0, 1.0, 1.0), 100.0, 1.0, 0.0, 0.0, 0.0}},
        {"Red", {dvec3(1.0, 0.0, 0.0), 2.0, 1.0, 0.0, 0.0, 0.5}},
        {"Green", {dvec3(0.0, 1.0, 0.0), 100.0, 1.0, 0.0, 0.6, 0.0}},
        {"Bug", {dvec3(1.0, 0.0, 1.0), 100.0, 1.0, 0.0, 0.0, 1.0}},
        {"Copper", {dvec3(0.7038, 0.27048, 0.0828), 100.0, 1.0, 0.0, 0.5, 0.5}},
        {"Mirror", {dvec3(1.0, 1.0, 1.0), 50.0, 0.13511, 0.0, 1.0, 1.0}},
        {"MirrorB", {dvec3(0.7038, 0.27048, 0.0828), 50.0, 1.0, 0.0, 0.6, 0.6}},
    };
    return m;
}

Material::Material()   // Material.cpp:28-35
    : color(1.0, 0.0, 1.0), ns(100.0), ni(1.0), transparency(0.0), metalness(0.5), smoothness(0.5), emission(0.0) {}

Material::Material(dvec3 color, double ns, double ni, double transparency, double metalness,
                   double smoothness, dvec3 emission, Sampler colorFn, Sampler normalFn, Sampler nsFn,
                   Sampler niFn, Sampler transparencyFn, Sampler metalnessFn, Sampler smoothnessFn)
    : colorFn(colorFn), normalFn(normalFn), nsFn(nsFn), niFn(niFn), transparencyFn(transparencyFn),
      metalnessFn(metalnessFn), smoothnessFn(smoothnessFn), color(color), ns(ns), ni(ni),
      transparency(transparency), metalness(metalness), smoothness(smoothness), emission(emission) {}

Material::Material(const std::string& premade, dvec3 emission, Sampler colorFn, Sampler normalFn,
                   Sampler nsFn, Sampler niFn, Sampler transparencyFn, Sampler metalnessFn,
                   Sampler smoothnessFn)
    : colorFn(colorFn), normalFn(normalFn), nsFn(nsFn), niFn(niFn), transparencyFn(transparencyFn),
      metalnessFn(metalnessFn), smoothnessFn(smoothnessFn), emission(emission) {
    const materialStats& s = presets().at(premade);
    color = s.color; ns = s.ns; ni = s.ni; transparency = s.transparency;
    metalness = s.metalness; smoothness = s.smoothness;
}

// ---- motion functions, Animatable.cpp:5-35 ------------------------------------
dmat4 noMovement(double) { return dmat4(1.0); }
dmat4 oscilateX(double t) { return translate(dmat4(1.0), dvec3(std::sin(t) * 5.0, 0.0, 0.0)); }
dmat4 oscilateY(double t) { return translate(dmat4(1.0), dvec3(0.0, std::sin(t) * 5.0, 0.0)); }
dmat4 rotateY(double t) { return rotate(dmat4(1.0), t / 3.0, dvec3(0.0, 1.0, 0.0)); }
dmat4 circle0(double t) {
    t += 0.000;
    return translate(dmat4(1.0), dvec3(std::sin(t) * 15, 15, std::cos(t) * 15));
}
dmat4 circle1(double t) {
    t += 0.333 * 2.0 * pi;
    return translate(dmat4(1.0), dvec3(std::sin(t) * 15, 15, std::cos(t) * 15));
}
dmat4 circle2(double t) {
    t += 0.666 * 2.0 * pi;
    return translate(dmat4(1.0), dvec3(std::sin(t) * 15, 15, std::cos(t) * 15));
}

dvec3 transformNormal(const dvec3& x, const dmat4& model) {
    dmat4 normalMat = inverse(transpose(model));
    dvec4 r = normalMat * dvec4(x, 1.0);
    return dvec3(r.x, r.y, r.z);
}
dvec3 transformPos(const dvec3& x, const dmat4& model) {
    dvec4 p = model * dvec4(x, 1.0);
    dvec4 q = p / p.w;
    return dvec3(q.x, q.y, q.z);
}

// ---- shapes ---------------------------------------------------------------------
void Triangle::redoAABB(double) {
    dvec3 a = vertices[0].position, b = vertices[1].position, c = vertices[2].position;
    dvec3 minB = rth::min(a, rth::min(b, c)) - dvec3(epsilon);
    dvec3 maxB = rth::max(a, rth::max(b, c)) + dvec3(epsilon);
    boundingBox = AABB(minB, maxB);
}

Sphere::Sphere(const dvec3& origin, double radius, Material* mat, MovementFunction mf)
    : Shape(AABB(origin - dvec3(radius), origin + dvec3(radius)), mat), Animatable(mf),
      radius(radius), origin(origin), r2(radius * radius) {}

void Sphere::redoAABB(double currentTime) {
    dvec3 t = transformPos(origin, getModel(currentTime));
    boundingBox = AABB(t - dvec3(radius), t + dvec3(radius));
}

double solveLensForD(double r, double f, double n) {
    double numerator = n * r * r;
    double denom = f * (n - 1) * (n - 1);
    return numerator / denom;
}

// Biconvex.cpp:29-51
AABB Biconvex::setupSpheres(const dvec3 origin, dvec3 forward, double sphereRadius, double focalLength) {
    glass = new Material("Glass");
    d = solveLensForD(sphereRadius, focalLength, 1.54);
    this->origin = origin;
    this->forward = normalize(forward);
    dvec3 s1Origin = origin - (this->forward * (sphereRadius - d));
    dvec3 s2Origin = origin + (this->forward * (sphereRadius - d));
    s1 = new Sphere(s1Origin, sphereRadius, glass, noMovement);
    s2 = new Sphere(s2Origin, sphereRadius, glass, noMovement);
    double lbsr = std::sqrt((sphereRadius - (d / 2)) * (8.0 * d)) / 2.0;
    return AABB(origin - dvec3(lbsr), origin + dvec3(lbsr));
}

Biconvex::Biconvex(const dvec3 origin, dvec3 forward, double lensRadius, double focalLength, MovementFunction mf)
    : Shape(AABB()), Animatable(mf), s1(nullptr), s2(nullptr), d(0), glass(nullptr) {
    boundingBox = setupSpheres(origin, forward, lensRadius, focalLength);
}

Biconvex::~Biconvex() { delete s1; delete s2; delete glass; }

// ---- lights ---------------------------------------------------------------------
static void put3(double* dst, dvec3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

rtx_light PointLight::flatten() const {
    rtx_light l;
    std::memset(&l, 0, sizeof l);
    l.kind = RTX_LIGHT_POINT; l.multisample = multisample;
    put3(l.color, color); put3(l.position, position);
    return l;
}

DirectionalLight::DirectionalLight(dvec3 direction, dvec3 color) : Light(color) {
    this->direction = normalize(direction);
    this->reverseDirection = -this->direction;
}
rtx_light DirectionalLight::flatten() const {
    rtx_light l;
    std::memset(&l, 0, sizeof l);
    l.kind = RTX_LIGHT_DIRECTIONAL; l.multisample = multisample;
    put3(l.color, color); put3(l.direction, reverseDirection);
    return l;
}

// SquareLight.cpp:8-41 (without the constructor's debug prints)
SquareLight::SquareLight(dvec3 position, dvec3 normal, double width, double height, dvec3 color)
    : Light(color, true) {
    dvec3 worldUp(0.0, 1.0, 0.0);
    this->normal = normalize(normal);
    this->right = normalize(cross(worldUp, this->normal));
    if (std::isinf(this->right.x) || std::isnan(this->right.x)) this->right = dvec3(1.0, 0.0, 0.0);
    this->up = normalize(cross(this->normal, this->right));
    this->position = position;
    this->width = width;
    this->height = height;
    this->bottomLeft = this->position - ((this->right * 0.5 * width) + (this->up * 0.5 * height));
    this->topRight = bottomLeft + (this->right * width) + (this->up * height);
}
rtx_light SquareLight::flatten() const {
    rtx_light l;
    std::memset(&l, 0, sizeof l);
    l.kind = RTX_LIGHT_SQUARE; l.multisample = multisample;
    put3(l.color, color); put3(l.position, position); put3(l.direction, normal);
    put3(l.right, right); put3(l.up, up); put3(l.bottom_left, bottomLeft);
    l.width = width; l.height = height;
    return l;
}

// ---- KD construction ---------------------------------------------------------------
// buildKDTree(shapes, box) of KDTree.hpp with the reference's node classes as the result.  The
// tree itself comes from buildFlatKD (kdbuild.cpp), which defines it; here its flat arrays are
// turned back into KDBranch / KDLeaf objects for callers that want to walk pointers.
namespace {
KDNode* inflate(const FlatKD& kd, int32_t node, const std::vector<Shape*>& shapes) {
    if (node < 0) {
        const uint32_t leaf = (uint32_t)~node;
        std::vector<Shape*> held;
        held.reserve(kd.leafCount[leaf]);
        for (uint32_t r = 0; r < kd.leafCount[leaf]; ++r) held.push_back(shapes[kd.leafRefs[kd.leafFirst[leaf] + r]]);
        return new KDLeaf(std::move(held));
    }
    AABB area;
    for (int k = 0; k < 3; ++k) { area.min[k] = kd.bmin[3 * (size_t)node + k]; area.max[k] = kd.bmax[3 * (size_t)node + k]; }
    KDNode* lesser = inflate(kd, kd.lesser[node], shapes);
    KDNode* greater = inflate(kd, kd.greater[node], shapes);
    return new KDBranch(greater, lesser, area, Axis(kd.axis[node]), kd.value[node]);
}
}  // namespace

KDNode* buildKDTree(std::vector<Shape*> shapes, AABB box, Axis axis, int recursionWithoutChange, int layer) {
    FlatKD kd;
    buildFlatKD(shapes, box, kd, 0, axis, recursionWithoutChange, layer);
    return inflate(kd, kd.root, shapes);
}

AABB redoAABBs(FrameInfo& fi) {
    AABB toReturn(dvec3(0.0), dvec3(0.0));
    for (Shape* s : fi.shapes) {
        s->redoAABB(fi.currentTime);
        toReturn.encompass(s->boundingBox);
    }
    return toReturn;
}

rtx_camera makeCamera(dvec3 eye, dvec3 lookat, double fovDegrees) {
    dvec3 camForward = normalize(lookat - eye);
    dvec3 camUp(0.0, 1.0, 0.0);
    dvec3 camRight = cross(camForward, camUp);
    camUp = cross(camRight, camForward);
    double viewPortHeight = 2.0;
    double focal = (viewPortHeight / 2.0) / std::tan((fovDegrees / 2.0) * 0.01745329251994329576923690768489);
    rtx_camera c;
    put3(c.eye, eye); put3(c.right, camRight); put3(c.up, camUp); put3(c.forward, camForward);
    c.focal = focal;
    return c;
}

// ---- flattener ------------------------------------------------------------------
namespace {

struct Flattener {
    FlatScene& out;
    const FrameInfo& fi;
    std::unordered_map<const Shape*, uint32_t> shapeId;
    std::unordered_map<const Material*, uint32_t> materialId;
    std::unordered_map<const Texture*, uint32_t> textureId;
    uint32_t maxDepth = 0;

    Flattener(FlatScene& o, const FrameInfo& f) : out(o), fi(f) {}

    rtx_sampler sampler(const Sampler& s) {
        rtx_sampler r;
        r.kind = s.kind; r.texture = 0;
        if (s.kind == RTX_SAMPLER_TEXTURE_VEC3 || s.kind == RTX_SAMPLER_TEXTURE_DOUBLE) {
            if (!s.texture) throw std::runtime_error("texture sampler without a texture");
            auto it = textureId.find(s.texture);
            if (it == textureId.end()) {
                rtx_texture t;
                t.width = (uint32_t)s.texture->width; t.height = (uint32_t)s.texture->height;
                t.offset = out.texels.size();
                out.texels.insert(out.texels.end(), s.texture->data.begin(), s.texture->data.end());
                it = textureId.emplace(s.texture, (uint32_t)out.textures.size()).first;
                out.textures.push_back(t);
            }
            r.texture = it->second;
        }
        return r;
    }

    uint32_t material(const Material* m) {
        if (!m) throw std::runtime_error("shape without a material");
        auto it = materialId.find(m);
        if (it != materialId.end()) return it->second;
        rtx_material r;
        std::memset(&r, 0, sizeof r);
        put3(r.color, m->color);
        r.ns = m->ns; r.ni = m->ni; r.transparency = m->transparency;
        r.metalness = m->metalness; r.smoothness = m->smoothness;
        put3(r.emission, m->emission);
        r.color_fn = sampler(m->colorFn); r.ns_fn = sampler(m->nsFn); r.ni_fn = sampler(m->niFn);
        r.transparency_fn = sampler(m->transparencyFn); r.metalness_fn = sampler(m->metalnessFn);
        r.smoothness_fn = sampler(m->smoothnessFn);
        uint32_t id = (uint32_t)out.materials.size();
        out.materials.push_back(r);
        materialId.emplace(m, id);
        return id;
    }

    void pushBox(const AABB& b) {
        for (int i = 0; i < 3; ++i) { out.primMin.push_back(b.min[i]); out.primMax.push_back(b.max[i]); }
    }

    uint32_t addSphere(const Sphere* s) {
        uint32_t idx = (uint32_t)out.sphRadius.size();
        dmat4 model = s->getModel(fi.currentTime);
        dvec3 c = transformPos(s->origin, model);
        dmat4 nm = inverse(transpose(model));
        for (int i = 0; i < 3; ++i) out.sphCenter.push_back(c[i]);
        out.sphRadius.push_back(s->radius);
        out.sphR2.push_back(s->r2);
        for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row) out.sphNormalMat.push_back(nm[col][row]);
        return idx;
    }

    void addPrim(uint8_t type, uint32_t index, uint32_t mat, const AABB& box) {
        out.primType.push_back(type); out.primIndex.push_back(index); out.primMaterial.push_back(mat);
        pushBox(box);
    }

    int32_t walk(const KDNode* n, uint32_t depth) {
        if (depth > maxDepth) maxDepth = depth;
        if (n->isLeaf) {
            const KDLeaf* l = static_cast<const KDLeaf*>(n);
            uint32_t id = (uint32_t)out.leafFirst.size();
            out.leafFirst.push_back((uint32_t)out.leafRefs.size());
            out.leafCount.push_back((uint32_t)l->shapes.size());
            for (const Shape* s : l->shapes) out.leafRefs.push_back(shapeId.at(s));
            if (l->shapes.size() > out.stats.maxLeaf) out.stats.maxLeaf = (uint32_t)l->shapes.size();
            return ~(int32_t)id;
        }
        const KDBranch* b = static_cast<const KDBranch*>(n);
        uint32_t id = (uint32_t)out.branchValue.size();
        for (int i = 0; i < 3; ++i) { out.branchMin.push_back(b->area.min[i]); out.branchMax.push_back(b->area.max[i]); }
        out.branchValue.push_back(b->value);
        out.branchAxis.push_back((int32_t)b->axis);
        out.branchLesser.push_back(0); out.branchGreater.push_back(0);
        int32_t lesser = walk(b->lesser, depth + 1);
        int32_t greater = walk(b->greater, depth + 1);
        out.branchLesser[id] = lesser; out.branchGreater[id] = greater;
        return (int32_t)id;
    }

    void run(const AABB* sceneBox, int threads) {
        const uint32_t nShapes = (uint32_t)fi.shapes.size();
        // The tree is built from the shapes' boxes alone, the arrays below from everything else:
        // the two go side by side (the builder's own threads are busy for a part of its run only).
        FlatKD kd;
        std::future<void> kdDone;
        if (sceneBox) kdDone = std::async(std::launch::async, [&]() { buildFlatKD(fi.shapes, *sceneBox, kd, threads); });
        out.triPos.reserve(9 * (size_t)nShapes); out.triNrm.reserve(9 * (size_t)nShapes); out.triUv.reserve(6 * (size_t)nShapes);
        out.primType.reserve(nShapes); out.primIndex.reserve(nShapes); out.primMaterial.reserve(nShapes);
        out.primMin.reserve(3 * (size_t)nShapes); out.primMax.reserve(3 * (size_t)nShapes);
        std::vector<const Biconvex*> lenses;
        for (uint32_t i = 0; i < nShapes; ++i) {
            const Shape* s = fi.shapes[i];
            if (!sceneBox) shapeId.emplace(s, i);
            if (const Triangle* t = dynamic_cast<const Triangle*>(s)) {
                uint32_t idx = (uint32_t)(out.triPos.size() / 9);
                for (uint32_t v = 0; v < 3; ++v) {
                    Vertex vert = (*t)[v];
                    for (int k = 0; k < 3; ++k) { out.triPos.push_back(vert.position[k]); out.triNrm.push_back(vert.normal[k]); }
                    out.triUv.push_back(vert.texCoords.x); out.triUv.push_back(vert.texCoords.y);
                }
                addPrim(RTX_PRIM_TRIANGLE, idx, material(t->mat), t->boundingBox);
            } else if (const Sphere* sp = dynamic_cast<const Sphere*>(s)) {
                addPrim(RTX_PRIM_SPHERE, addSphere(sp), material(sp->mat), sp->boundingBox);
            } else if (const Biconvex* bc = dynamic_cast<const Biconvex*>(s)) {
                addPrim(RTX_PRIM_BICONVEX, (uint32_t)lenses.size(), 0, bc->boundingBox);
                lenses.push_back(bc);
            } else {
                throw std::runtime_error("unsupported Shape subclass");
            }
        }
        // the private inner spheres of each lens get primitive ids after the shapes
        for (const Biconvex* bc : lenses) {
            for (const Sphere* sp : {bc->s1, bc->s2}) {
                out.lensSpheres.push_back((uint32_t)out.primType.size());
                addPrim(RTX_PRIM_SPHERE, addSphere(sp), material(sp->mat), sp->boundingBox);
            }
        }
        for (const Light* l : fi.lights) out.lights.push_back(l->flatten());
        if (sceneBox) {
            kdDone.get();
            out.branchMin.swap(kd.bmin); out.branchMax.swap(kd.bmax); out.branchValue.swap(kd.value);
            out.branchAxis.swap(kd.axis); out.branchLesser.swap(kd.lesser); out.branchGreater.swap(kd.greater);
            out.leafFirst.swap(kd.leafFirst); out.leafCount.swap(kd.leafCount); out.leafRefs.swap(kd.leafRefs);
            out.stats = kd.stats;
            root = kd.root;
            return;
        }
        if (!fi.kdTree) throw std::runtime_error("FrameInfo::kdTree is null");
        root = walk(fi.kdTree, 0);

        out.stats.branches = (uint32_t)out.branchValue.size();
        out.stats.leaves = (uint32_t)out.leafFirst.size();
        out.stats.leafRefs = (uint32_t)out.leafRefs.size();
        out.stats.depth = maxDepth;
    }
    int32_t root = 0;
};

}  // namespace

void FlatScene::build(const FrameInfo& fi) {
    *this = FlatScene();
    Flattener f(*this, fi);
    f.run(nullptr, 0);
    finish(fi, f.root);
}

void FlatScene::build(const FrameInfo& fi, const AABB& sceneBox, int threads) {
    *this = FlatScene();
    Flattener f(*this, fi);
    f.run(&sceneBox, threads);
    finish(fi, f.root);
}

void FlatScene::finish(const FrameInfo& fi, int32_t root) {
    std::memset(&d, 0, sizeof d);
    d.abi_version = RTX_ABI_VERSION;
    d.kd_root = root;                                   // pre-order: node 0 of its kind
    d.n_branches = (uint32_t)branchValue.size();
    d.n_leaves = (uint32_t)leafFirst.size();
    d.n_leaf_refs = (uint32_t)leafRefs.size();
    d.branch_min = branchMin.data(); d.branch_max = branchMax.data(); d.branch_value = branchValue.data();
    d.branch_axis = branchAxis.data(); d.branch_lesser = branchLesser.data(); d.branch_greater = branchGreater.data();
    d.leaf_first = leafFirst.data(); d.leaf_count = leafCount.data(); d.leaf_refs = leafRefs.data();
    d.n_shapes = (uint32_t)fi.shapes.size();
    d.n_prims = (uint32_t)primType.size();
    d.prim_type = primType.data(); d.prim_index = primIndex.data(); d.prim_material = primMaterial.data();
    d.prim_aabb_min = primMin.data(); d.prim_aabb_max = primMax.data();
    d.n_triangles = (uint32_t)(triPos.size() / 9);
    d.tri_position = triPos.data(); d.tri_normal = triNrm.data(); d.tri_uv = triUv.data();
    d.n_spheres = (uint32_t)sphRadius.size();
    d.sphere_center = sphCenter.data(); d.sphere_radius = sphRadius.data(); d.sphere_r2 = sphR2.data();
    d.sphere_normal_matrix = sphNormalMat.data();
    d.n_lenses = (uint32_t)(lensSpheres.size() / 2);
    d.lens_spheres = lensSpheres.data();
    d.n_materials = (uint32_t)materials.size(); d.materials = materials.data();
    d.n_textures = (uint32_t)textures.size(); d.textures = textures.data();
    d.n_texel_bytes = texels.size(); d.texels = texels.data();
    d.n_lights = (uint32_t)this->lights.size(); d.lights = this->lights.data();
}

}  // namespace rth
