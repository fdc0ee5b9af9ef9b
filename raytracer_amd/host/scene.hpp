// scene.hpp -- host scene API: the reference's public scene classes, kept as
// the way a frame is described, minus everything that moved behind the C ABI
// (rayHit / rayAABB / traverseKDTree live on the GPU, see include/rtx.h).
//
// Mirrors, with the same constructor signatures and field names:
//   Ray (Raytracer/Ray.hpp), AABB / Shape / HitResult (Shape.hpp:13-65),
//   Vertex / Triangle (Triangle.hpp:18-71), Sphere (Sphere.hpp:17-29),
//   Biconvex (Biconvex.hpp:16-35), Animatable + motion functions
//   (Animatable.hpp, Animatable.cpp:5-35), Material (Material.hpp:25-85),
//   Texture (Texture.hpp:17-33), procedural samplers (Procedural.hpp),
//   Light / PointLight / DirectionalLight / SquareLight (Light.hpp:13-26 ...),
//   KDNode / KDBranch / KDLeaf / buildKDTree (KDTree.hpp:14-58),
//   FrameInfo (Raytracer.hpp:60-67 + lights, Raytracer-Noah-Desktop.cpp:104-111),
//   redoAABBs (Raytracer.cpp:368-375).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "vecmath.hpp"

namespace rth {

// ---- Texture / samplers -----------------------------------------------------
class Texture;

// Stands in for the reference's std::function<...(dvec2)> material callbacks
// (Material.hpp:70-76): a callable cannot cross a C ABI, so a sampler is the tag
// of the factory that made it plus, for textures, the image it reads.
struct Sampler {
    uint32_t kind = RTX_SAMPLER_CONST;
    const Texture* texture = nullptr;
    Sampler() {}
    Sampler(std::nullptr_t) {}
    Sampler(uint32_t k, const Texture* t = nullptr) : kind(k), texture(t) {}
    bool operator!=(std::nullptr_t) const { return kind != RTX_SAMPLER_CONST; }
};

extern const Sampler ryCheckers10x10;       // Procedural.cpp:4-14
extern const Sampler doubleCheckers10x10;   // Procedural.cpp:16-25

class Texture {
public:
    // takes a copy of width*height*3 bytes of RGB8 (the reference forces 3 channels,
    // Texture.cpp:17)
    Texture(const unsigned char* data, int width, int height, int nrComponents = 3);
    Sampler getDoubleSampler() const { return Sampler(RTX_SAMPLER_TEXTURE_DOUBLE, this); }
    Sampler getVec3Sampler() const { return Sampler(RTX_SAMPLER_TEXTURE_VEC3, this); }
    std::vector<unsigned char> data;
    int width, height, nrComponents;
};

// ---- Material ----------------------------------------------------------------
struct materialStats {
    dvec3 color = dvec3(1.0, 0.0, 1.0);
    double ns = 100.0, ni = 1.0, transparency = 0.0, metalness = 0.7, smoothness = 0.5;
};

class Material {
public:
    Material();
    Material(dvec3 color, double ns = 100.0, double ni = 1.0, double transparency = 0.0,
             double metalness = 0.5, double smoothness = 0.5, dvec3 emission = dvec3(0.0),
             Sampler colorFn = nullptr, Sampler normalFn = nullptr, Sampler nsFn = nullptr,
             Sampler niFn = nullptr, Sampler transparencyFn = nullptr,
             Sampler metalnessFn = nullptr, Sampler smoothnessFn = nullptr);
    // throws std::out_of_range for an unknown preset, like materials.at() (Material.cpp:92)
    Material(const std::string& premade, dvec3 emission = dvec3(0.0), Sampler colorFn = nullptr,
             Sampler normalFn = nullptr, Sampler nsFn = nullptr, Sampler niFn = nullptr,
             Sampler transparencyFn = nullptr, Sampler metalnessFn = nullptr,
             Sampler smoothnessFn = nullptr);
    Material* setColor(dvec3 newColor) { color = newColor; return this; }
    dvec3 getEmission() const { return emission; }

    Sampler colorFn, normalFn, nsFn, niFn, transparencyFn, metalnessFn, smoothnessFn;
    dvec3 color;
    double ns, ni, transparency, metalness, smoothness;
    dvec3 emission;
};

// ---- Ray / AABB / Shape --------------------------------------------------------
class Ray {
public:
    Ray(const dvec3& origin, const dvec3& direction) {   // Ray.cpp:3-8
        this->direction = normalize(direction);
        this->inverseDirection = -this->direction;
        this->origin = origin;
    }
    dvec3 origin, direction, inverseDirection;
};

struct AABB {
    dvec3 min, max, mid;
    AABB() : min(0.0), max(0.0) {}                        // mid left unset in the reference
    AABB(dvec3 mn, dvec3 mx) : min(mn), max(mx), mid((mn + mx) / 2.0) {}
    void encompass(const AABB& o) { max = rth::max(max, o.max); min = rth::min(min, o.min); }
};

class Shape {
public:
    Shape(AABB boundingBox, Material* material = nullptr) : mat(material), boundingBox(boundingBox) {}
    virtual ~Shape() {}
    virtual void redoAABB(const double currentTime) = 0;
    Material* mat;
    AABB boundingBox;
};

// ---- Animatable ----------------------------------------------------------------
typedef dmat4 (*MovementFunction)(double currentTime);
dmat4 noMovement(double currentTime);
dmat4 oscilateX(double currentTime);
dmat4 oscilateY(double currentTime);
dmat4 rotateY(double currentTime);
dmat4 circle0(double currentTime);
dmat4 circle1(double currentTime);
dmat4 circle2(double currentTime);

class Animatable {
public:
    explicit Animatable(MovementFunction mf) : movementFunction(mf) {}
    dmat4 getModel(double currentTime) const { return movementFunction(currentTime); }
private:
    MovementFunction movementFunction;
};

dvec3 transformNormal(const dvec3& x, const dmat4& model);  // CoordinateHelpers.cpp:12-16
dvec3 transformPos(const dvec3& x, const dmat4& model);     // CoordinateHelpers.cpp:17-20

// ---- Triangle ------------------------------------------------------------------
struct Vertex {
    dvec3 position, normal, tangent, bitangent;
    dvec2 texCoords;
    Vertex() : position(1.0), normal(1.0), tangent(1.0), bitangent(1.0), texCoords(1.0) {}
    Vertex(dvec3 a) : position(a) {}
    Vertex(dvec3 a, dvec2 uv) : position(a), texCoords(uv) {}
    Vertex(dvec3 a, dvec3 n) : position(a), normal(n) {}
    Vertex(dvec3 a, dvec3 n, dvec2 uv) : position(a), normal(n), texCoords(uv) {}
};

class Triangle : public Shape {
    Vertex vertices[3];
public:
    Triangle(Vertex a, Vertex b, Vertex c, Material* material)
        : Shape(AABB(rth::min(a.position, rth::min(b.position, c.position)),
                     rth::max(a.position, rth::max(b.position, c.position))), material) {
        vertices[0] = a; vertices[1] = b; vertices[2] = c;
    }
    Vertex operator[](uint32_t index) const { return vertices[index]; }
    void redoAABB(double currentTime) override;            // Triangle.cpp:73-82
};

// ---- Sphere --------------------------------------------------------------------
class Sphere : public Shape, public Animatable {
public:
    Sphere(const dvec3& origin, double radius, Material* material, MovementFunction movementFunction);
    void redoAABB(double currentTime) override;            // Sphere.cpp:32-35
    double radius;
    dvec3 origin;
    double r2;
};

// ---- Biconvex ------------------------------------------------------------------
double solveLensForD(double r, double f, double n);        // Biconvex.cpp:3-10

class Biconvex : public Shape, public Animatable {
public:
    Biconvex(const dvec3 origin, dvec3 forward, double lensRadius, double focalLength,
             MovementFunction movementFunction);
    ~Biconvex() override;
    // the reference leaves this pure virtual (Biconvex.hpp:16-35), which makes the class
    // impossible to instantiate; the lens does not move, so the box stays as built
    void redoAABB(double) override {}
    Sphere* s1;
    Sphere* s2;
    double d;
    dvec3 origin, forward;
private:
    Material* glass;
    AABB setupSpheres(const dvec3 origin, dvec3 forward, double lensRadius, double focalLength);
};

// ---- Lights --------------------------------------------------------------------
class Light {
public:
    Light(const dvec3& color, bool multisample = false) : color(color), multisample(multisample) {}
    virtual ~Light() {}
    virtual rtx_light flatten() const = 0;
    dvec3 color;
    bool multisample;
};

class PointLight : public Light {
public:
    PointLight(const dvec3& position, const dvec3& color) : Light(color), position(position) {}
    rtx_light flatten() const override;
    dvec3 position;
};

class DirectionalLight : public Light {
public:
    DirectionalLight(dvec3 direction, dvec3 color);        // DirectionalLight.cpp:4-8
    rtx_light flatten() const override;
    dvec3 direction, reverseDirection;
};

class SquareLight : public Light {
public:
    SquareLight(dvec3 position, dvec3 normal, double width, double height, dvec3 color = dvec3(1.0));
    rtx_light flatten() const override;
    dvec3 position, normal, bottomLeft, topRight, right, up;
    double width, height;
};

// ---- KD tree -------------------------------------------------------------------
enum Axis { X = 0, Y, Z };

class KDNode {
public:
    explicit KDNode(bool isLeaf) : isLeaf(isLeaf) {}
    virtual ~KDNode() {}
    bool isLeaf;
};
class KDBranch : public KDNode {
public:
    KDBranch(KDNode* greater, KDNode* lesser, AABB area, Axis axis, double value)
        : KDNode(false), greater(greater), lesser(lesser), area(area), axis(axis), value(value) {}
    ~KDBranch() override { delete greater; delete lesser; }
    KDNode* greater;
    KDNode* lesser;
    AABB area;
    Axis axis;
    double value;
};
class KDLeaf : public KDNode {
public:
    explicit KDLeaf(std::vector<Shape*> shapes) : KDNode(true), shapes(std::move(shapes)) {}
    std::vector<Shape*> shapes;
};

// buildKDTree of KDTree.hpp:57 (construction rules of KDTree.cpp:68-126).  The reference sorts
// with std::sort, whose order among equal keys is unspecified; here equal keys keep their
// incoming order, so the tree is a function of the input order alone.  The tree is built by
// buildFlatKD below and handed back as node objects.
KDNode* buildKDTree(std::vector<Shape*> shapes, AABB box, Axis axis = X,
                    int recursionWithoutChange = 0, int layer = 0);

// ---- FrameInfo -----------------------------------------------------------------
struct FrameInfo {
    std::vector<Shape*> shapes;
    std::vector<Light*> lights;
    KDNode* kdTree = nullptr;
    dvec3 camPosition;
    double currentTime = 0.0;
};

AABB redoAABBs(FrameInfo& fi);                              // Raytracer.cpp:368-375

// camera basis of Raytracer.cpp:642-660 (fov in degrees; focal = 1/tan(fov/2))
rtx_camera makeCamera(dvec3 eye, dvec3 lookat, double fovDegrees = 90.0);

// ---- flattener -----------------------------------------------------------------
struct KDStats { uint32_t branches = 0, leaves = 0, leafRefs = 0, depth = 0, maxLeaf = 0; };

// The KD tree as the flattened arrays of rtx_scene_desc: branches and leaves numbered in
// pre-order (lesser subtree first), child >= 0 branch, < 0 leaf ~child.
struct FlatKD {
    std::vector<double> bmin, bmax, value;          // bmin/bmax: 3 per branch
    std::vector<int32_t> axis, lesser, greater;
    std::vector<uint32_t> leafFirst, leafCount, leafRefs;
    int32_t root = ~(int32_t)0;
    KDStats stats;
};

// buildKDTree + flattening in one step: the same tree, built from compact records and on
// several threads (threads <= 0: one per core, at most 16).  See kdbuild.cpp.
// axis / recursionWithoutChange / layer: the state of the reference's recursion to start from.
void buildFlatKD(const std::vector<Shape*>& shapes, const AABB& box, FlatKD& out, int threads = 0, Axis axis = X,
                 int recursionWithoutChange = 0, int layer = 0);

// Owns the SoA arrays an rtx_scene_desc points into.
class FlatScene {
public:
    // Walks fi.kdTree (which must have been built from fi.shapes at fi.currentTime)
    // and fills the arrays; throws std::runtime_error on an unsupported Shape subclass.
    void build(const FrameInfo& fi);
    // The same with the tree built here by buildFlatKD from fi.shapes inside sceneBox
    // (fi.kdTree is not used).
    void build(const FrameInfo& fi, const AABB& sceneBox, int threads = 0);
    const rtx_scene_desc& desc() const { return d; }
    KDStats stats;

    std::vector<double> branchMin, branchMax, branchValue;
    std::vector<int32_t> branchAxis, branchLesser, branchGreater;
    std::vector<uint32_t> leafFirst, leafCount, leafRefs;
    std::vector<uint8_t> primType;
    std::vector<uint32_t> primIndex, primMaterial;
    std::vector<double> primMin, primMax;
    std::vector<double> triPos, triNrm, triUv;
    std::vector<double> sphCenter, sphRadius, sphR2, sphNormalMat;
    std::vector<uint32_t> lensSpheres;
    std::vector<rtx_material> materials;
    std::vector<rtx_texture> textures;
    std::vector<uint8_t> texels;
    std::vector<rtx_light> lights;
private:
    void finish(const FrameInfo& fi, int32_t root);
    rtx_scene_desc d;
};

// ---- Model (Model.hpp / Model.cpp) ------------------------------------------------
// Triangle soup of a binary FBX file as the reference's importer settings produce it
// (see fbx.cpp): 9 doubles of position and of normal and 6 of uv per triangle, and the
// slot of the triangle's material in materialTextures (-1: the mesh has no material).
struct FbxMesh {
    std::vector<double> position, normal, uv;
    std::vector<int32_t> material;
    std::vector<std::string> materialTextures;      // per slot: file name of the diffuse texture, or empty
};
void loadFbx(const std::string& path, const dmat4& modelTransform, FbxMesh& out);

// Model(modelname, pos, rotation) of Model.cpp:19-30; the file is
// <modelsRoot>/<modelname>/<modelname>.fbx.  textureLoader maps a texture path to a decoded
// Texture (or nullptr when it cannot be had; the material is then constant white).
class Model {
public:
    Model(const std::string& modelsRoot, const std::string& modelname, dvec3 pos, dvec3 rotation,
          const std::function<Texture*(const std::string&)>& textureLoader = nullptr);
    std::vector<Triangle*> children;                        // Model.hpp: the mesh's triangles
    std::vector<std::unique_ptr<Material>> modelMaterials;
    std::vector<std::unique_ptr<Triangle>> owned;           // storage behind `children`
    dmat4 transform;
    std::string directory;
    FbxMesh mesh;
private:
    std::unique_ptr<Material> fallback;
};

// ---- output ---------------------------------------------------------------------
// PNG (RGB8, rows top first) through zlib; false when the file cannot be written.
bool writePng(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

}  // namespace rth
