// fbx.cpp -- geometry reader for binary FBX files and the Model class built on it.
//
// The reference imports its meshes through assimp (Model.cpp:63-200:
// aiProcess_Triangulate | GenSmoothNormals | FlipUVs | CalcTangentSpace | DropNormals) and
// keeps, per mesh vertex, the position, a normal and the first UV set.  assimp is a
// third-party library that is not part of the reference's sources, so this file reads
// the files itself and applies the same steps by their documented meaning:
//   * binary FBX node records (32-bit offsets below version 7500, 64-bit from there),
//     array properties raw or zlib-deflated;
//   * Objects/Geometry: Vertices, PolygonVertexIndex (a negative entry closes a polygon,
//     its vertex is ~entry), first LayerElementUV (ByPolygonVertex or ByVertice, Direct or
//     IndexToDirect), LayerElementMaterial (AllSame or ByPolygon);
//   * Objects/Model: Lcl Translation / Rotation (degrees, applied X then Y then Z) /
//     Scaling and PreRotation; Connections attach geometries and materials to models and
//     textures to materials;
//   * polygons are triangulated as fans (Triangulate), the file's normals are ignored
//     and every vertex gets the normalised sum of the unit normals of all faces that meet
//     at its position (DropNormals + GenSmoothNormals), v becomes 1 - v (FlipUVs);
//   * as Model::processNode does (Model.cpp:118-137), a mesh is placed with its own node's
//     local matrix times 0.01 with the last element reset to 1 -- parents are not applied --
//     and then with the Model's transform (Model.cpp:19-25).  assimp works in single
//     precision: control points and matrices are rounded to float on the way.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <unordered_map>

#include "scene.hpp"

namespace rth {
namespace {

struct Prop {
    char type = 0;
    int64_t i = 0;
    double d = 0.0;
    std::string s;
    std::vector<double> da;
    std::vector<int64_t> ia;
};

struct Node {
    std::string name;
    std::vector<Prop> props;
    std::vector<Node> children;
    const Node* child(const char* n) const {
        for (const Node& c : children) if (c.name == n) return &c;
        return nullptr;
    }
};

struct Reader {
    const uint8_t* b;
    size_t size;
    uint32_t version = 0;

    template <class T> T get(size_t& pos) const {
        if (pos + sizeof(T) > size) throw std::runtime_error("FBX: truncated file");
        T v;
        std::memcpy(&v, b + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }

    std::vector<uint8_t> arrayBytes(size_t& pos, size_t count, size_t elem) const {
        uint32_t encoding = get<uint32_t>(pos), clen = get<uint32_t>(pos);
        if (pos + clen > size) throw std::runtime_error("FBX: truncated array");
        // the element count comes from the file: it is checked against the bytes that are really
        // there before anything of that size is allocated (deflate expands at most 1032-fold)
        const uint64_t want = (uint64_t)count * elem;
        if (encoding == 0 ? want != clen : want > (uint64_t)clen * 1032u + 64u)
            throw std::runtime_error("FBX: array length mismatch");
        std::vector<uint8_t> out(want);
        if (encoding == 0) {
            std::memcpy(out.data(), b + pos, clen);
        } else {
            uLongf n = (uLongf)out.size();
            if (uncompress(out.data(), &n, b + pos, clen) != Z_OK || n != out.size())
                throw std::runtime_error("FBX: cannot inflate an array");
        }
        pos += clen;
        return out;
    }

    Prop prop(size_t& pos) const {
        Prop p;
        p.type = (char)get<uint8_t>(pos);
        switch (p.type) {
        case 'Y': p.i = get<int16_t>(pos); break;
        case 'C': p.i = get<int8_t>(pos); break;
        case 'I': p.i = get<int32_t>(pos); break;
        case 'L': p.i = get<int64_t>(pos); break;
        case 'F': p.d = get<float>(pos); break;
        case 'D': p.d = get<double>(pos); break;
        case 'S': case 'R': {
            uint32_t n = get<uint32_t>(pos);
            if (pos + n > size) throw std::runtime_error("FBX: truncated string");
            p.s.assign((const char*)b + pos, n);
            pos += n;
            break;
        }
        case 'f': case 'd': case 'l': case 'i': case 'b': {
            uint32_t count = get<uint32_t>(pos);
            size_t elem = p.type == 'd' || p.type == 'l' ? 8 : p.type == 'b' ? 1 : 4;
            std::vector<uint8_t> raw = arrayBytes(pos, count, elem);
            if (p.type == 'd') { p.da.resize(count); std::memcpy(p.da.data(), raw.data(), raw.size()); }
            else if (p.type == 'f') { p.da.resize(count); for (uint32_t k = 0; k < count; ++k) { float f; std::memcpy(&f, raw.data() + 4 * k, 4); p.da[k] = f; } }
            else if (p.type == 'l') { p.ia.resize(count); std::memcpy(p.ia.data(), raw.data(), raw.size()); }
            else if (p.type == 'i') { p.ia.resize(count); for (uint32_t k = 0; k < count; ++k) { int32_t v; std::memcpy(&v, raw.data() + 4 * k, 4); p.ia[k] = v; } }
            else { p.ia.resize(count); for (uint32_t k = 0; k < count; ++k) p.ia[k] = raw[k]; }
            break;
        }
        default: throw std::runtime_error("FBX: unknown property type");
        }
        return p;
    }

    // false at a null record (end of a child list)
    bool node(size_t& pos, Node& out, int depth) const {
        if (depth > 64) throw std::runtime_error("FBX: nesting too deep");
        uint64_t end, nprops, plen;
        if (version >= 7500) { end = get<uint64_t>(pos); nprops = get<uint64_t>(pos); plen = get<uint64_t>(pos); }
        else { end = get<uint32_t>(pos); nprops = get<uint32_t>(pos); plen = get<uint32_t>(pos); }
        uint8_t nl = get<uint8_t>(pos);
        if (pos + nl > size) throw std::runtime_error("FBX: truncated name");
        out.name.assign((const char*)b + pos, nl);
        pos += nl;
        if (end == 0) return false;
        if (end > size || pos + plen > end) throw std::runtime_error("FBX: bad node extent");
        // only the subtrees that are used are kept: others are skipped wholesale
        static const char* kept[] = {"Objects", "Connections", "Geometry", "Model", "Material", "Texture", "Vertices",
                                     "PolygonVertexIndex", "LayerElementUV", "LayerElementMaterial", "UV", "UVIndex",
                                     "Materials", "MappingInformationType", "ReferenceInformationType", "Properties70",
                                     "P", "C", "RelativeFilename", "FileName"};
        bool keep = depth == 0 ? (out.name == "Objects" || out.name == "Connections") : false;
        for (const char* k : kept) if (depth > 0 && out.name == k) keep = true;
        if (!keep) { pos = end; return true; }
        if (nprops > plen) throw std::runtime_error("FBX: bad property count");     // a property takes at least a byte
        out.props.reserve(nprops);
        for (uint64_t k = 0; k < nprops; ++k) out.props.push_back(prop(pos));
        while (pos < end) {
            Node c;
            if (!node(pos, c, depth + 1)) break;
            out.children.push_back(std::move(c));
        }
        pos = end;
        return true;
    }
};

dmat4 identity() { return dmat4(1.0); }

dmat4 mul(const dmat4& a, const dmat4& b) {
    dmat4 r(0.0);
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a[k][rr] * b[c][k];
            r[c][rr] = s;
        }
    return r;
}

dmat4 toFloat(dmat4 m) {
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) m[c][r] = (double)(float)m[c][r];
    return m;
}

dmat4 eulerXYZ(dvec3 deg) {
    const double k = 0.01745329251994329576923690768489;
    dmat4 m = identity();
    m = rotate(m, deg.z * k, dvec3(0.0, 0.0, 1.0));
    m = rotate(m, deg.y * k, dvec3(0.0, 1.0, 0.0));
    m = rotate(m, deg.x * k, dvec3(1.0, 0.0, 0.0));
    return m;                       // Rz * Ry * Rx: X is applied first
}

dmat4 scaling(dvec3 s) {
    dmat4 m = identity();
    m[0][0] = s.x; m[1][1] = s.y; m[2][2] = s.z;
    return m;
}

struct ModelNode {
    dvec3 t{0.0, 0.0, 0.0}, r{0.0, 0.0, 0.0}, s{1.0, 1.0, 1.0}, pre{0.0, 0.0, 0.0};
    dmat4 local() const { return mul(mul(mul(translate(identity(), t), eulerXYZ(pre)), eulerXYZ(r)), scaling(s)); }
};

std::string baseName(const std::string& path) {
    size_t at = path.find_last_of("/\\");
    std::string f = at == std::string::npos ? path : path.substr(at + 1);
    size_t z = f.find('\0');
    return z == std::string::npos ? f : f.substr(0, z);
}

struct PosKey {
    double x, y, z;
    bool operator<(const PosKey& o) const { return x != o.x ? x < o.x : y != o.y ? y < o.y : z < o.z; }
};

}  // namespace

void loadFbx(const std::string& path, const dmat4& modelTransform, FbxMesh& out) {
    out = FbxMesh();
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("FBX: cannot open " + path);
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + got);
    std::fclose(f);
    if (bytes.size() < 27 || std::memcmp(bytes.data(), "Kaydara FBX Binary  ", 20) != 0)
        throw std::runtime_error("FBX: " + path + " is not a binary FBX file");
    Reader rd{bytes.data(), bytes.size()};
    size_t pos = 23;
    rd.version = rd.get<uint32_t>(pos);

    Node objects, connections;
    while (pos < bytes.size()) {
        Node n;
        if (!rd.node(pos, n, 0)) break;
        if (n.name == "Objects") objects = std::move(n);
        else if (n.name == "Connections") connections = std::move(n);
    }

    std::unordered_map<int64_t, const Node*> geometry, material, texture;
    std::unordered_map<int64_t, ModelNode> model;
    std::vector<int64_t> geometryOrder;
    for (const Node& o : objects.children) {
        if (o.props.empty()) continue;
        int64_t id = o.props[0].i;
        if (o.name == "Geometry") { geometry[id] = &o; geometryOrder.push_back(id); }
        else if (o.name == "Material") material[id] = &o;
        else if (o.name == "Texture") texture[id] = &o;
        else if (o.name == "Model") {
            ModelNode m;
            if (const Node* p70 = o.child("Properties70"))
                for (const Node& p : p70->children) {
                    if (p.name != "P" || p.props.size() < 7) continue;
                    dvec3 v(p.props[4].d, p.props[5].d, p.props[6].d);
                    const std::string& key = p.props[0].s;
                    if (key == "Lcl Translation") m.t = v;
                    else if (key == "Lcl Rotation") m.r = v;
                    else if (key == "Lcl Scaling") m.s = v;
                    else if (key == "PreRotation") m.pre = v;
                }
            model[id] = m;
        }
    }
    // child -> parents; material order per model as connected; diffuse texture per material
    std::unordered_map<int64_t, int64_t> geometryModel;
    std::unordered_map<int64_t, std::vector<int64_t>> modelMaterials;
    std::unordered_map<int64_t, int64_t> materialTexture;
    for (const Node& c : connections.children) {
        if (c.name != "C" || c.props.size() < 3) continue;
        int64_t child = c.props[1].i, parent = c.props[2].i;
        if (geometry.count(child) && model.count(parent)) geometryModel[child] = parent;
        else if (material.count(child) && model.count(parent)) modelMaterials[parent].push_back(child);
        else if (texture.count(child) && material.count(parent)) {
            bool diffuse = c.props.size() < 4 || c.props[3].s == "DiffuseColor";
            if (diffuse && !materialTexture.count(parent)) materialTexture[parent] = child;
        }
    }
    std::unordered_map<int64_t, int32_t> materialSlot;
    auto slotOf = [&](int64_t matId) {
        auto it = materialSlot.find(matId);
        if (it != materialSlot.end()) return it->second;
        int32_t slot = (int32_t)out.materialTextures.size();
        std::string file;
        auto t = materialTexture.find(matId);
        if (t != materialTexture.end()) {
            const Node* tn = texture[t->second];
            const Node* fn = tn->child("RelativeFilename");
            if (!fn || fn->props.empty() || fn->props[0].s.empty()) fn = tn->child("FileName");
            if (fn && !fn->props.empty()) file = baseName(fn->props[0].s);
        }
        out.materialTextures.push_back(file);
        materialSlot[matId] = slot;
        return slot;
    };

    const dmat4 modelTx = toFloat(modelTransform);
    for (int64_t gid : geometryOrder) {
        const Node& g = *geometry[gid];
        const Node* vn = g.child("Vertices");
        const Node* pn = g.child("PolygonVertexIndex");
        if (!vn || !pn || vn->props.empty() || pn->props.empty()) continue;
        const std::vector<double>& verts = vn->props[0].da;
        const std::vector<int64_t>& poly = pn->props[0].ia;
        const size_t nControl = verts.size() / 3;

        dmat4 nodeTx = identity();
        std::vector<int64_t> mats;
        auto gm = geometryModel.find(gid);
        if (gm != geometryModel.end()) {
            nodeTx = toFloat(model[gm->second].local()) * 0.01;          // Model.cpp:121
            nodeTx[3][3] = 1.0;                                           // Model.cpp:122
            nodeTx = toFloat(nodeTx);
            mats = modelMaterials[gm->second];
        }
        const dmat4 tx = toFloat(mul(modelTx, nodeTx));
        const dmat4 normalTx = inverse(transpose(tx));

        // UV layer
        const Node* uvLayer = g.child("LayerElementUV");
        const std::vector<double>* uv = nullptr;
        const std::vector<int64_t>* uvIndex = nullptr;
        bool uvByVertex = false;
        if (uvLayer) {
            if (const Node* u = uvLayer->child("UV")) if (!u->props.empty()) uv = &u->props[0].da;
            if (const Node* u = uvLayer->child("UVIndex")) if (!u->props.empty()) uvIndex = &u->props[0].ia;
            if (const Node* m = uvLayer->child("MappingInformationType"))
                if (!m->props.empty() && (m->props[0].s == "ByVertice" || m->props[0].s == "ByVertex")) uvByVertex = true;
            if (const Node* r = uvLayer->child("ReferenceInformationType"))
                if (!r->props.empty() && r->props[0].s == "Direct") uvIndex = nullptr;
        }
        // material layer
        const std::vector<int64_t>* polyMaterial = nullptr;
        bool materialByPolygon = false;
        if (const Node* ml = g.child("LayerElementMaterial")) {
            if (const Node* m = ml->child("Materials")) if (!m->props.empty()) polyMaterial = &m->props[0].ia;
            if (const Node* m = ml->child("MappingInformationType"))
                if (!m->props.empty() && m->props[0].s == "ByPolygon") materialByPolygon = true;
        }

        const size_t firstTriangle = out.material.size();
        std::vector<uint32_t> corner;           // polygon-vertex positions of the open polygon
        std::vector<uint32_t> triControl;       // control point of every emitted triangle corner
        size_t polygon = 0;
        for (size_t k = 0; k < poly.size(); ++k) {
            int64_t e = poly[k];
            bool last = e < 0;
            int64_t cp = last ? ~e : e;
            if (cp < 0 || (size_t)cp >= nControl) throw std::runtime_error("FBX: polygon vertex out of range");
            corner.push_back((uint32_t)k);
            if (!last) continue;
            int32_t slot = -1;
            if (!mats.empty()) {
                int64_t mi = 0;
                if (polyMaterial && !polyMaterial->empty())
                    mi = materialByPolygon && polygon < polyMaterial->size() ? (*polyMaterial)[polygon] : (*polyMaterial)[0];
                if (mi >= 0 && (size_t)mi < mats.size()) slot = slotOf(mats[(size_t)mi]);
            }
            for (size_t t = 1; t + 1 < corner.size(); ++t) {
                const uint32_t tri[3] = {corner[0], corner[t], corner[t + 1]};
                for (uint32_t pv : tri) {
                    int64_t raw = poly[pv];
                    uint32_t c = (uint32_t)(raw < 0 ? ~raw : raw);
                    triControl.push_back(c);
                    dvec3 p((double)(float)verts[3 * c], (double)(float)verts[3 * c + 1], (double)(float)verts[3 * c + 2]);
                    dvec3 w = transformPos(p, tx);
                    out.position.push_back(w.x); out.position.push_back(w.y); out.position.push_back(w.z);
                    double u = 0.0, v = 0.0;
                    if (uv) {
                        size_t at = uvByVertex ? c : pv;
                        if (uvIndex) at = at < uvIndex->size() && (*uvIndex)[at] >= 0 ? (size_t)(*uvIndex)[at] : (size_t)-1;
                        if (at != (size_t)-1 && 2 * at + 1 < uv->size()) {
                            u = (double)(float)(*uv)[2 * at];
                            v = (double)(1.0f - (float)(*uv)[2 * at + 1]);        // aiProcess_FlipUVs
                        }
                    }
                    out.uv.push_back(u); out.uv.push_back(v);
                }
                out.material.push_back(slot);
            }
            corner.clear();
            ++polygon;
        }
        // smooth normals: per position, the sum of the unit normals of the faces around it
        const size_t nTri = out.material.size() - firstTriangle;
        std::map<PosKey, dvec3> sum;
        std::vector<dvec3> faceNormal(nTri);
        auto local = [&](uint32_t c) { return dvec3((double)(float)verts[3 * c], (double)(float)verts[3 * c + 1], (double)(float)verts[3 * c + 2]); };
        for (size_t t = 0; t < nTri; ++t) {
            dvec3 a = local(triControl[3 * t]), b2 = local(triControl[3 * t + 1]), c = local(triControl[3 * t + 2]);
            dvec3 n = cross(b2 - a, c - a);
            double len = length(n);
            faceNormal[t] = len > 0.0 ? n / len : dvec3(0.0, 0.0, 0.0);
            for (dvec3 p : {a, b2, c}) {
                auto it = sum.find(PosKey{p.x, p.y, p.z});
                if (it == sum.end()) sum.emplace(PosKey{p.x, p.y, p.z}, faceNormal[t]);
                else it->second = it->second + faceNormal[t];
            }
        }
        for (size_t t = 0; t < nTri; ++t)
            for (int k = 0; k < 3; ++k) {
                dvec3 p = local(triControl[3 * t + k]);
                dvec3 n = sum[PosKey{p.x, p.y, p.z}];
                double len = length(n);
                n = len > 0.0 ? n / len : faceNormal[t];
                dvec3 w = transformNormal(dvec3((double)(float)n.x, (double)(float)n.y, (double)(float)n.z), tx);
                double wl = length(w);
                if (wl > 0.0) w = normalize(w);                         // Model.cpp:159
                (void)normalTx;
                out.normal.push_back(w.x); out.normal.push_back(w.y); out.normal.push_back(w.z);
            }
    }
}

// ---- Model (Model.cpp:19-30) ---------------------------------------------------------
Model::Model(const std::string& modelsRoot, const std::string& modelname, dvec3 pos, dvec3 rotation,
             const std::function<Texture*(const std::string&)>& textureLoader) {
    dmat4 t(1.0);
    t = rotate(t, (double)(float)rotation.x, dvec3(1.0, 0.0, 0.0));
    t = rotate(t, (double)(float)rotation.y, dvec3(0.0, 1.0, 0.0));
    t = rotate(t, (double)(float)rotation.z, dvec3(0.0, 0.0, 1.0));
    t = translate(t, dvec3((double)(float)pos.x, (double)(float)pos.y, (double)(float)pos.z));
    transform = t;
    directory = modelsRoot + "/" + modelname + "/";
    loadFbx(directory + modelname + ".fbx", transform, mesh);
    // Model.cpp:73-98: a material with a diffuse texture is PlainWhiteTees sampled from it,
    // one without is "Bug"; a mesh with no material at all gets PlainWhiteTees (:186-188)
    for (const std::string& file : mesh.materialTextures) {
        Texture* tex = (!file.empty() && textureLoader) ? textureLoader(directory + file) : nullptr;
        if (tex) modelMaterials.emplace_back(new Material("PlainWhiteTees", dvec3(0.0), tex->getVec3Sampler()));
        else modelMaterials.emplace_back(new Material(file.empty() ? "Bug" : "PlainWhiteTees"));
    }
    fallback.reset(new Material("PlainWhiteTees"));
    const size_t n = mesh.material.size();
    children.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        Vertex v[3];
        for (int k = 0; k < 3; ++k) {
            const double* p = &mesh.position[9 * i + 3 * k];
            const double* nn = &mesh.normal[9 * i + 3 * k];
            const double* t2 = &mesh.uv[6 * i + 2 * k];
            v[k] = Vertex(dvec3(p[0], p[1], p[2]), dvec3(nn[0], nn[1], nn[2]), dvec2(t2[0], t2[1]));
        }
        int32_t slot = mesh.material[i];
        Material* m = slot >= 0 ? modelMaterials[(size_t)slot].get() : fallback.get();
        owned.emplace_back(new Triangle(v[0], v[1], v[2], m));
        children.push_back(owned.back().get());
    }
}

}  // namespace rth
