// vecmath.hpp -- the small FP64 vector/matrix subset the host scene API needs.
//
// The reference builds on glm 0.9.9.8 (Raytracer/packages.config:5), which is a
// third-party dependency that is not part of the reference tree.  These types
// carry the same names and the same published evaluation order for the
// operations the scene classes use (dot = (x+y)+z, normalize = v*(1/sqrt(dot)),
// min(x,y) = (y<x)?y:x, column-major matrices, cofactor inverse), so that
// per-frame host results (sphere centres, normal matrices, AABBs, KD split
// planes) are reproducible by the CPU oracle and the device code.
#pragma once
#include <cmath>

namespace rth {

struct dvec2 {
    double x, y;
    dvec2() : x(0), y(0) {}
    explicit dvec2(double s) : x(s), y(s) {}
    dvec2(double a, double b) : x(a), y(b) {}
};

struct dvec3 {
    double x, y, z;
    dvec3() : x(0), y(0), z(0) {}
    explicit dvec3(double s) : x(s), y(s), z(s) {}
    dvec3(double a, double b, double c) : x(a), y(b), z(c) {}
    double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

struct dvec4 {
    double x, y, z, w;
    dvec4() : x(0), y(0), z(0), w(0) {}
    dvec4(double a, double b, double c, double d) : x(a), y(b), z(c), w(d) {}
    dvec4(const dvec3& v, double d) : x(v.x), y(v.y), z(v.z), w(d) {}
    double& operator[](int i) { return (&x)[i]; }
    double operator[](int i) const { return (&x)[i]; }
};

inline dvec3 operator+(dvec3 a, dvec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline dvec3 operator-(dvec3 a, dvec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline dvec3 operator-(dvec3 a) { return {-a.x, -a.y, -a.z}; }
inline dvec3 operator*(dvec3 a, dvec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline dvec3 operator*(dvec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline dvec3 operator*(double s, dvec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline dvec3 operator/(dvec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline dvec4 operator+(dvec4 a, dvec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline dvec4 operator-(dvec4 a, dvec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline dvec4 operator*(dvec4 a, dvec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline dvec4 operator*(dvec4 a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline dvec4 operator/(dvec4 a, double s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }

inline double dot(dvec3 a, dvec3 b) { double tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
inline dvec3 cross(dvec3 x, dvec3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline double length(dvec3 a) { return std::sqrt(dot(a, a)); }
inline double distance(dvec3 a, dvec3 b) { return length(b - a); }
inline dvec3 normalize(dvec3 a) { return a * (1.0 / std::sqrt(dot(a, a))); }
inline double gmin(double x, double y) { return (y < x) ? y : x; }
inline double gmax(double x, double y) { return (x < y) ? y : x; }
inline dvec3 min(dvec3 a, dvec3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
inline dvec3 max(dvec3 a, dvec3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }

constexpr double pi = 3.14159265358979323846264338327950288;
constexpr double epsilon = 2.220446049250313e-16;   // glm::epsilon<double>()

struct dmat4 {
    dvec4 c[4];     // columns
    dmat4() {}
    explicit dmat4(double d) {
        c[0] = {d, 0, 0, 0}; c[1] = {0, d, 0, 0}; c[2] = {0, 0, d, 0}; c[3] = {0, 0, 0, d};
    }
    dvec4& operator[](int i) { return c[i]; }
    const dvec4& operator[](int i) const { return c[i]; }
};

// mat4 * vec4: (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
inline dvec4 operator*(const dmat4& m, dvec4 v) {
    dvec4 mul0 = m[0] * v.x, mul1 = m[1] * v.y, add0 = mul0 + mul1;
    dvec4 mul2 = m[2] * v.z, mul3 = m[3] * v.w, add1 = mul2 + mul3;
    return add0 + add1;
}
inline dmat4 operator*(const dmat4& m, double s) {
    dmat4 r; for (int i = 0; i < 4; ++i) r[i] = m[i] * s; return r;
}

inline dmat4 transpose(const dmat4& m) {
    dmat4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r[i][j] = m[j][i];
    return r;
}

// translate(m, v): m[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]
inline dmat4 translate(const dmat4& m, dvec3 v) {
    dmat4 r = m;
    r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
    return r;
}

// rotate(m, angle, axis): axis-angle matrix, then m * R column by column
inline dmat4 rotate(const dmat4& m, double angle, dvec3 v) {
    const double a = angle, c = std::cos(a), s = std::sin(a);
    dvec3 axis = normalize(v);
    dvec3 temp = (1.0 - c) * axis;
    double R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = temp[0] * axis[1] + s * axis[2];
    R[0][2] = temp[0] * axis[2] - s * axis[1];
    R[1][0] = temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = temp[1] * axis[2] + s * axis[0];
    R[2][0] = temp[2] * axis[0] + s * axis[1];
    R[2][1] = temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    dmat4 r;
    r[0] = m[0] * R[0][0] + m[1] * R[0][1] + m[2] * R[0][2];
    r[1] = m[0] * R[1][0] + m[1] * R[1][1] + m[2] * R[1][2];
    r[2] = m[0] * R[2][0] + m[1] * R[2][1] + m[2] * R[2][2];
    r[3] = m[3];
    return r;
}

// cofactor inverse, in the term order of glm's compute_inverse<4,4>
inline dmat4 inverse(const dmat4& m) {
    double c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    double c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    double c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    double c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    double c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    double c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    double c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    double c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    double c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    double c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    double c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    double c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    double c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    double c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    double c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    double c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    double c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    double c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    dvec4 f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11);
    dvec4 f3(c12, c12, c14, c15), f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    dvec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]);
    dvec4 v1(m[1][1], m[0][1], m[0][1], m[0][1]);
    dvec4 v2(m[1][2], m[0][2], m[0][2], m[0][2]);
    dvec4 v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    dvec4 i0 = v1 * f0 - v2 * f1 + v3 * f2;
    dvec4 i1 = v0 * f0 - v2 * f3 + v3 * f4;
    dvec4 i2 = v0 * f1 - v1 * f3 + v3 * f5;
    dvec4 i3 = v0 * f2 - v1 * f4 + v2 * f5;
    dvec4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
    dmat4 inv;
    inv[0] = i0 * sa; inv[1] = i1 * sb; inv[2] = i2 * sa; inv[3] = i3 * sb;
    dvec4 row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    dvec4 d0 = m[0] * row0;
    double d1 = (d0.x + d0.y) + (d0.z + d0.w);
    return inv * (1.0 / d1);
}

}  // namespace rth
