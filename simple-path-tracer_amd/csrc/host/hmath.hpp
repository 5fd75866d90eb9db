// Host-side f32 vector/affine math in glam's operation order (glam 0.20, the
// reference's math crate): dot = (x*x' + y*y') + z*z'; Mat3 * v = (c0*x + c1*y) + c2*z;
// Affine point = M*p + t; normalize = v / sqrt(dot).  Used only for scene set-up
// (instance transforms, camera basis, tangents); the per-sample math lives in the
// HIP kernels.
#pragma once
#include <algorithm>
#include <cmath>

namespace spt_host {

struct V3 {
    float x = 0, y = 0, z = 0;
    V3() = default;
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return a * s; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a / length(a); }
inline V3 vmin(V3 a, V3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

struct M3 {  // three columns
    V3 c0{1, 0, 0}, c1{0, 1, 0}, c2{0, 0, 1};
};
inline V3 operator*(const M3& m, V3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
inline M3 operator*(const M3& a, const M3& b) { return {a * b.c0, a * b.c1, a * b.c2}; }
inline M3 transpose(const M3& m) {
    return {{m.c0.x, m.c1.x, m.c2.x}, {m.c0.y, m.c1.y, m.c2.y}, {m.c0.z, m.c1.z, m.c2.z}};
}
inline float determinant(const M3& m) { return dot(m.c2, cross(m.c0, m.c1)); }
// glam Mat3A::inverse: cross products of the columns scaled by 1/det, transposed
inline M3 inverse(const M3& m) {
    V3 t0 = cross(m.c1, m.c2), t1 = cross(m.c2, m.c0), t2 = cross(m.c0, m.c1);
    float det = dot(m.c2, t2);
    float inv = 1.0f / det;
    return transpose(M3{t0 * inv, t1 * inv, t2 * inv});
}

struct Affine {
    M3 m;
    V3 t;
    V3 point(V3 p) const { return m * p + t; }
    V3 vector(V3 v) const { return m * v; }
};
inline Affine operator*(const Affine& a, const Affine& b) { return {a.m * b.m, a.m * b.t + a.t}; }
inline Affine inverse(const Affine& a) {
    M3 mi = inverse(a.m);
    return {mi, -(mi * a.t)};
}
inline Affine from_scale(V3 s) { return {M3{{s.x, 0, 0}, {0, s.y, 0}, {0, 0, s.z}}, {}}; }
inline Affine from_translation(V3 t) { return {M3{}, t}; }
inline Affine from_rotation_x(float a) {
    float s = std::sin(a), c = std::cos(a);
    return {M3{{1, 0, 0}, {0, c, s}, {0, -s, c}}, {}};
}
inline Affine from_rotation_y(float a) {
    float s = std::sin(a), c = std::cos(a);
    return {M3{{c, 0, -s}, {0, 1, 0}, {s, 0, c}}, {}};
}
inline Affine from_rotation_z(float a) {
    float s = std::sin(a), c = std::cos(a);
    return {M3{{c, s, 0}, {-s, c, 0}, {0, 0, 1}}, {}};
}

struct Box {
    V3 lo{3.40282347e38f, 3.40282347e38f, 3.40282347e38f};
    V3 hi{-3.40282347e38f, -3.40282347e38f, -3.40282347e38f};
    void grow(V3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    void grow(const Box& b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
    V3 centroid() const { return (lo + hi) * 0.5f; }
    float area() const {
        V3 d = hi - lo;
        if (d.x < 0 || d.y < 0 || d.z < 0) return 0.0f;
        return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
    }
};

}  // namespace spt_host
