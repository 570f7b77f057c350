// Host-side value types. Every operation keeps the operand order and rounding points of
// the reference's Vector3D (src/vector/vector3d.cpp:5-152) and RGBAColor
// (src/image/PNG.cpp:32-97), because BVH leaf order and hit bits depend on them.
// Build with -ffp-contract=off (no fused multiply-add).
#pragma once
#include <cmath>

namespace rth {

struct Vec3 {
  float x = 0.0f, y = 0.0f, z = 0.0f;
  Vec3() = default;
  Vec3(float x1, float y1, float z1) : x(x1), y(y1), z(z1) {}
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(const Vec3 &a) { return Vec3(-a.x, -a.y, -a.z); }
// scalar * v and v * scalar both multiply component-by-scalar (vector3d.cpp:46-52)
inline Vec3 operator*(float s, const Vec3 &v) { return Vec3(v.x * s, v.y * s, v.z * s); }
inline Vec3 operator*(const Vec3 &v, float s) { return Vec3(v.x * s, v.y * s, v.z * s); }
inline Vec3 operator/(const Vec3 &v, float s) { return Vec3(v.x / s, v.y / s, v.z / s); }

// vector3d.cpp:83-85 — left-to-right sum of products
inline float dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vector3d.cpp:74-81
inline float magnitude(const Vec3 &v) { return std::sqrt(dot(v, v)); }
inline Vec3 normalized(const Vec3 &v) {
  float invMag = 1.0f / magnitude(v);
  return Vec3(v.x * invMag, v.y * invMag, v.z * invMag);
}
// vector3d.cpp:134-140 — note the explicit -1 * on the j component
inline Vec3 cross(const Vec3 &v1, const Vec3 &v2) {
  float i = v1.y * v2.z - v2.y * v1.z;
  float j = -1 * (v1.x * v2.z - v2.x * v1.z);
  float k = v1.x * v2.y - v2.x * v1.y;
  return Vec3(i, j, k);
}
// vector3d.cpp:150-152
inline float maxDimension(const Vec3 &v) { return std::max(v.z, std::max(v.x, v.y)); }

struct Color {
  float r = 0.0f, g = 0.0f, b = 0.0f, a = 1.0f;  // RGBAColor() default is opaque black (PNG.h:13)
  Color() = default;
  Color(float r1, float g1, float b1, float a1 = 1.0f) : r(r1), g(g1), b(b1), a(a1) {}
};

}  // namespace rth
