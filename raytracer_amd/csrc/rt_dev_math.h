// Device-side value types for the gfx950 render loop. Operand order and rounding points follow the
// reference's Vector3D (src/vector/vector3d.cpp) and RGBAColor (src/image/PNG.cpp:32-106): hit decisions
// are compared bit-for-bit with the CPU path, so this translation unit is built with -ffp-contract=off
// and uses IEEE division / square root (hipcc's defaults; never -ffast-math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

namespace rtd {

constexpr float kInf = __builtin_huge_valf();
constexpr double kPi = 3.14159265358979323846;      // M_PI
constexpr double kInvPi = 0.318309886183790671538;  // M_1_PI

struct V3 {
  float x, y, z;
};
RT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 v3(const float *p) { return v3(p[0], p[1], p[2]); }
RT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RT_DEV V3 operator*(float s, V3 v) { return v3(v.x * s, v.y * s, v.z * s); }
RT_DEV V3 operator*(V3 v, float s) { return v3(v.x * s, v.y * s, v.z * s); }
RT_DEV V3 operator/(V3 v, float s) { return v3(v.x / s, v.y / s, v.z / s); }
RT_DEV V3 operator-(V3 v, float s) { return v3(v.x - s, v.y - s, v.z - s); }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // left-to-right, vector3d.cpp:83-85
RT_DEV float magnitude(V3 v) { return sqrtf(dot(v, v)); }
RT_DEV V3 normalized(V3 v) {  // vector3d.cpp:78-81
  float invMag = 1.0f / magnitude(v);
  return v3(v.x * invMag, v.y * invMag, v.z * invMag);
}
RT_DEV V3 cross(V3 v1, V3 v2) {  // vector3d.cpp:134-140
  float i = v1.y * v2.z - v2.y * v1.z;
  float j = -1 * (v1.x * v2.z - v2.x * v1.z);
  float k = v1.x * v2.y - v2.x * v1.y;
  return v3(i, j, k);
}
// std::min / std::max as the reference uses them (src/acceleration/BVH.cpp:393-404): compare-and-select, so a
// NaN in the FIRST argument propagates and one in the second is ignored. fminf/fmaxf behave differently.
RT_DEV float rmin(float a, float b) { return (b < a) ? b : a; }
RT_DEV float rmax(float a, float b) { return (a < b) ? b : a; }
RT_DEV float clampf(float val, float low, float high) { return rmax(rmin(val, high), low); }  // math_utils.cpp:54-56
RT_DEV float cosineTheta(V3 a, V3 b) { return clampf(dot(a, b), -1, 1); }
RT_DEV float sineTheta(V3 a, V3 b) {
  float c = cosineTheta(a, b);
  return sqrtf(1 - c * c);
}
RT_DEV float tangentTheta(V3 a, V3 b) {
  float c = cosineTheta(a, b);
  float s = sineTheta(a, b);
  return s / c;
}
RT_DEV float cosinePhi(V3 v, V3 n) {  // vector3d.cpp:102-106: the scalar .z is subtracted from every component
  V3 vc = v - dot(v, n) * n;
  vc = (vc - vc.z) / magnitude(vc);
  return vc.x;
}
RT_DEV float clipDot(V3 a, V3 b) { return rmax(0.0f, dot(a, b)); }

struct RGBA {
  float r, g, b, a;
};
RT_DEV RGBA rgba(float r, float g, float b, float a) { RGBA c; c.r = r; c.g = g; c.b = b; c.a = a; return c; }
RT_DEV RGBA rgba(float4 v) { return rgba(v.x, v.y, v.z, v.w); }
RT_DEV float4 f4(RGBA c) { return make_float4(c.r, c.g, c.b, c.a); }
RT_DEV RGBA operator*(RGBA c, float s) { return rgba(s * c.r, s * c.g, s * c.b, c.a); }                         // PNG.cpp:32-37
RT_DEV RGBA operator*(RGBA c1, RGBA c2) { return rgba(c1.r * c2.r, c1.g * c2.g, c1.b * c2.b, rmax(c1.a, c2.a)); }  // :56-58
RT_DEV void mulAssign(RGBA &x, float s) { x.r *= s; x.g *= s; x.b *= s; }                                         // :49-54
RT_DEV void mulAssign(RGBA &x, RGBA c) { x.r *= c.r; x.g *= c.g; x.b *= c.b; x.a = rmax(x.a, c.a); }             // :60-66
RT_DEV void addAssign(RGBA &x, RGBA o) { x.r += o.r; x.g += o.g; x.b += o.b; x.a = rmax(o.a, x.a); }             // :79-85
RT_DEV bool hasNaN(RGBA c) { return isnan(c.r) || isnan(c.g) || isnan(c.b); }

// Counter-based random stream: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
// with key (seed, 0) and counter (pixel, sample, block, 0). A camera sample's draw number d is word d % 4 of block
// d / 4, so two draws coincide only if (seed, pixel, sample, d) coincide: the generator is a bijection of the counter
// for a fixed key. The only state a path carries between kernels is d (how many draws it has consumed); everything
// else is recomputed from the path's pixel and sample index, so images do not depend on the tile partition, the pass
// size or the GPU count. 24-bit floats in [0,1).
struct Rng {
  uint32_t seed, pixel, sample;
  uint32_t draws;     // draws consumed so far by this camera sample
  uint32_t block;     // block held in w, 0xffffffff = none
  uint32_t w[4];
};
RT_DEV void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
RT_DEV Rng rngInit(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t draws) {
  Rng r;
  r.seed = seed; r.pixel = pixel; r.sample = sample; r.draws = draws; r.block = 0xffffffffu;
  r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0u;
  return r;
}
RT_DEV float rngNext(Rng &r) {
  const uint32_t block = r.draws >> 2;
  if (block != r.block) {
    philox4x32_10(r.seed, 0u, r.pixel, r.sample, block, 0u, r.w);
    r.block = block;
  }
  const uint32_t lane = r.draws & 3u;
  const uint32_t word = lane == 0u ? r.w[0] : lane == 1u ? r.w[1] : lane == 2u ? r.w[2] : r.w[3];
  ++r.draws;
  return (float)(word >> 8) * (1.0f / 16777216.0f);
}

}  // namespace rtd
