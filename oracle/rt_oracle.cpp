// TEST INFRASTRUCTURE ONLY — never linked, imported or called by the product path.
//
// CPU restatement of the reference's render loop (RichardXiao13/Raytracer), statement by statement,
// over the same RtSceneDesc the HIP library consumes (include/rt_amd.h; entry points exported with an
// `rto_` prefix). It is the checker for the GPU path and the `cpu_baseline` subject of bench.py.
//
// Pinning: with RT_RNG_REF_MT19937 (one std::mt19937, default seed, consumed in the reference's -t1
// pixel order) its float framebuffer is compared bit-for-bit with the compiled reference
// (oracle/_ref/ref_harness render ... 1); primary hits are compared bit-for-bit with
// `ref_harness hits`. Fixtures made that way live in tests/golden/.
//
// Numerics: compile with -ffp-contract=off on x86-64 baseline (no FMA), like the reference build.
// Each function cites the reference lines it follows (paths relative to the reference's root).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <math.h>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"

namespace {

constexpr float INF = std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------- src/vector/vector3d.{h,cpp}
struct V3 {
  float x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(float a, float b, float c) : x(a), y(b), z(c) {}
  explicit V3(const float *p) : x(p[0]), y(p[1]), z(p[2]) {}
};
inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3 &a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(float s, const V3 &v) { return V3(v.x * s, v.y * s, v.z * s); }
inline V3 operator*(const V3 &v, float s) { return V3(v.x * s, v.y * s, v.z * s); }
inline V3 operator/(float s, const V3 &v) { return V3(s / v.x, s / v.y, s / v.z); }
inline V3 operator/(const V3 &v, float s) { return V3(v.x / s, v.y / s, v.z / s); }
inline V3 operator-(const V3 &v, float s) { return V3(v.x - s, v.y - s, v.z - s); }
inline float dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vector3d.cpp:83-85
inline float magnitude(const V3 &v) { return std::sqrt(dot(v, v)); }                        // :74-76
inline V3 normalized(const V3 &v) {                                                         // :78-81
  float invMag = 1.0f / magnitude(v);
  return V3(v.x * invMag, v.y * invMag, v.z * invMag);
}
inline V3 cross(const V3 &v1, const V3 &v2) {  // :134-140
  float i = v1.y * v2.z - v2.y * v1.z;
  float j = -1 * (v1.x * v2.z - v2.x * v1.z);
  float k = v1.x * v2.y - v2.x * v1.y;
  return V3(i, j, k);
}
inline float clampf(float val, float low, float high) { return std::max(std::min(val, high), low); }  // math_utils.cpp:54-56
inline float cosineTheta(const V3 &a, const V3 &b) { return clampf(dot(a, b), -1, 1); }             // vector3d.cpp:87-89
inline float sineTheta(const V3 &a, const V3 &b) {                                                   // :91-94
  float c = cosineTheta(a, b);
  return std::sqrt(1 - c * c);
}
inline float tangentTheta(const V3 &a, const V3 &b) {  // :96-100
  float c = cosineTheta(a, b);
  float s = sineTheta(a, b);
  return s / c;
}
inline float cosinePhi(const V3 &v, const V3 &n) {  // :102-106 (subtracts the scalar .z from every component)
  V3 vc = v - dot(v, n) * n;
  vc = (vc - vc.z) / magnitude(vc);
  return vc.x;
}
inline float clipDot(const V3 &a, const V3 &b) { return std::max(0.0f, dot(a, b)); }  // :120-122

// ---------------------------------------------------------------- src/image/PNG.{h,cpp} colour type
struct RGBA {
  float r, g, b, a;
  RGBA() : r(0), g(0), b(0), a(1.0f) {}
  RGBA(float r1, float g1, float b1, float a1) : r(r1), g(g1), b(b1), a(a1) {}
  explicit RGBA(const float *p) : r(p[0]), g(p[1]), b(p[2]), a(p[3]) {}
};
inline RGBA operator*(const RGBA &c, float s) { return RGBA(s * c.r, s * c.g, s * c.b, c.a); }          // PNG.cpp:32-37
inline RGBA operator*(const RGBA &c1, const RGBA &c2) {                                                   // :56-58
  return RGBA(c1.r * c2.r, c1.g * c2.g, c1.b * c2.b, std::max(c1.a, c2.a));
}
inline void mulAssign(RGBA &x, float s) { x.r *= s; x.g *= s; x.b *= s; }                                  // :49-54
inline void mulAssign(RGBA &x, const RGBA &c) { x.r *= c.r; x.g *= c.g; x.b *= c.b; x.a = std::max(x.a, c.a); }  // :60-66
inline void addAssign(RGBA &x, const RGBA &o) { x.r += o.r; x.g += o.g; x.b += o.b; x.a = std::max(o.a, x.a); }  // :79-85
inline bool hasNaN(const RGBA &c) { return std::isnan(c.r) || std::isnan(c.g) || std::isnan(c.b); }        // :87-89
inline RGBA clipColor(const RGBA &c) {                                                                     // :99-106
  return RGBA(std::max(0.0f, std::min(c.r, 1.0f)), std::max(0.0f, std::min(c.g, 1.0f)),
              std::max(0.0f, std::min(c.b, 1.0f)), std::max(0.0f, std::min(c.a, 1.0f)));
}
inline float exponentialExposure(float channel, float exposure) {  // PNG.cpp:20-22; exp(float) is the float overload
  return 1.0 - std::exp(-1 * channel * exposure);
}
inline float linearToGamma(float channel) {  // PNG.cpp:7-12
  if (channel < 0.0031308) return 12.92 * channel;
  return 1.055 * std::pow(channel, 1 / 2.4) - 0.055;
}

// ---------------------------------------------------------------- samplers
struct Sampler {
  virtual ~Sampler() {}
  virtual float next() = 0;
};
// src/bsdf/math_utils.h:8-19 as constructed at src/scene/raytracer.cpp:33
struct MtSampler : Sampler {
  std::mt19937 rng;
  std::uniform_real_distribution<float> dist{0, 1.0};
  float next() override { return dist(rng); }
};
// Counter-based stream shared with the HIP kernels: Philox4x32-10 (Salmon et al., SC'11), key (seed, 0), counter
// (pixel, sample, block, 0); draw number d of a camera sample is word d % 4 of block d / 4.
inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c[4]) {
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
struct CounterSampler : Sampler {
  uint32_t seedWord = 0, pixel = 0, sample = 0, draws = 0, block = 0xffffffffu;
  uint32_t w[4] = {0, 0, 0, 0};
  void seed(uint32_t seed, uint32_t pixel_, uint32_t sample_) {
    seedWord = seed; pixel = pixel_; sample = sample_; draws = 0; block = 0xffffffffu;
  }
  float next() override {
    const uint32_t b = draws >> 2;
    if (b != block) {
      w[0] = pixel; w[1] = sample; w[2] = b; w[3] = 0;
      philox4x32_10(seedWord, 0u, w);
      block = b;
    }
    const uint32_t word = w[draws & 3u];
    ++draws;
    return (float)(word >> 8) * (1.0f / 16777216.0f);
  }
};

// ---------------------------------------------------------------- scene copy
struct Texture { int w = 0, h = 0; std::vector<float> px; };

enum LobeKind { LAMBERT, OREN_NAYAR, SPECULAR_REFLECTION, FRESNEL_SPECULAR, MICROFACET };
enum FresnelKind { FRESNEL_DIELECTRIC, FRESNEL_CONDUCTOR };
struct Lobe {
  LobeKind kind;
  float Kr = 0, Kt = 0, etaI = 0, etaT = 0;  // FresnelSpecular etas / Lambert & OrenNayar Kr
  float A = 0, B = 0;                       // OrenNayar
  float alpha = 0;                          // TrowbridgeReitz
  FresnelKind fresnel = FRESNEL_DIELECTRIC;
  float fEtaI = 0, fEtaT = 0, fK = 0;
};
struct Material { int type; std::vector<Lobe> lobes; };

struct Counters {
  uint64_t raysClosest = 0, raysAny = 0, inner = 0, prims = 0, planes = 0, samples = 0, raysExtend = 0;
  uint64_t innerExtend = 0, primExtend = 0;  // the part of inner/prims spent on path-extension queries
  bool extending = false;
  void add(const Counters &o) {
    raysClosest += o.raysClosest; raysAny += o.raysAny; inner += o.inner; prims += o.prims; planes += o.planes; samples += o.samples;
    raysExtend += o.raysExtend; innerExtend += o.innerExtend; primExtend += o.primExtend;
  }
};

struct Hit {
  float t = INF;
  V3 point, normal;
  int prim = -1;  // BVH position; numPrims + k for plane k; -1 = nullptr
};

}  // namespace

// Per-pixel state of the sample loop that outlives a call (progressive rendering, RtRenderParams::keepAccumulator).
struct PixelAcc {
  RGBA sum = RGBA(0, 0, 0, 0);
  int hits = 0, taken = 0;
  float lumSum = 0, lumSqSum = 0;
};

struct RtoScene {
  std::vector<RtBvhNode> nodes;
  int numPrims = 0;
  std::vector<int32_t> kind, materialOf, textureOf, origIndex;
  std::vector<float> hit, normals, texCoords, color;
  std::vector<RtPlane> planes;
  std::vector<Material> materials;
  std::vector<RtLight> lights;
  std::vector<Texture> textures;
  RtCamera camera;
  RtOptions options;
  std::vector<float> framebuffer;
  std::vector<PixelAcc> acc;  // per-pixel sums of the sample loop, kept for RtRenderParams::keepAccumulator
  int threads = 1;
};

namespace {

thread_local std::string g_error;
thread_local Counters *tl_counters = nullptr;
Counters g_dummyCounters;
inline Counters &ctr() { return tl_counters ? *tl_counters : g_dummyCounters; }

// ---------------------------------------------------------------- src/bsdf/math_utils.cpp
V3 transformToWorld(float x, float y, float z, const V3 &normal) {  // :3-21
  V3 majorAxis;
  if (std::abs(normal.x) < 1.0f / sqrtf(3.0)) majorAxis = V3(1, 0, 0);
  else if (std::abs(normal.y) < 1.0f / sqrtf(3.0)) majorAxis = V3(0, 1, 0);
  else majorAxis = V3(0, 0, 1);
  V3 u = normalized(cross(normal, majorAxis));
  V3 v = cross(normal, u);
  V3 w = normal;
  return u * x + v * y + w * z;
}
V3 sampleHemisphere(const V3 &n, Sampler &sampler) {  // :23-36
  float rand = sampler.next();
  float r = sqrtf(rand);
  float theta = sampler.next() * 2.0f * M_PI;
  float x = r * cosf(theta);
  float y = r * sinf(theta);
  float z = sqrtf(1.0f - x * x - y * y);
  return normalized(transformToWorld(x, y, z, n));
}
V3 sphericalToUV(const V3 &point, const Texture &tex) {  // :38-48
  V3 np = normalized(point);
  float cosPhi = np.z;
  float phi = std::acos(cosPhi);
  float theta = M_PI + std::atan2(np.y, np.x);
  float width = tex.w;
  float height = tex.h;
  float y = (height - 1) - clampf(phi * 0.5 * M_1_PI * height, 0, height - 1);
  float x = clampf(theta * 0.5 * M_1_PI * width, 0, width - 1);
  return V3(x, y, 0.0f);
}
V3 faceForward(const V3 &w, const V3 &n) { return dot(w, n) < 0.0f ? -n : n; }                       // :50-52
V3 reflect(const V3 &incident, const V3 &normal) { return 2.0f * dot(normal, incident) * normal - incident; }  // :58-60
bool refract(const V3 &incident, const V3 &normal, float eta, V3 *wt) {  // :62-73
  float cosThetaI = dot(normal, incident);
  float sin2ThetaI = std::max(0.0f, 1.0f - cosThetaI * cosThetaI);
  float sin2ThetaT = eta * eta * sin2ThetaI;
  if (sin2ThetaT >= 1) return false;
  float cosThetaT = std::sqrt(1.0f - sin2ThetaT);
  *wt = eta * -incident + (eta * cosThetaI - cosThetaT) * normal;
  return true;
}
float fresnelDielectric(float cosThetaI, float etaI, float etaT) {  // :75-98
  cosThetaI = clampf(cosThetaI, -1, 1);
  bool entering = cosThetaI > 0.0f;
  if (!entering) {
    std::swap(etaI, etaT);
    cosThetaI = std::abs(cosThetaI);
  }
  float sinThetaI = std::sqrt(std::max(0.0f, 1 - cosThetaI * cosThetaI));
  float sinThetaT = etaI / etaT * sinThetaI;
  if (sinThetaT >= 1) return 1;
  float cosThetaT = std::sqrt(std::max(0.0f, 1 - sinThetaT * sinThetaT));
  float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
  float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
  return (Rparl * Rparl + Rperp * Rperp) / 2;
}
float fresnelConductor(float cosThetaI, float etaI, float etaT, float k) {  // :101-123
  cosThetaI = clampf(cosThetaI, -1, 1);
  float eta = etaT / etaI;
  float etak = k / etaI;
  float cosThetaI2 = cosThetaI * cosThetaI;
  float sinThetaI2 = 1.0f - cosThetaI2;
  float eta2 = eta * eta;
  float etak2 = etak * etak;
  float t0 = eta2 - etak2 - sinThetaI2;
  float a2plusb2 = std::sqrt(t0 * t0 + 4 * eta2 * etak2);
  float t1 = a2plusb2 + cosThetaI2;
  float a = std::sqrt(0.5f * (a2plusb2 + t0));
  float t2 = 2.0f * a * cosThetaI;
  float Rs = (t1 - t2) / (t1 + t2);
  float t3 = cosThetaI2 * a2plusb2 + sinThetaI2 * sinThetaI2;
  float t4 = 2.0f * a * cosThetaI * sinThetaI2;
  float Rp = Rs * (t3 - t4) / (t3 + t4);
  return (Rs + Rp) / 2;
}
float fresnelEvaluate(const Lobe &l, float cosThetaI) {  // :125-131
  if (l.fresnel == FRESNEL_DIELECTRIC) return fresnelDielectric(cosThetaI, l.fEtaI, l.fEtaT);
  return fresnelConductor(std::abs(cosThetaI), l.fEtaI, l.fEtaT, l.fK);
}

// ---------------------------------------------------------------- src/bsdf/microfacets.cpp (Trowbridge-Reitz)
float trDistribution(float alpha, const V3 &wh, const V3 &n) {  // :9-23
  float tanTheta2 = tangentTheta(wh, n);
  tanTheta2 *= tanTheta2;
  if (std::isinf(tanTheta2)) return 0;
  float cosTheta4 = cosineTheta(wh, n);
  cosTheta4 = cosTheta4 * cosTheta4 * cosTheta4 * cosTheta4;
  float alpha2 = alpha * alpha;
  float t = tanTheta2 / alpha2;
  return 1 / (M_PI * alpha2 * cosTheta4 * (1 + t) * (1 + t));
}
float trLambda(float alpha, const V3 &w, const V3 &n) {  // :25-32
  float absTanTheta = std::abs(tangentTheta(w, n));
  if (std::isinf(absTanTheta)) return 0;
  float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
  return (-1 + std::sqrt(1 + alpha2Tan2Theta)) / 2;
}
V3 trSampleWh(float alpha, const V3 &wo, const V3 &n, Sampler &sampler) {  // :34-47
  float rand = sampler.next();
  float theta = atanf(alpha * sqrtf(rand / (1.0f - rand)));
  float phi = sampler.next() * 2.0f * M_PI;
  float x = sinf(phi) * cosf(theta);
  float y = sinf(phi) * sinf(theta);
  float z = cosf(phi);
  // faceForward returns +-n, not the sampled half vector (math_utils.cpp:50-52)
  return faceForward(normalized(transformToWorld(x, y, z, n)), n);
}
float trPdf(float alpha, const V3 &wh, const V3 &n) { return trDistribution(alpha, wh, n) * std::abs(cosineTheta(wh, n)); }  // :5-7
float trGeometry(float alpha, const V3 &wo, const V3 &wi, const V3 &n) {  // microfacets.h:14-16
  return 1 / (1 + trLambda(alpha, wo, n) + trLambda(alpha, wi, n));
}

// ---------------------------------------------------------------- src/bsdf/BDF.cpp lobes
float lobeFunc(const Lobe &l, const V3 &wo, const V3 &wi, const V3 &n) {
  switch (l.kind) {
    case LAMBERT: return l.Kr * M_1_PI;  // BDF.h:86-88
    case OREN_NAYAR: {                   // BDF.cpp:74-90
      float maxCos;
      float phiDiff = std::acos(cosinePhi(wi, n)) - std::acos(cosinePhi(wo, n));
      maxCos = std::max(0.0f, phiDiff);
      float sinAlpha = sineTheta(wi, n);
      float tanBeta = tangentTheta(wo, n);
      float absCosO = std::abs(cosineTheta(wo, n));
      float absCosI = std::abs(cosineTheta(wi, n));
      if (absCosI > absCosO) {
        tanBeta = sinAlpha / absCosI;
        sinAlpha = sineTheta(wo, n);
      }
      return l.Kr * M_1_PI * (l.A + l.B * maxCos * sinAlpha * tanBeta);
    }
    case MICROFACET: {  // BDF.cpp:92-104
      float cosThetaO = std::abs(cosineTheta(wo, n));
      float cosThetaI = std::abs(cosineTheta(wi, n));
      V3 wh = wi + wo;
      if (cosThetaI == 0 || cosThetaO == 0) return 0;
      if (wh.x == 0 && wh.y == 0 && wh.z == 0) return 0;
      wh = normalized(wh);
      float F = fresnelEvaluate(l, dot(wi, wh));
      return l.Kr * trDistribution(l.alpha, wh, n) * trGeometry(l.alpha, wo, wi, n) * F / (4 * cosThetaI * cosThetaO);
    }
    default: return 0.0f;  // specular lobes: BDF.h:30-32, 66-68
  }
}
float basePdf(const V3 &wo, const V3 &wi, const V3 &n) {  // BDF.cpp:20-23
  return (dot(wo, wi) > 0) ? std::abs(dot(wi, n)) * M_1_PI * 2 : 0;
}
float lobePdf(const Lobe &l, const V3 &wo, const V3 &wi, const V3 &n) {
  switch (l.kind) {
    case LAMBERT:
    case OREN_NAYAR: return basePdf(wo, wi, n);
    case MICROFACET: {  // BDF.cpp:117-122
      if (dot(wo, wi) < 0) return 0;
      V3 wh = normalized(wo + wi);
      return trPdf(l.alpha, wh, n) / (4 * std::abs(dot(wo, wh)));
    }
    default: return 0.0f;
  }
}
float lobeSample(const Lobe &l, const V3 &wo, V3 *wi, const V3 &n, Sampler &sampler, float *pdf) {
  switch (l.kind) {
    case LAMBERT:
    case OREN_NAYAR:  // BDF.cpp:13-18
      *wi = sampleHemisphere(n, sampler);
      *pdf = basePdf(wo, *wi, n);
      return lobeFunc(l, wo, *wi, n);
    case SPECULAR_REFLECTION: {  // BDF.cpp:25-31
      *wi = reflect(wo, n);
      *pdf = 1.0f;
      float cosThetaI = dot(*wi, n);
      return fresnelEvaluate(l, cosThetaI) * l.Kr / std::abs(cosThetaI);
    }
    case FRESNEL_SPECULAR: {  // BDF.cpp:49-72
      float sample = sampler.next();
      float cosThetaI = dot(wo, n);
      float Fr = fresnelDielectric(cosThetaI, l.etaI, l.etaT);
      if (sample < Fr) {
        *wi = reflect(wo, n);
        *pdf = Fr;
        return Fr * l.Kr / std::abs(cosThetaI);
      } else {
        bool entering = cosThetaI > 0;
        float etaI = entering ? l.etaI : l.etaT;
        float etaT = entering ? l.etaT : l.etaI;
        if (refract(wo, faceForward(wo, n), etaI / etaT, wi) == false) return 0.0f;
        *pdf = 1.0f - Fr;
        float Ft = l.Kt * (1.0f - Fr);
        Ft *= (etaI * etaI) / (etaT * etaT);
        return Ft / std::abs(dot(*wi, n));
      }
    }
    case MICROFACET: {  // BDF.cpp:106-115
      V3 wh = trSampleWh(l.alpha, wo, n, sampler);
      *wi = reflect(wo, wh);
      *pdf = trPdf(l.alpha, wh, n) / (4 * std::abs(dot(wo, wh)));
      return lobeFunc(l, wo, *wi, n);
    }
  }
  return 0.0f;
}
// BSDF::sampleFunc, BDF.cpp:133-157. Every lobe's `type` differs from the bare PERFECT_SPECULAR bit, so the
// mixing branch at :146 is always taken.
float bsdfSample(const Material &m, const V3 &wo, V3 *wi, const V3 &n, Sampler &sampler, float *pdf) {
  int numBDFs = (int)m.lobes.size();
  float rand = std::floor(sampler.next() * numBDFs);
  int idx = std::min(numBDFs - 1, static_cast<int>(rand));
  const Lobe &bdf = m.lobes[idx];
  float contribution = lobeSample(bdf, wo, wi, n, sampler, pdf);
  if (*pdf == 0) return 0;
  for (int i = 0; i < numBDFs; ++i) {
    if (i != idx) {
      *pdf += lobePdf(m.lobes[i], wo, *wi, n);
      contribution += lobeFunc(m.lobes[i], wo, *wi, n);
    }
  }
  *pdf /= numBDFs;
  return contribution;
}

Material makeMaterial(const RtMaterial &m) {  // src/scene/Material.cpp:4-36
  Material out;
  out.type = m.type;
  auto microfacet = [&](FresnelKind fk, float fI, float fT, float fk_) {
    Lobe l; l.kind = MICROFACET; l.Kr = m.Kr;
    l.alpha = std::sqrt(std::min(1e-4f, m.roughness));  // roughnessToAlpha, microfacets.cpp:49-51 (min, not max)
    l.fresnel = fk; l.fEtaI = fI; l.fEtaT = fT; l.fK = fk_;
    return l;
  };
  if (m.type == RT_MAT_DEFAULT) {
    Lobe l; l.kind = LAMBERT; l.Kr = 1.0f;  // Material() always uses Kd = 1
    out.lobes.push_back(l);
  } else if (m.type == RT_MAT_METAL) {
    out.lobes.push_back(microfacet(FRESNEL_CONDUCTOR, 1.0f, m.eta, m.Ka));
  } else if (m.type == RT_MAT_MIRROR) {
    Lobe l; l.kind = SPECULAR_REFLECTION; l.Kr = m.Kr; l.fresnel = FRESNEL_DIELECTRIC; l.fEtaI = 1.0f; l.fEtaT = m.eta;
    out.lobes.push_back(l);
  } else if (m.type == RT_MAT_PLASTIC) {
    Lobe o; o.kind = OREN_NAYAR; o.Kr = m.Kd;
    float sigma = 20;  // BDF.h:108-115
    sigma *= M_PI / 180;
    float sigma2 = sigma * sigma;
    o.A = 1.0f - 0.5 * sigma2 / (sigma2 + 0.33);
    o.B = 0.45 * sigma2 / (sigma2 + 0.09);
    out.lobes.push_back(o);
    out.lobes.push_back(microfacet(FRESNEL_DIELECTRIC, 1.0f, m.eta, 0.0f));
  } else if (m.type == RT_MAT_GLASS) {
    Lobe l; l.kind = FRESNEL_SPECULAR; l.Kr = m.Kr; l.Kt = m.Kt; l.etaI = 1.0f; l.etaT = m.eta;
    out.lobes.push_back(l);
  }
  return out;
}

// ---------------------------------------------------------------- primitives, src/scene/Object.cpp
Hit intersectTriangle(const RtoScene &s, int i, const V3 &origin, const V3 &direction) {  // :195-213
  const float *h = &s.hit[(size_t)i * 12];
  V3 p1(h), normal(h + 3), e1(h + 6), e2(h + 9);
  V3 nd = normalized(direction);
  float t = dot((p1 - origin), normal) / dot(nd, normal);
  if (t < 0) return Hit();
  V3 ip = t * nd + origin;
  float b2 = dot(e1, ip - p1);
  float b3 = dot(e2, ip - p1);
  float b1 = 1.0f - b3 - b2;
  if (b1 < 0 || b1 > 1 || b2 < 0 || b2 > 1 || b3 < 0 || b3 > 1) return Hit();
  const float *nn = &s.normals[(size_t)i * 9];
  Hit out;
  out.t = t; out.point = ip; out.prim = i;
  out.normal = normalized(V3(nn) * b1 + V3(nn + 3) * b2 + V3(nn + 6) * b3);
  return out;
}
Hit intersectSphere(const RtoScene &s, int i, const V3 &origin, const V3 &direction) {  // :75-99
  const float *h = &s.hit[(size_t)i * 12];
  V3 center(h);
  float r = h[3];
  float radiusSquared = r * r;
  V3 dfs = center - origin;
  bool inside = dot(dfs, dfs) < radiusSquared;
  V3 nd = normalized(direction);
  float tc = dot(dfs, nd);
  if (inside == false && tc < 0) return Hit();
  V3 d = origin + tc * nd - center;
  float distanceSquared = dot(d, d);
  if (inside == false && radiusSquared < distanceSquared) return Hit();
  float tOffset = std::sqrt(radiusSquared - distanceSquared);
  float t = inside ? tc + tOffset : tc - tOffset;
  Hit out;
  out.t = t; out.point = t * nd + origin; out.normal = normalized(out.point - center); out.prim = i;
  return out;
}
Hit intersectPlane(const RtoScene &s, int k, const V3 &origin, const V3 &direction) {  // :133-141
  const RtPlane &p = s.planes[k];
  V3 nd = normalized(direction);
  V3 normal(p.normal), point(p.point);
  float t = dot((point - origin), normal) / dot(nd, normal);
  if (t < 0) return Hit();
  Hit out;
  out.t = t; out.point = t * nd + origin; out.normal = normal; out.prim = s.numPrims + k;
  return out;
}

// ---------------------------------------------------------------- src/acceleration/BVH.cpp traversal
float intersectAABB(const V3 &origin, const V3 &inv, const float *mn, const float *mx) {  // :390-409
  float tx_near = (mn[0] - origin.x) * inv.x;
  float tx_far = (mx[0] - origin.x) * inv.x;
  float tmin = std::min(tx_near, tx_far);
  float tmax = std::max(tx_near, tx_far);
  float ty_near = (mn[1] - origin.y) * inv.y;
  float ty_far = (mx[1] - origin.y) * inv.y;
  tmin = std::max(tmin, std::min(ty_near, ty_far));
  tmax = std::min(tmax, std::max(ty_near, ty_far));
  float tz_near = (mn[2] - origin.z) * inv.z;
  float tz_far = (mx[2] - origin.z) * inv.z;
  tmin = std::max(tmin, std::min(tz_near, tz_far));
  tmax = std::min(tmax, std::max(tz_near, tz_far));
  if (tmax >= tmin && tmax > 0) return tmin;
  return INF;
}
Hit bvhClosest(const RtoScene &s, const V3 &origin, const V3 &direction) {  // :283-340
  if (s.nodes.empty()) return Hit();
  Counters &c = ctr();
  int to_visit[64];
  float distances[64];
  V3 inv = 1.0f / direction;
  to_visit[0] = 0;
  distances[0] = intersectAABB(origin, inv, s.nodes[0].aabbMin, s.nodes[0].aabbMax);
  int stackIdx = 0;
  float minDistance = INF;
  Hit closest;
  while (stackIdx >= 0) {
    float dist = distances[stackIdx];
    int nodeIdx = to_visit[stackIdx];
    const RtBvhNode &sub = s.nodes[nodeIdx];
    if (dist < minDistance) {
      if (sub.numObjects > 0) {
        int end = sub.rightOrStart + sub.numObjects;
        for (int i = sub.rightOrStart; i < end; ++i) {
          ++c.prims;
          if (c.extending) ++c.primExtend;
          Hit info = s.kind[i] == RT_PRIM_TRIANGLE ? intersectTriangle(s, i, origin, direction)
                                                   : intersectSphere(s, i, origin, direction);
          if (info.t < minDistance) {
            minDistance = info.t;
            closest = info;
          }
        }
      } else {
        ++c.inner;
        if (c.extending) ++c.innerExtend;
        int leftIdx = nodeIdx + 1;
        int rightIdx = sub.rightOrStart;
        const RtBvhNode &left = s.nodes[leftIdx];
        const RtBvhNode &right = s.nodes[rightIdx];
        float leftDistance = intersectAABB(origin, inv, left.aabbMin, left.aabbMax);
        float rightDistance = intersectAABB(origin, inv, right.aabbMin, right.aabbMax);
        if (rightDistance < leftDistance) {
          std::swap(leftIdx, rightIdx);
          std::swap(leftDistance, rightDistance);
        }
        if (leftDistance != INF) {
          if (rightDistance != INF) {
            to_visit[stackIdx] = rightIdx;
            distances[stackIdx++] = rightDistance;
          }
          to_visit[stackIdx] = leftIdx;
          distances[stackIdx++] = leftDistance;
        }
      }
    }
    --stackIdx;
  }
  return closest;
}

// ---------------------------------------------------------------- src/scene/raytracer.cpp:174-196
Hit findClosestObject(const RtoScene &s, const V3 &origin, const V3 &direction) {
  ++ctr().raysClosest;
  Hit closest = bvhClosest(s, origin, direction);
  for (int k = 0; k < (int)s.planes.size(); ++k) {
    ++ctr().planes;
    Hit info = intersectPlane(s, k, origin, direction);
    if (info.t < closest.t) closest = info;
  }
  return closest;
}
Hit findAnyObject(const RtoScene &s, const V3 &origin, const V3 &direction) {
  ++ctr().raysAny;
  Hit closest = bvhClosest(s, origin, direction);
  if (closest.prim >= 0) return closest;
  for (int k = 0; k < (int)s.planes.size(); ++k) {
    ++ctr().planes;
    Hit info = intersectPlane(s, k, origin, direction);
    if (info.t < closest.t) return info;
  }
  return closest;
}

// ---------------------------------------------------------------- textures / getColor, src/scene/Object.cpp
RGBA texel(const Texture &t, float row, float col) {
  // PNG::getPixel(unsigned row, unsigned col) (src/image/PNG.cpp:108-113) asserts the bounds and the float ->
  // unsigned conversion of a negative value is undefined; this restatement clamps instead of aborting.
  if (t.w <= 0 || t.h <= 0) return RGBA(0, 0, 0, 1);
  long r = (long)row, c = (long)col;
  if (!(row >= 0)) r = 0;
  if (!(col >= 0)) c = 0;
  r = std::min<long>(r, t.h - 1);
  c = std::min<long>(c, t.w - 1);
  return RGBA(&t.px[((size_t)r * t.w + c) * 4]);
}
RGBA getColor(const RtoScene &s, int prim, const V3 &ip) {
  if (prim >= s.numPrims) {  // Plane::getColor :143-158
    const RtPlane &p = s.planes[prim - s.numPrims];
    if (p.texture < 0) return RGBA(p.color);
    const Texture &tex = s.textures[p.texture];
    V3 tp = transformToWorld(ip.y, ip.x, ip.z, V3(0, 0, 1));
    tp = tp / tp.z;
    float x = p.texZoom * (tp.x - p.texTopLeft[0]) + p.texShift[0];
    float y = p.texZoom * (tp.y - p.texTopLeft[1]) + p.texShift[1];
    float width = tex.w;
    float height = tex.h;
    x = x > 0 ? std::fmod(x, width) : width - std::fmod(-x, width);
    y = y > 0 ? std::fmod(y, height) : height - std::fmod(-y, height);
    x = clampf(x, 0, width - 1);
    y = clampf(height - y, 0, height - 1);
    return texel(tex, y, x);
  }
  int ti = s.textureOf[prim];
  if (ti < 0) return RGBA(&s.color[(size_t)prim * 4]);
  const Texture &tex = s.textures[ti];
  const float *h = &s.hit[(size_t)prim * 12];
  if (s.kind[prim] == RT_PRIM_SPHERE) {  // Sphere::getColor :101-109
    V3 tc = sphericalToUV(ip - V3(h), tex);
    unsigned x = static_cast<unsigned>(tc.x);
    unsigned y = static_cast<unsigned>(tc.y);
    return texel(tex, (float)y, (float)x);
  }
  // Triangle::getColor :215-224
  V3 p1(h), e1(h + 6), e2(h + 9);
  float b2 = dot(e1, ip - p1);
  float b3 = dot(e2, ip - p1);
  float b1 = 1.0f - b3 - b2;
  const float *tcs = s.texCoords.empty() ? nullptr : &s.texCoords[(size_t)prim * 9];
  V3 t1 = tcs ? V3(tcs) : V3(), t2 = tcs ? V3(tcs + 3) : V3(), t3 = tcs ? V3(tcs + 6) : V3();
  V3 tc = t1 * b1 + t2 * b2 + t3 * b3;
  return texel(tex, tc.y, tc.x);
}

// ---------------------------------------------------------------- lights, src/scene/Object.cpp:7-55
RGBA emittedLight(const RtoScene &s, const RtLight &l, const V3 &direction) {
  if (l.type != RT_LIGHT_ENVIRONMENT) return RGBA(0, 0, 0, 0);  // Object.h:61-63
  if (l.texture < 0) return RGBA(l.color);
  const Texture &tex = s.textures[l.texture];
  V3 mc = sphericalToUV(V3(l.v) + direction * l.radius, tex);
  unsigned x = static_cast<unsigned>(mc.x);
  unsigned y = static_cast<unsigned>(mc.y);
  return texel(tex, (float)y, (float)x);
}
RGBA lightIntensity(const RtoScene &s, const RtLight &l, const V3 &point, const V3 &n, Sampler &sampler) {
  if (l.type == RT_LIGHT_DISTANT) {  // :7-16
    V3 direction(l.v);
    if (findAnyObject(s, point, direction).prim >= 0) return RGBA(0, 0, 0, 1);
    return RGBA(l.color) * clipDot(n, direction);
  }
  if (l.type == RT_LIGHT_POINT) {  // :18-34
    const V3 lightDirection = V3(l.v) - point;
    float intersectToBulbDist = magnitude(lightDirection);
    Hit info = findClosestObject(s, point, lightDirection);
    float objectToIntersect = magnitude(point - info.point);
    if (info.prim >= 0 && objectToIntersect < intersectToBulbDist) return RGBA(0, 0, 0, 1);
    float distance = magnitude(lightDirection);
    float invDistance = 1.0f / (distance * distance);
    return RGBA(l.color) * clipDot(n, lightDirection) * invDistance;
  }
  // environment :36-46
  V3 wi = sampleHemisphere(n, sampler);
  if (findAnyObject(s, point, wi).prim >= 0) return RGBA(0, 0, 0, 1);
  return emittedLight(s, l, wi) * clipDot(n, wi);
}

// ---------------------------------------------------------------- src/scene/raytracer.cpp:198-251
RGBA illuminate(const RtoScene &s, const Hit &info, Sampler &sampler) {
  RGBA L;
  for (const RtLight &l : s.lights) addAssign(L, lightIntensity(s, l, info.point, info.normal, sampler));
  return getColor(s, info.prim, info.point) * L;
}
RGBA raytrace(const RtoScene &s, const V3 &origin, const V3 &direction, Sampler &sampler) {
  RGBA L(0, 0, 0, 0);
  RGBA beta(1, 1, 1, 1);
  V3 rayOrigin = origin;
  V3 rayDirection = direction;
  for (int bounces = 0; bounces < s.options.maxBounces; ++bounces) {
    ++ctr().raysExtend;
    ctr().extending = true;
    Hit info = findClosestObject(s, rayOrigin, rayDirection);
    ctr().extending = false;
    if (info.prim < 0) {
      for (const RtLight &l : s.lights) addAssign(L, emittedLight(s, l, rayDirection));
      break;
    }
    V3 point = info.point;
    V3 wo = -normalized(rayDirection);
    V3 outNormal = faceForward(wo, info.normal);
    info.point = info.point + s.options.bias * outNormal;
    bool isPlane = info.prim >= s.numPrims;
    int matIdx = isPlane ? s.planes[info.prim - s.numPrims].material : s.materialOf[info.prim];
    const Material &material = s.materials[matIdx];
    addAssign(L, beta * illuminate(s, info, sampler));
    if (material.type == RT_MAT_METAL)
      mulAssign(beta, isPlane ? RGBA(s.planes[info.prim - s.numPrims].color) : RGBA(&s.color[(size_t)info.prim * 4]));
    float pdf = 0.0f;
    V3 wi;
    float contribution = bsdfSample(material, wo, &wi, info.normal, sampler, &pdf);
    if (pdf == 0 || contribution == 0) break;
    mulAssign(beta, contribution * std::abs(dot(wi, info.normal)) / pdf);
    bool exiting = dot(wi, outNormal) > 0;
    rayOrigin = point + outNormal * (exiting ? s.options.bias : -s.options.bias);
    rayDirection = wi;
  }
  return L;
}

// ---------------------------------------------------------------- camera samples, src/scene/raytracer.cpp:13-172
float getRayScaleX(float x, int w, int h) { return (2 * x - w) / std::max(w, h); }
float getRayScaleY(float y, int w, int h) { return (h - 2 * y) / std::max(w, h); }

struct CameraMode {
  int mode;  // 0 default, 1 fisheye, 2 dof (selection at raytracer.cpp:290-299)
  float invForwardLength = 0;
  V3 normalizedForward;
};

// One iteration of the `for i < numRays` loop. Returns false when the fisheye sample is rejected (`continue`).
bool cameraSample(const RtoScene &s, const CameraMode &cm, int x, int y, int allowAA, Sampler &sampler, RGBA *color) {
  const RtOptions &o = s.options;
  V3 eye(s.camera.eye), forward(s.camera.forward), right(s.camera.right), up(s.camera.up);
  float Sx = getRayScaleX(x + (sampler.next() - 0.5f) * allowAA, o.width, o.height);
  float Sy = getRayScaleY(y + (sampler.next() - 0.5f) * allowAA, o.width, o.height);
  ++ctr().samples;
  if (cm.mode == 0) {  // :44-47
    *color = raytrace(s, eye, forward + Sx * right + Sy * up, sampler);
    return true;
  }
  if (cm.mode == 1) {  // :92-103
    Sx *= cm.invForwardLength;
    Sy *= cm.invForwardLength;
    float r_2 = Sx * Sx + Sy * Sy;
    if (r_2 > 1) return false;
    V3 forwardCopy = std::sqrt(1 - r_2) * cm.normalizedForward;
    *color = raytrace(s, eye, forwardCopy + Sx * right + Sy * up, sampler);
    return true;
  }
  // :142-153
  V3 rayDirection = forward + Sx * right + Sy * up;
  V3 intersectionPoint = o.focus / magnitude(rayDirection) * rayDirection + eye;
  float weight = sampler.next() * 2 * M_PI;
  float r = sampler.next() - 0.5f;
  V3 origin = r * std::cos(weight) * o.lens / magnitude(right) * right + r * std::sin(weight) * o.lens / magnitude(up) * up + eye;
  rayDirection = intersectionPoint - origin;
  *color = raytrace(s, origin, rayDirection, sampler);
  return true;
}

CameraMode cameraMode(const RtoScene &s) {
  CameraMode cm;
  V3 forward(s.camera.forward);
  if (s.options.fisheye) {
    cm.mode = 1;
    cm.invForwardLength = 1.0 / magnitude(forward);  // :75
    cm.normalizedForward = normalized(forward);
  } else if (s.options.focus > 0) {
    cm.mode = 2;
  } else {
    cm.mode = 0;
  }
  return cm;
}

// The per-pixel body of threadTaskDefault/Fisheye/DOF (:36-59 and twins): samples [first, end) of the loop
// `for i < numRays`, added to `acc`; with threshold > 0 the pixel stops at a batch boundary once its luminance estimate
// has converged (the rule of RtRenderParams::adaptiveThreshold). All-zero extensions = the reference's loop.
RGBA renderPixel(const RtoScene &s, const CameraMode &cm, int x, int y, Sampler *shared, uint32_t seed, PixelAcc &acc, int first, int end,
                 float threshold = 0.0f, int batch = 16, int minSamples = 16) {
  const RtOptions &o = s.options;
  int allowAA = std::min(1, o.numRays - 1);
  CounterSampler cs;
  for (int i = first; i < end; ++i) {
    Sampler *sampler = shared;
    if (!sampler) {
      cs.seed(seed, (uint32_t)(y * o.width + x), (uint32_t)i);
      sampler = &cs;
    }
    RGBA color;
    bool traced = cameraSample(s, cm, x, y, allowAA, *sampler, &color);
    ++acc.taken;
    if (traced && hasNaN(color) == false && color.a != 0) {
      addAssign(acc.sum, color);
      ++acc.hits;
      if (threshold > 0) {
        float yv = (0.2126f * color.r + 0.7152f * color.g) + 0.0722f * color.b;
        acc.lumSum += yv;
        acc.lumSqSum += yv * yv;
      }
    }
    if (threshold > 0 && (i + 1 - first) % batch == 0 && acc.taken < o.numRays && acc.taken >= minSamples) {
      if (acc.hits == 0) break;
      if (acc.hits >= 2) {
        float mean = acc.lumSum / (float)acc.hits;
        float var = std::max(0.0f, acc.lumSqSum - acc.lumSum * mean) / (float)(acc.hits - 1);
        float se = std::sqrt(var / (float)acc.hits);
        if (se <= threshold * (mean + 0.05f)) break;
      }
    }
  }
  RGBA avg = acc.sum;
  float invNumRays = 1.0 / std::max(1, acc.taken);  // = 1.0 / numRays once the whole loop has run (:39)
  if (acc.hits != 0) {
    mulAssign(avg, 1.0f / acc.hits);
    avg.a = acc.hits * invNumRays;
  }
  return clipColor(avg);
}

bool inPartition(const RtRenderParams &p, int W, int x, int y) {
  if (p.partCount <= 1) return true;
  int tw = p.tileW > 0 ? p.tileW : 32, th = p.tileH > 0 ? p.tileH : 32;
  int tilesX = (W + tw - 1) / tw;
  int tile = (y / th) * tilesX + (x / tw);
  return tile % p.partCount == p.partIndex;
}

int fail(const std::string &m) { g_error = m; return RT_ERR_INVALID; }

}  // namespace

extern "C" {

const char *rto_last_error(void) { return g_error.c_str(); }
const char *rto_version(void) { return "rt_oracle 1 (CPU restatement, test infrastructure)"; }

RtoScene *rto_scene_create(const RtSceneDesc *d, int /*device*/) {
  if (!d) { fail("null desc"); return nullptr; }
  // The restatement walks the reference's tree only: a description without nodes asks for the HIP library's own
  // device-side BVH build, which has no counterpart in the reference.
  if (d->numPrims > 0 && (!d->nodes || d->numNodes <= 0)) { fail("the CPU restatement needs the host-built BVH nodes"); return nullptr; }
  auto s = std::make_unique<RtoScene>();
  s->nodes.assign(d->nodes, d->nodes + d->numNodes);
  s->numPrims = d->numPrims;
  size_t n = (size_t)d->numPrims;
  s->kind.assign(d->primKind, d->primKind + n);
  s->hit.assign(d->primHit, d->primHit + n * 12);
  s->normals.assign(d->primNormals, d->primNormals + n * 9);
  if (d->primTexCoords) s->texCoords.assign(d->primTexCoords, d->primTexCoords + n * 9);
  s->color.assign(d->primColor, d->primColor + n * 4);
  s->materialOf.assign(d->primMaterial, d->primMaterial + n);
  s->textureOf.assign(d->primTexture, d->primTexture + n);
  s->origIndex.assign(d->primOrigIndex, d->primOrigIndex + n);
  s->planes.assign(d->planes, d->planes + d->numPlanes);
  for (int i = 0; i < d->numMaterials; ++i) s->materials.push_back(makeMaterial(d->materials[i]));
  s->lights.assign(d->lights, d->lights + d->numLights);
  for (int i = 0; i < d->numTextures; ++i) {
    Texture t;
    t.w = d->textures[i].width; t.h = d->textures[i].height;
    if (t.w > 0 && t.h > 0) t.px.assign(d->textures[i].rgba, d->textures[i].rgba + (size_t)t.w * t.h * 4);
    s->textures.push_back(std::move(t));
  }
  s->camera = d->camera;
  s->options = d->options;
  s->threads = 1;
  return s.release();
}

void rto_scene_destroy(RtoScene *s) { delete s; }

int rto_scene_set_view(RtoScene *s, const RtCamera *camera, const RtOptions *options) {
  if (!s) return fail("null scene");
  if (camera) s->camera = *camera;
  if (options) s->options = *options;
  return RT_OK;
}

// Not part of rt_amd.h: how many host threads rto_render / rto_trace_* may use (counter RNG mode only).
void rto_set_threads(RtoScene *s, int threads) { if (s) s->threads = std::max(1, threads); }

static void parallelRows(int threads, int rows, Counters *total, const std::function<void(int)> &rowFn) {
  threads = std::max(1, std::min(threads, rows));
  std::atomic<int> next{0};
  std::vector<Counters> per(threads);
  auto work = [&](int t) {
    tl_counters = &per[t];
    for (int y; (y = next.fetch_add(1)) < rows;) rowFn(y);
    tl_counters = nullptr;
  };
  if (threads == 1) work(0);
  else {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
    for (auto &th : pool) th.join();
  }
  if (total) for (auto &c : per) total->add(c);
}

int rto_trace_primary(RtoScene *s, int32_t x0, int32_t y0, int32_t w, int32_t h, int32_t *primId, float *t) {
  if (!s || !primId || !t) return fail("null argument");
  V3 eye(s->camera.eye), forward(s->camera.forward), right(s->camera.right), up(s->camera.up);
  int W = s->options.width, H = s->options.height;
  parallelRows(s->threads, h, nullptr, [&](int row) {
    int y = y0 + row;
    for (int col = 0; col < w; ++col) {
      int x = x0 + col;
      float Sx = getRayScaleX((float)x, W, H);
      float Sy = getRayScaleY((float)y, W, H);
      Hit info = findClosestObject(*s, eye, forward + Sx * right + Sy * up);
      size_t j = (size_t)row * w + col;
      t[j] = info.t;
      primId[j] = info.prim < 0 ? -1 : (info.prim < s->numPrims ? s->origIndex[info.prim] : info.prim);
    }
  });
  return RT_OK;
}

int rto_trace_closest(RtoScene *s, int64_t n, const float *origins, const float *directions, int32_t *primId, float *t,
                      float *point, float *normal) {
  if (!s || !origins || !directions) return fail("null argument");
  const int chunk = 1024;
  int rows = (int)((n + chunk - 1) / chunk);
  parallelRows(s->threads, rows, nullptr, [&](int row) {
    for (int64_t j = (int64_t)row * chunk; j < std::min<int64_t>(n, (int64_t)(row + 1) * chunk); ++j) {
      Hit info = findClosestObject(*s, V3(origins + j * 3), V3(directions + j * 3));
      if (t) t[j] = info.t;
      if (primId) primId[j] = info.prim < 0 ? -1 : (info.prim < s->numPrims ? s->origIndex[info.prim] : info.prim);
      if (point) { point[j * 3] = info.prim < 0 ? 0 : info.point.x; point[j * 3 + 1] = info.prim < 0 ? 0 : info.point.y; point[j * 3 + 2] = info.prim < 0 ? 0 : info.point.z; }
      if (normal) { normal[j * 3] = info.prim < 0 ? 0 : info.normal.x; normal[j * 3 + 1] = info.prim < 0 ? 0 : info.normal.y; normal[j * 3 + 2] = info.prim < 0 ? 0 : info.normal.z; }
    }
  });
  return RT_OK;
}

int rto_trace_any(RtoScene *s, int64_t n, const float *origins, const float *directions, uint8_t *occluded) {
  if (!s || !origins || !directions || !occluded) return fail("null argument");
  const int chunk = 1024;
  int rows = (int)((n + chunk - 1) / chunk);
  parallelRows(s->threads, rows, nullptr, [&](int row) {
    for (int64_t j = (int64_t)row * chunk; j < std::min<int64_t>(n, (int64_t)(row + 1) * chunk); ++j)
      occluded[j] = findAnyObject(*s, V3(origins + j * 3), V3(directions + j * 3)).prim >= 0 ? 1 : 0;
  });
  return RT_OK;
}

int rto_render(RtoScene *s, const RtRenderParams *params, float *rgbaOut, RtStats *stats) {
  if (!s || !params) return fail("null argument");
  const RtOptions &o = s->options;
  int W = o.width, H = o.height;
  if (W <= 0 || H <= 0) return fail("bad image size");
  for (const Material &m : s->materials)
    if (m.lobes.empty()) return fail("material without BSDF lobes");
  s->framebuffer.assign((size_t)W * H * 4, 0.0f);
  CameraMode cm = cameraMode(*s);
  Counters total;
  const int sFirst = params->firstSample;
  if (sFirst < 0 || sFirst >= o.numRays || params->sampleCount < 0) return fail("firstSample / sampleCount outside 0 .. numRays");
  const int sEnd = params->sampleCount > 0 ? std::min(o.numRays, sFirst + params->sampleCount) : o.numRays;
  const bool adaptive = params->adaptiveThreshold > 0.0f;
  const int batch = params->adaptiveBatch > 0 ? params->adaptiveBatch : 16;
  const int minSamples = params->adaptiveMinSamples > 0 ? params->adaptiveMinSamples : batch;
  if (adaptive && (sFirst != 0 || params->keepAccumulator)) return fail("adaptive sampling starts at sample 0 with an empty accumulator");
  if (params->keepAccumulator && s->acc.size() != (size_t)W * H) return fail("keepAccumulator needs a previous render of the same image size");
  if (!params->keepAccumulator) s->acc.assign((size_t)W * H, PixelAcc());
  if (params->rngMode == RT_RNG_REF_MT19937) {
    // one worker, one generator, row-major pixel order: the reference at -t1 (raytracer.cpp:33, 269-273)
    MtSampler sampler;
    tl_counters = &total;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        RGBA px = renderPixel(*s, cm, x, y, &sampler, 0, s->acc[(size_t)y * W + x], sFirst, sEnd);
        std::memcpy(&s->framebuffer[((size_t)y * W + x) * 4], &px, 16);
      }
    tl_counters = nullptr;
  } else {
    parallelRows(s->threads, H, &total, [&](int y) {
      for (int x = 0; x < W; ++x) {
        if (!inPartition(*params, W, x, y)) continue;
        RGBA px = renderPixel(*s, cm, x, y, nullptr, params->seed, s->acc[(size_t)y * W + x], sFirst, sEnd,
                              adaptive ? params->adaptiveThreshold : 0.0f, batch, minSamples);
        std::memcpy(&s->framebuffer[((size_t)y * W + x) * 4], &px, 16);
      }
    });
  }
  if (o.exposure >= 0 && !params->skipExposure) {  // Scene::expose, raytracer.cpp:303-312
    for (size_t i = 0; i < (size_t)W * H; ++i) {
      float *p = &s->framebuffer[i * 4];
      p[0] = exponentialExposure(p[0], o.exposure);
      p[1] = exponentialExposure(p[1], o.exposure);
      p[2] = exponentialExposure(p[2], o.exposure);
    }
    if (params->partCount > 1)  // pixels outside the partition stay exactly 0
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
          if (!inPartition(*params, W, x, y)) std::memset(&s->framebuffer[((size_t)y * W + x) * 4], 0, 16);
  }
  if (rgbaOut) std::memcpy(rgbaOut, s->framebuffer.data(), s->framebuffer.size() * sizeof(float));
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->cameraSamples = total.samples;
    stats->raysClosest = total.raysClosest;
    stats->raysAny = total.raysAny;
    stats->innerExpansions = total.inner;
    stats->primTests = total.prims;
    stats->planeTests = total.planes;
    stats->raysExtend = total.raysExtend;
    stats->raysShadow = total.raysClosest + total.raysAny - total.raysExtend;
    stats->innerExtend = total.innerExtend; stats->primExtend = total.primExtend;
    stats->innerShadow = total.inner - total.innerExtend; stats->primShadow = total.prims - total.primExtend;
  }
  return RT_OK;
}

const float *rto_framebuffer_device(RtoScene *s) { return s ? s->framebuffer.data() : nullptr; }

// Test hook: one Philox4x32-10 block, for the known-answer vectors.
void rto_philox4x32_10(const uint32_t key[2], const uint32_t counter[4], uint32_t out[4]) {
  uint32_t c[4] = {counter[0], counter[1], counter[2], counter[3]};
  philox4x32_10(key[0], key[1], c);
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}
// Test hook: the first n draws of camera sample (seed, pixel, sample) as the kernels and rto_render see them.
void rto_counter_draws(uint32_t seed, uint32_t pixel, uint32_t sample, int32_t n, float *out) {
  CounterSampler cs;
  cs.seed(seed, pixel, sample);
  for (int32_t i = 0; i < n; ++i) out[i] = cs.next();
}

int rto_device_count(void) { return 0; }  // the CPU restatement uses no GPU

int rto_framebuffer_add(RtoScene *dst, RtoScene *src) {  // host mirror of rt_framebuffer_add
  if (!dst || !src || dst == src) return fail("rto_framebuffer_add needs two different scenes");
  if (dst->framebuffer.size() != src->framebuffer.size() || dst->framebuffer.empty()) return fail("framebuffers differ in size");
  for (size_t i = 0; i < dst->framebuffer.size(); ++i) dst->framebuffer[i] = dst->framebuffer[i] + src->framebuffer[i];
  return RT_OK;
}

int rto_framebuffer_gather(RtoScene *dst, RtoScene *const *srcs, int32_t numSrcs) {  // host mirror of rt_framebuffer_gather
  for (int32_t i = 0; i < numSrcs; ++i)
    if (int rc = rto_framebuffer_add(dst, srcs[i])) return rc;
  return RT_OK;
}

int rto_framebuffer_read(RtoScene *s, float *out) {
  if (!s || !out) return fail("null argument");
  std::copy(s->framebuffer.begin(), s->framebuffer.end(), out);
  return RT_OK;
}

int rto_resolve_rgba8(RtoScene *s, uint8_t *out) {  // PNG::saveToFile, src/image/PNG.cpp:149-157
  if (!s || !out) return fail("null argument");
  for (size_t i = 0; i < s->framebuffer.size() / 4; ++i) {
    const float *p = &s->framebuffer[i * 4];
    out[i * 4 + 0] = static_cast<unsigned char>(std::round(linearToGamma(p[0]) * 255));
    out[i * 4 + 1] = static_cast<unsigned char>(std::round(linearToGamma(p[1]) * 255));
    out[i * 4 + 2] = static_cast<unsigned char>(std::round(linearToGamma(p[2]) * 255));
    out[i * 4 + 3] = static_cast<unsigned char>(std::round(p[3] * 255));
  }
  return RT_OK;
}

}  // extern "C"
