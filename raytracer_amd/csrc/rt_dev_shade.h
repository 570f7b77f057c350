// Shading math on the device: BSDF lobes (src/bsdf/BDF.cpp), Trowbridge-Reitz (src/bsdf/microfacets.cpp), Fresnel
// terms, hemisphere sampling and texture addressing (src/bsdf/math_utils.cpp, src/scene/Object.cpp getColor
// overrides). Sub-expressions the reference evaluates in double (because of M_PI / M_1_PI / double literals) are
// evaluated in double here too; libm calls (sinf, cosf, atanf, acosf, atan2f, fmodf) go to the device math library,
// which is why shaded images are compared with a tolerance and not bit-for-bit.
#pragma once
#include "rt_dev_scene.h"

namespace rtd {

// src/bsdf/math_utils.cpp:3-21
RT_DEV V3 transformToWorld(float x, float y, float z, V3 normal) {
  const float inv_sqrt3 = 1.0f / sqrtf(3.0f);
  V3 majorAxis;
  if (fabsf(normal.x) < inv_sqrt3) majorAxis = v3(1, 0, 0);
  else if (fabsf(normal.y) < inv_sqrt3) majorAxis = v3(0, 1, 0);
  else majorAxis = v3(0, 0, 1);
  V3 u = normalized(cross(normal, majorAxis));
  V3 v = cross(normal, u);
  V3 w = normal;
  return u * x + v * y + w * z;
}

// src/bsdf/math_utils.cpp:23-36 (two draws)
RT_DEV V3 sampleHemisphere(V3 n, Rng &rng) {
  float rand = rngNext(rng);
  float r = sqrtf(rand);
  float theta = (float)((double)(rngNext(rng) * 2.0f) * kPi);
  float x = r * cosf(theta);
  float y = r * sinf(theta);
  float z = sqrtf(1.0f - x * x - y * y);
  return normalized(transformToWorld(x, y, z, n));
}

// src/bsdf/math_utils.cpp:38-48; returns (x, y) texel coordinates as floats
RT_DEV void sphericalToUV(V3 point, int32_t texW, int32_t texH, float &outX, float &outY) {
  V3 np = normalized(point);
  float phi = acosf(np.z);
  float theta = (float)(kPi + (double)atan2f(np.y, np.x));
  float width = (float)texW;
  float height = (float)texH;
  outY = (height - 1) - clampf((float)((double)phi * 0.5 * kInvPi * (double)height), 0, height - 1);
  outX = clampf((float)((double)theta * 0.5 * kInvPi * (double)width), 0, width - 1);
}

RT_DEV V3 faceForward(V3 w, V3 n) { return dot(w, n) < 0.0f ? -n : n; }                                  // :50-52
RT_DEV V3 reflect(V3 incident, V3 normal) { return (2.0f * dot(normal, incident)) * normal - incident; }  // :58-60
RT_DEV bool refract(V3 incident, V3 normal, float eta, V3 &wt) {                                          // :62-73
  float cosThetaI = dot(normal, incident);
  float sin2ThetaI = rmax(0.0f, 1.0f - cosThetaI * cosThetaI);
  float sin2ThetaT = eta * eta * sin2ThetaI;
  if (sin2ThetaT >= 1) return false;
  float cosThetaT = sqrtf(1.0f - sin2ThetaT);
  wt = eta * -incident + (eta * cosThetaI - cosThetaT) * normal;
  return true;
}
RT_DEV float fresnelDielectric(float cosThetaI, float etaI, float etaT) {  // :75-98
  cosThetaI = clampf(cosThetaI, -1, 1);
  bool entering = cosThetaI > 0.0f;
  if (!entering) {
    float tmp = etaI; etaI = etaT; etaT = tmp;
    cosThetaI = fabsf(cosThetaI);
  }
  float sinThetaI = sqrtf(rmax(0.0f, 1 - cosThetaI * cosThetaI));
  float sinThetaT = etaI / etaT * sinThetaI;
  if (sinThetaT >= 1) return 1;
  float cosThetaT = sqrtf(rmax(0.0f, 1 - sinThetaT * sinThetaT));
  float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
  float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
  return (Rparl * Rparl + Rperp * Rperp) / 2;
}
RT_DEV float fresnelConductor(float cosThetaI, float etaI, float etaT, float k) {  // :101-123
  cosThetaI = clampf(cosThetaI, -1, 1);
  float eta = etaT / etaI;
  float etak = k / etaI;
  float cosThetaI2 = cosThetaI * cosThetaI;
  float sinThetaI2 = 1.0f - cosThetaI2;
  float eta2 = eta * eta;
  float etak2 = etak * etak;
  float t0 = eta2 - etak2 - sinThetaI2;
  float a2plusb2 = sqrtf(t0 * t0 + 4 * eta2 * etak2);
  float t1 = a2plusb2 + cosThetaI2;
  float a = sqrtf(0.5f * (a2plusb2 + t0));
  float t2 = 2.0f * a * cosThetaI;
  float Rs = (t1 - t2) / (t1 + t2);
  float t3 = cosThetaI2 * a2plusb2 + sinThetaI2 * sinThetaI2;
  float t4 = 2.0f * a * cosThetaI * sinThetaI2;
  float Rp = Rs * (t3 - t4) / (t3 + t4);
  return (Rs + Rp) / 2;
}
RT_DEV float fresnelEvaluate(const DevLobe &l, float cosThetaI) {  // :125-131
  if (l.fresnel == FRESNEL_DIELECTRIC) return fresnelDielectric(cosThetaI, l.fEtaI, l.fEtaT);
  return fresnelConductor(fabsf(cosThetaI), l.fEtaI, l.fEtaT, l.fK);
}

// src/bsdf/microfacets.cpp
RT_DEV float trDistribution(float alpha, V3 wh, V3 n) {  // :9-23
  float tanTheta2 = tangentTheta(wh, n);
  tanTheta2 *= tanTheta2;
  if (isinf(tanTheta2)) return 0;
  float cosTheta4 = cosineTheta(wh, n);
  cosTheta4 = cosTheta4 * cosTheta4 * cosTheta4 * cosTheta4;
  float alpha2 = alpha * alpha;
  float t = tanTheta2 / alpha2;
  float onePlusT = 1 + t;
  return (float)(1 / (kPi * (double)alpha2 * (double)cosTheta4 * (double)onePlusT * (double)onePlusT));
}
RT_DEV float trLambda(float alpha, V3 w, V3 n) {  // :25-32
  float absTanTheta = fabsf(tangentTheta(w, n));
  if (isinf(absTanTheta)) return 0;
  float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
  return (-1 + sqrtf(1 + alpha2Tan2Theta)) / 2;
}
RT_DEV V3 trSampleWh(float alpha, V3 wo, V3 n, Rng &rng) {  // :34-47 (two draws)
  float rand = rngNext(rng);
  float theta = atanf(alpha * sqrtf(rand / (1.0f - rand)));
  float phi = (float)((double)(rngNext(rng) * 2.0f) * kPi);
  float x = sinf(phi) * cosf(theta);
  float y = sinf(phi) * sinf(theta);
  float z = cosf(phi);
  // faceForward(w, n) returns +-n: the "half vector" is the normal itself, oriented like the sampled direction
  return faceForward(normalized(transformToWorld(x, y, z, n)), n);
}
RT_DEV float trPdf(float alpha, V3 wh, V3 n) { return trDistribution(alpha, wh, n) * fabsf(cosineTheta(wh, n)); }
RT_DEV float trGeometry(float alpha, V3 wo, V3 wi, V3 n) { return 1 / (1 + trLambda(alpha, wo, n) + trLambda(alpha, wi, n)); }

// lobes, src/bsdf/BDF.cpp / BDF.h
RT_DEV float lobeFunc(const DevLobe &l, V3 wo, V3 wi, V3 n) {
  if (l.kind == LOBE_LAMBERT) return (float)((double)l.Kr * kInvPi);  // BDF.h:86-88
  if (l.kind == LOBE_OREN_NAYAR) {                                     // BDF.cpp:74-90
    float phiDiff = acosf(cosinePhi(wi, n)) - acosf(cosinePhi(wo, n));
    float maxCos = rmax(0.0f, phiDiff);
    float sinAlpha = sineTheta(wi, n);
    float tanBeta = tangentTheta(wo, n);
    float absCosO = fabsf(cosineTheta(wo, n));
    float absCosI = fabsf(cosineTheta(wi, n));
    if (absCosI > absCosO) {
      tanBeta = sinAlpha / absCosI;
      sinAlpha = sineTheta(wo, n);
    }
    return (float)((double)l.Kr * kInvPi * (double)(l.A + l.B * maxCos * sinAlpha * tanBeta));
  }
  if (l.kind == LOBE_MICROFACET) {  // BDF.cpp:92-104
    float cosThetaO = fabsf(cosineTheta(wo, n));
    float cosThetaI = fabsf(cosineTheta(wi, n));
    V3 wh = wi + wo;
    if (cosThetaI == 0 || cosThetaO == 0) return 0;
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return 0;
    wh = normalized(wh);
    float F = fresnelEvaluate(l, dot(wi, wh));
    return l.Kr * trDistribution(l.alpha, wh, n) * trGeometry(l.alpha, wo, wi, n) * F / (4 * cosThetaI * cosThetaO);
  }
  return 0.0f;
}
RT_DEV float basePdf(V3 wo, V3 wi, V3 n) {  // BDF.cpp:20-23 (tests dot(wo, wi), and carries a factor 2)
  return (dot(wo, wi) > 0) ? (float)((double)fabsf(dot(wi, n)) * kInvPi * 2) : 0.0f;
}
RT_DEV float lobePdf(const DevLobe &l, V3 wo, V3 wi, V3 n) {
  if (l.kind == LOBE_LAMBERT || l.kind == LOBE_OREN_NAYAR) return basePdf(wo, wi, n);
  if (l.kind == LOBE_MICROFACET) {  // BDF.cpp:117-122
    if (dot(wo, wi) < 0) return 0;
    V3 wh = normalized(wo + wi);
    return trPdf(l.alpha, wh, n) / (4 * fabsf(dot(wo, wh)));
  }
  return 0.0f;
}
RT_DEV float lobeSample(const DevLobe &l, V3 wo, V3 &wi, V3 n, Rng &rng, float &pdf) {
  if (l.kind == LOBE_LAMBERT || l.kind == LOBE_OREN_NAYAR) {  // BDF.cpp:13-18
    wi = sampleHemisphere(n, rng);
    pdf = basePdf(wo, wi, n);
    return lobeFunc(l, wo, wi, n);
  }
  if (l.kind == LOBE_SPECULAR_REFLECTION) {  // BDF.cpp:25-31
    wi = reflect(wo, n);
    pdf = 1.0f;
    float cosThetaI = dot(wi, n);
    return fresnelEvaluate(l, cosThetaI) * l.Kr / fabsf(cosThetaI);
  }
  if (l.kind == LOBE_FRESNEL_SPECULAR) {  // BDF.cpp:49-72
    float sample = rngNext(rng);
    float cosThetaI = dot(wo, n);
    float Fr = fresnelDielectric(cosThetaI, l.etaI, l.etaT);
    if (sample < Fr) {
      wi = reflect(wo, n);
      pdf = Fr;
      return Fr * l.Kr / fabsf(cosThetaI);
    }
    bool entering = cosThetaI > 0;
    float etaI = entering ? l.etaI : l.etaT;
    float etaT = entering ? l.etaT : l.etaI;
    if (refract(wo, faceForward(wo, n), etaI / etaT, wi) == false) return 0.0f;
    pdf = 1.0f - Fr;
    float Ft = l.Kt * (1.0f - Fr);
    Ft *= (etaI * etaI) / (etaT * etaT);
    return Ft / fabsf(dot(wi, n));
  }
  // microfacet, BDF.cpp:106-115
  V3 wh = trSampleWh(l.alpha, wo, n, rng);
  wi = reflect(wo, wh);
  pdf = trPdf(l.alpha, wh, n) / (4 * fabsf(dot(wo, wh)));
  return lobeFunc(l, wo, wi, n);
}

// BSDF::sampleFunc, src/bsdf/BDF.cpp:133-157 (the branch at :146 is always taken: no lobe's type equals the bare
// PERFECT_SPECULAR bit)
RT_DEV float bsdfSample(const DevMaterial &m, V3 wo, V3 &wi, V3 n, Rng &rng, float &pdf) {
  int numBDFs = m.numLobes;
  float rand = floorf(rngNext(rng) * numBDFs);
  int idx = min(numBDFs - 1, (int)rand);
  pdf = 0.0f;
  wi = v3(0, 0, 0);
  float contribution = lobeSample(m.lobes[idx], wo, wi, n, rng, pdf);
  if (pdf == 0) return 0;
  for (int i = 0; i < numBDFs; ++i) {
    if (i != idx) {
      pdf += lobePdf(m.lobes[i], wo, wi, n);
      contribution += lobeFunc(m.lobes[i], wo, wi, n);
    }
  }
  pdf /= numBDFs;
  return contribution;
}

// PNG::getPixel(row, col) with float arguments converted to unsigned (src/image/PNG.cpp:108-113). The reference asserts
// the bounds; out-of-range coordinates are clamped here instead of aborting.
RT_DEV RGBA texel(const DevScene &S, int32_t tex, float row, float col) {
  DevTexture t = S.textures[tex];
  if (t.width <= 0 || t.height <= 0) return rgba(0, 0, 0, 1);
  int64_t r = (row >= 0) ? (int64_t)row : 0;
  int64_t c = (col >= 0) ? (int64_t)col : 0;
  r = r < t.height - 1 ? r : t.height - 1;
  c = c < t.width - 1 ? c : t.width - 1;
  return rgba(S.texels[t.offset + r * t.width + c]);
}

// Object::getColor overrides, src/scene/Object.cpp:101-109 (sphere), :143-158 (plane), :215-224 (triangle)
// What shading reads of the object that was hit, loaded in one batch right after the hit is known (one memory round
// trip instead of one per use).
struct ObjectData {
  float4 a, b, c;   // hit record: triangle p1, normal, e1, e2 / sphere center, r
  float4 color;     // Object::color
  int32_t texture;  // -1 = none
  int32_t material;
};

RT_DEV RGBA getColor(const DevScene &S, int32_t prim, V3 ip, const ObjectData &od) {
  if (prim >= S.numPrims) {
    const DevPlane &p = S.planes[prim - S.numPrims];
    if (p.texture < 0) return rgba(p.color[0], p.color[1], p.color[2], p.color[3]);
    DevTexture tex = S.textures[p.texture];
    V3 tp = transformToWorld(ip.y, ip.x, ip.z, v3(0, 0, 1));
    tp = tp / tp.z;
    float x = p.texZoom * (tp.x - p.texTopLeft[0]) + p.texShift[0];
    float y = p.texZoom * (tp.y - p.texTopLeft[1]) + p.texShift[1];
    float width = (float)tex.width;
    float height = (float)tex.height;
    x = x > 0 ? fmodf(x, width) : width - fmodf(-x, width);
    y = y > 0 ? fmodf(y, height) : height - fmodf(-y, height);
    x = clampf(x, 0, width - 1);
    y = clampf(height - y, 0, height - 1);
    return texel(S, p.texture, y, x);
  }
  int32_t ti = od.texture;
  if (ti < 0) return rgba(od.color);
  float4 a = od.a, b = od.b, c = od.c;
  bool sphere = (__float_as_uint(a.w) | __float_as_uint(b.x) | __float_as_uint(b.y)) == 0u;
  if (sphere) {
    DevTexture tex = S.textures[ti];
    float x, y;
    sphericalToUV(ip - v3(a.x, a.y, a.z), tex.width, tex.height, x, y);
    return texel(S, ti, (float)(uint32_t)y, (float)(uint32_t)x);
  }
  V3 p1 = v3(a.x, a.y, a.z), e1 = v3(b.z, b.w, c.x), e2 = v3(c.y, c.z, c.w);
  float b2 = dot(e1, ip - p1);
  float b3 = dot(e2, ip - p1);
  float b1 = 1.0f - b3 - b2;
  V3 t1 = v3(0, 0, 0), t2 = t1, t3 = t1;
  if (S.primTexCoords) {
    const float *tc = S.primTexCoords + 9 * (int64_t)prim;
    t1 = v3(tc); t2 = v3(tc + 3); t3 = v3(tc + 6);
  }
  V3 tcoord = t1 * b1 + t2 * b2 + t3 * b3;
  return texel(S, ti, tcoord.y, tcoord.x);
}

// Light::emittedLight, src/scene/Object.h:61-63 and src/scene/Object.cpp:48-55
RT_DEV RGBA emittedLight(const DevScene &S, const DevLight &l, V3 direction) {
  if (l.type != RT_LIGHT_ENVIRONMENT) return rgba(0, 0, 0, 0);
  if (l.texture < 0) return rgba(l.color[0], l.color[1], l.color[2], l.color[3]);
  DevTexture tex = S.textures[l.texture];
  float x, y;
  sphericalToUV(v3(l.v) + direction * l.radius, tex.width, tex.height, x, y);
  return texel(S, l.texture, (float)(uint32_t)y, (float)(uint32_t)x);
}

}  // namespace rtd
