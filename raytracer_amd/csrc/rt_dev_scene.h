// HBM layout of a scene, as the kernels see it.
//
//   nodes      one 64-byte record per INNER node of the reference's BVH (src/acceleration/BVH.h:49-61): the boxes of
//              both children plus a packed reference to each, so that one inner step is one 64-byte fetch instead of
//              the two 32-byte fetches (left at idx+1, right far away) the pre-order array needs. Inner records are
//              numbered in pre-order, which keeps a subtree contiguous. Four float4 per record:
//                [0] lmin.xyz lmax.x   [1] lmax.yz rmin.xy   [2] rmin.z rmax.xyz   [3] leftRef rightRef 0 0 (as ints)
//   primHit    48 bytes per BVH object, in BVH order, three float4:
//                triangle: p1.xyz n.x | n.yz e1.xy | e1.z e2.xyz        (src/scene/Object.h:162-165)
//                sphere:   c.xyz 0    | 0 0 r 0    | 0 0 0 0            (src/scene/Object.h:119-120)
//              A triangle's unit normal is never the all-zero vector (a degenerate one is NaN), so three zero words in
//              the normal slot identify a sphere without a separate kind array in the hot loop.
//   primShade  64 bytes per BVH object, 64-byte aligned, four float4 — everything shading reads of an object besides
//              its hit record, in one memory sector:
//                n1.xyz n2.x | n2.yz n3.xy | n3.z color.rgb | color.a material texture originalIndex (last three as ints)
//   child reference (int32):
//                >= 0                              index of an inner record
//                bit31 | count<<24 | start         leaf with 1..127 objects starting below 2^24
//                bit31 | index (count field 0)     leaf described by bigLeaves[index] = {start, count}
#pragma once
#include "rt_amd.h"
#include "rt_dev_math.h"

namespace rtd {

constexpr int kMaxStack = 64;  // STACK_SIZE of the reference (src/acceleration/BVH.cpp:10)

struct BigLeaf {
  int32_t start, count;
};

struct DevPlane {  // src/scene/Object.h:123-143
  float normal[3], point[3];
  float texTopLeft[3], texZoom, texShift[3];
  float color[4];
  int32_t material, texture;
};

enum { LOBE_LAMBERT = 0, LOBE_OREN_NAYAR = 1, LOBE_SPECULAR_REFLECTION = 2, LOBE_FRESNEL_SPECULAR = 3, LOBE_MICROFACET = 4 };
enum { FRESNEL_DIELECTRIC = 0, FRESNEL_CONDUCTOR = 1 };

struct DevLobe {
  int32_t kind, fresnel;
  float Kr, Kt, etaI, etaT;  // FresnelSpecular: Kr, Kt, etas; Lambert / OrenNayar / Microfacet / SpecularReflection: Kr
  float A, B;                // OrenNayar
  float alpha;               // TrowbridgeReitz
  float fEtaI, fEtaT, fK;    // Fresnel term of Microfacet / SpecularReflection
};

struct DevMaterial {  // Material -> lobes, src/scene/Material.cpp:4-36
  int32_t type, numLobes;
  DevLobe lobes[2];
};

struct DevLight {  // src/scene/Object.h:54-105
  int32_t type;
  float color[4];
  float v[3];
  float radius;
  int32_t texture;
  int32_t envSlot;  // index among the environment lights (for the per-path sampled direction), -1 otherwise
};

struct DevTexture {
  int32_t width, height;
  int64_t offset;  // first texel in `texels`
};

struct DevScene {
  // traversal
  const float4 *nodes;
  const float4 *primHit;
  const BigLeaf *bigLeaves;
  int32_t rootRef;  // packed reference of the root; 0x7fffffff = no BVH (no objects)
  float rootMin[3], rootMax[3];
  int32_t numPrims;
  int32_t hasSpheres;
  int32_t boxesFinite;  // every node box coordinate is finite (enables the min/max slab test)
  float cutEps;         // point-light queries: how far outside its box an accepted hit point can lie (see buildScene); 0 = unknown, no box is culled against the bulb
  int32_t numPlanes;
  const DevPlane *planes;
  // shading
  const float4 *primShade;     // 64-byte shading record per object (see above), read once per hit
  const float *primTexCoords;  // 9 per object or nullptr
  const DevMaterial *materials;
  int32_t numLights, numEnvLights, numPointLights, numAnyLights;
  const DevLight *lights;
  const DevTexture *textures;
  const float4 *texels;
  // view
  float eye[3], forward[3], right[3], up[3];
  int32_t width, height;
  float bias, exposure;
  int32_t maxBounces, numRays;
  int32_t cameraMode;  // 0 default, 1 fisheye, 2 depth of field (src/scene/raytracer.cpp:290-299)
  float focus, lens;
  float invForwardLength;  // fisheye: (float)(1.0 / |forward|), raytracer.cpp:75
  float normalizedForward[3];
  float magRight, magUp;   // dof: magnitude(camera.right/up), raytracer.cpp:148-149
};

constexpr int32_t kNoBvh = 0x7fffffff;
RT_DEV bool refIsLeaf(int32_t ref) { return ref < 0; }

}  // namespace rtd
