// libraytracer_amd.so — the render loop of RichardXiao13/Raytracer as a wavefront path tracer for gfx950 (MI355X),
// behind the C ABI of include/rt_amd.h.
//
// One frame = passes of P paths (camera samples). A path owns one slot of SoA state in HBM for the whole pass; the
// kernels below are launched once per bounce over index queues of live slots:
//
//   k_raygen      Scene::threadTaskDefault/Fisheye/DOF sample set-up (src/scene/raytracer.cpp:44-47, 92-103, 142-153)
//   k_extend      Scene::findClosestObject for every live path (src/scene/raytracer.cpp:216)           <- traversal
//   k_shade       miss -> environment; hit -> geometry, getColor, light sampling set-up, BSDF sample (:217-246)
//   k_shadow      the visibility queries of Light::intensity (src/scene/Object.cpp:7-46), one per hit and light  <- traversal
//   k_direct      L += beta * (getColor * sum of the visible lights' intensities) (:198-206, :231)
//   k_advance     queue bookkeeping and ray counters
//   k_accumulate  per-pixel sample average bookkeeping (:48-51), in sample order
//   k_resolve     hits/alpha/clip (:54-59) and Scene::expose (:303-312) into the float RGBA framebuffer
//
// Built with -ffp-contract=off: primary hit ids and t bits are compared bit-for-bit with the CPU path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "rt_amd.h"
#include "rt_bvh_build.h"
#include "rt_dev_shade.h"
#include "rt_dev_traverse.h"

using namespace rtd;

namespace {

constexpr int kBlock = 256;       // 4 wavefronts
// Register budget of the traversal kernels: waves per SIMD the compiler must leave room for, and the scalar-register cap
// that goes with it (256-thread blocks are admitted per CU by 800 / (sgprs rounded up to 16, + 16)).
#ifndef RT_TRAV_WAVES
#define RT_TRAV_WAVES 7
#endif
#ifndef RT_TRAV_SGPRS
#define RT_TRAV_SGPRS 96
#endif
#define RT_TRAV_ATTR __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_TRAV_WAVES, 8), amdgpu_num_sgpr(RT_TRAV_SGPRS)))
constexpr int kStackLds = 8;      // stack entries per lane kept in LDS (8 B each -> 16 KiB per block)

// Every group of fields that one kernel updates with atomics while others only read sits in a 128-byte line of its own:
// a read of a line that thousands of wavefronts are adding to queues behind those adds.
struct Ticket {
  alignas(128) uint32_t next;  // next unclaimed queue index of one segment
};
struct TicketSet {
  Ticket seg[kQueueSegments];
};
struct Counts {
  alignas(128) uint32_t nActive;  // live paths of this bounce (written between kernels, and by k_raygen for fisheye)
  alignas(128) uint32_t nHit;     // appended by k_shade; {nHit, nNext} are added to as ONE 64-bit word
  uint32_t nNext;
  TicketSet ticket[3];            // [1] light visibility; [0] / [2] path extension of even / odd bounces (two, because the rays of a
                                  // bounce can be traced before the counters of the previous one are reset), [2] also trace-only kernels
  alignas(128) unsigned long long skipPoint;  // visibility queries not traced (see ShadowSource), by query kind
  unsigned long long skipAny;
  unsigned long long planePoint, planeAny;    // visibility queries a plane answered before any BVH node was touched
  alignas(128) unsigned long long raysClosest, raysAny, samples, raysExtend, raysShadow;
  unsigned long long inner[2], prims[2], planes[2];  // [0] path extension, [1] light visibility
  float innerWaveIters[2], primWaveIters[2];
};

struct PathBuf {
  float4 *rayO, *rayD, *L;
  // Path weight. A bounce reads beta[bounce & 1] (what the path carried into the bounce; all ones at bounce 0 and not
  // stored) and writes beta[(bounce + 1) & 1]: k_direct, which runs after k_shade has updated the weight, still finds the
  // value from before the update without a copy being kept for it.
  float4 *beta[2];
  uint32_t *rng;
  float *hitT;
  int32_t *hitPrim;
  float4 *shPoint, *shColor;
  float4 *envWi;   // [numEnvLights][P] hemisphere direction drawn for each environment light
  float4 *lightC;  // [numLights][P] what each light contributes if it turns out to be visible
  uint32_t *qA, *qB, *qHit;
  // [numLights][rowsCap] what k_shade decided about the visibility query of each queue position, 64 positions per entry:
  // bit p of {x, y} (low, high word) and of {z, w} form a 2-bit code per position (ShadowCode)
  uint4 *shadowCode;
  int32_t rowsCap;
  Counts *counts;
  int32_t P, npix;
  int32_t spp;  // samples per pixel in this pass; path slot = pixel position * spp + sample, so a pixel's samples are neighbours
  const int32_t *active;  // adaptive sampling: positions (in the pixel list) of the pixels this pass samples; nullptr = all, in order
};
static_assert(offsetof(Counts, nNext) == offsetof(Counts, nHit) + 4 && offsetof(Counts, nHit) % 8 == 0, "nHit/nNext form one 64-bit word");

// Appends the lanes with `pred` to a queue with one atomic per wavefront; lane order is kept inside the wavefront.
// Must be reached by every lane of the wavefront.
RT_DEV uint32_t queueSlot(uint32_t *counter, bool pred) {
  unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return 0u;
  uint32_t lane = __lane_id();
  int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Appends to two queues at once with ONE atomic per block: `counters` is {countA, countB} as a 64-bit word. Lane order is
// kept inside the block. Must be reached by every thread of the block.
RT_DEV void queueSlots2(uint32_t *counters, bool predA, bool predB, uint32_t &slotA, uint32_t &slotB) {
  __shared__ uint32_t waveA[kBlock / 64], waveB[kBlock / 64];
  __shared__ unsigned long long blockBase;
  const unsigned long long maskA = __ballot(predA), maskB = __ballot(predB);
  const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
  if (lane == 0) { waveA[wave] = (uint32_t)__popcll(maskA); waveB[wave] = (uint32_t)__popcll(maskB); }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long add = 0;
    for (int w = 0; w < kBlock / 64; ++w) add += (unsigned long long)waveA[w] | ((unsigned long long)waveB[w] << 32);
    blockBase = add ? atomicAdd(reinterpret_cast<unsigned long long *>(counters), add) : 0ull;
  }
  __syncthreads();
  uint32_t a = (uint32_t)blockBase, b = (uint32_t)(blockBase >> 32);
  for (uint32_t w = 0; w < wave; ++w) { a += waveA[w]; b += waveB[w]; }
  const unsigned long long below = (1ull << lane) - 1ull;
  slotA = a + (uint32_t)__popcll(maskA & below);
  slotB = b + (uint32_t)__popcll(maskB & below);
}

template <bool COUNT>
RT_DEV void flushCounters(Counts *c, const TravCounters &tc, int kind) {
  if (COUNT) {
    // wave reduction, then one atomic per wavefront and counter
    unsigned long long a = tc.inner, b = tc.prims, p = tc.planes;
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off); b += __shfl_down(b, off); p += __shfl_down(p, off);
    }
    float wi = tc.innerWaveIters, wp = tc.primWaveIters;
    for (int off = 32; off > 0; off >>= 1) { wi += __shfl_down(wi, off); wp += __shfl_down(wp, off); }
    if (__lane_id() == 0) {
      atomicAdd(&c->inner[kind], a); atomicAdd(&c->prims[kind], b); atomicAdd(&c->planes[kind], p);
      atomicAdd(&c->innerWaveIters[kind], wi); atomicAdd(&c->primWaveIters[kind], wp);
    }
  }
}

RT_DEV float getRayScaleX(float x, int w, int h) { return (2 * x - w) / max(w, h); }  // src/scene/raytracer.cpp:13-15
RT_DEV float getRayScaleY(float y, int w, int h) { return (h - 2 * y) / max(w, h); }  // :17-19

// ---------------------------------------------------------------------------------------------------- k_raygen
__global__ __launch_bounds__(kBlock) void k_raygen(DevScene S, PathBuf B, const int32_t *pixList, int32_t sampleBase,
                                                   int32_t sppThis, uint32_t seed) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool valid = i < (int64_t)B.npix * sppThis;
  bool alive = false;
  if (valid) {
    int32_t p = (int32_t)(i / sppThis), s = (int32_t)(i % sppThis);
    int32_t pixel = pixList[B.active ? B.active[p] : p];
    int32_t x = pixel % S.width, y = pixel / S.width;
    Rng rng = rngInit(seed, (uint32_t)pixel, (uint32_t)(sampleBase + s), 0u);
    int allowAA = min(1, S.numRays - 1);
    float Sx = getRayScaleX(x + (rngNext(rng) - 0.5f) * allowAA, S.width, S.height);
    float Sy = getRayScaleY(y + (rngNext(rng) - 0.5f) * allowAA, S.width, S.height);
    V3 eye = v3(S.eye), forward = v3(S.forward), right = v3(S.right), up = v3(S.up);
    V3 o = eye, d;
    alive = true;
    if (S.cameraMode == 0) {
      d = forward + Sx * right + Sy * up;
    } else if (S.cameraMode == 1) {  // fisheye, raytracer.cpp:95-103
      Sx *= S.invForwardLength;
      Sy *= S.invForwardLength;
      float r_2 = Sx * Sx + Sy * Sy;
      if (r_2 > 1) alive = false;  // `continue`: the sample neither traces nor counts
      V3 forwardCopy = sqrtf(1 - r_2) * v3(S.normalizedForward);
      d = forwardCopy + Sx * right + Sy * up;
    } else {  // depth of field, raytracer.cpp:144-151
      V3 rayDirection = forward + Sx * right + Sy * up;
      V3 focusPoint = (S.focus / magnitude(rayDirection)) * rayDirection + eye;
      float weight = (float)((double)(rngNext(rng) * 2) * kPi);
      float r = rngNext(rng) - 0.5f;
      o = (r * cosf(weight) * S.lens / S.magRight) * right + (r * sinf(weight) * S.lens / S.magUp) * up + eye;
      d = focusPoint - o;
    }
    B.rng[i] = rng.draws;
    // A live sample starts with L = (0, 0, 0, 0) and a path weight of (1, 1, 1, 1) (raytracer.cpp:209-210): the first bounce's
    // kernels know that and neither read nor expect them, and they write L for every sample they see. Only a sample that
    // never enters the queue (or a scene without a single bounce) needs its L written here.
    if (!alive || S.maxBounces <= 0) B.L[i] = make_float4(0, 0, 0, 0);
    if (alive) {
      // without a lens every camera ray starts at the eye: the first bounce's kernels take it from the scene
      if (S.cameraMode == 2) B.rayO[i] = make_float4(o.x, o.y, o.z, 0);
      B.rayD[i] = make_float4(d.x, d.y, d.z, 0);
    }
  }
  if (S.cameraMode == 1) {  // fisheye drops samples: compact the survivors
    uint32_t q = queueSlot(&B.counts->nActive, alive);
    if (alive) B.qA[q] = (uint32_t)i;
  } else if (valid) {
    B.qA[i] = (uint32_t)i;  // every sample lives: the queue is the identity and k_begin_pass has set its length
  }
}

// ---------------------------------------------------------------------------------------------------- k_extend
struct ExtendSource {
  static constexpr bool kPlanesDone = false;
  static constexpr bool kVisibilitySplit = false;
  static constexpr bool kFiltered = false;
  const PathBuf &B;
  const uint32_t *queue;
  const uint32_t *count;  // length of the queue: nActive, or nNext when the next bounce's queue is traced before the swap
  const float *eye;       // camera rays of a camera without a lens: their common origin (k_raygen has not stored it); else nullptr
  RT_DEV uint32_t total() const { return *count; }
  RT_DEV uint32_t fetch(uint32_t idx, V3 &o, V3 &d, int &mode, float &bulbDist) const {
    uint32_t slot = queue[idx];
    float4 d4 = B.rayD[slot];
    if (eye) {
      o = v3(eye);
    } else {
      float4 o4 = B.rayO[slot];
      o = v3(o4.x, o4.y, o4.z);
    }
    d = v3(d4.x, d4.y, d4.z);
    mode = TRAV_CLOSEST; bulbDist = 0.0f;
    return slot;
  }
  RT_DEV void store(uint32_t slot, HitRec h, bool, V3, V3) const {
    B.hitT[slot] = h.t;
    B.hitPrim[slot] = h.prim;
  }
};

template <bool COUNT>
__global__ RT_TRAV_ATTR void k_extend(DevScene S, PathBuf B, const uint32_t *queue, int32_t bounce) {
  __shared__ uint2 ldsStack[kStackLds * kBlock];

  __shared__ uint32_t ldsPool[(kBlock / 64) * kPoolFields * kPoolSlots];
  TravCounters tc{};
  ExtendSource src{B, queue, &B.counts->nActive, (bounce == 0 && S.cameraMode != 2) ? S.eye : nullptr};
  traverseQueue<ExtendSource, false, COUNT, kStackLds, kBlock>(S, src, &B.counts->ticket[(bounce & 1) * 2].seg[0].next, ldsStack, ldsPool, tc);
  flushCounters<COUNT>(B.counts, tc, 0);
}

// ---------------------------------------------------------------------------------------------------- k_shade
// What k_shade decides about the visibility query of (queue position, light), two bits per query:
enum ShadowCode {
  SHADOW_NONE = 0,    // no query: the path missed, or the position is beyond the queue
  SHADOW_TRACE = 1,   // k_shadow has to trace it
  SHADOW_PLANE = 2,   // one of the scene's planes alone settles it: hidden, contribution already replaced
  SHADOW_ZERO = 3     // the unshadowed contribution is exactly (0, 0, 0): visible or not, it adds the same nothing
};

// The code k_shade left for (queue position pos, light k).
RT_DEV int shadowCodeAt(const PathBuf &B, int k, uint32_t pos) {
  const uint4 w = B.shadowCode[(int64_t)k * B.rowsCap + (pos >> 6)];
  const uint32_t bit = pos & 31u;
  return (int)((((pos & 32u) ? w.y : w.x) >> bit) & 1u) | (int)(((((pos & 32u) ? w.w : w.z) >> bit) & 1u) << 1);
}
// Contributions that are not stored because the code says what they are: a light hidden by a plane contributes
// RGBAColor(0, 0, 0, 1); a contribution of exactly zero is (0, 0, 0, alpha of the light's colour), the scalar factors
// leave alpha alone (src/image/PNG.cpp:32-37) - unless the colour came out of an environment map, whose alpha is the
// texel's. (A stored zero may be -0; added to a sum that starts at +0 and is never -0, both zeros give the same bits.)
RT_DEV bool contributionImplied(const DevLight &l, int code) {
  return code == SHADOW_PLANE || (code == SHADOW_ZERO && !(l.type == RT_LIGHT_ENVIRONMENT && l.texture >= 0));
}
RT_DEV RGBA impliedContribution(const DevLight &l, int code) { return rgba(0, 0, 0, code == SHADOW_PLANE ? 1.0f : l.color[3]); }

#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 5
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_SHADE_WAVES, 8))) void k_shade(DevScene S, PathBuf B, const uint32_t *queue, uint32_t *queueNext,
                                                  int32_t bounce, const int32_t *pixList, int32_t sampleBase, uint32_t seed) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  bool isHit = false, continues = false;
  uint32_t slot = 0;
  // state of a hit that the three parts below share
  Rng rng{};
  V3 nd = v3(0, 0, 0), point = nd, normal = nd, outNormal = nd, biased = nd;
  RGBA beta = rgba(0, 0, 0, 0), objColor = beta;
  int32_t matIdx = 0;
  const uint32_t nActive = B.counts->nActive;
  // ------------------------------------------------------------------ part 1: the hit (or the miss)
  if (i < nActive) {
    slot = queue[i];
    // Loads that depend only on the path slot, then loads that depend only on the object, each group issued together:
    // the kernel is bound by the latency of these scattered reads, so they must not wait for one another.
    const float4 d4 = B.rayD[slot];
    // camera rays of a camera without a lens start at the eye, which k_raygen does not store per ray (bounce is uniform)
    const float4 o4 = (bounce == 0 && S.cameraMode != 2) ? make_float4(S.eye[0], S.eye[1], S.eye[2], 0) : B.rayO[slot];
    int32_t prim = B.hitPrim[slot];
    float t = B.hitT[slot];
    // the path's random stream is a function of its pixel and sample index; only the draw count is stored
    const uint32_t pixPos = slot / (uint32_t)B.spp, sampleInPass = slot - pixPos * (uint32_t)B.spp;
    rng = rngInit(seed, (uint32_t)pixList[B.active ? B.active[pixPos] : (int32_t)pixPos], (uint32_t)sampleBase + sampleInPass, B.rng[slot]);
    const float4 beta4 = bounce == 0 ? make_float4(1, 1, 1, 1) : B.beta[bounce & 1][slot];  // bounce is uniform
    V3 o = v3(o4.x, o4.y, o4.z), d = v3(d4.x, d4.y, d4.z);
    const bool onObject = prim >= 0 && prim < S.numPrims;
    ObjectData od;
    od.a = od.b = od.c = od.color = make_float4(0, 0, 0, 0);
    od.texture = -1; od.material = 0;
    float nn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (S.numPrims > 0) {  // uniform; a lane without an object reads object 0 and ignores it
      const int64_t pc = onObject ? prim : 0;
      const float4 *P = S.primHit + 3 * pc;
      od.a = P[0]; od.b = P[1]; od.c = P[2];
      const float4 *Q = S.primShade + 4 * pc;
      const float4 q0 = Q[0], q1 = Q[1], q2 = Q[2], q3 = Q[3];
      nn[0] = q0.x; nn[1] = q0.y; nn[2] = q0.z; nn[3] = q0.w; nn[4] = q1.x; nn[5] = q1.y; nn[6] = q1.z; nn[7] = q1.w; nn[8] = q2.x;
      od.color = make_float4(q2.y, q2.z, q2.w, q3.x);
      od.material = __float_as_int(q3.y);
      od.texture = __float_as_int(q3.z);
    }
    if (prim < 0) {
      // miss: add every light's emittedLight(direction) (raytracer.cpp:217-223); only environment lights emit
      RGBA L = bounce == 0 ? rgba(0, 0, 0, 0) : rgba(B.L[slot]);
      for (int k = 0; k < S.numLights; ++k) addAssign(L, emittedLight(S, S.lights[k], d));
      B.L[slot] = f4(L);
    } else {
      isHit = true;
      nd = normalized(d);
      point = t * nd + o;  // IntersectionInfo::point as the primitive computed it
      if (prim >= S.numPrims) {
        const DevPlane &pl = S.planes[prim - S.numPrims];
        normal = v3(pl.normal);
        matIdx = pl.material;
        objColor = rgba(pl.color[0], pl.color[1], pl.color[2], pl.color[3]);
      } else {
        float4 a = od.a, b = od.b, c = od.c;
        bool sphere = (__float_as_uint(a.w) | __float_as_uint(b.x) | __float_as_uint(b.y)) == 0u;
        if (sphere) {
          normal = normalized(point - v3(a.x, a.y, a.z));  // Object.cpp:98
        } else {
          // shading normal from the barycentrics, Object.cpp:206-212
          V3 p1 = v3(a.x, a.y, a.z), e1 = v3(b.z, b.w, c.x), e2 = v3(c.y, c.z, c.w);
          V3 q = point - p1;
          float b2 = dot(e1, q);
          float b3 = dot(e2, q);
          float b1 = 1.0f - b3 - b2;
          normal = normalized(v3(nn) * b1 + v3(nn + 3) * b2 + v3(nn + 6) * b3);
        }
        matIdx = od.material;
        objColor = rgba(od.color);
      }
      outNormal = faceForward(-nd, normal);
      biased = point + S.bias * outNormal;  // raytracer.cpp:228
      beta = rgba(beta4);
      // Inputs of the direct-light term L += beta * (getColor(biased) * sum_lights intensity), finished by k_direct once
      // visibility is known.
      RGBA color = getColor(S, prim, biased, od);
      B.shColor[slot] = f4(color);  // the weight k_direct needs is beta[bounce & 1], which this kernel only reads
    }
  }
  // ------------------------------------------------------------------ part 2: the lights, in scene order
  // Each light's unshadowed intensity (src/scene/Object.cpp:12-16, 27-34, 41-46) is evaluated here, so that environment
  // lights draw their hemisphere direction before the BSDF draws, and so is everything about its visibility query that
  // needs no traversal: a contribution of exactly (0, 0, 0) adds the same nothing whether the light is visible or not (the
  // alpha of the sum is 1 from the start, src/scene/raytracer.cpp:199), and a plane that alone hides the light is found
  // with the expressions the traversal engine uses for queries it prepares itself. The loop is reached by every lane of
  // the wavefront: what was decided for 64 queue positions goes out as two 64-bit ballots.
  bool anyTraced = false;
  for (int k = 0; k < S.numLights; ++k) {
    int code = SHADOW_NONE;
    if (isHit) {
      const DevLight &l = S.lights[k];
      RGBA lc = rgba(l.color[0], l.color[1], l.color[2], l.color[3]);
      RGBA c;
      V3 toLight;          // direction of the visibility query, as k_shadow forms it
      float bulbDist = 0.0f;
      if (l.type == RT_LIGHT_DISTANT) {
        c = lc * clipDot(normal, v3(l.v));
        toLight = v3(l.v);  // Object.cpp:7-10
      } else if (l.type == RT_LIGHT_POINT) {
        V3 lightDirection = v3(l.v) - biased;
        float distance = magnitude(lightDirection);
        float invDistance = 1.0f / (distance * distance);
        c = lc * clipDot(normal, lightDirection) * invDistance;
        toLight = lightDirection;  // Object.cpp:18-25
        bulbDist = distance;
      } else {
        V3 wi = sampleHemisphere(normal, rng);
        c = emittedLight(S, l, wi) * clipDot(normal, wi);
        toLight = wi;  // Object.cpp:36-39
      }
      float4 c4 = f4(c);
      if (c4.x == 0.0f && c4.y == 0.0f && c4.z == 0.0f) {
        code = SHADOW_ZERO;
      } else {
        code = SHADOW_TRACE;
        if (S.numPlanes > 0) {
          V3 qnd = normalized(toLight);
          float tPlane; int32_t plane;
          closestPlane(S, biased, qnd, tPlane, plane);
          if (planeSettlesVisibility(l.type == RT_LIGHT_POINT, biased, qnd, bulbDist, tPlane, plane)) {
            code = SHADOW_PLANE;
            c4 = make_float4(0, 0, 0, 1);  // what a hidden light contributes
          }
        }
      }
      if (!contributionImplied(l, code)) B.lightC[(int64_t)k * B.P + slot] = c4;
      if (code == SHADOW_TRACE) {  // what k_shadow reads of a query it traces
        anyTraced = true;
        if (l.type == RT_LIGHT_ENVIRONMENT) B.envWi[(int64_t)l.envSlot * B.P + slot] = make_float4(toLight.x, toLight.y, toLight.z, 0);
      }
    }
    const unsigned long long lo = __ballot(code & 1), hi = __ballot(code & 2);
    // lane 0 holds the wavefront's first queue position; a wavefront wholly beyond the queue has no entry (its row index
    // may lie beyond this light's rows)
    if (__lane_id() == 0 && i < nActive)
      B.shadowCode[(int64_t)k * B.rowsCap + (i >> 6)] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
  }
  if (anyTraced) B.shPoint[slot] = make_float4(biased.x, biased.y, biased.z, 0);
  // ------------------------------------------------------------------ part 3: the next ray
  if (isHit) {
    const DevMaterial &mat = S.materials[matIdx];
    if (mat.type == RT_MAT_METAL) mulAssign(beta, objColor);  // raytracer.cpp:233-234
    if (bounce + 1 < S.maxBounces) {
      // The reference samples the BSDF on the last bounce too but never uses the result.
      float pdf = 0.0f;
      V3 wi;
      float contribution = bsdfSample(mat, -nd, wi, normal, rng, pdf);
      if (!(pdf == 0 || contribution == 0)) {
        continues = true;
        mulAssign(beta, contribution * fabsf(dot(wi, normal)) / pdf);
        bool exiting = dot(wi, outNormal) > 0;
        V3 no = point + outNormal * (exiting ? S.bias : -S.bias);
        B.rayO[slot] = make_float4(no.x, no.y, no.z, 0);
        B.rayD[slot] = make_float4(wi.x, wi.y, wi.z, 0);
      }
    }
    if (bounce + 1 < S.maxBounces) {  // nothing reads them after the last bounce
      B.beta[(bounce + 1) & 1][slot] = f4(beta);
      B.rng[slot] = rng.draws;
    }
  }
  uint32_t qh, qn;
  queueSlots2(&B.counts->nHit, isHit, continues, qh, qn);
  if (isHit && S.numLights == 0) B.qHit[qh] = slot;  // with lights, k_direct finds the hits through the visibility codes
  if (continues) queueNext[qn] = slot;
}

// ---------------------------------------------------------------------------------------------------- k_shadow
// One visibility query per (queue position, light), light-major so that a wavefront traces rays toward the same light.
// k_shade has sorted them out already (ShadowCode): the traversal engine classifies 64 positions at a time, and only
// SHADOW_TRACE queries are loaded and traced here, the others are counted by kind. A light that turns out to be hidden has its stored contribution replaced by RGBAColor(0, 0, 0, 1).
// The reference traces the zero-contribution queries too; they are reported apart.
struct ShadowSource {
  static constexpr bool kPlanesDone = true;  // k_shade has tested them
  static constexpr bool kVisibilitySplit = true;  // store() acts on `occluded` only, and writing the same zero twice is harmless
  static constexpr bool kFiltered = true;    // and has left a code for every (queue position, light)
  const DevScene &S;
  const PathBuf &B;
  const uint32_t *queue;  // the queue k_shade ran over
  uint32_t nActive;       // its length
  uint32_t zeroPoint = 0, zeroAny = 0, planePoint = 0, planeAny = 0, traced = 0;
  RT_DEV uint32_t total() const { return nActive * (uint32_t)S.numLights; }
  RT_DEV int classify(uint32_t idx) const {
    const uint32_t k = idx / nActive, pos = idx - k * nActive;
    return shadowCodeAt(B, (int)k, pos);
  }
  RT_DEV void passedOver(uint32_t idx, int code) {
    const bool point = S.lights[idx / nActive].type == RT_LIGHT_POINT;
    if (code == SHADOW_ZERO) { if (point) ++zeroPoint; else ++zeroAny; }
    else { if (point) ++planePoint; else ++planeAny; }
  }
  RT_DEV uint32_t fetch(uint32_t idx, V3 &o, V3 &d, int &mode, float &bulbDist) {
    const uint32_t k = idx / nActive;
    const uint32_t slot = queue[idx - k * nActive];
    const DevLight &l = S.lights[k];
    ++traced;
    float4 p4 = B.shPoint[slot];
    o = v3(p4.x, p4.y, p4.z);
    mode = TRAV_ANY; bulbDist = 0.0f;
    if (l.type == RT_LIGHT_DISTANT) {
      d = v3(l.v);  // Object.cpp:7-10
    } else if (l.type == RT_LIGHT_POINT) {
      d = v3(l.v) - o;  // Object.cpp:18-25: closest hit along the un-normalised vector to the bulb
      bulbDist = magnitude(d);
      mode = TRAV_POINT_LIGHT;
    } else {
      float4 w4 = B.envWi[(int64_t)l.envSlot * B.P + slot];  // Object.cpp:36-39, direction drawn in k_shade
      d = v3(w4.x, w4.y, w4.z);
    }
    return idx;
  }
  RT_DEV void store(uint32_t idx, HitRec, bool occluded, V3, V3) const {
    if (!occluded) return;
    const uint32_t k = idx / nActive;
    const uint32_t slot = queue[idx - k * nActive];
    B.lightC[(int64_t)k * B.P + slot] = make_float4(0, 0, 0, 1);
  }
};

RT_DEV void flushShadowKinds(Counts *c, const ShadowSource &src) {
  uint32_t sp = src.zeroPoint, sa = src.zeroAny, pp = src.planePoint, pa = src.planeAny;
  for (int off = 32; off > 0; off >>= 1) {
    sp += __shfl_down(sp, off); sa += __shfl_down(sa, off); pp += __shfl_down(pp, off); pa += __shfl_down(pa, off);
  }
  if (__lane_id() == 0) {  // one wavefront = up to four atomics per launch, on a line of their own
    if (sp) atomicAdd(&c->skipPoint, (unsigned long long)sp);
    if (sa) atomicAdd(&c->skipAny, (unsigned long long)sa);
    if (pp) atomicAdd(&c->planePoint, (unsigned long long)pp);
    if (pa) atomicAdd(&c->planeAny, (unsigned long long)pa);
  }
}

template <bool COUNT>
__global__ RT_TRAV_ATTR void k_shadow(DevScene S, PathBuf B, const uint32_t *queue) {
  __shared__ uint2 ldsStack[kStackLds * kBlock];
  __shared__ uint32_t ldsPool[(kBlock / 64) * kPoolFields * kPoolSlots];
  TravCounters tc{};
  ShadowSource src{S, B, queue, B.counts->nActive};
  traverseQueue<ShadowSource, true, COUNT, kStackLds, kBlock>(S, src, &B.counts->ticket[1].seg[0].next, ldsStack, ldsPool, tc);
  if (COUNT) tc.planes += (src.traced + src.planePoint + src.planeAny) * (uint32_t)S.numPlanes;  // tested by k_shade
  flushCounters<COUNT>(B.counts, tc, 1);
  flushShadowKinds(B.counts, src);
}

// k_shadow of one bounce and k_extend of the next in ONE launch. Both need nothing but what k_shade has left (the
// visibility queries of this bounce; the rays and the queue of the next), and they touch disjoint data. A persistent
// traversal kernel ends with its wavefronts running dry one after the other; here a wavefront that is out of visibility
// queries goes straight on to the next bounce's rays, so that one tail per bounce disappears. Used for short launches
// (a part of a frame on one of several GPUs), where the tails are a tenth of the time.
template <bool COUNT>
__global__ RT_TRAV_ATTR void k_shadow_then_extend(DevScene S, PathBuf B, const uint32_t *queue, const uint32_t *queueNext, int32_t bounce) {
  __shared__ uint2 ldsStack[kStackLds * kBlock];
  __shared__ uint32_t ldsPool[(kBlock / 64) * kPoolFields * kPoolSlots];
  {
    TravCounters tc{};
    ShadowSource src{S, B, queue, B.counts->nActive};
    traverseQueue<ShadowSource, true, COUNT, kStackLds, kBlock>(S, src, &B.counts->ticket[1].seg[0].next, ldsStack, ldsPool, tc);
    if (COUNT) tc.planes += (src.traced + src.planePoint + src.planeAny) * (uint32_t)S.numPlanes;  // tested by k_shade
    flushCounters<COUNT>(B.counts, tc, 1);
    flushShadowKinds(B.counts, src);
  }
  {
    TravCounters tc{};
    ExtendSource src{B, queueNext, &B.counts->nNext, nullptr};
    // the next bounce's ticket counters: those of this bounce's rays have not been reset yet
    traverseQueue<ExtendSource, false, COUNT, kStackLds, kBlock>(S, src, &B.counts->ticket[((bounce + 1) & 1) * 2].seg[0].next, ldsStack, ldsPool, tc);
    flushCounters<COUNT>(B.counts, tc, 0);
  }
}

// L += beta * (getColor(point) * sum over lights, in scene order) — src/scene/raytracer.cpp:198-206, 231
// Runs over the queue k_shade ran over: the codes it left say which positions are hits (a hit has a code for every
// light) and which contributions were not stored because the code implies them. A scene without lights has no codes;
// its hits (the sum is RGBAColor(0, 0, 0, 1) then, which still raises alpha) come from the hit queue.
__global__ __launch_bounds__(kBlock) void k_direct(DevScene S, PathBuf B, const uint32_t *queue, int32_t bounce) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  uint32_t slot;
  RGBA Lsum = rgba(0, 0, 0, 1);  // `RGBAColor L;`
  if (S.numLights == 0) {
    if (i >= B.counts->nHit) return;
    slot = B.qHit[i];
  } else {
    if (i >= B.counts->nActive) return;
    int code = shadowCodeAt(B, 0, i);
    if (code == SHADOW_NONE) return;  // a miss
    slot = queue[i];
    for (int k = 0; k < S.numLights; ++k) {
      if (k > 0) code = shadowCodeAt(B, k, i);
      const DevLight &l = S.lights[k];
      addAssign(Lsum, contributionImplied(l, code) ? impliedContribution(l, code) : rgba(B.lightC[(int64_t)k * B.P + slot]));
    }
  }
  RGBA L = bounce == 0 ? rgba(0, 0, 0, 0) : rgba(B.L[slot]);  // a path's L is first written at its first bounce
  const RGBA weight = bounce == 0 ? rgba(1, 1, 1, 1) : rgba(B.beta[bounce & 1][slot]);  // the weight the path carried into the bounce
  addAssign(L, weight * (rgba(B.shColor[slot]) * Lsum));
  B.L[slot] = f4(L);
}

// ---------------------------------------------------------------------------------------------------- bookkeeping
__global__ void k_advance(PathBuf B, int32_t numPointLights, int32_t numAnyLights) {
  Counts *c = B.counts;
  c->raysClosest += (unsigned long long)c->nActive + (unsigned long long)c->nHit * numPointLights;
  c->raysAny += (unsigned long long)c->nHit * numAnyLights;
  c->raysExtend += c->nActive;
  c->raysShadow += (unsigned long long)c->nHit * (numPointLights + numAnyLights);
  c->nActive = c->nNext;
  c->nNext = 0;
  c->nHit = 0;
  for (int g = 0; g < kQueueSegments; ++g) { c->ticket[0].seg[g].next = 0; c->ticket[1].seg[g].next = 0; c->ticket[2].seg[g].next = 0; }
}
__global__ void k_begin_pass(PathBuf B, uint32_t nActive) {
  Counts *c = B.counts;
  c->nActive = nActive; c->nNext = 0; c->nHit = 0;
  for (int g = 0; g < kQueueSegments; ++g) { c->ticket[0].seg[g].next = 0; c->ticket[1].seg[g].next = 0; c->ticket[2].seg[g].next = 0; }
}
__global__ void k_count_samples(PathBuf B, unsigned long long n) { B.counts->samples += n; }  // rejected fisheye samples count too

// Per-pixel state of the sample loop that outlives a pass (and, for progressive rendering, a call).
struct PixelAcc {
  float4 *sum;       // running colour sum of the valid samples (raytracer.cpp:48-51)
  int32_t *hits;     // number of valid samples
  int32_t *taken;    // number of samples taken (the loop counter `i` of raytracer.cpp:41)
  float *lumSum, *lumSqSum;  // adaptive sampling: sum and sum of squares of the valid samples' luminance
};

// Per pixel of this pass: fold the pass's samples, in sample order, into the running sum (raytracer.cpp:48-51). With
// adaptive sampling (threshold > 0) the pixel then decides whether it goes on (see RtRenderParams): pixels that do are
// appended to `nextActive`.
__global__ __launch_bounds__(kBlock) void k_accumulate(PathBuf B, PixelAcc A, int32_t nPix, int32_t sppThis, float threshold, int32_t minSamples,
                                                       int32_t maxSamples, int32_t *nextActive, uint32_t *nextCount) {
  // A pixel's samples are neighbours in memory (slot = pixel position * spp + sample) and a lane sums ONE pixel in sample
  // order, so a wavefront reads its 64 pixels' samples through LDS: chunks of kChunk samples per pixel are loaded with
  // consecutive lanes on consecutive samples (128-byte runs) and read back transposed, one pixel per lane.
  constexpr int kChunk = 8, kRow = 65;  // 65: rows of 64 float4 plus one of padding, so the transposed writes spread over the banks
  __shared__ float4 tile[kBlock / 64][kChunk * kRow];
  const int32_t p = blockIdx.x * kBlock + threadIdx.x;
  const int32_t lane = (int32_t)__lane_id(), wave = threadIdx.x >> 6;
  const int32_t waveBase = p - lane;
  const bool valid = p < nPix;
  bool goesOn = false;
  int32_t fp = 0;
  RGBA avg = rgba(0, 0, 0, 0);
  int hits = 0;
  float s1 = 0.0f, s2 = 0.0f;
  if (valid) {
    fp = B.active ? B.active[p] : p;
    avg = rgba(A.sum[fp]);
    hits = A.hits[fp];
    if (threshold > 0) { s1 = A.lumSum[fp]; s2 = A.lumSqSum[fp]; }
  }
  for (int s0 = 0; s0 < sppThis; s0 += kChunk) {
    const int ch = min(kChunk, sppThis - s0);
    for (int e = lane; e < 64 * ch; e += 64) {
      const int px = e / ch, sm = e - px * ch;
      if (waveBase + px < nPix) tile[wave][sm * kRow + px] = B.L[(int64_t)(waveBase + px) * sppThis + s0 + sm];
    }
    __syncthreads();
    if (valid) {
      for (int sm = 0; sm < ch; ++sm) {
        RGBA color = rgba(tile[wave][sm * kRow + lane]);
        if (hasNaN(color) == false && color.a != 0) {
          addAssign(avg, color);
          ++hits;
          if (threshold > 0) {
            float y = (0.2126f * color.r + 0.7152f * color.g) + 0.0722f * color.b;
            s1 += y;
            s2 += y * y;
          }
        }
      }
    }
    __syncthreads();
  }
  if (valid) {
    const int32_t taken = A.taken[fp] + sppThis;
    A.sum[fp] = f4(avg);
    A.hits[fp] = hits;
    A.taken[fp] = taken;
    if (threshold > 0) {
      A.lumSum[fp] = s1; A.lumSqSum[fp] = s2;
      goesOn = taken < maxSamples;
      if (goesOn && taken >= minSamples) {
        if (hits == 0) {
          goesOn = false;
        } else if (hits >= 2) {
          float mean = s1 / (float)hits;
          float var = rmax(0.0f, s2 - s1 * mean) / (float)(hits - 1);
          float se = sqrtf(var / (float)hits);
          goesOn = !(se <= threshold * (mean + 0.05f));
        }
      }
    }
  }
  if (threshold > 0) {
    uint32_t q = queueSlot(nextCount, goesOn);
    if (goesOn) nextActive[q] = fp;
  }
}

// raytracer.cpp:54-59 then Scene::expose (:303-312)
__global__ __launch_bounds__(kBlock) void k_resolve(DevScene S, int32_t npix, const int32_t *pixList, PixelAcc A, float4 *fb, float4 *packed,
                                                    int32_t applyExposure) {
  int32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= npix) return;
  RGBA avg = rgba(A.sum[p]);
  int hits = A.hits[p];
  float invNumRays = (float)(1.0 / (double)max(1, A.taken[p]));  // = 1.0 / numRays once the whole loop has run (raytracer.cpp:39)
  if (hits != 0) {
    mulAssign(avg, 1.0f / hits);
    avg.a = hits * invNumRays;
  }
  // clipColor, src/image/PNG.cpp:99-106
  RGBA c = rgba(rmax(0.0f, rmin(avg.r, 1.0f)), rmax(0.0f, rmin(avg.g, 1.0f)), rmax(0.0f, rmin(avg.b, 1.0f)),
                rmax(0.0f, rmin(avg.a, 1.0f)));
  if (applyExposure) {  // exponentialExposure, src/image/PNG.cpp:20-22 (float exp, double subtraction)
    c.r = (float)(1.0 - (double)expf(-1 * c.r * S.exposure));
    c.g = (float)(1.0 - (double)expf(-1 * c.g * S.exposure));
    c.b = (float)(1.0 - (double)expf(-1 * c.b * S.exposure));
  }
  fb[pixList[p]] = f4(c);
  packed[p] = f4(c);
}

// PNG::saveToFile's conversion, src/image/PNG.cpp:7-12, 149-157
RT_DEV float linearToGamma(float channel) {
  if ((double)channel < 0.0031308) return (float)(12.92 * (double)channel);
  return (float)(1.055 * pow((double)channel, 1 / 2.4) - 0.055);
}
// fb[idx[i]] += src[i], component by component: tile partitions own disjoint pixels and a frame holds exact zeros
// outside its own partition, so the sum is the other partition's value, bit for bit
__global__ __launch_bounds__(kBlock) void k_scatter_add(float4 *fb, const float4 *src, const int32_t *idx, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float4 b = src[i];
  float4 *d = fb + idx[i];
  float4 a = *d;
  *d = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

__global__ __launch_bounds__(kBlock) void k_srgb8(const float4 *fb, uchar4 *out, int32_t n) {
  int32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float4 c = fb[i];
  uchar4 o;
  o.x = (unsigned char)roundf(linearToGamma(c.x) * 255);
  o.y = (unsigned char)roundf(linearToGamma(c.y) * 255);
  o.z = (unsigned char)roundf(linearToGamma(c.z) * 255);
  o.w = (unsigned char)roundf(c.w * 255);
  out[i] = o;
}

// dst[i] = src[order[i]] for records of `dwords` 32-bit words: puts per-object arrays into the object order of a
// device-built BVH
__global__ __launch_bounds__(kBlock) void k_gather_records(const uint32_t *src, uint32_t *dst, const uint32_t *order, int64_t n, int32_t dwords) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n * dwords) return;
  int64_t rec = i / dwords;
  int32_t w = (int32_t)(i - rec * dwords);
  dst[i] = src[(int64_t)order[rec] * dwords + w];
}

// ---------------------------------------------------------------------------------------------------- trace-only kernels
RT_DEV int32_t reportedId(const DevScene &S, int32_t prim) {
  return prim < 0 ? -1 : (prim < S.numPrims ? __float_as_int(S.primShade[4 * (int64_t)prim + 3].w) : prim);
}

struct PrimarySource {
  static constexpr bool kPlanesDone = false;
  static constexpr bool kVisibilitySplit = false;
  static constexpr bool kFiltered = false;
  const DevScene &S;
  int32_t x0, y0, w, h;
  int32_t *outPrim;
  float *outT;
  // 8x8 pixel blocks per 64 consecutive queries: neighbouring lanes walk the same part of the tree
  RT_DEV uint32_t total() const { return (uint32_t)(((w + 7) / 8) * ((h + 7) / 8)) * 64u; }
  RT_DEV uint32_t fetch(uint32_t idx, V3 &o, V3 &d, int &mode, float &bulbDist) const {
    int32_t tilesX = (w + 7) / 8;
    int32_t tile = (int32_t)(idx / 64u), l = (int32_t)(idx & 63u);
    int32_t col = (tile % tilesX) * 8 + (l & 7), row = (tile / tilesX) * 8 + (l >> 3);
    mode = TRAV_CLOSEST; bulbDist = 0.0f;
    o = v3(S.eye);
    if (col >= w || row >= h) {
      // padding lane of a partial block: a NaN origin fails every box and plane comparison, so the query ends at once
      o = v3(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
      d = v3(1, 0, 0);
      return 0xffffffffu;
    }
    float Sx = getRayScaleX((float)(x0 + col), S.width, S.height);
    float Sy = getRayScaleY((float)(y0 + row), S.width, S.height);
    d = v3(S.forward) + Sx * v3(S.right) + Sy * v3(S.up);
    return (uint32_t)(row * w + col);
  }
  RT_DEV void store(uint32_t j, HitRec hr, bool, V3, V3) const {
    if (j == 0xffffffffu) return;
    outPrim[j] = reportedId(S, hr.prim);
    outT[j] = hr.t;
  }
};

__global__ __launch_bounds__(kBlock) void k_trace_primary(DevScene S, int32_t x0, int32_t y0, int32_t w, int32_t h,
                                                          int32_t *outPrim, float *outT, uint32_t *ticket) {
  __shared__ uint2 ldsStack[kStackLds * kBlock];
  __shared__ uint32_t ldsPool[(kBlock / 64) * kPoolFields * kPoolSlots];
  TravCounters tc{};
  PrimarySource src{S, x0, y0, w, h, outPrim, outT};
  traverseQueue<PrimarySource, false, false, kStackLds, kBlock>(S, src, ticket, ldsStack, ldsPool, tc);
}

template <bool ANY>
struct RaysSource {
  static constexpr bool kPlanesDone = false;
  static constexpr bool kVisibilitySplit = false;
  static constexpr bool kFiltered = false;
  const DevScene &S;
  int64_t n;
  const float *origins, *directions;
  int32_t *outPrim;
  float *outT, *outPoint, *outNormal;
  uint8_t *outOccluded;
  RT_DEV uint32_t total() const { return (uint32_t)n; }
  RT_DEV uint32_t fetch(uint32_t j, V3 &o, V3 &d, int &mode, float &bulbDist) const {
    o = v3(origins + 3 * (int64_t)j); d = v3(directions + 3 * (int64_t)j);
    mode = ANY ? TRAV_ANY : TRAV_CLOSEST; bulbDist = 0.0f;
    return j;
  }
  RT_DEV void store(uint32_t j, HitRec hr, bool occluded, V3 o, V3 nd) const {
    if (ANY) { outOccluded[j] = occluded ? 1 : 0; return; }
    if (outPrim) outPrim[j] = reportedId(S, hr.prim);
    if (outT) outT[j] = hr.t;
    V3 point = v3(0, 0, 0), normal = v3(0, 0, 0);
    if (hr.prim >= 0 && (outPoint || outNormal)) {
      point = hr.t * nd + o;
      if (hr.prim >= S.numPrims) {
        normal = v3(S.planes[hr.prim - S.numPrims].normal);
      } else {
        const float4 *P = S.primHit + 3 * (int64_t)hr.prim;
        float4 a = P[0], b = P[1], c = P[2];
        if ((__float_as_uint(a.w) | __float_as_uint(b.x) | __float_as_uint(b.y)) == 0u) {
          normal = normalized(point - v3(a.x, a.y, a.z));
        } else {
          V3 p1 = v3(a.x, a.y, a.z), e1 = v3(b.z, b.w, c.x), e2 = v3(c.y, c.z, c.w);
          V3 q = point - p1;
          float b2 = dot(e1, q), b3 = dot(e2, q);
          float b1 = 1.0f - b3 - b2;
          const float4 *Q = S.primShade + 4 * (int64_t)hr.prim;
          const float4 q0 = Q[0], q1 = Q[1], q2 = Q[2];
          normal = normalized(v3(q0.x, q0.y, q0.z) * b1 + v3(q0.w, q1.x, q1.y) * b2 + v3(q1.z, q1.w, q2.x) * b3);
        }
      }
    }
    if (outPoint) { outPoint[3 * (int64_t)j] = point.x; outPoint[3 * (int64_t)j + 1] = point.y; outPoint[3 * (int64_t)j + 2] = point.z; }
    if (outNormal) { outNormal[3 * (int64_t)j] = normal.x; outNormal[3 * (int64_t)j + 1] = normal.y; outNormal[3 * (int64_t)j + 2] = normal.z; }
  }
};

template <bool ANY>
__global__ __launch_bounds__(kBlock) void k_trace_rays(DevScene S, int64_t n, const float *origins, const float *directions,
                                                       int32_t *outPrim, float *outT, float *outPoint, float *outNormal,
                                                       uint8_t *outOccluded, uint32_t *ticket) {
  __shared__ uint2 ldsStack[kStackLds * kBlock];
  __shared__ uint32_t ldsPool[(kBlock / 64) * kPoolFields * kPoolSlots];
  TravCounters tc{};
  RaysSource<ANY> src{S, n, origins, directions, outPrim, outT, outPoint, outNormal, outOccluded};
  traverseQueue<RaysSource<ANY>, ANY, false, kStackLds, kBlock>(S, src, ticket, ldsStack, ldsPool, tc);
}

// ---------------------------------------------------------------------------------------------------- host side
thread_local std::string g_error;

int failMsg(int code, const std::string &msg) { g_error = msg; return code; }

// Called inside a catch (...) of an entry point: no C++ exception leaves the library. Host memory that ran out becomes
// RT_ERR_NOMEM, anything else RT_ERR_INVALID with what() as the message.
int caught() noexcept {
  try {
    throw;
  } catch (const std::bad_alloc &) {
    try { g_error = "out of host memory"; } catch (...) {}
    return RT_ERR_NOMEM;
  } catch (const std::exception &e) {
    try { g_error = std::string("unexpected failure: ") + e.what(); } catch (...) {}
    return RT_ERR_INVALID;
  } catch (...) {
    try { g_error = "unexpected failure"; } catch (...) {}
    return RT_ERR_INVALID;
  }
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      g_error = std::string(#expr) + ": " + hipGetErrorString(e_);                                      \
      return RT_ERR_DEVICE;                                                                             \
    }                                                                                                   \
  } while (0)

struct EventPair {
  hipEvent_t a, b;
  int category;
};
enum { CAT_RAYGEN, CAT_EXTEND, CAT_SHADE, CAT_SHADOW, CAT_DIRECT, CAT_ACCUM, CAT_RESOLVE, CAT_TOTAL, CAT_COUNT };

}  // namespace

struct PeerSlot {  // destination-side state of a gather from one other scene
  int key[6] = {-1, -1, -1, -1, -1, -1};
  int32_t *idx = nullptr;
  float4 *staging = nullptr;
  int64_t cap = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
};

struct RtScene {
  int device = 0;
  hipStream_t stream = nullptr;
  DevScene dev{};
  std::vector<void *> allocs;
  int32_t numPointLights = 0, numAnyLights = 0, numEnvLights = 0;
  int numCUs = 256;
  int blocksPerCU = 4;  // resident blocks of a traversal kernel per CU (occupancy query at scene creation)
  size_t extraLds = 0;  // RT_TRAV_EXTRA_LDS: unused dynamic LDS per traversal block, to lower occupancy in experiments
  // render state (allocated on first use, grown when needed)
  PathBuf buf{};
  void *arena = nullptr;
  size_t arenaBytes = 0;
  int64_t arenaP = 0;
  int32_t *pixList = nullptr;
  // per-pixel state of the sample loop (kept between calls for progressive rendering) and the active-pixel lists of
  // adaptive sampling: ONE allocation, carved in ensureRenderBuffers
  void *pixState = nullptr;
  PixelAcc acc{};
  int32_t *activeA = nullptr, *activeB = nullptr;
  uint32_t *activeCount = nullptr;  // two counters, 128 bytes apart
  float4 *fb = nullptr;
  float4 *packed = nullptr;  // this partition's resolved pixels in pixel-list order: what a gather moves to another GPU
  int64_t fbPixels = 0, pixCap = 0;
  // the pixel list on the device belongs to this partition (a repeated frame does not rebuild and re-upload it)
  int pixKey[6] = {-1, -1, -1, -1, -1, -1};
  int64_t pixCount = 0;
  bool accValid = false;  // the per-pixel sums hold the samples of a finished rt_render on this partition (what keepAccumulator adds to)
  Counts *counts = nullptr;
  // staging of the trace-only and conversion entry points: grow-only, so a repeated call of the same size allocates nothing
  void *work = nullptr;
  size_t workBytes = 0;
  void *pinned = nullptr;  // page-locked host staging for results that go to pageable host memory: grow-only as well
  size_t pinnedBytes = 0;
  hipEvent_t workEvA = nullptr, workEvB = nullptr;
  float lastKernelMs = 0.0f;
  float bvhBuildMs = 0.0f;  // device BVH build of rt_scene_create (0: the scene came with a host-built tree)
  std::vector<EventPair> events;
  size_t eventsUsed = 0;
  std::vector<PeerSlot> peers;
  std::vector<int> peerEnabled;
};

namespace {

template <typename T>
int upload(RtScene *s, const T *host, size_t n, const T **dev) {
  *dev = nullptr;
  if (n == 0) return RT_OK;
  void *p = nullptr;
  HIP_TRY(hipMalloc(&p, n * sizeof(T)));
  s->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
  *dev = static_cast<const T *>(p);
  return RT_OK;
}

// Material -> lobes, src/scene/Material.cpp:4-36 with the constants of src/bsdf/BDF.h:108-115 and
// src/bsdf/microfacets.cpp:49-51 evaluated on the host exactly as the constructors do.
bool makeMaterial(const RtMaterial &m, DevMaterial &out) {
  std::memset(&out, 0, sizeof out);
  out.type = m.type;
  auto microfacet = [&](int fresnel, float fI, float fT, float fK) {
    DevLobe l{};
    l.kind = LOBE_MICROFACET; l.Kr = m.Kr;
    l.alpha = std::sqrt(std::min(1e-4f, m.roughness));
    l.fresnel = fresnel; l.fEtaI = fI; l.fEtaT = fT; l.fK = fK;
    return l;
  };
  switch (m.type) {
    case RT_MAT_DEFAULT: {
      DevLobe l{}; l.kind = LOBE_LAMBERT; l.Kr = 1.0f;
      out.lobes[0] = l; out.numLobes = 1;
      return true;
    }
    case RT_MAT_METAL:
      out.lobes[0] = microfacet(FRESNEL_CONDUCTOR, 1.0f, m.eta, m.Ka); out.numLobes = 1;
      return true;
    case RT_MAT_MIRROR: {
      DevLobe l{}; l.kind = LOBE_SPECULAR_REFLECTION; l.Kr = m.Kr; l.fresnel = FRESNEL_DIELECTRIC; l.fEtaI = 1.0f; l.fEtaT = m.eta;
      out.lobes[0] = l; out.numLobes = 1;
      return true;
    }
    case RT_MAT_PLASTIC: {
      DevLobe o{}; o.kind = LOBE_OREN_NAYAR; o.Kr = m.Kd;
      float sigma = 20;
      sigma *= M_PI / 180;
      float sigma2 = sigma * sigma;
      o.A = 1.0f - 0.5 * sigma2 / (sigma2 + 0.33);
      o.B = 0.45 * sigma2 / (sigma2 + 0.09);
      out.lobes[0] = o;
      out.lobes[1] = microfacet(FRESNEL_DIELECTRIC, 1.0f, m.eta, 0.0f);
      out.numLobes = 2;
      return true;
    }
    case RT_MAT_GLASS: {
      DevLobe l{}; l.kind = LOBE_FRESNEL_SPECULAR; l.Kr = m.Kr; l.Kt = m.Kt; l.etaI = 1.0f; l.etaT = m.eta;
      out.lobes[0] = l; out.numLobes = 1;
      return true;
    }
    default: return false;
  }
}

void setView(RtScene *s, const RtCamera &cam, const RtOptions &o) {
  DevScene &d = s->dev;
  std::memcpy(d.eye, cam.eye, 12); std::memcpy(d.forward, cam.forward, 12);
  std::memcpy(d.right, cam.right, 12); std::memcpy(d.up, cam.up, 12);
  d.width = o.width; d.height = o.height; d.bias = o.bias; d.exposure = o.exposure;
  d.maxBounces = o.maxBounces; d.numRays = o.numRays; d.focus = o.focus; d.lens = o.lens;
  d.cameraMode = o.fisheye ? 1 : (o.focus > 0 ? 2 : 0);  // src/scene/raytracer.cpp:290-299
  auto mag = [](const float *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
  float fl = mag(cam.forward);
  d.invForwardLength = (float)(1.0 / (double)fl);  // raytracer.cpp:75
  float inv = 1.0f / fl;
  d.normalizedForward[0] = cam.forward[0] * inv; d.normalizedForward[1] = cam.forward[1] * inv; d.normalizedForward[2] = cam.forward[2] * inv;
  d.magRight = mag(cam.right); d.magUp = mag(cam.up);
}

int validateDesc(const RtSceneDesc *d) {
  if (!d) return failMsg(RT_ERR_INVALID, "null scene description");
  if (d->numPrims < 0 || d->numNodes < 0 || d->numPlanes < 0 || d->numLights < 0 || d->numMaterials < 0 || d->numTextures < 0)
    return failMsg(RT_ERR_INVALID, "negative count in scene description");
  if (d->numPrims > 0 && !d->nodes && (d->numNodes != 0 || !d->primAabbMin || !d->primAabbMax))
    return failMsg(RT_ERR_INVALID, "a scene without BVH nodes needs primAabbMin / primAabbMax (and numNodes == 0) for the device build");
  if (d->numPrims > 0 && !d->nodes && d->numPrims > (1 << 24))
    return failMsg(RT_ERR_INVALID, "the device BVH build takes at most 2^24 objects: build the tree on the host");
  if (d->numPrims > 0 && ((d->nodes && d->numNodes <= 0) || !d->primKind || !d->primHit || !d->primNormals || !d->primColor ||
                          !d->primMaterial || !d->primTexture || !d->primOrigIndex))
    return failMsg(RT_ERR_INVALID, "missing array in scene description");
  if (d->numPrims == 0 && d->numNodes != 0) return failMsg(RT_ERR_INVALID, "BVH nodes without objects");
  // The kernels index with these values: check every one on the host, before anything reaches the GPU.
  for (int32_t i = 0; i < d->numNodes; ++i) {
    const RtBvhNode &n = d->nodes[i];
    if (n.numObjects > 0) {
      if (n.rightOrStart < 0 || (int64_t)n.rightOrStart + n.numObjects > d->numPrims)
        return failMsg(RT_ERR_INVALID, "BVH leaf " + std::to_string(i) + " points outside the object array");
    } else {
      if (i + 1 >= d->numNodes || n.rightOrStart <= i + 1 || n.rightOrStart >= d->numNodes)
        return failMsg(RT_ERR_INVALID, "BVH inner node " + std::to_string(i) + " has a child outside the node array");
    }
  }
  bool texturedTriangle = false;
  for (int32_t i = 0; i < d->numPrims; ++i) {
    if (d->primKind[i] != RT_PRIM_TRIANGLE && d->primKind[i] != RT_PRIM_SPHERE) return failMsg(RT_ERR_INVALID, "bad primKind");
    if (d->primMaterial[i] < 0 || d->primMaterial[i] >= d->numMaterials) return failMsg(RT_ERR_INVALID, "object material index out of range");
    if (d->primTexture[i] >= d->numTextures) return failMsg(RT_ERR_INVALID, "object texture index out of range");
    texturedTriangle |= d->primKind[i] == RT_PRIM_TRIANGLE && d->primTexture[i] >= 0;
  }
  if (texturedTriangle && !d->primTexCoords) return failMsg(RT_ERR_INVALID, "textured triangles need primTexCoords");
  for (int32_t i = 0; i < d->numPlanes; ++i) {
    if (d->planes[i].material < 0 || d->planes[i].material >= d->numMaterials) return failMsg(RT_ERR_INVALID, "plane material index out of range");
    if (d->planes[i].texture >= d->numTextures) return failMsg(RT_ERR_INVALID, "plane texture index out of range");
  }
  for (int32_t i = 0; i < d->numLights; ++i) {
    if (d->lights[i].type < RT_LIGHT_DISTANT || d->lights[i].type > RT_LIGHT_ENVIRONMENT) return failMsg(RT_ERR_INVALID, "bad light type");
    if (d->lights[i].texture >= d->numTextures) return failMsg(RT_ERR_INVALID, "light texture index out of range");
  }
  for (int32_t i = 0; i < d->numTextures; ++i)
    if (d->textures[i].width < 0 || d->textures[i].height < 0 || (d->textures[i].width > 0 && d->textures[i].height > 0 && !d->textures[i].rgba))
      return failMsg(RT_ERR_INVALID, "bad texture");
  if (d->options.width <= 0 || d->options.height <= 0) return failMsg(RT_ERR_INVALID, "bad image size");
  return RT_OK;
}

int buildScene(RtScene *s, const RtSceneDesc *d) {
  DevScene &dev = s->dev;
  // ---- BVH: pre-order 32-byte nodes -> one 64-byte record per inner node
  std::vector<int32_t> innerOrdinal(d->numNodes, -1);
  int32_t numInner = 0;
  for (int32_t i = 0; i < d->numNodes; ++i)
    if (d->nodes[i].numObjects <= 0) innerOrdinal[i] = numInner++;
  std::vector<BigLeaf> bigLeaves;
  auto childRef = [&](int32_t node) -> int32_t {
    const RtBvhNode &n = d->nodes[node];
    if (n.numObjects <= 0) return innerOrdinal[node];
    if (n.numObjects < 128 && n.rightOrStart < (1 << 24)) return (int32_t)(0x80000000u | ((uint32_t)n.numObjects << 24) | (uint32_t)n.rightOrStart);
    bigLeaves.push_back({n.rightOrStart, n.numObjects});
    return (int32_t)(0x80000000u | (uint32_t)(bigLeaves.size() - 1));
  };
  std::vector<float4> nodes((size_t)numInner * 4);
  for (int32_t i = 0; i < d->numNodes; ++i) {
    if (innerOrdinal[i] < 0) continue;
    const RtBvhNode &l = d->nodes[i + 1];
    const RtBvhNode &r = d->nodes[d->nodes[i].rightOrStart];
    float4 *N = &nodes[(size_t)innerOrdinal[i] * 4];
    N[0] = make_float4(l.aabbMin[0], l.aabbMin[1], l.aabbMin[2], l.aabbMax[0]);
    N[1] = make_float4(l.aabbMax[1], l.aabbMax[2], r.aabbMin[0], r.aabbMin[1]);
    N[2] = make_float4(r.aabbMin[2], r.aabbMax[0], r.aabbMax[1], r.aabbMax[2]);
    int32_t refs[4] = {childRef(i + 1), childRef(d->nodes[i].rightOrStart), 0, 0};
    std::memcpy(&N[3], refs, 16);
  }
  if (bigLeaves.size() >= (1u << 24)) return failMsg(RT_ERR_INVALID, "too many oversized BVH leaves");
  if (numInner >= kMaxInnerNodes) return failMsg(RT_ERR_INVALID, "BVH has more inner nodes than 32-bit record offsets address");
  if (d->numPrims >= kMaxObjects) return failMsg(RT_ERR_INVALID, "more objects than 32-bit record offsets address");
  {
    // The traversal stack has the reference's 64 entries and, unlike the reference, no way to write past them: refuse
    // trees deeper than that before they reach the GPU.
    int maxDepth = 0;
    std::vector<std::pair<int32_t, int>> walk;
    if (d->numNodes > 0) walk.push_back({0, 1});
    while (!walk.empty()) {
      auto [node, depth] = walk.back();
      walk.pop_back();
      maxDepth = std::max(maxDepth, depth);
      if (d->nodes[node].numObjects <= 0) { walk.push_back({node + 1, depth + 1}); walk.push_back({d->nodes[node].rightOrStart, depth + 1}); }
    }
    if (maxDepth > kMaxTreeDepth) return failMsg(RT_ERR_INVALID, "BVH deeper than the 64-entry traversal stack (depth " + std::to_string(maxDepth) + ")");
  }
  const bool buildOnDevice = d->numPrims > 0 && !d->nodes;  // no host-built tree: a linear BVH is built below, on the GPU
  if (d->numNodes > 0) {
    dev.rootRef = childRef(0);
    std::memcpy(dev.rootMin, d->nodes[0].aabbMin, 12);
    std::memcpy(dev.rootMax, d->nodes[0].aabbMax, 12);
  } else {
    dev.rootRef = kNoBvh;
  }
  if (bigLeaves.empty()) bigLeaves.push_back({0, 0});
  // ---- objects
  size_t n = (size_t)d->numPrims;
  std::vector<float4> primHit(n * 3);
  int hasSpheres = 0;
  for (size_t i = 0; i < n; ++i) {
    const float *h = d->primHit + i * 12;
    if (d->primKind[i] == RT_PRIM_SPHERE) {
      hasSpheres = 1;
      primHit[i * 3 + 0] = make_float4(h[0], h[1], h[2], 0.0f);
      primHit[i * 3 + 1] = make_float4(0.0f, 0.0f, h[3], 0.0f);
      primHit[i * 3 + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
      primHit[i * 3 + 0] = make_float4(h[0], h[1], h[2], h[3]);
      primHit[i * 3 + 1] = make_float4(h[4], h[5], h[6], h[7]);
      primHit[i * 3 + 2] = make_float4(h[8], h[9], h[10], h[11]);
      uint32_t nb[3];
      std::memcpy(nb, h + 3, 12);
      if ((nb[0] | nb[1] | nb[2]) == 0u) return failMsg(RT_ERR_INVALID, "triangle with an all-zero normal");
    }
  }
  dev.numPrims = d->numPrims;
  dev.hasSpheres = hasSpheres;
  // How far outside its object's box can the point of an ACCEPTED hit lie? (What a point-light query needs to know before
  // it may skip a box beyond the bulb: rt_dev_traverse.h.) A sphere hit lies on the sphere up to the rounding of its
  // coordinates. A triangle hit passes b2 = dot(e1, P - p1), b3 = dot(e2, P - p1) in [0, 1] (src/scene/Object.cpp:203-208):
  // the rounding of those sums (and of P - p1, in units of the coordinates' size) moves the accepted region by
  // |e| * size * 2^-22 barycentric units, i.e. by that times the triangle's diameter. G = |e| * diameter is the triangle's
  // aspect ratio, 1 for a well-shaped one. A triangle whose edge vectors are not finite (zero area: the reference's
  // set-up divides by zero, Object.cpp:191-192) is accepted wherever a ray meets its plane: no bound, no cut (cutEps = 0).
  {
    double extent = 0.0, worst = 0.0;
    bool bounded = d->numNodes > 0 || buildOnDevice;
    for (size_t i = 0; i < n && bounded; ++i) {
      const float *h = d->primHit + i * 12;
      if (d->primKind[i] == RT_PRIM_SPHERE) {
        for (int k = 0; k < 3; ++k) extent = std::max(extent, (double)std::fabs(h[k]) + std::fabs(h[3]));
        continue;
      }
      const double nx = h[3], ny = h[4], nz = h[5], ax = h[6], ay = h[7], az = h[8], bx = h[9], by = h[10], bz = h[11];
      // the edges u = p2 - p1, v = p3 - p1 from their dual vectors e1, e2 (e1.u = 1, e1.v = 0, e2.u = 0, e2.v = 1, all in the plane)
      const double c1[3] = {by * nz - bz * ny, bz * nx - bx * nz, bx * ny - by * nx};  // e2 x n
      const double c2[3] = {ny * az - nz * ay, nz * ax - nx * az, nx * ay - ny * ax};  // n x e1
      const double d1 = ax * c1[0] + ay * c1[1] + az * c1[2], d2 = bx * c2[0] + by * c2[1] + bz * c2[2];
      const double u[3] = {c1[0] / d1, c1[1] / d1, c1[2] / d1}, v[3] = {c2[0] / d2, c2[1] / d2, c2[2] / d2};
      auto len = [](double x, double y, double z) { return std::sqrt(x * x + y * y + z * z); };
      const double diam = std::max({len(u[0], u[1], u[2]), len(v[0], v[1], v[2]), len(u[0] - v[0], u[1] - v[1], u[2] - v[2])});
      const double G = std::max(len(ax, ay, az), len(bx, by, bz)) * diam;
      if (!std::isfinite(G) || !std::isfinite(diam)) { bounded = false; break; }
      worst = std::max(worst, G);
      for (int k = 0; k < 3; ++k) extent = std::max(extent, std::fabs((double)h[k]) + diam);
    }
    const double eps = std::ldexp(std::max(1.0, worst) * extent, -20);
    dev.cutEps = (bounded && std::isfinite(eps) && eps > 0.0 && eps < 1e30) ? (float)eps : 0.0f;
    if (std::getenv("RT_NO_POINT_LIGHT_CUT")) dev.cutEps = 0.0f;
  }
  dev.boxesFinite = std::getenv("RT_EXACT_SLAB_TEST") ? 0 : 1;  // debugging aid: force the compare-and-select slab test
  for (int32_t i = 0; i < d->numNodes; ++i)
    for (int k = 0; k < 3; ++k)
      if (!std::isfinite(d->nodes[i].aabbMin[k]) || !std::isfinite(d->nodes[i].aabbMax[k])) { dev.boxesFinite = 0; dev.cutEps = 0.0f; }
  int rc;
  if ((rc = upload(s, nodes.data(), nodes.size(), &dev.nodes))) return rc;
  if ((rc = upload(s, primHit.data(), primHit.size(), &dev.primHit))) return rc;
  if ((rc = upload(s, bigLeaves.data(), bigLeaves.size(), &dev.bigLeaves))) return rc;
  const uint32_t *order = nullptr;  // device build: object at tree position i = input object order[i]
  // Re-orders an uploaded per-object array (records of `dwords` words) into tree order; the input-order copy is released.
  auto reorder = [&](const void **arr, int32_t dwords) -> int {
    if (!order || !*arr) return RT_OK;
    void *dst = nullptr;
    HIP_TRY(hipMalloc(&dst, n * (size_t)dwords * 4));
    s->allocs.push_back(dst);
    k_gather_records<<<(unsigned)(((int64_t)n * dwords + kBlock - 1) / kBlock), kBlock, 0, s->stream>>>(static_cast<const uint32_t *>(*arr), static_cast<uint32_t *>(dst), order,
                                                                            (int64_t)n, dwords);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    void *old = const_cast<void *>(*arr);
    s->allocs.erase(std::find(s->allocs.begin(), s->allocs.end(), old));
    (void)hipFree(old);
    *arr = dst;
    return RT_OK;
  };
  if (buildOnDevice) {
    const float *dMin = nullptr, *dMax = nullptr;
    if ((rc = upload(s, d->primAabbMin, n * 3, &dMin))) return rc;
    if ((rc = upload(s, d->primAabbMax, n * 3, &dMax))) return rc;
    DeviceBvh bvh;
    std::string err;
    // Clustering by surface area unless RT_DEVICE_BVH=lbvh asks for the linear BVH, which is also what inputs get that the
    // clustering is unsuitable for.
    const char *kind = std::getenv("RT_DEVICE_BVH");
    bool built = false, unsuitable = false;
    if (!(kind && std::strcmp(kind, "lbvh") == 0)) {
      built = buildPloc(s->stream, dMin, dMax, d->numPrims, &bvh, err, &unsuitable);
      if (!built && !unsuitable) return failMsg(RT_ERR_DEVICE, err);
    }
    if (!built && !buildLbvh(s->stream, dMin, dMax, d->numPrims, &bvh, err)) return failMsg(RT_ERR_DEVICE, err);
    s->allocs.push_back(bvh.records);
    s->allocs.push_back(bvh.order);
    for (const float *p : {dMin, dMax}) {
      s->allocs.erase(std::find(s->allocs.begin(), s->allocs.end(), static_cast<void *>(const_cast<float *>(p))));
      (void)hipFree(const_cast<float *>(p));
    }
    dev.nodes = bvh.records;
    dev.rootRef = bvh.rootRef;
    std::memcpy(dev.rootMin, bvh.rootMin, 12);
    std::memcpy(dev.rootMax, bvh.rootMax, 12);
    if (!bvh.boxesFinite) { dev.boxesFinite = 0; dev.cutEps = 0.0f; }
    s->bvhBuildMs = bvh.buildMs;
    order = bvh.order;
    if ((rc = reorder(reinterpret_cast<const void **>(&dev.primHit), 12))) return rc;
  }
  {
    std::vector<float4> shade(n * 4);
    for (size_t i = 0; i < n; ++i) {
      const float *nn = d->primNormals + i * 9, *col = d->primColor + i * 4;
      char *rec = reinterpret_cast<char *>(&shade[i * 4]);  // copied as bytes: three of the words are integers
      std::memcpy(rec, nn, 36);
      std::memcpy(rec + 36, col, 16);
      std::memcpy(rec + 52, &d->primMaterial[i], 4);
      std::memcpy(rec + 56, &d->primTexture[i], 4);
      std::memcpy(rec + 60, &d->primOrigIndex[i], 4);
    }
    if ((rc = upload(s, shade.data(), shade.size(), &dev.primShade))) return rc;  // hipMalloc: 256-byte aligned
    if ((rc = reorder(reinterpret_cast<const void **>(&dev.primShade), 16))) return rc;
  }
  if (d->primTexCoords) {
    if ((rc = upload(s, d->primTexCoords, n * 9, &dev.primTexCoords))) return rc;
    if ((rc = reorder(reinterpret_cast<const void **>(&dev.primTexCoords), 9))) return rc;
  }
  if (order) {
    s->allocs.erase(std::find(s->allocs.begin(), s->allocs.end(), static_cast<void *>(const_cast<uint32_t *>(order))));
    (void)hipFree(const_cast<uint32_t *>(order));
  }
  // ---- planes, materials, lights, textures
  std::vector<DevPlane> planes(d->numPlanes);
  for (int32_t i = 0; i < d->numPlanes; ++i) {
    const RtPlane &p = d->planes[i];
    DevPlane q{};
    std::memcpy(q.normal, p.normal, 12); std::memcpy(q.point, p.point, 12);
    std::memcpy(q.texTopLeft, p.texTopLeft, 12); q.texZoom = p.texZoom; std::memcpy(q.texShift, p.texShift, 12);
    std::memcpy(q.color, p.color, 16); q.material = p.material; q.texture = p.texture;
    planes[i] = q;
  }
  dev.numPlanes = d->numPlanes;
  if ((rc = upload(s, planes.data(), planes.size(), &dev.planes))) return rc;
  std::vector<DevMaterial> mats(d->numMaterials);
  std::vector<bool> used(d->numMaterials, false);
  for (size_t i = 0; i < n; ++i) used[d->primMaterial[i]] = true;
  for (int32_t i = 0; i < d->numPlanes; ++i) used[d->planes[i].material] = true;
  for (int32_t i = 0; i < d->numMaterials; ++i)
    if (!makeMaterial(d->materials[i], mats[i]) && used[i]) return failMsg(RT_ERR_INVALID, "material " + std::to_string(i) + " has an unsupported type");
  if ((rc = upload(s, mats.data(), mats.size(), &dev.materials))) return rc;
  std::vector<DevLight> lights(d->numLights);
  for (int32_t i = 0; i < d->numLights; ++i) {
    const RtLight &l = d->lights[i];
    DevLight q{};
    q.type = l.type; std::memcpy(q.color, l.color, 16); std::memcpy(q.v, l.v, 12); q.radius = l.radius; q.texture = l.texture;
    q.envSlot = -1;
    if (l.type == RT_LIGHT_ENVIRONMENT) { q.envSlot = s->numEnvLights++; s->numAnyLights++; }
    else if (l.type == RT_LIGHT_DISTANT) s->numAnyLights++;
    else s->numPointLights++;
    lights[i] = q;
  }
  dev.numLights = d->numLights; dev.numEnvLights = s->numEnvLights; dev.numPointLights = s->numPointLights; dev.numAnyLights = s->numAnyLights;
  if ((rc = upload(s, lights.data(), lights.size(), &dev.lights))) return rc;
  std::vector<DevTexture> texs(d->numTextures);
  size_t totalTexels = 0;
  for (int32_t i = 0; i < d->numTextures; ++i) {
    texs[i].width = d->textures[i].width; texs[i].height = d->textures[i].height; texs[i].offset = (int64_t)totalTexels;
    totalTexels += (size_t)d->textures[i].width * d->textures[i].height;
  }
  std::vector<float4> texels(totalTexels);
  for (int32_t i = 0; i < d->numTextures; ++i) {
    size_t cnt = (size_t)d->textures[i].width * d->textures[i].height;
    if (cnt) std::memcpy(&texels[texs[i].offset], d->textures[i].rgba, cnt * 16);
  }
  if ((rc = upload(s, texs.data(), texs.size(), &dev.textures))) return rc;
  if ((rc = upload(s, texels.data(), texels.size(), &dev.texels))) return rc;
  setView(s, d->camera, d->options);
  void *c = nullptr;
  HIP_TRY(hipMalloc(&c, sizeof(Counts)));
  s->allocs.push_back(c);
  s->counts = static_cast<Counts *>(c);
  HIP_TRY(hipMemset(c, 0, sizeof(Counts)));
  return RT_OK;
}

bool inPartition(const RtRenderParams &p, int tilesX, int tw, int th, int x, int y) {
  if (p.partCount <= 1) return true;
  int tile = (y / th) * tilesX + (x / tw);
  return tile % p.partCount == p.partIndex;
}

// Pixels of this call's partition, ordered so that 64 consecutive entries are an 8x8 block wherever possible.
std::vector<int32_t> buildPixelList(const RtRenderParams &p, int W, int H) {
  int tw = p.tileW > 0 ? p.tileW : 32, th = p.tileH > 0 ? p.tileH : 32;
  int tilesX = (W + tw - 1) / tw;
  std::vector<int32_t> list;
  list.reserve((size_t)W * H / std::max(1, p.partCount) + 64);
  for (int by = 0; by < H; by += 8)
    for (int bx = 0; bx < W; bx += 8)
      for (int y = by; y < std::min(by + 8, H); ++y)
        for (int x = bx; x < std::min(bx + 8, W); ++x)
          if (inPartition(p, tilesX, tw, th, x, y)) list.push_back(y * W + x);
  return list;
}

int ensureRenderBuffers(RtScene *s, int64_t P, int64_t npix, int64_t fbPixels) {
  if (s->fbPixels < fbPixels) {
    if (s->fb) (void)hipFree(s->fb);
    s->fb = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->fb), fbPixels * sizeof(float4)));
    s->fbPixels = fbPixels;
  }
  if (s->pixCap < npix) {
    if (s->pixList) (void)hipFree(s->pixList);
    if (s->pixState) (void)hipFree(s->pixState);
    if (s->packed) (void)hipFree(s->packed);
    s->pixList = nullptr; s->pixState = nullptr; s->packed = nullptr;
    s->pixKey[0] = -1;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->pixList), npix * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&s->pixState, (size_t)npix * (sizeof(float4) + 6 * 4) + 256));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->packed), npix * sizeof(float4)));
    char *ps = static_cast<char *>(s->pixState);
    s->acc.sum = reinterpret_cast<float4 *>(ps); ps += (size_t)npix * sizeof(float4);
    s->acc.hits = reinterpret_cast<int32_t *>(ps); ps += (size_t)npix * 4;
    s->acc.taken = reinterpret_cast<int32_t *>(ps); ps += (size_t)npix * 4;
    s->acc.lumSum = reinterpret_cast<float *>(ps); ps += (size_t)npix * 4;
    s->acc.lumSqSum = reinterpret_cast<float *>(ps); ps += (size_t)npix * 4;
    s->activeA = reinterpret_cast<int32_t *>(ps); ps += (size_t)npix * 4;
    s->activeB = reinterpret_cast<int32_t *>(ps); ps += (size_t)npix * 4;
    s->activeCount = reinterpret_cast<uint32_t *>(ps);
    s->pixCap = npix;
  }
  if (s->arenaP < P) {
    if (s->arena) (void)hipFree(s->arena);
    s->arena = nullptr; s->arenaP = 0; s->arenaBytes = 0;
    int64_t nEnv = std::max(1, s->numEnvLights);
    int64_t nLights = std::max(1, s->dev.numLights);
    size_t bytes = (size_t)P * (sizeof(float4) * (7 + nEnv + nLights) + 4 * 6) + (size_t)((P + 63) / 64) * sizeof(uint4) * nLights + 4096;
    HIP_TRY(hipMalloc(&s->arena, bytes));
    s->arenaBytes = bytes;
    s->arenaP = P;
  }
  // carve with the capacity the arena was allocated for, so the layout is the same for calls with a smaller P
  int64_t cap = s->arenaP;
  PathBuf &b = s->buf;
  size_t f4b = (size_t)cap * sizeof(float4);
  char *base = static_cast<char *>(s->arena);
  b.rayO = reinterpret_cast<float4 *>(base); base += f4b;
  b.rayD = reinterpret_cast<float4 *>(base); base += f4b;
  b.beta[0] = reinterpret_cast<float4 *>(base); base += f4b;
  b.L = reinterpret_cast<float4 *>(base); base += f4b;
  b.shPoint = reinterpret_cast<float4 *>(base); base += f4b;
  b.shColor = reinterpret_cast<float4 *>(base); base += f4b;
  b.beta[1] = reinterpret_cast<float4 *>(base); base += f4b;
  b.envWi = reinterpret_cast<float4 *>(base); base += f4b * std::max(1, s->numEnvLights);
  b.lightC = reinterpret_cast<float4 *>(base); base += f4b * std::max(1, s->dev.numLights);
  b.rng = reinterpret_cast<uint32_t *>(base); base += (size_t)cap * 4;
  b.hitT = reinterpret_cast<float *>(base); base += (size_t)cap * 4;
  b.hitPrim = reinterpret_cast<int32_t *>(base); base += (size_t)cap * 4;
  b.qA = reinterpret_cast<uint32_t *>(base); base += (size_t)cap * 4;
  b.qB = reinterpret_cast<uint32_t *>(base); base += (size_t)cap * 4;
  b.qHit = reinterpret_cast<uint32_t *>(base); base += (size_t)cap * 4;
  base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(base) + 255) & ~(uintptr_t)255);
  b.shadowCode = reinterpret_cast<uint4 *>(base);
  b.rowsCap = (int32_t)((cap + 63) / 64);
  b.counts = s->counts;
  b.P = (int32_t)P;
  b.npix = (int32_t)npix;
  return RT_OK;
}

int beginEvent(RtScene *s, int category, bool enabled) {
  if (!enabled) return RT_OK;
  if (s->eventsUsed == s->events.size()) {
    EventPair e{};
    HIP_TRY(hipEventCreate(&e.a));
    HIP_TRY(hipEventCreate(&e.b));
    s->events.push_back(e);
  }
  s->events[s->eventsUsed].category = category;
  HIP_TRY(hipEventRecord(s->events[s->eventsUsed].a, s->stream));
  return RT_OK;
}
int endEvent(RtScene *s, bool enabled) {
  if (!enabled) return RT_OK;
  HIP_TRY(hipEventRecord(s->events[s->eventsUsed].b, s->stream));
  ++s->eventsUsed;
  return RT_OK;
}

constexpr int64_t kFusedTraceMaxPaths = (int64_t)100 << 20;  // passes below this size use k_shadow_then_extend
inline unsigned gridFor(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + kBlock - 1) / kBlock); }
// Traversal kernels are persistent: enough blocks to fill every CU once (8 blocks of 4 wavefronts = the 32-wave limit),
// fewer when the queue cannot feed them.
inline unsigned persistentGrid(const RtScene *s, int64_t n) {
  return (unsigned)std::min<int64_t>(gridFor(n), (int64_t)s->numCUs * s->blocksPerCU);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- no-allocation entry
// rt_trace_primary_static keeps everything it needs beyond the caller's buffers (the queue ticket and the device-side
// staging of ids and t) in this array, which is part of the code object: every GPU the library runs on holds its own
// copy at its own address from the moment the code object is loaded until the process ends. Host code never uses the
// array's name as an address; it asks the runtime for the device address once per device (resolveStaticScratch, at the
// first rt_trace_primary_static on that device).
constexpr int64_t kStaticTraceMaxPixels = (int64_t)1 << 22;  // 4 Mi pixels (2048 x 2048): 32 MiB of staging
constexpr size_t kStaticTicketBytes = sizeof(Ticket) * kQueueSegments;  // 1 KiB
constexpr size_t kStaticScratchBytes = kStaticTicketBytes + (size_t)kStaticTraceMaxPixels * 8;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
void *g_scratchPtr[kMaxDevices] = {};
std::mutex g_scratchMutex;

// With `device` current: the address of this device's copy of `scratch`. Returns hipSuccess and caches the address.
hipError_t resolveStaticScratch(int device) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratchMutex);
  if (g_scratchPtr[device]) return hipSuccess;
  void *ptr = nullptr;
  hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
  if (e != hipSuccess) return e;
  if (!ptr) return hipErrorInvalidSymbol;
  g_scratchPtr[device] = ptr;
  return hipSuccess;
}
}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_error.c_str(); }
int rt_device_count(void) {
  int count = 0;
  return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}
const char *rt_version(void) { return "raytracer_amd 0.1 (gfx950 wavefront path tracer)"; }

RtScene *rt_scene_create(const RtSceneDesc *desc, int device) try {
  if (validateDesc(desc) != RT_OK) return nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) { g_error = std::string("no HIP device available: ") + hipGetErrorString(e); return nullptr; }
  if (device < 0 || device >= count) { g_error = "device index out of range"; return nullptr; }
  if ((e = hipSetDevice(device)) != hipSuccess) { g_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
  auto s = std::make_unique<RtScene>();
  s->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->numCUs = prop.multiProcessorCount;
  {
    int nb = 0;
    if (const char *x = std::getenv("RT_TRAV_EXTRA_LDS")) s->extraLds = (size_t)std::max(0L, std::atol(x));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_shadow<false>, kBlock, s->extraLds) == hipSuccess && nb > 0) s->blocksPerCU = nb;
    if (const char *x = std::getenv("RT_TRAV_BLOCKS_PER_CU")) s->blocksPerCU = std::max(1, std::atoi(x));
  }
  if ((e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)) != hipSuccess) {
    g_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
    return nullptr;
  }
  int built;
  try { built = buildScene(s.get(), desc); } catch (...) { built = caught(); }
  if (built != RT_OK) {
    RtScene *raw = s.release();
    rt_scene_destroy(raw);
    return nullptr;
  }
  return s.release();
} catch (...) {
  (void)caught();
  return nullptr;
}

void rt_scene_destroy(RtScene *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (void *p : s->allocs) (void)hipFree(p);
  if (s->arena) (void)hipFree(s->arena);
  if (s->pixList) (void)hipFree(s->pixList);
  if (s->pixState) (void)hipFree(s->pixState);
  if (s->fb) (void)hipFree(s->fb);
  if (s->packed) (void)hipFree(s->packed);
  if (s->work) (void)hipFree(s->work);
  if (s->pinned) (void)hipHostFree(s->pinned);
  if (s->workEvA) (void)hipEventDestroy(s->workEvA);
  if (s->workEvB) (void)hipEventDestroy(s->workEvB);
  for (auto &ps : s->peers) {
    if (ps.stream) { (void)hipStreamSynchronize(ps.stream); (void)hipStreamDestroy(ps.stream); }
    if (ps.done) (void)hipEventDestroy(ps.done);
    if (ps.idx) (void)hipFree(ps.idx);
    if (ps.staging) (void)hipFree(ps.staging);
  }
  for (auto &e : s->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

int rt_scene_set_view(RtScene *s, const RtCamera *camera, const RtOptions *options) try {
  if (!s) return failMsg(RT_ERR_INVALID, "null scene");
  RtCamera cam;
  std::memcpy(cam.eye, s->dev.eye, 12); std::memcpy(cam.forward, s->dev.forward, 12);
  std::memcpy(cam.right, s->dev.right, 12); std::memcpy(cam.up, s->dev.up, 12);
  RtOptions o;
  o.width = s->dev.width; o.height = s->dev.height; o.bias = s->dev.bias; o.exposure = s->dev.exposure;
  o.maxBounces = s->dev.maxBounces; o.numRays = s->dev.numRays; o.fisheye = s->dev.cameraMode == 1;
  o.focus = s->dev.focus; o.lens = s->dev.lens;
  if (camera) cam = *camera;
  if (options) {
    if (options->width <= 0 || options->height <= 0) return failMsg(RT_ERR_INVALID, "bad image size");
    o = *options;
  }
  setView(s, cam, o);
  return RT_OK;
} catch (...) {
  return caught();
}

// Staging for the trace-only entry points (device side of the host buffers the caller passes). It only ever grows:
// a caller that repeats a call of the same size pays for no allocation after the first.
static int ensureWork(RtScene *s, size_t bytes) {
  if (s->workBytes >= bytes) return RT_OK;
  if (s->work) { (void)hipFree(s->work); s->work = nullptr; s->workBytes = 0; }
  bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  HIP_TRY(hipMalloc(&s->work, bytes));
  s->workBytes = bytes;
  return RT_OK;
}
static int ensurePinned(RtScene *s, size_t bytes) {
  if (s->pinnedBytes >= bytes) return RT_OK;
  if (s->pinned) { (void)hipHostFree(s->pinned); s->pinned = nullptr; s->pinnedBytes = 0; }
  bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  HIP_TRY(hipHostMalloc(&s->pinned, bytes, hipHostMallocDefault));
  s->pinnedBytes = bytes;
  return RT_OK;
}
// Page-locked staging of at least `bytes`, or false: the staging is only a faster route, so a host that cannot lock
// that much memory gets the plain copies instead of an error.
static bool havePinned(RtScene *s, size_t bytes) {
  if (ensurePinned(s, bytes) == RT_OK) return true;
  (void)hipGetLastError();
  g_error.clear();
  return false;
}
// Is `p` ordinary (pageable) host memory? The runtime copies into such memory through small staging buffers of its own,
// at a fraction of the link's rate.
static bool isPageableHost(const void *p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return true; }  // unknown to the runtime
  return a.type == hipMemoryTypeUnregistered;
}
// dst[0 .. bytes) = src, by a few host threads when it is worth starting them
static void hostCopy(void *dst, const void *src, size_t bytes) {
  const size_t kPiece = (size_t)1 << 20;
  const int threads = (int)std::min<size_t>(4, bytes / kPiece);
  if (threads <= 1) { std::memcpy(dst, src, bytes); return; }
  std::vector<std::thread> pool;
  pool.reserve((size_t)threads);
  const size_t per = (bytes + threads - 1) / threads;
  auto piece = [=](size_t b, size_t e) { std::memcpy(static_cast<char *>(dst) + b, static_cast<const char *>(src) + b, e - b); };
  for (int k = 1; k < threads; ++k) {
    const size_t b = (size_t)k * per, e = std::min(bytes, b + per);
    if (b >= e) break;
    try { pool.emplace_back(piece, b, e); } catch (const std::system_error &) { piece(b, e); }  // no thread to be had: copied here
  }
  piece(0, std::min(bytes, per));  // the calling thread takes the first piece
  for (auto &t : pool) t.join();
}
// A result of `bytes` at device address src to dst, which may be device memory, page-locked or pageable host memory.
// Large copies into pageable memory go through the scene's page-locked staging (one copy at the link's rate, then host
// copies) and are complete on return; everything else is queued on `stream` for the caller to wait for.
static int copyOut(RtScene *s, void *dst, const void *src, size_t bytes, hipStream_t stream) {
  if (bytes >= ((size_t)1 << 20) && isPageableHost(dst) && havePinned(s, bytes)) {
    HIP_TRY(hipMemcpyAsync(s->pinned, src, bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    hostCopy(dst, s->pinned, bytes);
    return RT_OK;
  }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream));
  return RT_OK;
}
static int ensureWorkEvents(RtScene *s) {
  if (!s->workEvA) HIP_TRY(hipEventCreate(&s->workEvA));
  if (!s->workEvB) HIP_TRY(hipEventCreate(&s->workEvB));
  return RT_OK;
}
static inline size_t alignUp256(size_t x) { return (x + 255) & ~(size_t)255; }

// Un-jittered primary rays of the rectangle into device arrays outPrim / outT, queue ticket at `ticket` (device word).
static void launchTracePrimary(RtScene *s, int32_t x0, int32_t y0, int32_t w, int32_t h, int32_t *outPrim, float *outT, uint32_t *ticket) {
  int64_t waves = (int64_t)((w + 7) / 8) * ((h + 7) / 8);
  k_trace_primary<<<persistentGrid(s, waves * 64), kBlock, 0, s->stream>>>(s->dev, x0, y0, w, h, outPrim, outT, ticket);
}

int rt_trace_primary(RtScene *s, int32_t x0, int32_t y0, int32_t w, int32_t h, int32_t *primId, float *t) try {
  if (!s || !primId || !t) return failMsg(RT_ERR_INVALID, "null argument");
  if (w <= 0 || h <= 0) return failMsg(RT_ERR_INVALID, "empty rectangle");
  HIP_TRY(hipSetDevice(s->device));
  int64_t n = (int64_t)w * h;
  int rc;
  if ((rc = ensureWork(s, 2 * alignUp256((size_t)n * 4)))) return rc;
  if ((rc = ensureWorkEvents(s))) return rc;
  int32_t *dPrim = static_cast<int32_t *>(s->work);
  float *dT = reinterpret_cast<float *>(static_cast<char *>(s->work) + alignUp256((size_t)n * 4));
  HIP_TRY(hipMemsetAsync(&s->counts->ticket[2], 0, sizeof(TicketSet), s->stream));
  HIP_TRY(hipEventRecord(s->workEvA, s->stream));
  launchTracePrimary(s, x0, y0, w, h, dPrim, dT, &s->counts->ticket[2].seg[0].next);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->workEvB, s->stream));
  const size_t tOff = alignUp256((size_t)n * 4);
  if (isPageableHost(primId) && isPageableHost(t) && havePinned(s, tOff + (size_t)n * 4)) {
    // one copy of both arrays into page-locked staging at the link's rate, then plain host copies
    HIP_TRY(hipMemcpyAsync(s->pinned, s->work, tOff + (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    hostCopy(primId, s->pinned, (size_t)n * 4);
    hostCopy(t, static_cast<const char *>(s->pinned) + tOff, (size_t)n * 4);
  } else {
    HIP_TRY(hipMemcpyAsync(primId, dPrim, n * 4, hipMemcpyDefault, s->stream));
    HIP_TRY(hipMemcpyAsync(t, dT, n * 4, hipMemcpyDefault, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  HIP_TRY(hipEventElapsedTime(&s->lastKernelMs, s->workEvA, s->workEvB));
  return RT_OK;
} catch (...) {
  return caught();
}

int rt_trace_primary_static(RtScene *s, int32_t x0, int32_t y0, int32_t w, int32_t h, int32_t *primId, float *t) try {
  if (!s || !primId || !t) return failMsg(RT_ERR_INVALID, "null argument");
  if (w <= 0 || h <= 0) return failMsg(RT_ERR_INVALID, "empty rectangle");
  const int64_t n = (int64_t)w * h;
  if (n > kStaticTraceMaxPixels)
    return failMsg(RT_ERR_INVALID, "rt_trace_primary_static: " + std::to_string(n) + " pixels exceed the static scratch buffer's " +
                                       std::to_string(kStaticTraceMaxPixels) + "; use rt_trace_primary or a smaller rectangle");
  HIP_TRY(hipSetDevice(s->device));
  // first call on this GPU: ask the runtime where this GPU's copy of the array lies (with that device current)
  if (hipError_t e = resolveStaticScratch(s->device); e != hipSuccess)
    return failMsg(RT_ERR_DEVICE, std::string("the static scratch buffer has no address on this device: hipGetSymbolAddress: ") + hipGetErrorString(e));
  char *base = static_cast<char *>(g_scratchPtr[s->device]);
  uint32_t *ticket = reinterpret_cast<uint32_t *>(base);
  int32_t *dPrim = reinterpret_cast<int32_t *>(base + kStaticTicketBytes);
  float *dT = reinterpret_cast<float *>(base + kStaticTicketBytes + (size_t)kStaticTraceMaxPixels * 4);
  // the buffer keeps the previous call's contents: the ticket must start at zero; ids and t are fully overwritten
  HIP_TRY(hipMemsetAsync(ticket, 0, kStaticTicketBytes, s->stream));
  launchTracePrimary(s, x0, y0, w, h, dPrim, dT, ticket);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(primId, dPrim, n * 4, hipMemcpyDefault, s->stream));
  HIP_TRY(hipMemcpyAsync(t, dT, n * 4, hipMemcpyDefault, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RT_OK;
} catch (...) {
  return caught();
}

int64_t rt_trace_primary_static_max_pixels(void) { return kStaticTraceMaxPixels; }

static int traceRays(RtScene *s, bool any, int64_t n, const float *origins, const float *directions, int32_t *primId, float *t,
                     float *point, float *normal, uint8_t *occluded) {
  if (!s || !origins || !directions) return failMsg(RT_ERR_INVALID, "null argument");
  if (n <= 0) return RT_OK;
  if (n >= ((int64_t)1 << 32) - ((int64_t)1 << 24)) return failMsg(RT_ERR_INVALID, "too many rays for one call (queue indices are 32-bit)");
  HIP_TRY(hipSetDevice(s->device));
  // carve the staging area: origins, directions, then whichever outputs were asked for
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t at = off; off += alignUp256(bytes); return at; };
  const size_t oO = carve((size_t)n * 12), oD = carve((size_t)n * 12);
  const size_t oOcc = any ? carve((size_t)n) : 0;
  const size_t oPrim = (!any && primId) ? carve((size_t)n * 4) : 0, oT = (!any && t) ? carve((size_t)n * 4) : 0;
  const size_t oP = (!any && point) ? carve((size_t)n * 12) : 0, oN = (!any && normal) ? carve((size_t)n * 12) : 0;
  int rc;
  if ((rc = ensureWork(s, off))) return rc;
  if ((rc = ensureWorkEvents(s))) return rc;
  char *base = static_cast<char *>(s->work);
  float *dO = reinterpret_cast<float *>(base + oO), *dD = reinterpret_cast<float *>(base + oD);
  uint8_t *dOcc = any ? reinterpret_cast<uint8_t *>(base + oOcc) : nullptr;
  int32_t *dPrim = (!any && primId) ? reinterpret_cast<int32_t *>(base + oPrim) : nullptr;
  float *dT = (!any && t) ? reinterpret_cast<float *>(base + oT) : nullptr;
  float *dP = (!any && point) ? reinterpret_cast<float *>(base + oP) : nullptr;
  float *dN = (!any && normal) ? reinterpret_cast<float *>(base + oN) : nullptr;
  uint32_t *ticket = &s->counts->ticket[2].seg[0].next;
  // Large batches in pageable host memory: rays and results cross the link in one copy each way through page-locked
  // staging laid out like the device staging; host threads move them between the caller's arrays and the staging.
  const bool staged = (size_t)n * 24 >= ((size_t)1 << 20) && isPageableHost(origins) && isPageableHost(directions) &&
                      (!occluded || isPageableHost(occluded)) && (!primId || isPageableHost(primId)) && (!t || isPageableHost(t)) &&
                      (!point || isPageableHost(point)) && (!normal || isPageableHost(normal)) && havePinned(s, off);
  char *pin = nullptr;
  const size_t inputsEnd = oD + (size_t)n * 12, outputsBegin = oD + alignUp256((size_t)n * 12);
  if (staged) {
    pin = static_cast<char *>(s->pinned);
    hostCopy(pin + oO, origins, (size_t)n * 12);
    hostCopy(pin + oD, directions, (size_t)n * 12);
    HIP_TRY(hipMemcpyAsync(base + oO, pin + oO, inputsEnd - oO, hipMemcpyHostToDevice, s->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(dO, origins, n * 12, hipMemcpyDefault, s->stream));
    HIP_TRY(hipMemcpyAsync(dD, directions, n * 12, hipMemcpyDefault, s->stream));
  }
  HIP_TRY(hipMemsetAsync(ticket, 0, sizeof(TicketSet), s->stream));
  HIP_TRY(hipEventRecord(s->workEvA, s->stream));
  if (any) k_trace_rays<true><<<persistentGrid(s, n), kBlock, 0, s->stream>>>(s->dev, n, dO, dD, nullptr, nullptr, nullptr, nullptr, dOcc, ticket);
  else k_trace_rays<false><<<persistentGrid(s, n), kBlock, 0, s->stream>>>(s->dev, n, dO, dD, dPrim, dT, dP, dN, nullptr, ticket);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->workEvB, s->stream));
  if (staged && off > outputsBegin) {
    HIP_TRY(hipMemcpyAsync(pin + outputsBegin, base + outputsBegin, off - outputsBegin, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (dOcc) hostCopy(occluded, pin + oOcc, (size_t)n);
    if (dPrim) hostCopy(primId, pin + oPrim, (size_t)n * 4);
    if (dT) hostCopy(t, pin + oT, (size_t)n * 4);
    if (dP) hostCopy(point, pin + oP, (size_t)n * 12);
    if (dN) hostCopy(normal, pin + oN, (size_t)n * 12);
  } else {
    if (dOcc) HIP_TRY(hipMemcpyAsync(occluded, dOcc, n, hipMemcpyDefault, s->stream));
    if (dPrim) HIP_TRY(hipMemcpyAsync(primId, dPrim, n * 4, hipMemcpyDefault, s->stream));
    if (dT) HIP_TRY(hipMemcpyAsync(t, dT, n * 4, hipMemcpyDefault, s->stream));
    if (dP) HIP_TRY(hipMemcpyAsync(point, dP, n * 12, hipMemcpyDefault, s->stream));
    if (dN) HIP_TRY(hipMemcpyAsync(normal, dN, n * 12, hipMemcpyDefault, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  HIP_TRY(hipEventElapsedTime(&s->lastKernelMs, s->workEvA, s->workEvB));
  return RT_OK;
}

int rt_trace_closest(RtScene *s, int64_t n, const float *origins, const float *directions, int32_t *primId, float *t, float *point,
                     float *normal) try {
  return traceRays(s, false, n, origins, directions, primId, t, point, normal, nullptr);
} catch (...) {
  return caught();
}

int rt_trace_any(RtScene *s, int64_t n, const float *origins, const float *directions, uint8_t *occluded) try {
  if (!occluded) return failMsg(RT_ERR_INVALID, "null argument");
  return traceRays(s, true, n, origins, directions, nullptr, nullptr, nullptr, nullptr, occluded);
} catch (...) {
  return caught();
}

int rt_render(RtScene *s, const RtRenderParams *params, float *rgbaOut, RtStats *stats) try {
  if (!s || !params) return failMsg(RT_ERR_INVALID, "null argument");
  if (params->rngMode != RT_RNG_COUNTER) return failMsg(RT_ERR_INVALID, "the HIP library only implements RT_RNG_COUNTER");
  if (params->partCount > 1 && (params->partIndex < 0 || params->partIndex >= params->partCount))
    return failMsg(RT_ERR_INVALID, "partIndex out of range");
  HIP_TRY(hipSetDevice(s->device));
  const DevScene &S = s->dev;
  int W = S.width, H = S.height;
  if (S.numRays <= 0) return failMsg(RT_ERR_INVALID, "numRays must be positive");
  // the window of the sample loop this call runs
  const int sFirst = params->firstSample;
  if (sFirst < 0 || sFirst >= S.numRays || params->sampleCount < 0) return failMsg(RT_ERR_INVALID, "firstSample / sampleCount outside 0 .. numRays");
  const int sEnd = params->sampleCount > 0 ? std::min(S.numRays, sFirst + params->sampleCount) : S.numRays;
  const bool adaptive = params->adaptiveThreshold > 0.0f;
  const int adaptiveBatch = params->adaptiveBatch > 0 ? params->adaptiveBatch : 16;
  const int adaptiveMin = params->adaptiveMinSamples > 0 ? params->adaptiveMinSamples : adaptiveBatch;
  if (adaptive && (sFirst != 0 || params->keepAccumulator)) return failMsg(RT_ERR_INVALID, "adaptive sampling starts at sample 0 with an empty accumulator");
  const int key[6] = {W, H, params->partIndex, params->partCount, params->tileW, params->tileH};
  const bool samePartition = s->pixList && std::memcmp(key, s->pixKey, sizeof key) == 0;
  std::vector<int32_t> pix;
  if (params->keepAccumulator && (!samePartition || !s->accValid))
    return failMsg(RT_ERR_INVALID, "keepAccumulator needs a previous rt_render on this scene with the same image size and tile partition");
  if (!samePartition) pix = buildPixelList(*params, W, H);
  int64_t npix = samePartition ? s->pixCount : (int64_t)pix.size();
  int64_t fbPixels = (int64_t)W * H;
  int sppPass = params->samplesPerPass;
  if (sppPass <= 0) {
    // Default: as many samples per pass as fit in half of the free HBM, at most 64 GiB of path state. The traversal
    // kernels run persistent wavefronts; a launch with few queries per wavefront spends a large part of its time
    // draining (measured on the 1080x1080 frame: 7 samples per pass 790 ms, 64 -> 635 ms, all 256 -> 618 ms).
    size_t freeB = 0, totalB = 0;
    HIP_TRY(hipMemGetInfo(&freeB, &totalB));
    const size_t avail = freeB + (s->arena ? s->arenaBytes : 0);
    const size_t budget = std::min<size_t>((size_t)64 << 30, avail / 2);
    const size_t perPath = sizeof(float4) * (7 + std::max(1, s->numEnvLights) + std::max(1, s->dev.numLights)) + 4 * 6 + (size_t)std::max(1, s->dev.numLights);
    sppPass = (int)std::min<size_t>((size_t)S.numRays, std::max<size_t>(1, budget / (perPath * (size_t)std::max<int64_t>(1, npix))));
  }
  bool sppPassChosenHere = params->samplesPerPass <= 0;
  sppPass = std::min(sppPass, sEnd - sFirst);
  if (adaptive) { sppPass = std::min(adaptiveBatch, S.numRays); sppPassChosenHere = false; }  // pixels decide at batch boundaries
  // path slots are 31-bit indices, and the visibility queue (one entry per hit and light) is indexed with 32 bits
  const int64_t maxPaths = std::min<int64_t>(((int64_t)1 << 31) - 1, (((int64_t)1 << 32) - ((int64_t)1 << 24)) / std::max(1, S.numLights));
  if (sppPassChosenHere) sppPass = (int)std::max<int64_t>(1, std::min<int64_t>(sppPass, maxPaths / std::max<int64_t>(1, npix)));
  int64_t P = std::max<int64_t>(1, npix * sppPass);
  if (P > maxPaths) return failMsg(RT_ERR_INVALID, "samplesPerPass too large for this image and light count");
  int rc = ensureRenderBuffers(s, P, std::max<int64_t>(1, npix), fbPixels);
  // Somebody else may have taken the memory between the query and the allocation: with a pass size chosen here, fall
  // back to smaller passes instead of failing the frame.
  while (rc == RT_ERR_DEVICE && sppPassChosenHere && sppPass > 1) {
    (void)hipGetLastError();
    sppPass = std::max(1, sppPass / 2);
    P = std::max<int64_t>(1, npix * sppPass);
    rc = ensureRenderBuffers(s, P, std::max<int64_t>(1, npix), fbPixels);
  }
  if (rc) return rc;
  PathBuf B = s->buf;
  bool timing = stats != nullptr;
  bool count = params->collectCounters != 0;
  s->eventsUsed = 0;
  hipStream_t st = s->stream;
  HIP_TRY(hipMemsetAsync(s->counts, 0, sizeof(Counts), st));
  HIP_TRY(hipMemsetAsync(s->fb, 0, fbPixels * sizeof(float4), st));
  if (npix > 0) {
    if (!samePartition) {
      s->pixKey[0] = -1;
      HIP_TRY(hipMemcpyAsync(s->pixList, pix.data(), npix * 4, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));  // `pix` is pageable host memory that goes out of scope
      std::memcpy(s->pixKey, key, sizeof key);
      s->pixCount = npix;
    }
    if (!params->keepAccumulator) HIP_TRY(hipMemsetAsync(s->acc.sum, 0, (size_t)s->pixCap * (sizeof(float4) + 4 * 4), st));  // sum, hits, taken, lumSum, lumSqSum: carved back to back for pixCap pixels
  } else {
    std::memcpy(s->pixKey, key, sizeof key);  // an empty partition: nothing to trace, nothing for a gather to move
    s->pixCount = 0;
  }
  if ((rc = beginEvent(s, CAT_TOTAL, timing))) return rc;
  size_t totalIdx = s->eventsUsed;
  if (timing) ++s->eventsUsed;  // the total pair stays open until the end
  uint32_t passes = 0, launchesExtend = 0, launchesShadow = 0, launchesFused = 0;
  int64_t nPixPass = npix;            // pixels that take samples in the next pass (adaptive sampling shrinks it)
  int32_t *activeCur = nullptr, *activeNext = s->activeA;
  s->accValid = false;
  for (int sBase = sFirst; nPixPass > 0 && sBase < sEnd; sBase += sppPass, ++passes) {
    int sppThis = std::min(sppPass, sEnd - sBase);
    int64_t n = nPixPass * sppThis;
    B.spp = sppThis;
    B.npix = (int32_t)nPixPass;
    B.active = activeCur;
    unsigned grid = gridFor(n);
    k_begin_pass<<<1, 1, 0, st>>>(B, S.cameraMode == 1 ? 0u : (uint32_t)n);
    if ((rc = beginEvent(s, CAT_RAYGEN, timing))) return rc;
    k_raygen<<<grid, kBlock, 0, st>>>(S, B, s->pixList, sBase, sppThis, params->seed);
    k_count_samples<<<1, 1, 0, st>>>(B, (unsigned long long)n);
    if ((rc = endEvent(s, timing))) return rc;
    uint32_t *qCur = B.qA, *qNext = B.qB;
    // Short launches (a part of a frame on one of several GPUs, a frame of few samples): the visibility queries of a
    // bounce and the rays of the next are traced in one launch (k_shadow_then_extend), timed as visibility work. Measured
    // on the bench frame: 1.3 ms off a pass of 37 M or 75 M paths, nothing off one of 150 M.
    const bool fuse = n < kFusedTraceMaxPaths && S.numLights > 0 && !std::getenv("RT_NO_FUSED_TRACE");
    for (int bounce = 0; bounce < S.maxBounces; ++bounce) {
      if (!(fuse && bounce > 0)) {  // otherwise these rays were traced together with the previous bounce's visibility queries
        if ((rc = beginEvent(s, CAT_EXTEND, timing))) return rc;
        if (count) k_extend<true><<<persistentGrid(s, n), kBlock, s->extraLds, st>>>(S, B, qCur, bounce);
        else k_extend<false><<<persistentGrid(s, n), kBlock, s->extraLds, st>>>(S, B, qCur, bounce);
        if ((rc = endEvent(s, timing))) return rc;
        ++launchesExtend;
      }
      if ((rc = beginEvent(s, CAT_SHADE, timing))) return rc;
      k_shade<<<grid, kBlock, 0, st>>>(S, B, qCur, qNext, bounce, s->pixList, sBase, params->seed);
      if ((rc = endEvent(s, timing))) return rc;
      if (S.numLights > 0) {
        if ((rc = beginEvent(s, CAT_SHADOW, timing))) return rc;
        unsigned gridShadow = persistentGrid(s, n * S.numLights);
        if (fuse && bounce + 1 < S.maxBounces) {
          ++launchesFused;
          if (count) k_shadow_then_extend<true><<<gridShadow, kBlock, s->extraLds, st>>>(S, B, qCur, qNext, bounce);
          else k_shadow_then_extend<false><<<gridShadow, kBlock, s->extraLds, st>>>(S, B, qCur, qNext, bounce);
        } else if (count) k_shadow<true><<<gridShadow, kBlock, s->extraLds, st>>>(S, B, qCur);
        else k_shadow<false><<<gridShadow, kBlock, s->extraLds, st>>>(S, B, qCur);
        if ((rc = endEvent(s, timing))) return rc;
        ++launchesShadow;
      }
      if ((rc = beginEvent(s, CAT_DIRECT, timing))) return rc;
      k_direct<<<grid, kBlock, 0, st>>>(S, B, qCur, bounce);
      if ((rc = endEvent(s, timing))) return rc;
      k_advance<<<1, 1, 0, st>>>(B, s->numPointLights, s->numAnyLights);
      std::swap(qCur, qNext);
    }
    if ((rc = beginEvent(s, CAT_ACCUM, timing))) return rc;
    uint32_t *nextCount = s->activeCount + (passes & 1u) * 32u;
    if (adaptive) HIP_TRY(hipMemsetAsync(nextCount, 0, sizeof(uint32_t), st));
    k_accumulate<<<gridFor(nPixPass), kBlock, 0, st>>>(B, s->acc, (int32_t)nPixPass, sppThis, adaptive ? params->adaptiveThreshold : 0.0f, adaptiveMin,
                                                       S.numRays, activeNext, nextCount);
    if ((rc = endEvent(s, timing))) return rc;
    if (adaptive) {
      // the next pass is sized by the number of pixels that go on: one small read-back per batch
      uint32_t left = 0;
      HIP_TRY(hipMemcpyAsync(&left, nextCount, sizeof left, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      nPixPass = left;
      activeCur = activeNext;
      activeNext = activeCur == s->activeA ? s->activeB : s->activeA;
    }
  }
  s->accValid = true;
  if (npix > 0) {
    if ((rc = beginEvent(s, CAT_RESOLVE, timing))) return rc;
    int applyExposure = (S.exposure >= 0 && !params->skipExposure) ? 1 : 0;
    k_resolve<<<gridFor(npix), kBlock, 0, st>>>(S, (int32_t)npix, s->pixList, s->acc, s->fb, s->packed, applyExposure);
    if ((rc = endEvent(s, timing))) return rc;
  }
  if (timing) HIP_TRY(hipEventRecord(s->events[totalIdx].b, st));
  HIP_TRY(hipGetLastError());
  if (rgbaOut && (rc = copyOut(s, rgbaOut, s->fb, fbPixels * sizeof(float4), st))) return rc;
  Counts hc{};
  HIP_TRY(hipMemcpyAsync(&hc, s->counts, sizeof(Counts), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->cameraSamples = hc.samples;
    stats->raysClosest = hc.raysClosest - hc.skipPoint - hc.planePoint;
    stats->raysAny = hc.raysAny - hc.skipAny - hc.planeAny;
    stats->raysSkipped = hc.skipPoint + hc.skipAny + hc.planePoint + hc.planeAny;
    stats->raysPlaneAnswered = hc.planePoint + hc.planeAny;
    stats->innerExpansions = hc.inner[0] + hc.inner[1];
    stats->primTests = hc.prims[0] + hc.prims[1];
    stats->planeTests = hc.planes[0] + hc.planes[1];
    stats->raysExtend = hc.raysExtend; stats->raysShadow = hc.raysShadow - hc.skipPoint - hc.skipAny - hc.planePoint - hc.planeAny;
    stats->innerExtend = hc.inner[0]; stats->primExtend = hc.prims[0];
    stats->innerShadow = hc.inner[1]; stats->primShadow = hc.prims[1];
    stats->laneUseInnerExtend = hc.innerWaveIters[0] > 0 ? hc.inner[0] / (64.0 * hc.innerWaveIters[0]) : 0.0;
    stats->laneUsePrimExtend = hc.primWaveIters[0] > 0 ? hc.prims[0] / (64.0 * hc.primWaveIters[0]) : 0.0;
    stats->laneUseInnerShadow = hc.innerWaveIters[1] > 0 ? hc.inner[1] / (64.0 * hc.innerWaveIters[1]) : 0.0;
    stats->laneUsePrimShadow = hc.primWaveIters[1] > 0 ? hc.prims[1] / (64.0 * hc.primWaveIters[1]) : 0.0;
    stats->passes = passes;
    stats->launchesExtend = launchesExtend;
    stats->launchesShadow = launchesShadow;
    stats->launchesFused = launchesFused;
    double ms[CAT_COUNT] = {0};
    for (size_t i = 0; i < s->eventsUsed; ++i) {
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, s->events[i].a, s->events[i].b));
      ms[s->events[i].category] += t;
    }
    stats->msTotal = ms[CAT_TOTAL]; stats->msRaygen = ms[CAT_RAYGEN]; stats->msExtend = ms[CAT_EXTEND];
    stats->msShade = ms[CAT_SHADE]; stats->msShadow = ms[CAT_SHADOW]; stats->msDirect = ms[CAT_DIRECT]; stats->msAccumulate = ms[CAT_ACCUM];
    stats->msResolve = ms[CAT_RESOLVE];
  }
  return RT_OK;
} catch (...) {
  return caught();
}

const float *rt_framebuffer_device(RtScene *s) { return s ? reinterpret_cast<const float *>(s->fb) : nullptr; }

int rt_framebuffer_gather(RtScene *dst, RtScene *const *srcs, int32_t numSrcs) try {
  if (!dst || (numSrcs > 0 && !srcs) || numSrcs < 0) return failMsg(RT_ERR_INVALID, "null argument");
  if (!dst->fb) return failMsg(RT_ERR_INVALID, "rt_render has not run on the destination scene");
  const int64_t n = (int64_t)dst->dev.width * dst->dev.height;
  for (int32_t i = 0; i < numSrcs; ++i) {
    const RtScene *src = srcs[i];
    if (!src || src == dst) return failMsg(RT_ERR_INVALID, "rt_framebuffer_gather needs scenes different from the destination");
    if (!src->fb || src->pixKey[0] < 0) return failMsg(RT_ERR_INVALID, "rt_render has not run on every source scene");
    if (n != (int64_t)src->dev.width * src->dev.height || n > dst->fbPixels || src->pixKey[0] != dst->dev.width || src->pixKey[1] != dst->dev.height)
      return failMsg(RT_ERR_INVALID, "the framebuffers have different sizes");
  }
  HIP_TRY(hipSetDevice(dst->device));
  if ((int32_t)dst->peers.size() < numSrcs) dst->peers.resize(numSrcs);
  // Copies and scatter kernels of the sources handled so far run on streams of their own and write dst->fb: whichever
  // way this function is left, they have finished before the caller can read, re-render or destroy anything.
  struct Drain {
    std::vector<PeerSlot> &slots;
    int32_t used = 0;
    ~Drain() { for (int32_t k = 0; k < used; ++k) if (slots[k].stream) (void)hipStreamSynchronize(slots[k].stream); }
  } drain{dst->peers};
  for (int32_t i = 0; i < numSrcs; ++i) {
    drain.used = i + 1;
    RtScene *src = srcs[i];
    PeerSlot &slot = dst->peers[i];
    const int64_t npix = src->pixCount;
    if (npix == 0) continue;
    if (!slot.stream) HIP_TRY(hipStreamCreateWithFlags(&slot.stream, hipStreamNonBlocking));
    if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    if (slot.cap < npix) {
      if (slot.idx) (void)hipFree(slot.idx);
      if (slot.staging) (void)hipFree(slot.staging);
      slot.idx = nullptr; slot.staging = nullptr; slot.cap = 0; slot.key[0] = -1;
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(&slot.idx), npix * sizeof(int32_t)));
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(&slot.staging), npix * sizeof(float4)));
      slot.cap = npix;
    }
    if (std::memcmp(slot.key, src->pixKey, sizeof slot.key) != 0) {
      // the positions of the source's pixels in the frame follow from its partition alone: rebuilt here and uploaded
      // once per partition, so that a gather moves only the pixels themselves between the GPUs
      RtRenderParams q{};
      q.partIndex = src->pixKey[2]; q.partCount = src->pixKey[3]; q.tileW = src->pixKey[4]; q.tileH = src->pixKey[5];
      std::vector<int32_t> pix = buildPixelList(q, src->pixKey[0], src->pixKey[1]);
      if ((int64_t)pix.size() != npix) return failMsg(RT_ERR_INVALID, "source partition does not match its pixel count");
      HIP_TRY(hipMemcpy(slot.idx, pix.data(), npix * sizeof(int32_t), hipMemcpyHostToDevice));
      std::memcpy(slot.key, src->pixKey, sizeof slot.key);
    }
    if (src->device != dst->device && std::find(dst->peerEnabled.begin(), dst->peerEnabled.end(), src->device) == dst->peerEnabled.end()) {
      hipError_t pe = hipDeviceEnablePeerAccess(src->device, 0);  // direct xGMI copies between the two GPUs
      if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();  // no direct path: the copy below still works, staged by the runtime
      }
      (void)hipGetLastError();
      dst->peerEnabled.push_back(src->device);
    }
    // rt_render has synchronised both scenes' streams; each source gets a stream of its own so that the copies of
    // different peers (different xGMI links) overlap
    HIP_TRY(hipMemcpyPeerAsync(slot.staging, dst->device, src->packed, src->device, npix * sizeof(float4), slot.stream));
    k_scatter_add<<<gridFor(npix), kBlock, 0, slot.stream>>>(dst->fb, slot.staging, slot.idx, npix);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(slot.done, slot.stream));
    HIP_TRY(hipStreamWaitEvent(dst->stream, slot.done, 0));
  }
  HIP_TRY(hipStreamSynchronize(dst->stream));
  return RT_OK;
} catch (...) {
  return caught();
}

int rt_framebuffer_add(RtScene *dst, RtScene *src) try {
  if (!dst || !src || dst == src) return failMsg(RT_ERR_INVALID, "rt_framebuffer_add needs two different scenes");
  RtScene *one[1] = {src};
  return rt_framebuffer_gather(dst, one, 1);
} catch (...) {
  return caught();
}

int rt_framebuffer_read(RtScene *s, float *rgbaOut) try {
  if (!s || !rgbaOut) return failMsg(RT_ERR_INVALID, "null argument");
  if (!s->fb) return failMsg(RT_ERR_INVALID, "rt_render has not run on this scene");
  HIP_TRY(hipSetDevice(s->device));
  int64_t n = (int64_t)s->dev.width * s->dev.height;
  if (n > s->fbPixels) return failMsg(RT_ERR_INVALID, "framebuffer is smaller than the current image size");
  if (int rc = copyOut(s, rgbaOut, s->fb, n * sizeof(float4), s->stream)) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RT_OK;
} catch (...) {
  return caught();
}

int rt_resolve_rgba8(RtScene *s, uint8_t *rgba8Out) try {
  if (!s || !rgba8Out) return failMsg(RT_ERR_INVALID, "null argument");
  if (!s->fb) return failMsg(RT_ERR_INVALID, "rt_render has not run on this scene");
  HIP_TRY(hipSetDevice(s->device));
  int64_t n = (int64_t)s->dev.width * s->dev.height;
  if (n > s->fbPixels) return failMsg(RT_ERR_INVALID, "framebuffer is smaller than the current image size");
  int rc;
  if ((rc = ensureWork(s, (size_t)n * 4))) return rc;
  uchar4 *d = static_cast<uchar4 *>(s->work);
  k_srgb8<<<gridFor(n), kBlock, 0, s->stream>>>(s->fb, d, (int32_t)n);
  HIP_TRY(hipGetLastError());
  if (int rc = copyOut(s, rgba8Out, d, n * 4, s->stream)) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RT_OK;
} catch (...) {
  return caught();
}

float rt_last_trace_kernel_ms(RtScene *s) { return s ? s->lastKernelMs : 0.0f; }
float rt_scene_bvh_build_ms(RtScene *s) { return s ? s->bvhBuildMs : 0.0f; }

}  // extern "C"
