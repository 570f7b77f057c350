// BVH traversal and primitive intersection for one ray per lane.
//
// Visit rule = the reference's BVH::findClosestObject (src/acceleration/BVH.cpp:283-340): children are tested with
// the UN-normalised direction's reciprocal, the nearer child (right only if strictly nearer) is visited first, a
// node is skipped when its entry distance is not below the current closest t (which is in NORMALISED units — the
// mismatch is the reference's and is kept), leaf objects are tested in array order and replace the closest hit only
// on a strict `<`. Two things differ from the CPU loop without changing what is visited or returned:
//   * the nearer child is kept in a register instead of being pushed and immediately popped;
//   * a child whose entry distance is already >= the closest t is not pushed at all. The closest t only shrinks, so
//     the reference would pop and skip that entry anyway.
// The per-lane stack keeps its top K entries in LDS (8 bytes each: entry distance + node reference, one column per
// lane, so no bank conflicts between lanes) and spills older entries to per-lane scratch memory.
#pragma once
#include <type_traits>

#include "rt_dev_scene.h"

namespace rtd {

struct HitRec {
  float t;
  int32_t prim;  // BVH position of the object, numPrims + k for plane k, -1 = miss
};

// src/acceleration/BVH.cpp:390-409
RT_DEV float intersectAABB(V3 o, V3 inv, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
  float tx_near = (mnx - o.x) * inv.x;
  float tx_far = (mxx - o.x) * inv.x;
  float tmin = rmin(tx_near, tx_far);
  float tmax = rmax(tx_near, tx_far);
  float ty_near = (mny - o.y) * inv.y;
  float ty_far = (mxy - o.y) * inv.y;
  tmin = rmax(tmin, rmin(ty_near, ty_far));
  tmax = rmin(tmax, rmax(ty_near, ty_far));
  float tz_near = (mnz - o.z) * inv.z;
  float tz_far = (mxz - o.z) * inv.z;
  tmin = rmax(tmin, rmin(tz_near, tz_far));
  tmax = rmin(tmax, rmax(tz_near, tz_far));
  return (tmax >= tmin && tmax > 0) ? tmin : kInf;
}

struct TravCounters {
  uint32_t planeAnsweredPoint, planeAnsweredAny;  // always counted: visibility queries the planes answered (no BVH traversal)
  uint32_t inner, prims, planes;
  // wave-level iteration counts (each active lane adds 1/active, so the wave sums to 1 per iteration): with `inner`
  // and `prims` they give the fraction of the 64 lanes doing useful work in the two hot loops
  float innerWaveIters, primWaveIters;
};

// Tests object i against the ray; returns true and updates the closest hit on a strict `<` (src/acceleration/BVH.cpp:309-313).
RT_DEV bool intersectPrim(const DevScene &S, float4 a, float4 b, float4 c, int32_t i, V3 o, V3 nd, float &tBest, int32_t &primBest) {
  float t;
  bool sphere = S.hasSpheres && ((__float_as_uint(a.w) | __float_as_uint(b.x) | __float_as_uint(b.y)) == 0u);
  if (!sphere) {
    // Triangle::intersect, src/scene/Object.cpp:195-213 (direction normalised once per ray: same expression)
    V3 p1 = v3(a.x, a.y, a.z), n = v3(a.w, b.x, b.y), e1 = v3(b.z, b.w, c.x), e2 = v3(c.y, c.z, c.w);
    t = dot(p1 - o, n) / dot(nd, n);
    if (t < 0) return false;
    V3 q = (t * nd + o) - p1;
    float b2 = dot(e1, q);
    float b3 = dot(e2, q);
    float b1 = 1.0f - b3 - b2;
    if (b1 < 0 || b1 > 1 || b2 < 0 || b2 > 1 || b3 < 0 || b3 > 1) return false;
  } else {
    // Sphere::intersect, src/scene/Object.cpp:75-99
    V3 center = v3(a.x, a.y, a.z);
    float r = b.z;
    float radiusSquared = r * r;
    V3 dfs = center - o;
    bool inside = dot(dfs, dfs) < radiusSquared;
    float tcen = dot(dfs, nd);
    if (inside == false && tcen < 0) return false;
    V3 dd = (o + tcen * nd) - center;
    float distanceSquared = dot(dd, dd);
    if (inside == false && radiusSquared < distanceSquared) return false;
    float tOffset = sqrtf(radiusSquared - distanceSquared);
    t = inside ? tcen + tOffset : tcen - tOffset;
  }
  if (t < tBest) { tBest = t; primBest = i; return true; }
  return false;
}

// Slab test for rays whose reciprocal direction is finite and non-zero in every component (and whose origin is
// finite): then no product below can be NaN, and IEEE minNum/maxNum (v_min_f32 / v_max_f32) return exactly what the
// reference's compare-and-select chain returns, up to the sign of a zero, which no later comparison can see.
RT_DEV float intersectAABBFinite(V3 o, V3 inv, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
  float tx_near = (mnx - o.x) * inv.x, tx_far = (mxx - o.x) * inv.x;
  float ty_near = (mny - o.y) * inv.y, ty_far = (mxy - o.y) * inv.y;
  float tz_near = (mnz - o.z) * inv.z, tz_far = (mxz - o.z) * inv.z;
  float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx_near, tx_far), __builtin_fminf(ty_near, ty_far)),
                               __builtin_fminf(tz_near, tz_far));
  float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx_near, tx_far), __builtin_fmaxf(ty_near, ty_far)),
                               __builtin_fmaxf(tz_near, tz_far));
  return (tmax >= tmin && tmax > 0) ? tmin : kInf;
}

// The two slab tests above without their final select: entry and exit parameter of the box, for callers that fold
// "the box is hit" into their own comparison chain (hit <=> tmax >= tmin && tmax > 0).
RT_DEV void slabExact(V3 o, V3 inv, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float &tmin, float &tmax) {
  float tx_near = (mnx - o.x) * inv.x, tx_far = (mxx - o.x) * inv.x;
  tmin = rmin(tx_near, tx_far); tmax = rmax(tx_near, tx_far);
  float ty_near = (mny - o.y) * inv.y, ty_far = (mxy - o.y) * inv.y;
  tmin = rmax(tmin, rmin(ty_near, ty_far)); tmax = rmin(tmax, rmax(ty_near, ty_far));
  float tz_near = (mnz - o.z) * inv.z, tz_far = (mxz - o.z) * inv.z;
  tmin = rmax(tmin, rmin(tz_near, tz_far)); tmax = rmin(tmax, rmax(tz_near, tz_far));
}
RT_DEV void slabFinite(V3 o, V3 inv, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float &tmin, float &tmax) {
  float tx_near = (mnx - o.x) * inv.x, tx_far = (mxx - o.x) * inv.x;
  float ty_near = (mny - o.y) * inv.y, ty_far = (mxy - o.y) * inv.y;
  float tz_near = (mnz - o.z) * inv.z, tz_far = (mxz - o.z) * inv.z;
  tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx_near, tx_far), __builtin_fminf(ty_near, ty_far)), __builtin_fminf(tz_near, tz_far));
  tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx_near, tx_far), __builtin_fmaxf(ty_near, ty_far)), __builtin_fmaxf(tz_near, tz_far));
}

// The closest plane along a ray (origin o, normalised direction nd). The reference tests the planes after the BVH
// (src/scene/raytracer.cpp:174-196), each with Plane::intersect (src/scene/Object.cpp:133-141), keeping the first plane in
// scene order that reaches the smallest t; that plane does not depend on the BVH part of the query. Returns the number of
// planes tested.
RT_DEV int closestPlane(const DevScene &S, V3 o, V3 nd, float &tPlane, int32_t &plane) {
  tPlane = kInf; plane = -1;
  for (int32_t k = 0; k < S.numPlanes; ++k) {
    const DevPlane &pl = S.planes[k];
    V3 pn = v3(pl.normal), pp = v3(pl.point);
    float t = dot(pp - o, pn) / dot(nd, pn);
    if (t < 0) continue;
    if (t < tPlane) { tPlane = t; plane = k; }
  }
  return S.numPlanes;
}
// A visibility query asks a yes/no question that a plane alone can settle: findAnyObject is non-null as soon as any plane
// is hit (src/scene/raytracer.cpp:177-183), and a point-light query is in shadow as soon as a plane lies nearer than the
// bulb (the closest hit overall is then nearer still; src/scene/Object.cpp:18-25). Such a query needs no traversal.
RT_DEV bool planeSettlesVisibility(bool pointLight, V3 o, V3 nd, float bulbDist, float tPlane, int32_t plane) {
  if (plane < 0) return false;
  if (!pointLight) return true;
  V3 hp = tPlane * nd + o;
  return magnitude(o - hp) < bulbDist;
}

#ifndef RT_CHUNK_MAX
#define RT_CHUNK_MAX 128
#endif
#ifndef RT_CHUNK_MAX_FILTERED
#define RT_CHUNK_MAX_FILTERED 256
#endif
#ifndef RT_CHUNK_DIV
#define RT_CHUNK_DIV 16
#endif
// A queue is handed out in kQueueSegments equal, contiguous segments with a ticket counter each (128 bytes apart). A
// workgroup starts in segment blockIdx.x % kQueueSegments — workgroups are dealt round-robin to the eight accelerator
// dies (XCDs), each with an L2 of its own — and moves on to the next segment when its own is used up. Queues are in
// image order, so a die's wavefronts trace rays that start in one band of the image and its L2 holds that band's
// part of the tree instead of a copy of what the other seven hold.
#ifndef RT_QUEUE_SEGMENTS
#define RT_QUEUE_SEGMENTS 8
#endif
constexpr int kQueueSegments = RT_QUEUE_SEGMENTS;
constexpr int kTicketStride = 32;  // dwords between the ticket counters of neighbouring segments
// Indices per atomic: at most 128 of a dense queue, 256 of a sparse one (where under half of them hold a query). Smaller
// chunks leave less for the last wavefronts of a launch (1/8 of the bench frame: k_extend 27.1 -> 25.9 ms; no difference on
// the whole frame), still smaller ones make the ticket counters hot. A prepare step takes up to 64 indices: a chunk must
// cover one.
constexpr uint32_t kTicketChunkMax = RT_CHUNK_MAX, kTicketChunkMin = 64, kChunksPerWave = RT_CHUNK_DIV;

enum { TRAV_CLOSEST = 0, TRAV_ANY = 1, TRAV_POINT_LIGHT = 2, TRAV_SKIP = 3 };
// Added to TRAV_POINT_LIGHT while a query travels through the engine: its answer does not depend on the order in which
// its subtrees are visited (see "SPLIT" in traverseQueue). A TRAV_ANY query never depends on it.
constexpr int TRAV_ORDER_FREE = 4;

// Scheduling thresholds of the traversal engine (lanes of one 64-lane wavefront).
#ifndef RT_POOL_SLOTS
#define RT_POOL_SLOTS 32
#endif
#ifndef RT_POOL_LOW
#define RT_POOL_LOW (RT_POOL_SLOTS / 4)
#endif
#ifndef RT_STAY_INNER
#define RT_STAY_INNER 3
#endif
#ifndef RT_STAY_PRIM
#define RT_STAY_PRIM 1
#endif
#ifndef RT_DONE_BREAK
#define RT_DONE_BREAK 12
#endif
#ifndef RT_TAKE_MIN
#define RT_TAKE_MIN 8
#endif
constexpr int kPoolSlots = RT_POOL_SLOTS;    // prepared queries a wavefront keeps in LDS (power of two)
constexpr int kPoolFields = 13;   // dwords per prepared query
constexpr int kPoolLow = RT_POOL_LOW;      // prepare more queries once at most this many are left
constexpr int kDoneBreak = RT_DONE_BREAK;  // a phase loop ends early once this many lanes have finished their query
constexpr int kTakeMin = RT_TAKE_MIN;      // lanes without work (done or idle) before results are written and new queries taken

// Limits the host checks before a scene reaches the GPU (rt_scene_create): node and object records are addressed
// with 32-bit byte offsets, and the per-lane stack has the reference's 64 entries (src/acceleration/BVH.cpp:10), which
// a tree of depth <= 64 cannot overflow (near-first descent keeps at most one pending sibling per level).
constexpr int64_t kMaxInnerNodes = (int64_t)1 << 26;      // * 64 B  < 4 GiB
constexpr int64_t kMaxObjects = ((int64_t)1 << 32) / 48;  // * 48 B  < 4 GiB
constexpr int kMaxTreeDepth = kMaxStack;

// Traversal engine. Every wavefront pulls queries from a queue (one atomic per chunk of 64-256 indices) and traces one
// query per lane. Each lane always has exactly one pending operation:
//     INNER   test the two child boxes of an inner node, descend / push / pop
//     PRIM    test one object of the current leaf
//     DONE    the BVH part of its query is over: the result has to be written out
//     IDLE    no query
// A wavefront keeps a small pool of PREPARED queries in LDS. Preparing is the expensive start of a query (load it,
// normalise the direction, three IEEE divisions, the planes, the root box) and runs for up to kPoolSlots queries at
// once, on all lanes, whatever their own query is doing meanwhile: their registers are not touched. A query that is
// decided at this point (a plane answers it, it misses the root box, its answer cannot matter) is written out at once
// and never occupies a lane. At every vote, lanes in DONE write their result (a few instructions) and lanes without a
// query take the next prepared one from the pool (14 LDS reads), so a lane is idle only until the current phase loop
// ends, not until two dozen others have finished too.
// The wavefront then votes: the more popular of INNER / PRIM runs in a tight loop of its own until too few of its lanes
// are left (or too many lanes are done), while the minority waits until it has become the majority. Incoherent rays
// therefore do not serialise the wavefront through both loop bodies every iteration. Per query, nodes and objects are
// visited in exactly the reference's order. The grid is sized to fill the GPU once (persistent wavefronts); a wavefront
// ends when the queue is exhausted, its pool is empty and its own lanes are done, so nothing ever waits on another wavefront.
//
// SPLIT. The last wavefronts of a launch run with most lanes idle, and a launch is over only when its longest query is: on
// the bench scene a few visibility queries per frame take 1 500 - 5 700 steps (paths that wandered 1e7 units out along an
// infinite plane: from there every box of the scene passes the slab test by a coin flip of rounding), a hundred times the
// average, at one memory round trip per step. Once the queue is exhausted and the pool empty, every lane that is still
// working on a visibility query whose answer cannot depend on the order of the visits hands the top entry of its stack to
// an idle lane, as a prepared query that starts at that subtree; the helpers do the same at the next vote, so the lanes
// at work double per vote. Only for sources whose store() ignores "not occluded" (Source::kVisibilitySplit): every part
// reports on its own, and the query is occluded if any part says so.
//   TRAV_ANY: the first object the loop accepts ends the query and until then the closest t is infinite, so no box is ever
//     culled by it: the query is occluded iff some object, reachable through boxes that pass the slab test, is hit.
//   TRAV_POINT_LIGHT, under the conditions checked at prepare time (TRAV_ORDER_FREE): the predicate |o - hit| < bulbDist
//     holds for every 0 <= t <= bulbDist / 2 (|o|_1 <= 2^14 bulbDist bounds the rounding of the hit point, bulbDist < 1e12
//     keeps its squares finite), so a hit the loop accepts without ending there has t > bulbDist / 2, and so has the
//     closest t as long as the query runs. (No accepted t is below zero by more than rounding: a triangle rejects t < 0,
//     and a sphere's t = tcen - tOffset with tcen >= tOffset up to the rounding of |centre - o| and the radius, which the
//     scene's extent bounds: S.cutEps, 2^-20 of the extent at least, must not exceed bulbDist / 64, and the root box
//     lies within bulbDist / 4 of the origin. S.cutEps is zero for scenes with unbounded objects: no split there.) Every box of
//     the tree lies inside the root box and the slab test is monotone in the box's coordinates (finite boxes, finite ray),
//     so no box is entered at a parameter beyond the root box's exit parameter, which is below bulbDist / 4: the
//     culling comparison `entry < closest t` (src/acceleration/BVH.cpp:302-305, parameter against distance as in the
//     reference) is true whenever it is made, in any order, and an occluder (t below the threshold of the predicate, hence
//     below the closest t) is accepted whenever it is tested. The query is in shadow iff some reachable object has such a t.
// Counting launches (COUNT) do not split, so that their step counts stay those of the reference's loop.
//
// Query kinds:
//   TRAV_CLOSEST      Scene::findClosestObject (src/scene/raytracer.cpp:187-196)
//   TRAV_ANY          the boolean of Scene::findAnyObject (:174-185): stops at the first object the closest-hit loop
//                     accepts; which object that is depends on the order, whether one exists does not
//   TRAV_POINT_LIGHT  the closest-hit query of PointLight::pointInShadow (src/scene/Object.cpp:18-25), stopping at the
//                     first accepted hit that already satisfies its predicate |origin - hit| < bulbDist. Every float
//                     operation between t and that length is monotone in t, and later accepted hits only have smaller
//                     t, so the final closest hit would satisfy the predicate too: same answer, less work.
//   TRAV_SKIP         (returned by Source::fetch) the answer of this query cannot matter: nothing is traced or stored
// RUNTIME_MODE = false compiles the TRAV_CLOSEST loop only.
//
// Source (by reference):
//   uint32_t total()                                                       number of queries in the queue
//   uint32_t fetch(uint32_t idx, V3 &o, V3 &d, int &mode, float &bulbDist) load query idx, return a tag for store()
//   void store(uint32_t tag, HitRec hit, bool occluded, V3 o, V3 nd)       deliver its result
//   static constexpr bool kPlanesDone                                      the planes were tested before the query was queued
//   static constexpr bool kVisibilitySplit                                 store() of a visibility query does nothing unless `occluded`, and may
//                                                                          be called several times for one tag (see "SPLIT" below)
//   static constexpr bool kFiltered                                        the queue is sparse and the source can tell cheaply where its
//                                                                          queries are; it then also has
//   int classify(uint32_t idx)                                             0 = nothing there, 1 = a query to fetch, other = see passedOver
//   void passedOver(uint32_t idx, int kind)                                called once for every index classified > 1
//
// LDS the caller provides: ldsStack = K * BLOCK uint2 (stack columns), ldsPool = (BLOCK / 64) * kPoolFields * kPoolSlots dwords.

// Lane states of the traversal engine (see traverseQueue).
enum { OP_IDLE = 0, OP_INNER = 1, OP_PRIM = 2, OP_DONE = 3, OP_DONE_EARLY = 4 };

// One INNER operation on the record (a, b, c, r) of the lane's current node: both child boxes, then descend, push or
// pop. A leaf reached here is only noted (pi = its reference, pe = 0); its object range is decoded when the lane's
// first PRIM operation runs. A free function with the lane state passed by reference (not a lambda of traverseQueue):
// the compiler keeps every one of these in registers this way.
template <bool EXACT, int K, int BLOCK>
RT_DEV void innerStep(float4 a, float4 b, float4 c, int4 r, V3 o, V3 inv, float lim, uint2 *stackCol, uint2 *spill, int &sp, int &lo,
                      int32_t &cur, int32_t &pi, int32_t &pe, int &op) {
  float tminL, tmaxL, tminR, tmaxR;
  if (EXACT) {
    slabExact(o, inv, a.x, a.y, a.z, a.w, b.x, b.y, tminL, tmaxL);
    slabExact(o, inv, b.z, b.w, c.x, c.y, c.z, c.w, tminR, tmaxR);
  } else {
    slabFinite(o, inv, a.x, a.y, a.z, a.w, b.x, b.y, tminL, tmaxL);
    slabFinite(o, inv, b.z, b.w, c.x, c.y, c.z, c.w, tminR, tmaxR);
  }
  // A child is visited if its box is hit (src/acceleration/BVH.cpp:390-409: tmax >= tmin && tmax > 0) at an entry
  // distance below the closest hit (`lim`); near child first, right only if strictly nearer (:322-333). A child that
  // cannot beat the closest hit any more is dropped here instead of at pop time.
  const bool hl = tmaxL >= tminL && tmaxL > 0 && tminL < lim;
  const bool hr = tmaxR >= tminR && tmaxR > 0 && tminR < lim;
  const bool both = hl && hr;
  const bool swap = both && tminR < tminL;
  if (both) {
    if (sp - lo == K) { spill[lo] = stackCol[(lo & (K - 1)) * BLOCK]; ++lo; }
    stackCol[(sp & (K - 1)) * BLOCK] = make_uint2(__float_as_uint(swap ? tminL : tminR), (uint32_t)(swap ? r.x : r.y));
    ++sp;
  }
  int32_t next = (hl && !swap) ? r.x : r.y;
  if (!(hl || hr)) {
    next = kNoBvh;
    while (sp > 0) {
      --sp;
      // A spilled entry is first copied back into its LDS slot (rare), so that the read below is always a plain LDS read
      uint2 *slot = stackCol + (sp & (K - 1)) * BLOCK;
      if (sp < lo) { lo = sp; *slot = spill[sp]; }
      uint2 e = *slot;
      if (__uint_as_float(e.x) < lim) { next = (int32_t)e.y; break; }
    }
  }
  if (next < 0) { pi = next; pe = 0; op = OP_PRIM; }
  else if (next == kNoBvh) op = OP_DONE;
  else cur = next;
}

// One PRIM operation on the record (a, b, c) of object pi (expanded inside traverseQueue, where the lane state lives).
#define RT_PRIM_STEP()                                                                                              \
  do {                                                                                                              \
    bool accepted = intersectPrim(S, a, b, c, pi, o, nd, tBest, primBest);                                          \
    ++pi;                                                                                                           \
    if (RUNTIME_MODE && accepted && mode != TRAV_CLOSEST) {                                                         \
      if (mode == TRAV_ANY) {                                                                                       \
        op = OP_DONE_EARLY;                                                                                         \
      } else {                                                                                                      \
        /* |point - info.point| < |center - point| with info.point = t * nd + origin (src/scene/Object.cpp:22-24) */ \
        V3 hp = tBest * nd + o;                                                                                     \
        if (magnitude(o - hp) < bulbDist) op = OP_DONE_EARLY;                                                       \
      }                                                                                                             \
    }                                                                                                               \
    if (op == OP_PRIM && pi == pe) popNext();                                                                       \
  } while (0)

template <class Source, bool RUNTIME_MODE, bool COUNT, int K, int BLOCK>
RT_DEV void traverseQueue(const DevScene &S, Source &src, uint32_t *ticket, uint2 *ldsStack, uint32_t *ldsPool, TravCounters &tc) {
  static_assert((K & (K - 1)) == 0, "K must be a power of two");
  static_assert((kPoolSlots & (kPoolSlots - 1)) == 0 && kPoolSlots <= 64, "pool size");
  constexpr bool kSplit = RUNTIME_MODE && Source::kVisibilitySplit && !COUNT;
  const uint32_t total = src.total();
  const uint32_t laneId = __lane_id();
  const char *nodeBase = reinterpret_cast<const char *>(S.nodes);
  const char *primBase = reinterpret_cast<const char *>(S.primHit);
  uint2 *stackCol = ldsStack + threadIdx.x;
  uint32_t *pool = ldsPool + (threadIdx.x >> 6) * (kPoolFields * kPoolSlots);  // [field][slot]
  uint2 spill[kMaxStack - K];
  int op = OP_IDLE;
  uint32_t item = 0;
  V3 o = v3(0, 0, 0), nd = o, inv = o;
  float tBest = kInf, bulbDist = 0.0f;
  float cutLimit = kInf;   // TRAV_POINT_LIGHT: boxes entered at or beyond this parameter (units of |d|) hold no occluder (see prepare)
  float tPlane = kInf;     // closest plane along the ray (found when the query was prepared)
  int32_t planeBest = -1;
  int32_t primBest = -1, cur = 0, pi = 0, pe = 0;
  int sp = 0, lo = 0, mode = TRAV_CLOSEST;
  bool finite = true;
  bool exhausted = false;  // wave-uniform: the queue has no more queries
  bool waveExact = false;  // wave-uniform: some lane holds a ray the min/max slab test is not valid for
  uint32_t chunkPos = 0, chunkEnd = 0;
  uint32_t seg = blockIdx.x % (uint32_t)kQueueSegments, segsLeft = (uint32_t)kQueueSegments;  // wave-uniform
  uint32_t poolHead = 0, poolCount = 0;  // wave-uniform ring of prepared queries
  const uint32_t chunkMax = Source::kFiltered ? (uint32_t)RT_CHUNK_MAX_FILTERED : kTicketChunkMax;
  const uint32_t kTicketChunk = min(chunkMax, max(kTicketChunkMin, total / (gridDim.x * (BLOCK / 64) * kChunksPerWave)));

  // object range [pi, pe) of a leaf reference (pe > 0 always: a leaf holds at least one object)
  auto setLeaf = [&](int32_t leaf) {
    int32_t count = (leaf >> 24) & 0x7f;
    pi = leaf & 0xffffff;
    if (count == 0) {
      BigLeaf bl = S.bigLeaves[leaf & 0x7fffffff];
      pi = bl.start; count = bl.count;
    }
    pe = pi + count;
  };
  // an inner node becomes an INNER operation, a leaf a run of PRIM operations
  auto enter = [&](int32_t next) {
    if (next >= 0) {
      cur = next; op = OP_INNER;
    } else {
      setLeaf(next);
      op = OP_PRIM;
    }
  };
  // next stack entry whose box entry distance is still below the closest hit (src/acceleration/BVH.cpp:302-305);
  // none left: the BVH part of the query is over
  auto popNext = [&]() {
    op = OP_DONE;
    while (sp > 0) {
      --sp;
      // A spilled entry is first copied back into its LDS slot (rare), so that the read below is always a plain LDS
      // read of one address; reading "LDS or spill" directly makes the compiler select between the two addresses
      // and issue a generic (flat) load.
      uint2 *slot = stackCol + (sp & (K - 1)) * BLOCK;
      if (sp < lo) { lo = sp; *slot = spill[sp]; }
      uint2 e = *slot;
      if (__uint_as_float(e.x) < tBest) { enter((int32_t)e.y); break; }
    }
  };
  // The end of a query. The closest plane was found when the query was prepared; the reference tests the planes after
  // the BVH and lets a plane win only on a strict `<` (src/scene/raytracer.cpp:190-195), which is this comparison.
  // Visibility queries: a plane that answers the question alone ended the query before it reached a lane, and object
  // hits of a point-light query were checked when they were accepted, so only TRAV_ANY is left to decide here.
  auto deliver = [&](uint32_t tag, bool early, int qmode, float tB, int32_t pB, float tP, int32_t plane, V3 qo, V3 qnd) {
    bool occluded = early;
    if (!early) {
      if (plane >= 0 && tP < tB) { tB = tP; pB = S.numPrims + plane; }
      if (!(RUNTIME_MODE && (qmode & 3) == TRAV_POINT_LIGHT)) occluded = pB >= 0;
    }
    HitRec h;
    h.t = pB >= 0 ? tB : kInf;
    h.prim = pB;
    src.store(tag, h, occluded, qo, qnd);
  };

  // INNER phase: every lane with op == OP_INNER expands its node once per iteration, until too few such lanes are left.
  // A leaf reached here is only noted (pi = its reference, pe = 0); its object range is decoded when the lane's first
  // PRIM operation runs, outside this loop.
  // One INNER operation on the record (a, b, c, r) of node `cur`.
  float4 a = make_float4(0, 0, 0, 0), b = a, c = a;  // the record the current step works on
  int4 r = make_int4(0, 0, 0, 0);
  while (true) {
    const int nWorking = __popcll(__ballot(op == OP_INNER || op == OP_PRIM));
    const bool service = 64 - nWorking >= kTakeMin || nWorking == 0;
    // ------------------------------------------------------------------ DONE lanes write their result and become idle
    if (service && __ballot(op >= OP_DONE) != 0ull) {
      if (op >= OP_DONE) {
        deliver(item, op == OP_DONE_EARLY, mode, tBest, primBest, tPlane, planeBest, o, nd);
        op = OP_IDLE;
        finite = true;
      }
    }
    // ------------------------------------------------------------------ PREPARE queries into the pool
    // Runs on lanes 0 .. n-1 whatever their own state: only the variables declared inside are written.
    if (service && !exhausted && poolCount <= (uint32_t)kPoolLow) {
      const uint32_t want = (uint32_t)kPoolSlots - poolCount;  // <= 64
      const uint32_t avail = chunkEnd - chunkPos;
      // Queue indices come from a wave-local chunk [chunkPos, chunkEnd); one atomic on a shared counter buys
      // kTicketChunk of them, so the counters (single addresses many wavefronts hit) see few operations.
      const bool needChunk = Source::kFiltered ? avail == 0u : avail < want;
      uint32_t idx = chunkPos + laneId;  // before the chunk changes: what is left of the old one is still handed out
      uint32_t base = 0, end = 0;
      if (needChunk) {
        while (segsLeft > 0) {
          const uint32_t segStart = (uint32_t)(((uint64_t)total * seg) / (uint32_t)kQueueSegments);
          const uint32_t segEnd = (uint32_t)(((uint64_t)total * (seg + 1)) / (uint32_t)kQueueSegments);
          uint32_t got = 0;
          if (laneId == 0) got = atomicAdd(ticket + seg * kTicketStride, kTicketChunk);
          got = __shfl(got, 0);
          if (got < segEnd - segStart) {
            base = segStart + got;
            end = min(base + kTicketChunk, segEnd);
            break;
          }
          seg = (seg + 1) % (uint32_t)kQueueSegments;
          --segsLeft;
        }
        if (segsLeft == 0) { exhausted = true; chunkPos = chunkEnd = 0; }
      }
      const bool gotChunk = needChunk && !exhausted;
      bool mine = false;  // this lane prepares query idx
      if constexpr (Source::kFiltered) {
        // The source knows cheaply which indices hold a query at all: a window of up to 64 indices is classified, and as
        // many of them are used up as the pool has room for wanted ones. A sparse queue fills the pool in one round.
        if (gotChunk) { chunkPos = base; chunkEnd = end; idx = chunkPos + laneId; }
        const uint32_t window = min(64u, chunkEnd - chunkPos);  // 0 once the queue is exhausted
        const int kind = laneId < window ? src.classify(idx) : 0;
        const unsigned long long mWanted = __ballot(kind == 1);
        uint32_t used = window;
        if ((uint32_t)__popcll(mWanted) > want) {  // stop in front of the first wanted index that does not fit
          const uint32_t rank = (uint32_t)__popcll(mWanted & ((1ull << laneId) - 1ull));
          used = (uint32_t)__ffsll((long long)__ballot(kind == 1 && rank == want)) - 1u;
        }
        if (laneId < used && kind > 1) src.passedOver(idx, kind);
        mine = laneId < used && kind == 1;
        chunkPos += used;
      } else {
        bool fresh = false;  // lanes >= avail got an index from a new chunk
        if (gotChunk) {
          fresh = true;
          if (laneId >= avail) idx = base + (laneId - avail);
          chunkPos = min(base + (want - avail), end);
          chunkEnd = end;
        } else if (!needChunk) {
          chunkPos += want;
        }
        mine = laneId < want && (laneId < avail || (fresh && idx < chunkEnd));
      }
      bool pooled = false;
      V3 qo = v3(0, 0, 0), qnd = qo, qinv = qo;
      float qtPlane = kInf, qbulb = 0.0f, qcut = kInf;
      int32_t qplane = -1, qfirst = kNoBvh;
      int qmode = TRAV_CLOSEST;
      bool qfinite = true;
      uint32_t qitem = 0;
      if (mine) {
        V3 d;
        qitem = src.fetch(idx, qo, d, qmode, qbulb);
        if (!(RUNTIME_MODE && qmode == TRAV_SKIP)) {
          qnd = normalized(d);
          // Planes first: the closest plane is found here and compared with the BVH's answer at the end. A visibility
          // query that a plane alone settles never reaches a lane; it is counted apart (TravCounters::planeAnswered*).
          // A source that has dealt with the planes itself (Source::kPlanesDone) only hands over queries no plane settles.
          bool planeAnswers = false;
          if (!Source::kPlanesDone) {
            const int tested = closestPlane(S, qo, qnd, qtPlane, qplane);
            if (COUNT) tc.planes += (uint32_t)tested;
            if (RUNTIME_MODE && qmode != TRAV_CLOSEST && planeSettlesVisibility(qmode == TRAV_POINT_LIGHT, qo, qnd, qbulb, qtPlane, qplane)) {
              planeAnswers = true;
              if (qmode == TRAV_POINT_LIGHT) ++tc.planeAnsweredPoint; else ++tc.planeAnsweredAny;
            }
          }
          if (planeAnswers) {
            deliver(qitem, true, qmode, qtPlane, S.numPrims + qplane, qtPlane, qplane, qo, qnd);
          } else {
            // A point-light query only asks whether the closest hit lies nearer than the bulb, and ends as soon as an
            // accepted hit is (RT_PRIM_STEP). Which boxes may be skipped because they lie beyond the bulb (parameter 1 in
            // units of |d|)? The slab test's entry parameter alone is no safe bound: an object test accepts a hit whose
            // point P = o + u d lies up to eps OUTSIDE the object's box (rounding of the barycentric test), and on an axis
            // the ray is almost parallel to, eps of position is eps / |d_k| of parameter: several percent, or anything.
            // What does hold, per axis, is u >= entry_k - eps / |d_k|; with margin = eps * max_k |1 / d_k| the box's entry
            // parameter (the largest entry_k) exceeds the u of every hit inside by at most `margin`. A box entered beyond
            // 1 + 2 margin therefore holds only hits beyond 1 + margin: none of them is an occluder, and none of them can
            // lower the closest hit below what it takes to cull a box that still may hold one (entry below 1 + margin), so
            // skipping them changes neither the answer nor what the reference's own culling (src/acceleration/BVH.cpp:302-305)
            // would have visited among the boxes that matter. Rays with a margin above 2 % get no cut, nor do bulbs nearer
            // than one unit: there t (in units of 1) is smaller than the parameter, and a hit BEYOND the bulb can make the
            // reference cull a box in front of it.
            if (S.rootRef != kNoBvh) {
              qinv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // src/acceleration/BVH.cpp:293
              const bool rayFinite = isfinite(qinv.x) && isfinite(qinv.y) && isfinite(qinv.z) && qinv.x != 0 && qinv.y != 0 && qinv.z != 0 &&
                                     isfinite(qo.x) && isfinite(qo.y) && isfinite(qo.z);
              qfinite = S.boxesFinite && rayFinite;
              if (RUNTIME_MODE && qmode == TRAV_POINT_LIGHT && S.cutEps > 0.0f && rayFinite && qbulb >= 1.0f) {  // cutEps > 0: every box is finite
                const float eps = S.cutEps + 9.5367431640625e-07f * (fabsf(qo.x) + fabsf(qo.y) + fabsf(qo.z) + qbulb);  // 2^-20
                const float margin = eps * __builtin_fmaxf(__builtin_fmaxf(fabsf(qinv.x), fabsf(qinv.y)), fabsf(qinv.z));
                if (margin <= 0.02f) {
                  // kept as the upper 16 bits of a float, rounded up (a larger limit only culls less)
                  const uint32_t bits = (__float_as_uint(1.0f + 2.0f * margin + 1.52587890625e-05f) + 0xffffu) & 0xffff0000u;
                  qcut = __uint_as_float(bits);
                }
              }
              float rootIn, rootOut;  // intersectAABB (src/acceleration/BVH.cpp:390-409), keeping the exit parameter
              slabExact(qo, qinv, S.rootMin[0], S.rootMin[1], S.rootMin[2], S.rootMax[0], S.rootMax[1], S.rootMax[2], rootIn, rootOut);
              const float rootDist = (rootOut >= rootIn && rootOut > 0) ? rootIn : kInf;
              if (rootDist < qcut) qfirst = S.rootRef;
              if (kSplit && qmode == TRAV_POINT_LIGHT && qfinite && qbulb < 1e12f && rootOut < 0.25f * qbulb &&
                  fabsf(qo.x) + fabsf(qo.y) + fabsf(qo.z) <= 16384.0f * qbulb && S.cutEps > 0.0f && S.cutEps <= 0.015625f * qbulb)
                qmode |= TRAV_ORDER_FREE;  // see "SPLIT" above
            }
            if (qfirst == kNoBvh) deliver(qitem, false, qmode, kInf, -1, qtPlane, qplane, qo, qnd);  // nothing to traverse
            else pooled = true;
          }
        }
      }
      const unsigned long long mPooled = __ballot(pooled);
      if (pooled) {
        const uint32_t e = (poolHead + poolCount + (uint32_t)__popcll(mPooled & ((1ull << laneId) - 1ull))) & (kPoolSlots - 1);
        uint32_t *q = pool + e;
        q[0 * kPoolSlots] = __float_as_uint(qo.x); q[1 * kPoolSlots] = __float_as_uint(qo.y); q[2 * kPoolSlots] = __float_as_uint(qo.z);
        q[3 * kPoolSlots] = __float_as_uint(qnd.x); q[4 * kPoolSlots] = __float_as_uint(qnd.y); q[5 * kPoolSlots] = __float_as_uint(qnd.z);
        q[6 * kPoolSlots] = __float_as_uint(qinv.x); q[7 * kPoolSlots] = __float_as_uint(qinv.y); q[8 * kPoolSlots] = __float_as_uint(qinv.z);
        // A visibility query that reaches the pool has no use for the planes any more (a plane that could answer it
        // has answered it above), so its bulb distance travels in the closest-plane slot.
        q[9 * kPoolSlots] = __float_as_uint(RUNTIME_MODE ? qbulb : qtPlane);
        q[10 * kPoolSlots] = qitem;
        q[11 * kPoolSlots] = (uint32_t)qfirst;
        // a visibility query has no plane to remember: its cut limit (upper half of a float) travels in those 16 bits
        q[12 * kPoolSlots] = (RUNTIME_MODE ? (__float_as_uint(qcut) >> 16) : (uint32_t)(qplane + 1)) | ((uint32_t)qmode << 16) | ((uint32_t)qfinite << 20);
      }
      poolCount += (uint32_t)__popcll(mPooled);
    }
    // ------------------------------------------------------------------ SPLIT: the end of a launch, lanes are idle
    if (kSplit && service && exhausted && poolCount == 0) {
      const bool gives = (op == OP_INNER || op == OP_PRIM) && (mode == TRAV_ANY || (mode & TRAV_ORDER_FREE) != 0) && sp > lo;
      const unsigned long long mGives = __ballot(gives), mIdle = __ballot(op == OP_IDLE);
      if (mGives != 0ull && mIdle != 0ull) {
        const uint32_t nGive = min((uint32_t)__popcll(mGives), min((uint32_t)__popcll(mIdle), (uint32_t)kPoolSlots));
        const uint32_t rank = (uint32_t)__popcll(mGives & ((1ull << laneId) - 1ull));
        if (gives && rank < nGive) {
          --sp;  // sp > lo: the top entry is in LDS; every entry of such a query is still wanted (nothing culls by the closest t)
          const uint2 e = stackCol[(sp & (K - 1)) * BLOCK];
          uint32_t *q = pool + ((poolHead + rank) & (kPoolSlots - 1));
          q[0 * kPoolSlots] = __float_as_uint(o.x); q[1 * kPoolSlots] = __float_as_uint(o.y); q[2 * kPoolSlots] = __float_as_uint(o.z);
          q[3 * kPoolSlots] = __float_as_uint(nd.x); q[4 * kPoolSlots] = __float_as_uint(nd.y); q[5 * kPoolSlots] = __float_as_uint(nd.z);
          q[6 * kPoolSlots] = __float_as_uint(inv.x); q[7 * kPoolSlots] = __float_as_uint(inv.y); q[8 * kPoolSlots] = __float_as_uint(inv.z);
          q[9 * kPoolSlots] = __float_as_uint(bulbDist);
          q[10 * kPoolSlots] = item;
          q[11 * kPoolSlots] = e.y;
          q[12 * kPoolSlots] = (__float_as_uint(cutLimit) >> 16) | ((uint32_t)mode << 16) | ((uint32_t)finite << 20);
        }
        poolCount = nGive;
      }
    }
    // ------------------------------------------------------------------ idle lanes TAKE prepared queries
    if (service) {
      const unsigned long long mIdle = __ballot(op == OP_IDLE);
      if (mIdle != 0ull && poolCount > 0) {
        const uint32_t rank = (uint32_t)__popcll(mIdle & ((1ull << laneId) - 1ull));
        if (op == OP_IDLE && rank < poolCount) {
          const uint32_t *q = pool + ((poolHead + rank) & (kPoolSlots - 1));
          o = v3(__uint_as_float(q[0 * kPoolSlots]), __uint_as_float(q[1 * kPoolSlots]), __uint_as_float(q[2 * kPoolSlots]));
          nd = v3(__uint_as_float(q[3 * kPoolSlots]), __uint_as_float(q[4 * kPoolSlots]), __uint_as_float(q[5 * kPoolSlots]));
          inv = v3(__uint_as_float(q[6 * kPoolSlots]), __uint_as_float(q[7 * kPoolSlots]), __uint_as_float(q[8 * kPoolSlots]));
          if (RUNTIME_MODE) { bulbDist = __uint_as_float(q[9 * kPoolSlots]); tPlane = kInf; }
          else tPlane = __uint_as_float(q[9 * kPoolSlots]);
          item = q[10 * kPoolSlots];
          const int32_t first = (int32_t)q[11 * kPoolSlots];
          const uint32_t packed = q[12 * kPoolSlots];
          planeBest = RUNTIME_MODE ? -1 : (int32_t)(packed & 0xffffu) - 1;
          if (RUNTIME_MODE) cutLimit = __uint_as_float(packed << 16);
          mode = (int)((packed >> 16) & 0xfu);
          finite = ((packed >> 20) & 1u) != 0u;
          tBest = kInf; primBest = -1; sp = 0; lo = 0;
          enter(first);
        }
        const uint32_t taken = min((uint32_t)__popcll(mIdle), poolCount);
        poolHead = (poolHead + taken) & (kPoolSlots - 1);
        poolCount -= taken;
        waveExact = __ballot(!finite) != 0ull;
        if (!exhausted && poolCount <= (uint32_t)kPoolLow && __ballot(op == OP_IDLE) != 0ull) continue;  // lanes still idle: prepare more at once
      }
    }
    const unsigned long long mInner = __ballot(op == OP_INNER), mPrim = __ballot(op == OP_PRIM);
    const int nInner = __popcll(mInner), nPrim = __popcll(mPrim);
    if (nInner + nPrim == 0) {
      if (exhausted && poolCount == 0) break;  // every lane idle, nothing prepared, nothing left in the queue
      continue;
    }
    // a phase ends once its lanes have fallen to 3/8 (INNER) or 1/8 (PRIM) of the lanes that had work at the vote:
    // measured best on MI355X (a PRIM phase is short, so running it almost dry costs little and saves votes)
    const int stayInner = ((nInner + nPrim) * RT_STAY_INNER) >> 3, stayPrim = ((nInner + nPrim) * RT_STAY_PRIM) >> 3;
    const bool canRefill = poolCount > 0 || !exhausted;  // a finished lane can be given a new query at the next vote
    if (nInner >= nPrim) {
      // ---------------------------------------------------------------- INNER: both child boxes of the current node
      int n = nInner;
      do {
        if (op == OP_INNER) {
          if (COUNT) { ++tc.inner; tc.innerWaveIters += 1.0f / (float)n; }
          const float4 *N = reinterpret_cast<const float4 *>(nodeBase + ((uint32_t)cur << 6));
          a = N[0]; b = N[1]; c = N[2];
          r = reinterpret_cast<const int4 *>(N)[3];
          const float lim = RUNTIME_MODE ? __builtin_fminf(tBest, cutLimit) : tBest;  // stack entries were pushed below cutLimit already
          if (waveExact) innerStep<true, K, BLOCK>(a, b, c, r, o, inv, lim, stackCol, spill, sp, lo, cur, pi, pe, op);  // wave-uniform: valid for every ray, just slower
          else innerStep<false, K, BLOCK>(a, b, c, r, o, inv, lim, stackCol, spill, sp, lo, cur, pi, pe, op);
        }
        n = __popcll(__ballot(op == OP_INNER));
        if (canRefill && __popcll(__ballot(op >= OP_DONE)) >= kDoneBreak) break;
      } while (n > stayInner);
    } else {
      // ---------------------------------------------------------------- PRIM: one object of the current leaf
      if (op == OP_PRIM && pe == 0) setLeaf(pi);  // leaves the INNER phase only noted
      int n = nPrim;
      do {
        if (op == OP_PRIM) {
          if (COUNT) { ++tc.prims; tc.primWaveIters += 1.0f / (float)n; }
          const float4 *P = reinterpret_cast<const float4 *>(primBase + (uint32_t)pi * 48u);
          a = P[0]; b = P[1]; c = P[2];
          RT_PRIM_STEP();
        }
        n = __popcll(__ballot(op == OP_PRIM));
      } while (n > stayPrim);
    }
  }
}

#undef RT_PRIM_STEP

}  // namespace rtd
