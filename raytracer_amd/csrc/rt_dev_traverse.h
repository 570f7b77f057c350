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

#ifndef RT_CHUNK_MAX
#define RT_CHUNK_MAX 256
#endif
#ifndef RT_CHUNK_DIV
#define RT_CHUNK_DIV 16
#endif
constexpr uint32_t kTicketChunkMax = RT_CHUNK_MAX, kTicketChunkMin = 64, kChunksPerWave = RT_CHUNK_DIV;  // a refill takes up to 64 indices: a chunk must cover one

enum { TRAV_CLOSEST = 0, TRAV_ANY = 1, TRAV_POINT_LIGHT = 2, TRAV_SKIP = 3 };

// Scheduling thresholds of the traversal engine (lanes of one 64-lane wavefront).
#ifndef RT_SERVICE_MIN
#define RT_SERVICE_MIN 24
#endif
#ifndef RT_STAY_INNER
#define RT_STAY_INNER 3
#endif
#ifndef RT_STAY_PRIM
#define RT_STAY_PRIM 1
#endif
constexpr int kServiceMin = RT_SERVICE_MIN;  // lanes without work (finished or idle) before results are written and the queue is visited

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
//     FINISH  planes + result write-out of a query whose BVH part is over
//     IDLE    no query
// The wavefront votes: once kServiceMin lanes have no work (their query is finished, or they never had one) a service
// step writes the finished results out and gives every such lane a new query from the queue; otherwise the more
// popular of INNER / PRIM runs in a tight loop of its own until too few of its lanes are left, while the minority
// waits until it has become the majority. Incoherent
// rays therefore do not serialise the wavefront through both loop bodies every iteration, and a few long traversals
// do not leave most lanes waiting: finished queries are REPLACED by new ones. Per query, nodes and objects are still
// visited in exactly the reference's order. The grid is sized to fill the GPU once (persistent wavefronts); a wavefront ends when the
// queue is exhausted and its own lanes are done, so nothing ever waits on another wavefront.
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
template <class Source, bool RUNTIME_MODE, bool COUNT, int K, int BLOCK>
RT_DEV void traverseQueue(const DevScene &S, Source &src, uint32_t *ticket, uint2 *ldsStack, TravCounters &tc) {
  static_assert((K & (K - 1)) == 0, "K must be a power of two");
  const uint32_t total = src.total();
  const uint32_t laneId = __lane_id();
  const char *nodeBase = reinterpret_cast<const char *>(S.nodes);
  const char *primBase = reinterpret_cast<const char *>(S.primHit);
  uint2 *stackCol = ldsStack + threadIdx.x;
  uint2 spill[kMaxStack - K];
  enum { OP_IDLE = 0, OP_INNER = 1, OP_PRIM = 2, OP_FINISH = 3, OP_FINISH_EARLY = 4 };
  int op = OP_IDLE;
  uint32_t item = 0;
  V3 o = v3(0, 0, 0), nd = o, inv = o;
  float tBest = kInf, bulbDist = 0.0f;
  float cullD = kInf;  // TRAV_POINT_LIGHT: boxes entered beyond this distance (in units of |d|) cannot matter, see below
  int32_t primBest = -1, cur = 0, pi = 0, pe = 0;
  int sp = 0, lo = 0, mode = TRAV_CLOSEST;
  bool finite = true;
  bool exhausted = false;  // wave-uniform: the queue has no more queries
  bool waveExact = false;  // wave-uniform: some lane holds a ray the min/max slab test is not valid for
  uint32_t chunkPos = 0, chunkEnd = 0;
  const uint32_t kTicketChunk = min(kTicketChunkMax, max(kTicketChunkMin, total / (gridDim.x * (BLOCK / 64) * kChunksPerWave)));

  // an inner node becomes an INNER operation, a leaf a run of PRIM operations
  auto enter = [&](int32_t next) {
    if (next >= 0) {
      cur = next; op = OP_INNER;
    } else {
      int32_t count = (next >> 24) & 0x7f;
      pi = next & 0xffffff;
      if (count == 0) {
        BigLeaf bl = S.bigLeaves[next & 0x7fffffff];
        pi = bl.start; count = bl.count;
      }
      pe = pi + count;
      op = OP_PRIM;
    }
  };
  // next stack entry whose box entry distance is still below the closest hit (src/acceleration/BVH.cpp:302-305);
  // none left: the BVH part of the query is over
  auto popNext = [&]() {
    op = OP_FINISH;
    while (sp > 0) {
      --sp;
      // A spilled entry is first copied back into its LDS slot (rare), so that the read below is always a plain LDS
      // read of one address; reading "LDS or spill" directly makes the compiler select between the two addresses
      // and issue a generic (flat) load.
      uint2 *slot = stackCol + (sp & (K - 1)) * BLOCK;
      if (sp < lo) { lo = sp; *slot = spill[sp]; }
      uint2 e = *slot;
      if (__uint_as_float(e.x) < (RUNTIME_MODE ? __builtin_fminf(tBest, cullD) : tBest)) { enter((int32_t)e.y); break; }
    }
  };

  while (true) {
    const unsigned long long mInner = __ballot(op == OP_INNER), mPrim = __ballot(op == OP_PRIM);
    const int nInner = __popcll(mInner), nPrim = __popcll(mPrim);
    const int nOther = 64 - nInner - nPrim;  // finished + idle lanes
    if (nOther >= kServiceMin || nInner + nPrim == 0) {
      const unsigned long long mFinish = __ballot(op >= OP_FINISH);
      const int nFinish = __popcll(mFinish), nIdle = nOther - nFinish;
      // one special step serves both: finished queries are written out and, in the same step, every idle lane
      // (the ones just finished included) takes a new query
      const bool doFinish = nFinish > 0, doRefill = !exhausted && nIdle + nFinish > 0;
      if (doFinish) {
        // -------------------------------------------------------------- FINISH: planes, result, lane becomes idle
        if (op >= OP_FINISH) {
          bool occluded = op == OP_FINISH_EARLY;
          if (op == OP_FINISH) {
            for (int32_t k = 0; k < S.numPlanes; ++k) {
              // Plane::intersect, src/scene/Object.cpp:133-141; planes lose ties to BVH objects (raytracer.cpp:192)
              if (COUNT) ++tc.planes;
              const DevPlane &pl = S.planes[k];
              V3 pn = v3(pl.normal), pp = v3(pl.point);
              float t = dot(pp - o, pn) / dot(nd, pn);
              if (t < 0) continue;
              if (t < tBest) {
                tBest = t; primBest = S.numPrims + k;
                if (RUNTIME_MODE && mode == TRAV_ANY) break;
              }
            }
            if (RUNTIME_MODE && mode == TRAV_POINT_LIGHT) {
              // object hits were checked when they were accepted; only a plane can still satisfy the predicate
              if (primBest >= S.numPrims) {
                V3 hp = tBest * nd + o;
                occluded = magnitude(o - hp) < bulbDist;
              }
            } else {
              occluded = primBest >= 0;
            }
          }
          HitRec h;
          h.t = primBest >= 0 ? tBest : kInf;
          h.prim = primBest;
          src.store(item, h, occluded, o, nd);
          op = OP_IDLE;
          finite = true;
        }
      }
      if (doRefill) {
        // -------------------------------------------------------------- REFILL idle lanes from the queue
        const unsigned long long mIdle = __ballot(op == OP_IDLE);
        int leader = __ffsll((long long)mIdle) - 1;
        // Queue indices come from a wave-local chunk [chunkPos, chunkEnd); one atomic on the shared counter buys
        // kTicketChunk of them, so the counter (a single address all wavefronts hit) sees few operations.
        const uint32_t avail = chunkEnd - chunkPos, want = (uint32_t)__popcll(mIdle);
        const uint32_t rank = (uint32_t)__popcll(mIdle & ((1ull << laneId) - 1ull));
        uint32_t idx = chunkPos + rank;
        if (avail < want) {
          uint32_t base = 0;
          if ((int)laneId == leader) base = atomicAdd(ticket, kTicketChunk);
          base = __shfl(base, leader);
          if (rank >= avail) idx = base + (rank - avail);
          chunkPos = base + (want - avail);
          chunkEnd = base + kTicketChunk;
          if (base >= total) { exhausted = true; chunkPos = chunkEnd = 0; }
        } else {
          chunkPos += want;
        }
        if (op == OP_IDLE && idx < total) {
          V3 d;
          item = src.fetch(idx, o, d, mode, bulbDist);
          nd = normalized(d);
          tBest = kInf; primBest = -1; sp = 0; lo = 0;
          const bool skip = RUNTIME_MODE && mode == TRAV_SKIP;
          op = skip ? OP_IDLE : OP_FINISH;
          // A point-light query only asks whether the closest hit lies nearer than the bulb. With |d| = bulbDist >= 1
          // the reference's culling is conservative (box distances are in units of |d|, t in units of 1, so a box is
          // never farther than its distance says) and its closest hit is the true one; every object in a box entered
          // beyond the bulb (distance > 1 in units of |d|, plus a 1 % margin for rounding) is hit beyond the bulb, which
          // answers the question like no hit at all. Such boxes are skipped. For bulbDist < 1 the reference's culling
          // can drop nearer hits depending on what it visited before, so nothing is skipped then.
          cullD = (RUNTIME_MODE && mode == TRAV_POINT_LIGHT && bulbDist >= 1.0f) ? 1.01f : kInf;
          if (S.rootRef != kNoBvh && !skip) {
            inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // src/acceleration/BVH.cpp:293
            finite = S.boxesFinite && isfinite(inv.x) && isfinite(inv.y) && isfinite(inv.z) && inv.x != 0 && inv.y != 0 &&
                     inv.z != 0 && isfinite(o.x) && isfinite(o.y) && isfinite(o.z);
            float rootDist = intersectAABB(o, inv, S.rootMin[0], S.rootMin[1], S.rootMin[2], S.rootMax[0], S.rootMax[1], S.rootMax[2]);
            if (rootDist < cullD) enter(S.rootRef);
          }
        }
        waveExact = __ballot(!finite) != 0ull;
        continue;
      }
      if (doFinish) { waveExact = __ballot(!finite) != 0ull; continue; }
      if (nInner + nPrim == 0) break;  // every lane idle and the queue exhausted
    }
    // a phase ends once its lanes have fallen to 3/8 (INNER) or 1/8 (PRIM) of the lanes that had work at the vote:
    // measured best on MI355X (a PRIM phase is short, so running it almost dry costs little and saves votes)
    const int stayInner = ((nInner + nPrim) * RT_STAY_INNER) >> 3, stayPrim = ((nInner + nPrim) * RT_STAY_PRIM) >> 3;
    if (nInner >= nPrim) {
      // ---------------------------------------------------------------- INNER: both child boxes of the current node
      int n = nInner;
      do {
        if (op == OP_INNER) {
          if (COUNT) { ++tc.inner; tc.innerWaveIters += 1.0f / (float)n; }
          const float4 *N = reinterpret_cast<const float4 *>(nodeBase + ((uint32_t)cur << 6));
          float4 a = N[0], b = N[1], c = N[2];
          int4 r = reinterpret_cast<const int4 *>(N)[3];
          float dl, dr;
          if (waveExact) {  // wave-uniform: the compare-and-select form is valid for every ray, just slower
            dl = intersectAABB(o, inv, a.x, a.y, a.z, a.w, b.x, b.y);
            dr = intersectAABB(o, inv, b.z, b.w, c.x, c.y, c.z, c.w);
          } else {
            dl = intersectAABBFinite(o, inv, a.x, a.y, a.z, a.w, b.x, b.y);
            dr = intersectAABBFinite(o, inv, b.z, b.w, c.x, c.y, c.z, c.w);
          }
          // Near child first, right only if strictly nearer (src/acceleration/BVH.cpp:322-333); a child that cannot
          // beat the closest hit any more is dropped here instead of at pop time.
          const float lim = RUNTIME_MODE ? __builtin_fminf(tBest, cullD) : tBest;
          bool hl = dl < lim, hr = dr < lim;
          bool both = hl && hr;
          bool swap = both && dr < dl;
          if (both) {
            if (sp - lo == K) { spill[lo] = stackCol[(lo & (K - 1)) * BLOCK]; ++lo; }
            stackCol[(sp & (K - 1)) * BLOCK] = make_uint2(__float_as_uint(swap ? dl : dr), (uint32_t)(swap ? r.x : r.y));
            ++sp;
          }
          if (hl || hr) enter((hl && !swap) ? r.x : r.y);
          else popNext();
        }
        n = __popcll(__ballot(op == OP_INNER));
      } while (n > stayInner);
    } else {
      // ---------------------------------------------------------------- PRIM: one object of the current leaf
      int n = nPrim;
      do {
        if (op == OP_PRIM) {
          if (COUNT) { ++tc.prims; tc.primWaveIters += 1.0f / (float)n; }
          const float4 *P = reinterpret_cast<const float4 *>(primBase + (uint32_t)pi * 48u);
          bool accepted = intersectPrim(S, P[0], P[1], P[2], pi, o, nd, tBest, primBest);
          ++pi;
          if (RUNTIME_MODE && accepted && mode != TRAV_CLOSEST) {
            if (mode == TRAV_ANY) {
              op = OP_FINISH_EARLY;
            } else {
              // |point - info.point| < |center - point| with info.point = t * nd + origin (src/scene/Object.cpp:22-24)
              V3 hp = tBest * nd + o;
              if (magnitude(o - hp) < bulbDist) op = OP_FINISH_EARLY;
            }
          }
          if (op == OP_PRIM && pi == pe) popNext();
        }
        n = __popcll(__ballot(op == OP_PRIM));
      } while (n > stayPrim);
    }
  }
}

}  // namespace rtd
