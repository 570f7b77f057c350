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
};

// Tests the objects [start, start+count) against the ray and keeps the closest (strict <, array order).
template <bool COUNT>
RT_DEV void intersectLeaf(const DevScene &S, int32_t start, int32_t count, V3 o, V3 nd, float &tBest, int32_t &primBest,
                          TravCounters &tc) {
  for (int32_t i = start; i < start + count; ++i) {
    if (COUNT) ++tc.prims;
    const float4 *P = S.primHit + 3 * (int64_t)i;
    float4 a = P[0], b = P[1], c = P[2];
    float t;
    bool sphere = S.hasSpheres && ((__float_as_uint(a.w) | __float_as_uint(b.x) | __float_as_uint(b.y)) == 0u);
    if (!sphere) {
      // Triangle::intersect, src/scene/Object.cpp:195-213 (direction normalised once per ray: same expression)
      V3 p1 = v3(a.x, a.y, a.z), n = v3(a.w, b.x, b.y), e1 = v3(b.z, b.w, c.x), e2 = v3(c.y, c.z, c.w);
      t = dot(p1 - o, n) / dot(nd, n);
      if (t < 0) continue;
      V3 q = (t * nd + o) - p1;
      float b2 = dot(e1, q);
      float b3 = dot(e2, q);
      float b1 = 1.0f - b3 - b2;
      if (b1 < 0 || b1 > 1 || b2 < 0 || b2 > 1 || b3 < 0 || b3 > 1) continue;
    } else {
      // Sphere::intersect, src/scene/Object.cpp:75-99
      V3 center = v3(a.x, a.y, a.z);
      float r = b.z;
      float radiusSquared = r * r;
      V3 dfs = center - o;
      bool inside = dot(dfs, dfs) < radiusSquared;
      float tcen = dot(dfs, nd);
      if (inside == false && tcen < 0) continue;
      V3 dd = (o + tcen * nd) - center;
      float distanceSquared = dot(dd, dd);
      if (inside == false && radiusSquared < distanceSquared) continue;
      float tOffset = sqrtf(radiusSquared - distanceSquared);
      t = inside ? tcen + tOffset : tcen - tOffset;
    }
    if (t < tBest) { tBest = t; primBest = i; }
  }
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

enum { TRAV_CLOSEST = 0, TRAV_ANY = 1, TRAV_POINT_LIGHT = 2 };

// One traversal query.
//   TRAV_CLOSEST      Scene::findClosestObject (src/scene/raytracer.cpp:187-196)
//   TRAV_ANY          the boolean of Scene::findAnyObject (:174-185): returns at the first object the closest-hit loop
//                     accepts; which objects get accepted first depends on the order, whether one exists does not
//   TRAV_POINT_LIGHT  the closest-hit query of PointLight::pointInShadow (src/scene/Object.cpp:18-25), stopping at the
//                     first accepted hit that already satisfies its predicate |origin - hit| < bulbDist. Every float
//                     operation between t and that length is monotone in t, and later accepted hits only have smaller
//                     t, so the final closest hit would satisfy the predicate too: same answer, less work.
// RUNTIME_MODE = false compiles the TRAV_CLOSEST loop only (the path-extension kernel).
template <bool RUNTIME_MODE, bool COUNT, int K, int BLOCK>
RT_DEV HitRec traverse(const DevScene &S, V3 o, V3 d, int mode, float bulbDist, uint2 *ldsStack, TravCounters &tc,
                       bool &stackOverflow) {
  const bool any = RUNTIME_MODE && mode == TRAV_ANY;
  const bool pointLight = RUNTIME_MODE && mode == TRAV_POINT_LIGHT;
  V3 nd = normalized(d);
  float tBest = kInf;
  int32_t primBest = -1;
  bool done = false;  // early exit taken
  if (S.rootRef != kNoBvh) {
    V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // src/acceleration/BVH.cpp:293
    const bool finite = S.boxesFinite && isfinite(inv.x) && isfinite(inv.y) && isfinite(inv.z) && inv.x != 0 && inv.y != 0 &&
                        inv.z != 0 && isfinite(o.x) && isfinite(o.y) && isfinite(o.z);
    float rootDist = intersectAABB(o, inv, S.rootMin[0], S.rootMin[1], S.rootMin[2], S.rootMax[0], S.rootMax[1], S.rootMax[2]);
    if (rootDist < kInf) {
      // Per-lane stack: entries [lo, sp) live in LDS (slot = index % K, one column per lane: entry slot s of lane l is
      // ldsStack[s * BLOCK + l]), entries [0, lo) have been spilled to per-lane scratch. The indices are plain locals
      // (registers); only the spill array is addressed dynamically.
      static_assert((K & (K - 1)) == 0, "K must be a power of two");
      uint2 spill[kMaxStack - K];
      int sp = 0, lo = 0;
      const int lane = threadIdx.x;
      int32_t cur = S.rootRef;
      bool alive = true;
      auto push = [&](float dist, int32_t ref) {
        if (sp == kMaxStack) { stackOverflow = true; return; }  // the reference would write past its arrays here
        if (sp - lo == K) { spill[lo] = ldsStack[(lo & (K - 1)) * BLOCK + lane]; ++lo; }
        ldsStack[(sp & (K - 1)) * BLOCK + lane] = make_uint2(__float_as_uint(dist), (uint32_t)ref);
        ++sp;
      };
      auto popNext = [&]() -> bool {
        while (sp > 0) {
          --sp;
          // A spilled entry is first copied back into its LDS slot (rare), so that the read below is always one
          // ds_read_b64 of one address; reading "LDS or spill" directly makes the compiler select between the two
          // addresses and issue a generic (flat) load.
          uint2 *slot = ldsStack + (sp & (K - 1)) * BLOCK + lane;
          if (sp < lo) { lo = sp; *slot = spill[sp]; }
          uint2 e = *slot;
          if (__uint_as_float(e.x) < tBest) { cur = (int32_t)e.y; return true; }
        }
        return false;
      };
      while (alive) {
        while (alive && cur >= 0) {
          if (COUNT) ++tc.inner;
          const float4 *N = S.nodes + 4 * (int64_t)cur;
          float4 a = N[0], b = N[1], c = N[2];
          int4 r = reinterpret_cast<const int4 *>(N)[3];
          float dl, dr;
          if (finite) {
            dl = intersectAABBFinite(o, inv, a.x, a.y, a.z, a.w, b.x, b.y);
            dr = intersectAABBFinite(o, inv, b.z, b.w, c.x, c.y, c.z, c.w);
          } else {
            dl = intersectAABB(o, inv, a.x, a.y, a.z, a.w, b.x, b.y);
            dr = intersectAABB(o, inv, b.z, b.w, c.x, c.y, c.z, c.w);
          }
          bool hl = dl < tBest, hr = dr < tBest;
          if (hl && hr) {
            bool swap = dr < dl;
            push(swap ? dl : dr, swap ? r.x : r.y);
            cur = swap ? r.y : r.x;
          } else if (hl) {
            cur = r.x;
          } else if (hr) {
            cur = r.y;
          } else {
            alive = popNext();
          }
        }
        if (!alive) break;
        int32_t start, count = (cur >> 24) & 0x7f;
        if (count != 0) {
          start = cur & 0xffffff;
        } else {
          BigLeaf bl = S.bigLeaves[cur & 0x7fffffff];
          start = bl.start; count = bl.count;
        }
        float tBefore = tBest;
        intersectLeaf<COUNT>(S, start, count, o, nd, tBest, primBest, tc);
        if (RUNTIME_MODE && tBest < tBefore) {
          if (any) { done = true; break; }
          if (pointLight) {
            // |point - info.point| < |center - point| with info.point = t * nd + origin (src/scene/Object.cpp:22-24)
            V3 hp = tBest * nd + o;
            if (magnitude(o - hp) < bulbDist) { done = true; break; }
          }
        }
        alive = popNext();
      }
    }
  }
  if (!done) {
    for (int32_t k = 0; k < S.numPlanes; ++k) {
      // Plane::intersect, src/scene/Object.cpp:133-141
      if (COUNT) ++tc.planes;
      const DevPlane &pl = S.planes[k];
      V3 pn = v3(pl.normal), pp = v3(pl.point);
      float t = dot(pp - o, pn) / dot(nd, pn);
      if (t < 0) continue;
      if (t < tBest) {
        tBest = t; primBest = S.numPrims + k;
        if (any) break;
      }
    }
  }
  HitRec h;
  h.t = primBest >= 0 ? tBest : kInf;
  h.prim = primBest;
  return h;
}

}  // namespace rtd
