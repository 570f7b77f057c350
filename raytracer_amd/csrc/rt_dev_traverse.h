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

template <int K, int BLOCK>
struct ShortStack {
  static_assert((K & (K - 1)) == 0, "K must be a power of two");
  uint2 *col;  // this lane's column in the block's LDS array: entry slot s lives at col[s * BLOCK]
  uint2 spill[kMaxStack - K];
  int sp, lo;  // entries [lo, sp) are in LDS (slot = index % K); entries [0, lo) are in `spill`
  bool overflow;
  RT_DEV void init(uint2 *ldsBase) { col = ldsBase + threadIdx.x; sp = 0; lo = 0; overflow = false; }
  RT_DEV void push(float dist, int32_t ref) {
    if (sp == kMaxStack) { overflow = true; return; }  // the reference would write past its arrays here
    if (sp - lo == K) { spill[lo] = col[(lo & (K - 1)) * BLOCK]; ++lo; }
    col[(sp & (K - 1)) * BLOCK] = make_uint2(__float_as_uint(dist), (uint32_t)ref);
    ++sp;
  }
  RT_DEV uint2 pop() {
    --sp;
    if (sp < lo) { lo = sp; return spill[sp]; }
    return col[(sp & (K - 1)) * BLOCK];
  }
};

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

// Scene::findClosestObject (ANY = false, src/scene/raytracer.cpp:187-196) or the boolean of Scene::findAnyObject
// (ANY = true, :174-185; returns at the first object the closest-hit loop would have accepted).
template <bool ANY, bool COUNT, int K, int BLOCK>
RT_DEV HitRec traverse(const DevScene &S, V3 o, V3 d, uint2 *ldsStack, TravCounters &tc, bool &stackOverflow) {
  V3 nd = normalized(d);
  float tBest = kInf;
  int32_t primBest = -1;
  if (S.rootRef != kNoBvh) {
    V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // src/acceleration/BVH.cpp:293
    float rootDist = intersectAABB(o, inv, S.rootMin[0], S.rootMin[1], S.rootMin[2], S.rootMax[0], S.rootMax[1], S.rootMax[2]);
    if (rootDist < kInf) {
      ShortStack<K, BLOCK> stk;
      stk.init(ldsStack);
      int32_t cur = S.rootRef;
      bool alive = true;
      auto popNext = [&]() -> bool {
        while (stk.sp > 0) {
          uint2 e = stk.pop();
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
          float dl = intersectAABB(o, inv, a.x, a.y, a.z, a.w, b.x, b.y);
          float dr = intersectAABB(o, inv, b.z, b.w, c.x, c.y, c.z, c.w);
          bool hl = dl < tBest, hr = dr < tBest;
          if (hl && hr) {
            bool swap = dr < dl;
            stk.push(swap ? dl : dr, swap ? r.x : r.y);
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
        intersectLeaf<COUNT>(S, start, count, o, nd, tBest, primBest, tc);
        if (ANY && primBest >= 0) break;
        alive = popNext();
      }
      stackOverflow = stackOverflow || stk.overflow;
    }
  }
  if (!(ANY && primBest >= 0)) {
    for (int32_t k = 0; k < S.numPlanes; ++k) {
      // Plane::intersect, src/scene/Object.cpp:133-141
      if (COUNT) ++tc.planes;
      const DevPlane &pl = S.planes[k];
      V3 pn = v3(pl.normal), pp = v3(pl.point);
      float t = dot(pp - o, pn) / dot(nd, pn);
      if (t < 0) continue;
      if (t < tBest) {
        tBest = t; primBest = S.numPrims + k;
        if (ANY) break;
      }
    }
  }
  HitRec h;
  h.t = primBest >= 0 ? tBest : kInf;
  h.prim = primBest;
  return h;
}

}  // namespace rtd
