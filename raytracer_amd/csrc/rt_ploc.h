// Bottom-up BVH by parallel locally-ordered clustering (Meister and Bittner, "Parallel Locally-Ordered Clustering for
// Bounding Volume Hierarchy Construction", TVCG 2018), the per-element steps. The kernels of rt_bvh_build.hip call these
// functions with one thread per element; tests/cpp/ploc_check.cpp calls the very same functions from plain loops on the
// CPU (under the address sanitizer), so that every index this code forms has been checked before it runs on a GPU.
//
// Why: a linear BVH splits at Morton-code bits, wherever they fall. Clustering merges, round after round, the pairs of
// neighbouring clusters (neighbours in Morton order) whose union has the smallest surface area — the surface-area
// heuristic of the reference's builder (src/acceleration/BVH.cpp:72-140) applied from the bottom — and gives trees of
// about the quality of a top-down binned-SAH build.
//
// Node numbering: 0 .. n-1 are the objects in Morton order (leaf i = input object order[i]), n .. 2n-2 the internal
// nodes in the order they were created. `clusters` holds the current roots, in Morton order of their first object.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define RT_PLOC_HD __host__ __device__ __forceinline__
#else
#define RT_PLOC_HD inline
#include <cmath>
#endif

namespace rtd {
namespace ploc {

struct Box {
  float mn[3], mx[3];
};

constexpr int32_t kNoParent = -1;

RT_PLOC_HD float fminNan(float a, float b) { return (b < a || a != a) ? b : a; }  // the other operand when one is NaN
RT_PLOC_HD float fmaxNan(float a, float b) { return (b > a || a != a) ? b : a; }
// Union of two boxes; a NaN coordinate (an object that cannot be hit anyway) does not spread to the ancestors.
RT_PLOC_HD Box unite(const Box &a, const Box &b) {
  Box r;
  for (int k = 0; k < 3; ++k) { r.mn[k] = fminNan(a.mn[k], b.mn[k]); r.mx[k] = fmaxNan(a.mx[k], b.mx[k]); }
  return r;
}
// Surface area / 2 of the union (what the reference's cost compares, src/acceleration/BVH.cpp:142-159); anything that
// is not a number counts as infinitely large.
RT_PLOC_HD float unionArea(const Box &a, const Box &b) {
  const Box u = unite(a, b);
  const float x = u.mx[0] - u.mn[0], y = u.mx[1] - u.mn[1], z = u.mx[2] - u.mn[2];
  const float area = x * y + y * z + x * z;
  return area == area ? area : __builtin_inff();
}

// Is the pair (i, j) a better partner choice than (i, bj)? A total order on unordered pairs, the same from both ends:
// smaller area first; among equal areas pairs whose lower position is even (so that a run of identical boxes pairs up
// 0-1, 2-3, ... in one round instead of one pair per round), then the closer pair, then the lower position. The pair that
// is smallest over all pairs is the best choice of both its members, so every round merges at least one pair.
RT_PLOC_HD bool betterPair(float area, uint32_t i, uint32_t j, float bestArea, uint32_t bj) {
  if (area != bestArea) return area < bestArea;
  const uint32_t lo = i < j ? i : j, blo = i < bj ? i : bj;
  const uint32_t par = lo & 1u, bpar = blo & 1u;
  if (par != bpar) return par < bpar;
  const uint32_t dist = i < j ? j - i : i - j, bdist = i < bj ? bj - i : i - bj;
  if (dist != bdist) return dist < bdist;
  return lo < blo;
}

// Round, step 1: cluster position i (of count >= 2) looks `radius` positions to either side and notes its best partner.
RT_PLOC_HD void nearest(uint32_t i, uint32_t count, int radius, const uint32_t *clusters, const Box *nodeBox, uint32_t *partner) {
  const Box me = nodeBox[clusters[i]];
  const uint32_t lo = i > (uint32_t)radius ? i - (uint32_t)radius : 0u;
  const uint32_t hi = i + (uint32_t)radius < count - 1u ? i + (uint32_t)radius : count - 1u;
  float bestArea = 0.0f;
  uint32_t best = 0xffffffffu;
  for (uint32_t j = lo; j <= hi; ++j) {
    if (j == i) continue;
    const float a = unionArea(me, nodeBox[clusters[j]]);
    if (best == 0xffffffffu || betterPair(a, i, j, bestArea, best)) { bestArea = a; best = j; }
  }
  partner[i] = best;  // count >= 2: at least one j exists
}

// Round, step 2: position i keeps its cluster, or (the lower position of a pair that chose each other) replaces it by a new
// internal node, or (the higher position of such a pair) drops out. `newId` is only used by the lower position of a pair:
// the caller hands out n + (number of internal nodes so far), atomically on the GPU. Returns 1 if the position survives.
template <class NextId>
RT_PLOC_HD uint32_t merge(uint32_t i, const uint32_t *clusters, const uint32_t *partner, Box *nodeBox, uint32_t n, uint32_t *child0,
                          uint32_t *child1, int32_t *parent, uint32_t *clustersOut, NextId nextId) {
  const uint32_t j = partner[i];
  const uint32_t ci = clusters[i];
  if (partner[j] != i) { clustersOut[i] = ci; return 1u; }
  if (i > j) { clustersOut[i] = 0xffffffffu; return 0u; }
  const uint32_t cj = clusters[j];
  const uint32_t id = nextId();  // n <= id <= 2n - 2: every merge takes one cluster away, n - 1 merges leave one
  child0[id - n] = ci; child1[id - n] = cj;
  parent[ci] = (int32_t)id; parent[cj] = (int32_t)id;
  parent[id] = kNoParent;
  nodeBox[id] = unite(nodeBox[ci], nodeBox[cj]);
  clustersOut[i] = id;
  return 1u;
}

// After the last round. One thread per object walks to the root; at every node the first to arrive stops and the second
// (which finds both children's figures written) fills in the node's object count and the number of inner records of
// its subtree: nodes spanning more than leafMax objects become records, smaller ones leaves.
template <class Arrive>
RT_PLOC_HD void countUp(uint32_t leaf, uint32_t n, int leafMax, const uint32_t *child0, const uint32_t *child1, const int32_t *parent,
                        uint32_t *objects, uint32_t *records, Arrive arriveSecond) {
  int32_t node = parent[leaf];
  while (node != kNoParent) {
    if (!arriveSecond((uint32_t)node - n)) return;
    const uint32_t l = child0[(uint32_t)node - n], r = child1[(uint32_t)node - n];
    const uint32_t cnt = objects[l] + objects[r];
    objects[node] = cnt;
    records[node] = cnt > (uint32_t)leafMax ? 1u + records[l] + records[r] : 0u;
    node = parent[node];
  }
}

// Where a node sits in the depth-first layout: position of its first object in the tree-ordered object array, its index
// among the inner records in pre-order (meaningful for record nodes only), and how many records lie above it.
struct Place {
  uint32_t firstObject, recordIndex, depth;
};
RT_PLOC_HD Place place(uint32_t node, uint32_t n, const uint32_t *child0, const uint32_t *child1, const int32_t *parent,
                       const uint32_t *objects, const uint32_t *records) {
  Place p{0u, 0u, 0u};
  uint32_t v = node;
  for (int32_t up = parent[v]; up != kNoParent; up = parent[v]) {
    const uint32_t l = child0[(uint32_t)up - n];
    const bool isRight = child1[(uint32_t)up - n] == v && l != v;
    p.recordIndex += 1u + (isRight ? records[l] : 0u);
    if (isRight) p.firstObject += objects[l];
    ++p.depth;
    v = (uint32_t)up;
  }
  return p;
}

}  // namespace ploc
}  // namespace rtd
