// Binned-SAH BVH builder: decision-for-decision the reference's
// src/acceleration/BVH.cpp:72-281 (findBestBucketSplit, partition, updateNodeBounds, flatten),
// because the leaf order it produces decides which primitive wins a tie in t
// (BVH.cpp:308-313) and therefore the hit ids. Works on an index permutation and
// reorders scene.prims once at the end.
#include <algorithm>
#include <limits>
#include <numeric>

#include "rth_scene.h"

namespace rth {
namespace {

constexpr int kBuckets = 16;     // N_BUCKETS, BVH.cpp:9
constexpr int kMinLeafSize = 4;  // MIN_LEAF_SIZE, BVH.cpp:11
constexpr float kInf = std::numeric_limits<float>::infinity();

struct Box {  // BVH.h:18-28, BVH.cpp:142-159
  Vec3 minPoint{kInf, kInf, kInf}, maxPoint{-kInf, -kInf, -kInf};
  void shrink(const Vec3 &p) {
    minPoint.x = std::min(minPoint.x, p.x); minPoint.y = std::min(minPoint.y, p.y); minPoint.z = std::min(minPoint.z, p.z);
  }
  void expand(const Vec3 &p) {
    maxPoint.x = std::max(maxPoint.x, p.x); maxPoint.y = std::max(maxPoint.y, p.y); maxPoint.z = std::max(maxPoint.z, p.z);
  }
  float surfaceArea() const {
    Vec3 e = maxPoint - minPoint;
    return 2 * (e.x * e.y + e.y * e.z + e.x * e.z);
  }
};

struct BuildNode {
  Vec3 aabbMin, aabbMax;
  int left = -1, right = -1;  // indices into the build-node pool
  int start = 0, numObjects = 0;
};

struct Builder {
  const std::vector<Prim> &prims;
  std::vector<int> order;  // order[i] = index into prims of the object at BVH position i
  std::vector<BuildNode> pool;

  explicit Builder(const std::vector<Prim> &p) : prims(p), order(p.size()) { std::iota(order.begin(), order.end(), 0); }
  const Prim &at(int i) const { return prims[order[i]]; }

  void updateNodeBounds(int n) {  // BVH.cpp:184-200
    BuildNode &node = pool[n];
    node.aabbMin = Vec3(kInf, kInf, kInf);
    node.aabbMax = Vec3(-kInf, -kInf, -kInf);
    int end = node.start + node.numObjects;
    for (int i = node.start; i < end; ++i) {
      const Prim &o = at(i);
      node.aabbMin.x = std::min(node.aabbMin.x, o.aabbMin.x);
      node.aabbMin.y = std::min(node.aabbMin.y, o.aabbMin.y);
      node.aabbMin.z = std::min(node.aabbMin.z, o.aabbMin.z);
      node.aabbMax.x = std::max(node.aabbMax.x, o.aabbMax.x);
      node.aabbMax.y = std::max(node.aabbMax.y, o.aabbMax.y);
      node.aabbMax.z = std::max(node.aabbMax.z, o.aabbMax.z);
    }
  }

  void findBestBucketSplit(int n, int &bestAxis, float &bestPosition, float &bestCost) {  // BVH.cpp:72-140
    const BuildNode &node = pool[n];
    bestAxis = -1; bestPosition = 0; bestCost = kInf;
    Box extent;
    int end = node.start + node.numObjects;
    for (int i = node.start; i < end; ++i) { extent.expand(at(i).centroid); extent.shrink(at(i).centroid); }
    const Vec3 &maxPoint = extent.maxPoint;
    const Vec3 &minPoint = extent.minPoint;
    for (int axis = 0; axis < 3; ++axis) {
      // Zero centroid extent: the reference computes scale = 16/0 = inf and (c-min)*inf = NaN, whose
      // int conversion is undefined (INT_MIN on x86 -> out-of-bounds write and a crash; 0 on aarch64,
      // where every candidate cost then comes out NaN and the axis is never chosen). Take the
      // well-defined outcome: the axis offers no split.
      if (!(maxPoint[axis] - minPoint[axis] > 0.0f)) continue;
      Box buckets[kBuckets];
      int bucketCount[kBuckets] = {0};
      float scale = kBuckets / (maxPoint[axis] - minPoint[axis]);
      for (int i = node.start; i < end; ++i) {
        const Prim &o = at(i);
        int bucketIdx = std::min(kBuckets - 1, static_cast<int>((o.centroid[axis] - minPoint[axis]) * scale));
        buckets[bucketIdx].shrink(o.aabbMin);
        buckets[bucketIdx].expand(o.aabbMax);
        bucketCount[bucketIdx]++;
      }
      int leftBoxCount[kBuckets - 1] = {0}, rightBoxCount[kBuckets - 1] = {0};
      float leftBoxArea[kBuckets - 1] = {0}, rightBoxArea[kBuckets - 1] = {0};
      int leftCount = 0, rightCount = 0;
      Box leftBox, rightBox;
      for (int i = 0; i < kBuckets - 1; ++i) {
        leftCount += bucketCount[i];
        leftBoxCount[i] = leftCount;
        rightCount += bucketCount[kBuckets - i - 1];
        rightBoxCount[kBuckets - i - 2] = rightCount;
        leftBox.expand(buckets[i].maxPoint);
        leftBox.shrink(buckets[i].minPoint);
        leftBoxArea[i] = leftBox.surfaceArea();
        rightBox.expand(buckets[kBuckets - i - 1].maxPoint);
        rightBox.shrink(buckets[kBuckets - i - 1].minPoint);
        rightBoxArea[kBuckets - i - 2] = rightBox.surfaceArea();
      }
      // The split plane is recomputed with a different expression than the bucket index above.
      scale = (maxPoint[axis] - minPoint[axis]) / kBuckets;
      for (int i = 0; i < kBuckets - 1; ++i) {
        float cost = leftBoxArea[i] * leftBoxCount[i] + rightBoxArea[i] * rightBoxCount[i];
        if (cost < bestCost) {
          bestCost = cost;
          bestAxis = axis;
          bestPosition = minPoint[axis] + (i + 1) * scale;
        }
      }
    }
  }

  int partition(int n) {  // BVH.cpp:231-281; returns the node count of the subtree
    if (pool[n].numObjects <= kMinLeafSize) return 1;
    int axis; float splitPosition, bestCost;
    findBestBucketSplit(n, axis, splitPosition, bestCost);
    Box parentBox; parentBox.minPoint = pool[n].aabbMin; parentBox.maxPoint = pool[n].aabbMax;
    float parentCost = parentBox.surfaceArea() * pool[n].numObjects;
    if (bestCost >= parentCost) return 1;
    int i = pool[n].start;
    int j = i + pool[n].numObjects;
    // Same algorithm object as the reference: libstdc++'s std::partition on random-access iterators.
    // It is not stable; the resulting order is part of the contract.
    auto middle = std::partition(order.begin() + i, order.begin() + j,
                                 [&](int idx) { return prims[idx].centroid[axis] < splitPosition; });
    int leftNumObjects = (int)std::distance(order.begin() + i, middle);
    if (leftNumObjects == 0 || leftNumObjects == pool[n].numObjects) return 1;
    int l = (int)pool.size(); pool.emplace_back();
    int r = (int)pool.size(); pool.emplace_back();
    pool[l].start = pool[n].start; pool[l].numObjects = leftNumObjects;
    pool[r].start = i + leftNumObjects; pool[r].numObjects = pool[n].numObjects - leftNumObjects;
    pool[n].left = l; pool[n].right = r; pool[n].numObjects = 0;
    updateNodeBounds(l);
    updateNodeBounds(r);
    int nl = partition(l);
    int nr = partition(r);
    return nl + nr + 1;
  }

  void flatten(int n, std::vector<BvhNode> &out) {  // BVH.cpp:411-428 (iterative pre-order DFS)
    struct Frame { int node; int slot; int stage; };
    std::vector<Frame> stack;
    stack.push_back({n, -1, 0});
    while (!stack.empty()) {
      Frame &f = stack.back();
      const BuildNode &b = pool[f.node];
      if (f.stage == 0) {
        f.slot = (int)out.size();
        BvhNode fn{};
        fn.aabbMin[0] = b.aabbMin.x; fn.aabbMin[1] = b.aabbMin.y; fn.aabbMin[2] = b.aabbMin.z;
        fn.aabbMax[0] = b.aabbMax.x; fn.aabbMax[1] = b.aabbMax.y; fn.aabbMax[2] = b.aabbMax.z;
        fn.numObjects = b.numObjects;
        fn.rightOrStart = b.start;
        out.push_back(fn);
        if (b.numObjects > 0) { stack.pop_back(); continue; }
        f.stage = 1;
        int left = b.left;
        stack.push_back({left, -1, 0});
      } else if (f.stage == 1) {
        out[f.slot].rightOrStart = (int)out.size();
        f.stage = 2;
        int right = b.right;
        stack.push_back({right, -1, 0});
      } else {
        stack.pop_back();
      }
    }
  }
};

}  // namespace

void buildBVH(Scene &scene) {
  scene.nodes.clear();
  scene.bvhBuilt = true;
  if (scene.prims.empty()) {
    scene.log.push_back("No triangles or spheres in scene. BVH not created.");
    return;
  }
  Builder b(scene.prims);
  b.pool.reserve(scene.prims.size() * 2 + 1);
  b.pool.emplace_back();
  b.pool[0].start = 0;
  b.pool[0].numObjects = (int)scene.prims.size();
  b.updateNodeBounds(0);
  int numNodes = b.partition(0);
  scene.nodes.reserve(numNodes);
  b.flatten(0, scene.nodes);
  std::vector<Prim> reordered;
  reordered.reserve(scene.prims.size());
  for (int idx : b.order) reordered.push_back(scene.prims[idx]);
  scene.prims.swap(reordered);
  scene.log.push_back("BVH created with " + std::to_string(numNodes) + " nodes on " +
                      std::to_string(scene.prims.size()) + " objects.");
}

}  // namespace rth
