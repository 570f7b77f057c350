// The top of the GPU-built BVH, built on the host: once bottom-up clustering (rt_ploc.h) has reduced the scene to a few
// thousand clusters, the levels above them — the ones every ray walks through — are split top-down with the surface-area
// heuristic over ALL split positions of the three axes (the reference bins into 16 buckets per axis,
// src/acceleration/BVH.cpp:72-140; with this few items the full sweep is affordable and at least as good). A cluster
// weighs as many objects as it holds. Plain C++ (no GPU code): tests/cpp/ploc_check.cpp runs it under sanitizers.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "rt_ploc.h"

namespace rtd {
namespace ploc {

struct TopTree {
  // new internal nodes, id = firstNewId + index (m - 1 of them for m items)
  std::vector<uint32_t> child0, child1, objects, records;
  std::vector<int32_t> parent;  // of the new nodes; kNoParent for the root
  std::vector<Box> box;
  std::vector<int32_t> parentOfItem;  // of the m items
  uint32_t root = 0;
};

// items: ids (node numbers of the clusters), their boxes, object counts and record counts. m >= 1.
inline TopTree buildTopTree(const std::vector<uint32_t> &ids, const std::vector<Box> &boxes, const std::vector<uint32_t> &objects,
                            const std::vector<uint32_t> &records, uint32_t firstNewId, int leafMax) {
  const uint32_t m = (uint32_t)ids.size();
  TopTree t;
  t.parentOfItem.assign(m, kNoParent);
  if (m == 1) { t.root = ids[0]; return t; }
  t.child0.reserve(m - 1); t.child1.reserve(m - 1);
  const float inf = std::numeric_limits<float>::infinity();
  auto centre = [&](uint32_t item, int axis) {
    const float c = 0.5f * (boxes[item].mn[axis] + boxes[item].mx[axis]);
    return c == c ? c : inf;  // NaN boxes sort to the end
  };
  auto halfArea = [&](const Box &b) {
    const float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
    const float a = x * y + y * z + x * z;
    return a == a ? a : inf;
  };
  std::vector<uint32_t> idx(m);
  for (uint32_t i = 0; i < m; ++i) idx[i] = i;
  std::vector<float> rightArea(m);
  // what a finished subtree reports to its parent
  struct Sub { uint32_t id; Box box; uint32_t objects, records; bool isItem; uint32_t index; };
  struct Frame { uint32_t b, e; int state; uint32_t mid; Sub left; };
  // explicit stack: an unlucky input can make the tree as deep as it has items
  std::vector<Frame> stack;
  std::vector<Sub> done;
  stack.push_back({0, m, 0, 0, {}});
  while (!stack.empty()) {
    Frame &f = stack.back();
    if (f.e - f.b == 1) {
      const uint32_t item = idx[f.b];
      done.push_back({ids[item], boxes[item], objects[item], records[item], true, item});
      stack.pop_back();
      continue;
    }
    if (f.state == 0) {
      // best split over the three axes: items sorted by box centre, every position
      float bestCost = inf;
      int bestAxis = -1;
      uint32_t bestPos = 0, bestOff = 0xffffffffu;
      const uint32_t centrePos = f.b + (f.e - f.b) / 2;
      for (int axis = 0; axis < 3; ++axis) {
        std::stable_sort(idx.begin() + f.b, idx.begin() + f.e, [&](uint32_t a, uint32_t c) { return centre(a, axis) < centre(c, axis); });
        Box acc = boxes[idx[f.e - 1]];
        for (uint32_t i = f.e - 1; i > f.b; --i) {  // rightArea[i] = half area of items i .. e-1
          acc = unite(acc, boxes[idx[i]]);
          rightArea[i] = halfArea(acc);
        }
        acc = boxes[idx[f.b]];
        uint64_t wl = 0, wAll = 0;
        for (uint32_t i = f.b; i < f.e; ++i) wAll += objects[idx[i]];
        for (uint32_t i = f.b; i + 1 < f.e; ++i) {  // split between i and i + 1
          acc = unite(acc, boxes[idx[i]]);
          wl += objects[idx[i]];
          const float cost = halfArea(acc) * (float)wl + rightArea[i + 1] * (float)(wAll - wl);
          // among equal costs (identical boxes) the most even split, so that ties do not build a chain
          const uint32_t off = i + 1 > centrePos ? i + 1 - centrePos : centrePos - (i + 1);
          if (cost < bestCost || (cost == bestCost && cost < inf && off < bestOff)) { bestCost = cost; bestAxis = axis; bestPos = i + 1; bestOff = off; }
        }
      }
      if (bestAxis < 0) {  // no finite cost anywhere (NaN or infinite boxes): split in the middle of whatever order is left
        bestPos = f.b + (f.e - f.b) / 2;
      } else {
        std::stable_sort(idx.begin() + f.b, idx.begin() + f.e, [&](uint32_t a, uint32_t c) { return centre(a, bestAxis) < centre(c, bestAxis); });
      }
      f.mid = bestPos;
      f.state = 1;
      const uint32_t b = f.b, mid = f.mid;
      stack.push_back({b, mid, 0, 0, {}});  // (f is invalid after this)
      continue;
    }
    if (f.state == 1) {
      f.left = done.back();
      done.pop_back();
      f.state = 2;
      const uint32_t mid = f.mid, e = f.e;
      stack.push_back({mid, e, 0, 0, {}});
      continue;
    }
    const Sub right = done.back();
    done.pop_back();
    const Sub left = f.left;
    const uint32_t k = (uint32_t)t.child0.size();
    const uint32_t id = firstNewId + k;
    t.child0.push_back(left.id); t.child1.push_back(right.id);
    t.box.push_back(unite(left.box, right.box));
    const uint32_t cnt = left.objects + right.objects;
    t.objects.push_back(cnt);
    t.records.push_back(cnt > (uint32_t)leafMax ? 1u + left.records + right.records : 0u);
    t.parent.push_back(kNoParent);
    for (const Sub &c : {left, right}) {
      if (c.isItem) t.parentOfItem[c.index] = (int32_t)id;
      else t.parent[c.id - firstNewId] = (int32_t)id;
    }
    done.push_back({id, t.box.back(), cnt, t.records.back(), false, 0});
    stack.pop_back();
  }
  t.root = done.back().id;
  return t;
}

}  // namespace ploc
}  // namespace rtd
