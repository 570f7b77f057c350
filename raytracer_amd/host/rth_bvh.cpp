// Host BVH build. The TREE is the reference's, decision for decision (src/acceleration/BVH.cpp:72-281: 16 centroid
// bins per axis, cost = areaL*countL + areaR*countR over 3 x 15 planes, leaf when <= 4 objects / no cheaper split / an
// empty side, objects regrouped with std::partition), because the leaf order it produces decides which object wins a
// tie in t (BVH.cpp:308-313) and therefore the hit ids. The SCHEDULE is this library's own: the tree is split top-down
// with all host threads binning each large node together until the pieces are small, then the pieces are built
// concurrently from a shared queue of subtrees. Bin boxes and counts are min / max / integer sums, so neither the thread count nor the
// order of the pieces changes a single bit of the result; node numbers are assigned afterwards by a pre-order walk
// (left child = index + 1, BVH.cpp:411-428).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <numeric>
#include <thread>

#include "rth_scene.h"
#include "rth_threads.h"

namespace rth {
namespace {

constexpr int kBins = 16;          // N_BUCKETS, BVH.cpp:9
constexpr int kLeafObjects = 4;    // MIN_LEAF_SIZE, BVH.cpp:11
constexpr int kSharedNode = 1 << 16;  // nodes with at least this many objects are binned by all threads together
constexpr int kTaskNode = 1 << 12;    // below that: a subtree with fewer objects than this is built by one thread in one go
constexpr float kInf = std::numeric_limits<float>::infinity();

struct Bounds {
  float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
  void add(const Vec3 &mn, const Vec3 &mx) {
    lo[0] = std::min(lo[0], mn.x); lo[1] = std::min(lo[1], mn.y); lo[2] = std::min(lo[2], mn.z);
    hi[0] = std::max(hi[0], mx.x); hi[1] = std::max(hi[1], mx.y); hi[2] = std::max(hi[2], mx.z);
  }
  void add(const Bounds &b) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
  }
  float area() const {  // Box::surfaceArea, BVH.cpp:142-159
    float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return 2 * (ex * ey + ey * ez + ex * ez);
  }
};

struct TreeNode {
  Bounds box;
  int left = -1, right = -1;
  int first = 0, count = 0;  // count > 0: leaf over order[first, first + count)
};

struct Bins {
  Bounds box[3][kBins];
  int count[3][kBins] = {};
  void add(const Bins &o) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < kBins; ++b) { box[a][b].add(o.box[a][b]); count[a][b] += o.count[a][b]; }
  }
};

struct Split { int axis = -1; float position = 0, cost = kInf; };

class TreeBuilder {
 public:
  explicit TreeBuilder(const std::vector<Prim> &p)
      : prims_(p.size()), order_(p.size()), nodeStore_(std::malloc((2 * p.size() + 1) * sizeof(TreeNode))),
        nodes_(static_cast<TreeNode *>(nodeStore_.get())) {
    // 2 n + 1 nodes at most, 40 bytes each: constructed one by one as the tree grows, not all of them up front
    if (!nodes_) throw std::bad_alloc();
    new (&nodes_[0]) TreeNode();
    // the build only looks at boxes and centroids: 36 bytes per object instead of the whole record
    parallelChunks((int64_t)p.size(), 1 << 14, [&](int64_t b, int64_t e, int) {
      for (int64_t i = b; i < e; ++i) {
        const Prim &o = p[(size_t)i];
        prims_[(size_t)i] = BuildPrim{o.aabbMin, o.aabbMax, o.centroid};
        order_[(size_t)i] = (int)i;
      }
    });
  }
  std::vector<int> &order() { return order_; }

  void build() {
    nodes_[0].first = 0; nodes_[0].count = (int)prims_.size();
    used_ = 1;
    nodes_[0].box = boundsOf(0, (int)prims_.size(), true);
    // top of the tree: one node at a time, every thread on it
    std::vector<int> open{0}, pieces;
    while (!open.empty()) {
      int n = open.back(); open.pop_back();
      if (nodes_[n].count < kSharedNode) { pieces.push_back(n); continue; }
      if (splitNode(n, true)) { open.push_back(nodes_[n].right); open.push_back(nodes_[n].left); }
    }
    // The rest: independent subtrees. A worker that takes a large one splits its root and hands the two halves back, so
    // the work spreads over the threads however unevenly the top of the tree came out; small ones are built whole.
    std::mutex m;
    std::condition_variable cv;
    int busy = 0;
    WorkerFailure failure;
    auto worker = [&]() {
      std::unique_lock<std::mutex> lock(m);
      while (true) {
        cv.wait(lock, [&] { return !pieces.empty() || busy == 0; });
        if (pieces.empty()) { cv.notify_all(); return; }  // nothing queued and nobody who could queue more
        const int n = pieces.back(); pieces.pop_back();
        ++busy;
        lock.unlock();
        bool split = false;
        try {
          if (nodes_[(size_t)n].count >= kTaskNode) split = splitNode(n, false);
          else buildSubtree(n);
        } catch (...) {  // kept for the calling thread; the queue is emptied so that every worker leaves
          failure.record();
          lock.lock();
          pieces.clear();
          --busy;
          cv.notify_all();
          return;
        }
        lock.lock();
        if (split) {
          try {
            pieces.push_back(nodes_[(size_t)n].right); pieces.push_back(nodes_[(size_t)n].left);
          } catch (...) {
            failure.record();
            pieces.clear();
            split = false;
          }
        }
        --busy;
        cv.notify_all();
      }
    };
    const int workers = std::max(1, std::min(hostThreads(), (int)prims_.size() / kTaskNode + 1));
    if (workers <= 1) { worker(); failure.rethrow(); return; }
    // the calling thread is one of the workers; with fewer threads than asked for, those there are share the work
    std::vector<std::thread> pool;
    pool.reserve((size_t)workers);
    for (int w = 1; w < workers; ++w) {
      try { pool.emplace_back(worker); } catch (const std::system_error &) { break; }
    }
    worker();
    for (auto &t : pool) t.join();
    failure.rethrow();
  }

  // pre-order numbering; returns the node count
  int flatten(std::vector<BvhNode> &out) const {
    out.clear();
    std::vector<std::pair<int, int>> todo;  // (tree node, slot of the parent whose right link it fills, or -1)
    todo.emplace_back(0, -1);
    while (!todo.empty()) {
      auto [n, parentSlot] = todo.back(); todo.pop_back();
      int slot = (int)out.size();
      if (parentSlot >= 0) out[(size_t)parentSlot].rightOrStart = slot;
      const TreeNode &t = nodes_[(size_t)n];
      BvhNode f{};
      for (int k = 0; k < 3; ++k) { f.aabbMin[k] = t.box.lo[k]; f.aabbMax[k] = t.box.hi[k]; }
      f.numObjects = t.count;
      f.rightOrStart = t.first;
      out.push_back(f);
      if (t.count == 0) {  // right subtree after the whole left subtree
        todo.emplace_back(t.right, slot);
        todo.emplace_back(t.left, -1);
      }
    }
    return (int)out.size();
  }

 private:
  struct BuildPrim { Vec3 aabbMin, aabbMax, centroid; };
  std::vector<BuildPrim> prims_;
  std::vector<int> order_;  // order_[i] = index into prims_ of the object at BVH position i
  struct FreeDeleter { void operator()(void *p) const { std::free(p); } };
  std::unique_ptr<void, FreeDeleter> nodeStore_;
  TreeNode *nodes_;
  std::atomic<int> used_{0};

  const BuildPrim &at(int i) const { return prims_[(size_t)order_[(size_t)i]]; }

  Bounds boundsOf(int first, int count, bool shared) const {  // updateNodeBounds, BVH.cpp:184-200
    auto range = [&](int64_t b, int64_t e) {
      Bounds r;
      for (int64_t i = b; i < e; ++i) { const BuildPrim &o = at(first + (int)i); r.add(o.aabbMin, o.aabbMax); }
      return r;
    };
    if (!shared) return range(0, count);
    // partial results are combined in data order: min / max keep the first of two equal values, and -0 == +0
    std::vector<Bounds> part((size_t)hostThreads());
    parallelChunks(count, 1 << 14, [&](int64_t b, int64_t e, int chunk) { part[(size_t)chunk] = range(b, e); });
    Bounds r;
    for (const Bounds &p : part) r.add(p);
    return r;
  }

  // findBestBucketSplit, BVH.cpp:72-140, with the three axes binned in one sweep over the objects
  Split bestSplit(const TreeNode &node, bool shared) const {
    const int first = node.first, count = node.count;
    // centroid extent
    auto centroidRange = [&](int64_t b, int64_t e) {
      Bounds r;
      for (int64_t i = b; i < e; ++i) { const Vec3 &c = at(first + (int)i).centroid; r.add(c, c); }
      return r;
    };
    Bounds ext;
    if (shared) {
      std::vector<Bounds> part((size_t)hostThreads());
      parallelChunks(count, 1 << 14, [&](int64_t b, int64_t e, int chunk) { part[(size_t)chunk] = centroidRange(b, e); });
      for (const Bounds &p : part) ext.add(p);
    } else {
      ext = centroidRange(0, count);
    }
    // An axis with zero centroid extent offers no split: the reference computes scale = 16/0 = inf and
    // (c - min) * inf = NaN there, whose int conversion is undefined (INT_MIN on x86: an out-of-bounds write; 0 on
    // aarch64, where every candidate cost then comes out NaN and the axis is never chosen).
    bool use[3]; float toBin[3];
    for (int a = 0; a < 3; ++a) {
      use[a] = ext.hi[a] - ext.lo[a] > 0.0f;
      toBin[a] = use[a] ? kBins / (ext.hi[a] - ext.lo[a]) : 0.0f;
    }
    auto binRange = [&](int64_t b, int64_t e, Bins &bins) {
      for (int64_t i = b; i < e; ++i) {
        const BuildPrim &o = at(first + (int)i);
        const float c[3] = {o.centroid.x, o.centroid.y, o.centroid.z};
        for (int a = 0; a < 3; ++a) {
          if (!use[a]) continue;
          // min(15, int(x)) for every x the reference can convert; a NaN or infinite centroid (for which that conversion
          // is undefined and the reference indexes out of bounds) lands in the first / last bin instead
          const float x = (c[a] - ext.lo[a]) * toBin[a];
          const int k = x >= 0.0f ? (x < (float)kBins ? static_cast<int>(x) : kBins - 1) : 0;
          bins.box[a][k].add(o.aabbMin, o.aabbMax);
          bins.count[a][k]++;
        }
      }
    };
    Bins bins;
    if (shared) {
      std::vector<Bins> part((size_t)hostThreads());
      parallelChunks(count, 1 << 14, [&](int64_t b, int64_t e, int chunk) { binRange(b, e, part[(size_t)chunk]); });
      for (const Bins &p : part) bins.add(p);
    } else {
      binRange(0, count, bins);
    }
    Split best;
    for (int a = 0; a < 3; ++a) {
      if (!use[a]) continue;
      // areas and counts left / right of each of the 15 planes
      float areaL[kBins - 1], areaR[kBins - 1];
      int countL[kBins - 1], countR[kBins - 1];
      Bounds l, r;
      int nl = 0, nr = 0;
      for (int i = 0; i < kBins - 1; ++i) {
        nl += bins.count[a][i]; countL[i] = nl;
        nr += bins.count[a][kBins - 1 - i]; countR[kBins - 2 - i] = nr;
        l.add(bins.box[a][i]); areaL[i] = l.area();
        r.add(bins.box[a][kBins - 1 - i]); areaR[kBins - 2 - i] = r.area();
      }
      // the plane position uses extent/16, not the reciprocal of the binning factor (BVH.cpp:127, 134)
      float width = (ext.hi[a] - ext.lo[a]) / kBins;
      for (int i = 0; i < kBins - 1; ++i) {
        float cost = areaL[i] * countL[i] + areaR[i] * countR[i];
        if (cost < best.cost) { best.cost = cost; best.axis = a; best.position = ext.lo[a] + (i + 1) * width; }
      }
    }
    return best;
  }

  // One step of BVH::partition (BVH.cpp:231-281): false = n stays a leaf.
  bool splitNode(int n, bool shared) {
    TreeNode &node = nodes_[(size_t)n];
    if (node.count <= kLeafObjects) return false;
    Split s = bestSplit(node, shared);
    if (s.cost >= node.box.area() * node.count) return false;
    // The reference regroups with libstdc++'s std::partition on random-access iterators; it is not stable, and the
    // order it leaves is part of the contract, so the same algorithm object is used.
    const int axis = s.axis; const float pos = s.position;
    const int nLeft = shared ? partitionShared(node.first, node.count, axis, pos)
                             : (int)(std::partition(order_.begin() + node.first, order_.begin() + node.first + node.count,
                                                    [&](int idx) { return prims_[(size_t)idx].centroid[axis] < pos; }) -
                                     (order_.begin() + node.first));
    if (nLeft == 0 || nLeft == node.count) return false;
    int l = used_.fetch_add(2), r = l + 1;
    new (&nodes_[l]) TreeNode();
    new (&nodes_[r]) TreeNode();
    nodes_[(size_t)l].first = node.first; nodes_[(size_t)l].count = nLeft;
    nodes_[(size_t)r].first = node.first + nLeft; nodes_[(size_t)r].count = node.count - nLeft;
    nodes_[(size_t)l].box = boundsOf(nodes_[(size_t)l].first, nodes_[(size_t)l].count, shared);
    nodes_[(size_t)r].box = boundsOf(nodes_[(size_t)r].first, nodes_[(size_t)r].count, shared);
    node.left = l; node.right = r; node.count = 0;
    return true;
  }

  // std::partition of order_[first, first + count) by "centroid[axis] < pos", by all threads, with the result of
  // libstdc++'s algorithm for random-access iterators: that one walks inward from both ends and swaps the i-th element
  // from the left that fails the test with the i-th element from the right that passes it, until the two walks meet —
  // at first + (number of passing elements). So: the elements left of that point that fail, in ascending order, change
  // places with the elements right of it that pass, in descending order. Returns the number of passing elements.
  int partitionShared(int first, int count, int axis, float pos) {
    const int threads = hostThreads();
    std::vector<uint8_t> pass((size_t)count);
    std::vector<int64_t> passing((size_t)threads, 0);
    parallelChunks(count, 1 << 14, [&](int64_t b, int64_t e, int chunk) {
      int64_t n = 0;
      for (int64_t i = b; i < e; ++i) {
        const bool p = at(first + (int)i).centroid[axis] < pos;
        pass[(size_t)i] = p; n += p;
      }
      passing[(size_t)chunk] = n;
    });
    int64_t mid = 0;
    for (int64_t n : passing) mid += n;
    // positions (relative to first) that have to move, per chunk and in ascending order
    std::vector<std::vector<int>> failLeft((size_t)threads), passRight((size_t)threads);
    parallelChunks(count, 1 << 14, [&](int64_t b, int64_t e, int chunk) {
      for (int64_t i = b; i < std::min(e, mid); ++i) if (!pass[(size_t)i]) failLeft[(size_t)chunk].push_back((int)i);
      for (int64_t i = std::max(b, mid); i < e; ++i) if (pass[(size_t)i]) passRight[(size_t)chunk].push_back((int)i);
    });
    std::vector<int> fromLeft, fromRight;
    for (const auto &v : failLeft) fromLeft.insert(fromLeft.end(), v.begin(), v.end());
    for (const auto &v : passRight) fromRight.insert(fromRight.end(), v.begin(), v.end());
    const int64_t swaps = (int64_t)fromLeft.size();  // == fromRight.size(): as many fail on the left as pass on the right
    parallelChunks(swaps, 1 << 12, [&](int64_t b, int64_t e, int) {
      for (int64_t i = b; i < e; ++i)
        std::swap(order_[(size_t)first + (size_t)fromLeft[(size_t)i]], order_[(size_t)first + (size_t)fromRight[(size_t)(swaps - 1 - i)]]);
    });
    return (int)mid;
  }

  void buildSubtree(int root) {
    std::vector<int> open{root};
    while (!open.empty()) {
      int n = open.back(); open.pop_back();
      if (splitNode(n, false)) { open.push_back(nodes_[(size_t)n].right); open.push_back(nodes_[(size_t)n].left); }
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
  TreeBuilder tb(scene.prims);
  tb.build();
  int numNodes = tb.flatten(scene.nodes);
  // Scene::objects end up in leaf order (BVH.cpp:258 reorders them in place)
  // through a buffer nobody has initialised (200 bytes per object: filling a fresh vector first would be the longest
  // single-threaded step of the build), gathered and copied back by all threads
  static_assert(std::is_trivially_copyable<Prim>::value, "Prim is moved with memcpy");
  struct FreeDeleter { void operator()(void *p) const { std::free(p); } };
  std::unique_ptr<void, FreeDeleter> raw(std::malloc(scene.prims.size() * sizeof(Prim)));
  if (!raw) throw std::bad_alloc();
  char *buffer = static_cast<char *>(raw.get());
  const std::vector<int> &order = tb.order();
  parallelChunks((int64_t)order.size(), 1 << 14, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; ++i) std::memcpy(buffer + (size_t)i * sizeof(Prim), &scene.prims[(size_t)order[(size_t)i]], sizeof(Prim));
  });
  parallelChunks((int64_t)order.size(), 1 << 14, [&](int64_t b, int64_t e, int) {
    std::memcpy(static_cast<void *>(&scene.prims[(size_t)b]), buffer + (size_t)b * sizeof(Prim), (size_t)(e - b) * sizeof(Prim));
  });
  scene.log.push_back("BVH created with " + std::to_string(numNodes) + " nodes on " +
                      std::to_string(scene.prims.size()) + " objects.");
}

}  // namespace rth
