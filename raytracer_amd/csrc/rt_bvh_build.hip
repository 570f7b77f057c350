// Linear BVH built on the GPU: Morton order, radix tree, bottom-up boxes, leaves of up to kLbvhLeafMax objects, emitted
// straight into the 64-byte inner records the traversal kernels read. See rt_bvh_build.h for when it is used.
//
// Steps (all on `stream`):
//   k_centre_bounds   bounding box of the object box centres (what the Morton grid spans)
//   k_morton          39-bit Morton code of every centre (13 bits per axis), value = object index
//   rocprim           radix sort of (code, index) pairs on the 39 code bits
//   k_radix_tree      one thread per internal node: its range of sorted positions and its split (Karras 2012, sec. 4);
//                     equal codes are told apart by their position, so every internal node has exactly two children
//   k_fit_boxes       one thread per object walks up; the second thread to reach a node unites the children's boxes
//   scan + k_emit     internal nodes that span more than kLbvhLeafMax objects become records; smaller ones become leaves
//                     (their objects are contiguous in sorted order)
// Depth: at most 39 levels of code bits plus log2(n) <= 24 levels inside a run of equal codes, below the traversal
// stack's 64 entries.
#include "rt_bvh_build.h"
#include "rt_ploc.h"
#include "rt_top_tree.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>  // before rocprim: its texture iterator calls the host memset
#include <vector>

#include <rocprim/rocprim.hpp>

namespace rtd {
namespace {

constexpr int kBuildBlock = 256;
constexpr int kMortonBitsPerAxis = 13;
constexpr int kMortonBits = 3 * kMortonBitsPerAxis;

inline unsigned gridOf(int64_t n) { return (unsigned)((n + kBuildBlock - 1) / kBuildBlock); }

// order-preserving map float -> uint32 (for atomicMin / atomicMax on floats)
__device__ __forceinline__ uint32_t orderedBits(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fromOrderedBits(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void k_init_bounds(uint32_t *bounds) {  // [0..2] min (ordered bits), [3..5] max
  if (threadIdx.x < 3) bounds[threadIdx.x] = 0xffffffffu;
  else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;
}

__global__ __launch_bounds__(kBuildBlock) void k_centre_bounds(const float *aabbMin, const float *aabbMax, int32_t n, uint32_t *bounds) {
  __shared__ uint32_t sMin[3], sMax[3];
  if (threadIdx.x < 3) { sMin[threadIdx.x] = 0xffffffffu; sMax[threadIdx.x] = 0u; }
  __syncthreads();
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i < n) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float c = (aabbMin[3 * (int64_t)i + k] + aabbMax[3 * (int64_t)i + k]) * 0.5f;
      if (isfinite(c)) { atomicMin(&sMin[k], orderedBits(c)); atomicMax(&sMax[k], orderedBits(c)); }
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) { atomicMin(&bounds[threadIdx.x], sMin[threadIdx.x]); atomicMax(&bounds[3 + threadIdx.x], sMax[threadIdx.x]); }
}

__device__ __forceinline__ uint64_t spreadBits3(uint32_t v) {  // 13 bits -> every third bit
  uint64_t x = v & 0x1fffu;
  x = (x | (x << 16)) & 0x0000ff0000ffull;
  x = (x | (x << 8)) & 0x00f00f00f00full;
  x = (x | (x << 4)) & 0x0c30c30c30c3ull;
  x = (x | (x << 2)) & 0x249249249249ull;
  return x;
}

__global__ __launch_bounds__(kBuildBlock) void k_morton(const float *aabbMin, const float *aabbMax, int32_t n, const uint32_t *bounds,
                                                        uint64_t *keys, uint32_t *vals) {
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= n) return;
  uint32_t cell[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float lo = fromOrderedBits(bounds[k]), hi = fromOrderedBits(bounds[3 + k]);
    const float c = (aabbMin[3 * (int64_t)i + k] + aabbMax[3 * (int64_t)i + k]) * 0.5f;
    float u = (hi > lo && isfinite(c)) ? (c - lo) / (hi - lo) : 0.0f;
    u = fminf(fmaxf(u, 0.0f), 1.0f);
    cell[k] = (uint32_t)fminf(u * (float)(1 << kMortonBitsPerAxis), (float)((1 << kMortonBitsPerAxis) - 1));
  }
  keys[i] = (spreadBits3(cell[0]) << 2) | (spreadBits3(cell[1]) << 1) | spreadBits3(cell[2]);
  vals[i] = (uint32_t)i;
}

// length of the common prefix of the sorted entries i and j; entries with equal codes are told apart by their position
__device__ __forceinline__ int prefixLen(const uint64_t *keys, int32_t n, int32_t i, int32_t j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = keys[i], b = keys[j];
  if (a != b) return __clzll((long long)(a ^ b));
  return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

constexpr uint32_t kLeafBit = 0x80000000u;  // child is the single object at this sorted position

// Karras 2012, algorithm of section 4. Internal node i; children and the range of sorted positions it spans.
__global__ __launch_bounds__(kBuildBlock) void k_radix_tree(const uint64_t *keys, int32_t n, uint32_t *left, uint32_t *right, int32_t *first,
                                                            int32_t *last, int32_t *parentOfNode, int32_t *parentOfLeaf) {
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= n - 1) return;
  const int d = prefixLen(keys, n, i, i + 1) - prefixLen(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int deltaMin = prefixLen(keys, n, i, i - d);
  int32_t lmax = 2;
  while (prefixLen(keys, n, i, i + lmax * d) > deltaMin) lmax <<= 1;
  int32_t l = 0;
  for (int32_t t = lmax >> 1; t >= 1; t >>= 1)
    if (prefixLen(keys, n, i, i + (l + t) * d) > deltaMin) l += t;
  const int32_t j = i + l * d;
  const int deltaNode = prefixLen(keys, n, i, j);
  int32_t s = 0;
  for (int32_t t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (prefixLen(keys, n, i, i + (s + t) * d) > deltaNode) s += t;
    if (t == 1) break;
  }
  const int32_t gamma = i + s * d + (d < 0 ? -1 : 0);
  const int32_t lo = d > 0 ? i : j, hi = d > 0 ? j : i;
  first[i] = lo; last[i] = hi;
  if (lo == gamma) { left[i] = kLeafBit | (uint32_t)gamma; parentOfLeaf[gamma] = i; }
  else { left[i] = (uint32_t)gamma; parentOfNode[gamma] = i; }
  if (hi == gamma + 1) { right[i] = kLeafBit | (uint32_t)(gamma + 1); parentOfLeaf[gamma + 1] = i; }
  else { right[i] = (uint32_t)(gamma + 1); parentOfNode[gamma + 1] = i; }
  if (i == 0) parentOfNode[0] = -1;
}

struct Box {
  float mn[3], mx[3];
};
__device__ __forceinline__ Box objectBox(const float *aabbMin, const float *aabbMax, uint32_t obj) {
  Box b;
#pragma unroll
  for (int k = 0; k < 3; ++k) { b.mn[k] = aabbMin[3 * (int64_t)obj + k]; b.mx[k] = aabbMax[3 * (int64_t)obj + k]; }
  return b;
}
// Union of two boxes (BVH::updateNodeBounds, src/acceleration/BVH.cpp:72-90) with fminf / fmaxf, which return the other
// operand when one is NaN: an object with NaN coordinates (a degenerate triangle) cannot be hit anyway and must not turn
// the boxes of its ancestors into NaN, which would hide the regular objects below them. (The reference's std::min /
// std::max chain keeps or drops a NaN depending on the order of the operands; its tree can lose such subtrees.)
__device__ __forceinline__ Box unite(const Box &a, const Box &b) {
  Box r;
#pragma unroll
  for (int k = 0; k < 3; ++k) { r.mn[k] = fminf(a.mn[k], b.mn[k]); r.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
  return r;
}

// One thread per object walks towards the root. At every node the first thread to arrive stops; the second one finds
// both children's boxes written (its own path, and the sibling's, published before the sibling's thread bumped the
// counter) and unites them.
__global__ __launch_bounds__(kBuildBlock) void k_fit_boxes(const float *aabbMin, const float *aabbMax, const uint32_t *order, int32_t n,
                                                           const uint32_t *left, const uint32_t *right, const int32_t *parentOfNode,
                                                           const int32_t *parentOfLeaf, uint32_t *arrived, Box *nodeBox) {
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= n) return;
  int32_t node = parentOfLeaf[i];
  while (node >= 0) {
    __threadfence();
    if (atomicAdd(&arrived[node], 1u) == 0u) return;
    __threadfence();
    const uint32_t l = left[node], r = right[node];
    const Box bl = (l & kLeafBit) ? objectBox(aabbMin, aabbMax, order[l & ~kLeafBit]) : nodeBox[l];
    const Box br = (r & kLeafBit) ? objectBox(aabbMin, aabbMax, order[r & ~kLeafBit]) : nodeBox[r];
    nodeBox[node] = unite(bl, br);
    node = parentOfNode[node];
  }
}

__global__ __launch_bounds__(kBuildBlock) void k_mark_records(const int32_t *first, const int32_t *last, int32_t numInternal, uint32_t *isRecord) {
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i < numInternal) isRecord[i] = (last[i] - first[i] + 1 > kLbvhLeafMax) ? 1u : 0u;
}

__device__ __forceinline__ int32_t leafRef(int32_t start, int32_t count) { return (int32_t)(0x80000000u | ((uint32_t)count << 24) | (uint32_t)start); }
inline int32_t leafRefHost(int32_t start, int32_t count) { return (int32_t)(0x80000000u | ((uint32_t)count << 24) | (uint32_t)start); }

__global__ __launch_bounds__(kBuildBlock) void k_emit(const float *aabbMin, const float *aabbMax, const uint32_t *order, int32_t numInternal,
                                                      const uint32_t *left, const uint32_t *right, const int32_t *first, const int32_t *last,
                                                      const uint32_t *isRecord, const uint32_t *recordIndex, const Box *nodeBox, float4 *records,
                                                      uint32_t *nonFinite) {
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= numInternal || !isRecord[i]) return;
  Box b[2];
  int32_t ref[2];
  const uint32_t child[2] = {left[i], right[i]};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (child[k] & kLeafBit) {
      const int32_t pos = (int32_t)(child[k] & ~kLeafBit);
      b[k] = objectBox(aabbMin, aabbMax, order[pos]);
      ref[k] = leafRef(pos, 1);
    } else {
      const int32_t c = (int32_t)child[k];
      b[k] = nodeBox[c];
      ref[k] = isRecord[c] ? (int32_t)recordIndex[c] : leafRef(first[c], last[c] - first[c] + 1);
    }
  }
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int a = 0; a < 3; ++a) bad |= !isfinite(b[k].mn[a]) || !isfinite(b[k].mx[a]);
  if (bad) atomicOr(nonFinite, 1u);
  float4 *R = records + 4 * (int64_t)recordIndex[i];
  R[0] = make_float4(b[0].mn[0], b[0].mn[1], b[0].mn[2], b[0].mx[0]);
  R[1] = make_float4(b[0].mx[1], b[0].mx[2], b[1].mn[0], b[1].mn[1]);
  R[2] = make_float4(b[1].mn[2], b[1].mx[0], b[1].mx[1], b[1].mx[2]);
  R[3] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), 0.0f, 0.0f);
}

struct Scratch {  // everything the build allocates besides its two results; freed on every way out
  std::vector<void *> ptrs;
  ~Scratch() { for (void *p : ptrs) (void)hipFree(p); }
  template <typename T>
  bool alloc(T **p, size_t count) {
    void *q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(1, count) * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); return false; }
    ptrs.push_back(q);
    *p = static_cast<T *>(q);
    return true;
  }
};

// ---------------------------------------------------------------------------------------------------- clustering build
// Kernels around the per-element steps of rt_ploc.h (one thread per element).
static_assert(sizeof(ploc::Box) == sizeof(Box), "one box layout");

__global__ __launch_bounds__(kBuildBlock) void k_ploc_init(const float *aabbMin, const float *aabbMax, const uint32_t *order, int32_t n,
                                                           ploc::Box *nodeBox, uint32_t *clusters, int32_t *parent, uint32_t *objects,
                                                           uint32_t *records) {
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= n) return;
  const Box b = objectBox(aabbMin, aabbMax, order[i]);
  ploc::Box pb;
#pragma unroll
  for (int k = 0; k < 3; ++k) { pb.mn[k] = b.mn[k]; pb.mx[k] = b.mx[k]; }
  nodeBox[i] = pb;
  clusters[i] = (uint32_t)i;
  parent[i] = ploc::kNoParent;
  objects[i] = 1u;
  records[i] = 0u;
}

__global__ __launch_bounds__(kBuildBlock) void k_ploc_nearest(uint32_t count, int radius, const uint32_t *clusters, const ploc::Box *nodeBox,
                                                              uint32_t *partner) {
  uint32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i < count) ploc::nearest(i, count, radius, clusters, nodeBox, partner);
}

__global__ __launch_bounds__(kBuildBlock) void k_ploc_merge(uint32_t count, const uint32_t *clusters, const uint32_t *partner, ploc::Box *nodeBox,
                                                            uint32_t n, uint32_t *child0, uint32_t *child1, int32_t *parent, uint32_t *clustersOut,
                                                            uint32_t *keep, uint32_t *made) {
  uint32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= count) return;
  keep[i] = ploc::merge(i, clusters, partner, nodeBox, n, child0, child1, parent, clustersOut, [=]() { return n + atomicAdd(made, 1u); });
}

__global__ __launch_bounds__(kBuildBlock) void k_ploc_compact(uint32_t count, const uint32_t *keep, const uint32_t *pos, const uint32_t *clustersOut,
                                                              uint32_t *clustersNext, uint32_t *countOut) {
  uint32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= count) return;
  if (keep[i]) clustersNext[pos[i]] = clustersOut[i];  // pos[i] < number kept <= count
  if (i == count - 1u) *countOut = pos[i] + keep[i];
}

__global__ __launch_bounds__(kBuildBlock) void k_ploc_count(uint32_t n, const uint32_t *child0, const uint32_t *child1, const int32_t *parent,
                                                            uint32_t *arrived, uint32_t *objects, uint32_t *records) {
  uint32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= n) return;
  // the first thread at a node publishes what it wrote below and stops; the second one sees both children's figures
  ploc::countUp(i, n, kLbvhLeafMax, child0, child1, parent, objects, records, [=](uint32_t k) {
    __threadfence();
    const bool second = atomicAdd(&arrived[k], 1u) == 1u;
    __threadfence();
    return second;
  });
}

__global__ __launch_bounds__(kBuildBlock) void k_ploc_place(uint32_t n, const uint32_t *child0, const uint32_t *child1, const int32_t *parent,
                                                            const uint32_t *objects, const uint32_t *records, const uint32_t *order,
                                                            uint32_t *firstObject, uint32_t *recordIndex, uint32_t *treeOrder, uint32_t *maxDepth) {
  uint32_t v = blockIdx.x * kBuildBlock + threadIdx.x;
  if (v >= 2u * n - 1u) return;
  const ploc::Place p = ploc::place(v, n, child0, child1, parent, objects, records);
  firstObject[v] = p.firstObject;
  recordIndex[v] = p.recordIndex;
  if (v < n) {
    if (p.firstObject < n) treeOrder[p.firstObject] = order[v];  // the objects' places are a permutation of 0 .. n-1 (tests/cpp/ploc_check.cpp)
    else atomicMax(maxDepth, 0xffffffffu);                        // never seen; reported as an unusable tree rather than written out of bounds
  }
  else if (records[v] > 0u) atomicMax(maxDepth, p.depth);
}

__global__ __launch_bounds__(kBuildBlock) void k_ploc_emit(uint32_t n, const uint32_t *child0, const uint32_t *child1, const uint32_t *objects,
                                                           const uint32_t *records, const uint32_t *firstObject, const uint32_t *recordIndex,
                                                           const ploc::Box *nodeBox, uint32_t numRecords, float4 *out, uint32_t *nonFinite) {
  uint32_t k = blockIdx.x * kBuildBlock + threadIdx.x;
  if (k >= n - 1u) return;
  const uint32_t id = n + k;
  if (records[id] == 0u) return;  // inside a leaf
  if (recordIndex[id] >= numRecords) { atomicOr(nonFinite, 2u); return; }  // never seen; the host refuses the tree
  const uint32_t child[2] = {child0[k], child1[k]};
  ploc::Box b[2];
  int32_t ref[2];
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    b[c] = nodeBox[child[c]];
    ref[c] = records[child[c]] > 0u ? (int32_t)recordIndex[child[c]] : leafRef((int32_t)firstObject[child[c]], (int32_t)objects[child[c]]);
#pragma unroll
    for (int a = 0; a < 3; ++a) bad |= !isfinite(b[c].mn[a]) || !isfinite(b[c].mx[a]);
  }
  if (bad) atomicOr(nonFinite, 1u);
  float4 *R = out + 4 * (int64_t)recordIndex[id];
  R[0] = make_float4(b[0].mn[0], b[0].mn[1], b[0].mn[2], b[0].mx[0]);
  R[1] = make_float4(b[0].mx[1], b[0].mx[2], b[1].mn[0], b[1].mn[1]);
  R[2] = make_float4(b[1].mn[2], b[1].mx[0], b[1].mx[1], b[1].mx[2]);
  R[3] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), 0.0f, 0.0f);
}

// what the host needs of the clusters that are left when the clustering stops: box, object count, record count
__global__ __launch_bounds__(kBuildBlock) void k_ploc_gather_roots(uint32_t count, const uint32_t *clusters, const ploc::Box *nodeBox,
                                                                   const uint32_t *objects, const uint32_t *records, ploc::Box *outBox,
                                                                   uint32_t *outObjects, uint32_t *outRecords) {
  uint32_t k = blockIdx.x * kBuildBlock + threadIdx.x;
  if (k >= count) return;
  const uint32_t id = clusters[k];
  outBox[k] = nodeBox[id]; outObjects[k] = objects[id]; outRecords[k] = records[id];
}
__global__ __launch_bounds__(kBuildBlock) void k_ploc_set_parents(uint32_t count, const uint32_t *clusters, const int32_t *parentOfItem, uint32_t numNodes,
                                                                  int32_t *parent) {
  uint32_t k = blockIdx.x * kBuildBlock + threadIdx.x;
  if (k < count && clusters[k] < numNodes) parent[clusters[k]] = parentOfItem[k];
}

constexpr uint32_t kPlocTopClusters = 2048;  // the clustering stops at this many clusters; the levels above them are built top-down on the host
constexpr int kPlocRadius = 16;       // positions searched to either side for a partner
constexpr uint32_t kPlocMaxRounds = 256;  // inputs that pair up slower than this (boxes growing geometrically along a line) get the linear BVH
constexpr uint32_t kPlocMaxDepth = 61;    // records above the deepest record; its leaves then sit at depth <= 63 of the 64-entry traversal stack

}  // namespace

bool buildLbvh(hipStream_t stream, const float *aabbMin, const float *aabbMax, int32_t n, DeviceBvh *out, std::string &error) {
  *out = DeviceBvh{};
  if (n < 1 || n > (1 << 24)) { error = "the device BVH build takes 1 .. 2^24 objects"; return false; }
  auto fail = [&](const char *what, hipError_t e) {
    error = std::string("device BVH build: ") + what + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string());
    if (out->records) (void)hipFree(out->records);
    if (out->order) (void)hipFree(out->order);
    *out = DeviceBvh{};
    return false;
  };
  Scratch sc;
  const int32_t numInternal = n - 1;
  uint32_t *bounds = nullptr, *vals = nullptr, *left = nullptr, *right = nullptr, *arrived = nullptr, *isRecord = nullptr, *recordIndex = nullptr, *flags = nullptr;
  uint64_t *keys = nullptr, *keysSorted = nullptr;
  int32_t *first = nullptr, *last = nullptr, *parentOfNode = nullptr, *parentOfLeaf = nullptr;
  Box *nodeBox = nullptr;
  if (!sc.alloc(&bounds, 8) || !sc.alloc(&keys, n) || !sc.alloc(&keysSorted, n) || !sc.alloc(&vals, n) || !sc.alloc(&left, numInternal) ||
      !sc.alloc(&right, numInternal) || !sc.alloc(&first, numInternal) || !sc.alloc(&last, numInternal) || !sc.alloc(&parentOfNode, numInternal) ||
      !sc.alloc(&parentOfLeaf, n) || !sc.alloc(&arrived, numInternal) || !sc.alloc(&isRecord, numInternal) || !sc.alloc(&recordIndex, numInternal + 1) ||
      !sc.alloc(&nodeBox, numInternal) || !sc.alloc(&flags, 2))
    return fail("out of device memory", hipSuccess);
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void **>(&out->order), (size_t)n * sizeof(uint32_t))) != hipSuccess) return fail("hipMalloc", e);

  // the two timing events are destroyed on every way out of this function
  struct Events {
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  if ((e = hipEventCreate(&ev.a)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreate(&ev.b)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventRecord(ev.a, stream)) != hipSuccess) return fail("hipEventRecord", e);
  k_init_bounds<<<1, 64, 0, stream>>>(bounds);
  k_centre_bounds<<<gridOf(n), kBuildBlock, 0, stream>>>(aabbMin, aabbMax, n, bounds);
  k_morton<<<gridOf(n), kBuildBlock, 0, stream>>>(aabbMin, aabbMax, n, bounds, keys, vals);
  if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the bounds / Morton-code kernels", e);
  {
    size_t tempBytes = 0;
    if ((e = rocprim::radix_sort_pairs(nullptr, tempBytes, keys, keysSorted, vals, out->order, (size_t)n, 0, kMortonBits, stream)) != hipSuccess)
      return fail("radix sort (size query)", e);
    void *temp = nullptr;
    if (!sc.alloc(reinterpret_cast<unsigned char **>(&temp), tempBytes)) return fail("out of device memory", hipSuccess);
    if ((e = rocprim::radix_sort_pairs(temp, tempBytes, keys, keysSorted, vals, out->order, (size_t)n, 0, kMortonBits, stream)) != hipSuccess)
      return fail("radix sort", e);
  }
  if ((e = hipMemsetAsync(flags, 0, 2 * sizeof(uint32_t), stream)) != hipSuccess) return fail("hipMemsetAsync", e);
  Box rootBox{};
  if (numInternal > 0) {
    if ((e = hipMemsetAsync(arrived, 0, (size_t)numInternal * sizeof(uint32_t), stream)) != hipSuccess) return fail("hipMemsetAsync", e);
    k_radix_tree<<<gridOf(numInternal), kBuildBlock, 0, stream>>>(keysSorted, n, left, right, first, last, parentOfNode, parentOfLeaf);
    k_fit_boxes<<<gridOf(n), kBuildBlock, 0, stream>>>(aabbMin, aabbMax, out->order, n, left, right, parentOfNode, parentOfLeaf, arrived, nodeBox);
    k_mark_records<<<gridOf(numInternal), kBuildBlock, 0, stream>>>(first, last, numInternal, isRecord);
    if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the tree / box kernels", e);
    size_t tempBytes = 0;
    if ((e = rocprim::exclusive_scan(nullptr, tempBytes, isRecord, recordIndex, 0u, (size_t)numInternal, rocprim::plus<uint32_t>(), stream)) != hipSuccess)
      return fail("scan (size query)", e);
    void *temp = nullptr;
    if (!sc.alloc(reinterpret_cast<unsigned char **>(&temp), tempBytes)) return fail("out of device memory", hipSuccess);
    if ((e = rocprim::exclusive_scan(temp, tempBytes, isRecord, recordIndex, 0u, (size_t)numInternal, rocprim::plus<uint32_t>(), stream)) != hipSuccess)
      return fail("scan", e);
    uint32_t lastIndex = 0, lastFlag = 0;
    if ((e = hipMemcpyAsync(&lastIndex, recordIndex + numInternal - 1, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
    if ((e = hipMemcpyAsync(&lastFlag, isRecord + numInternal - 1, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
    if ((e = hipMemcpyAsync(&rootBox, nodeBox, sizeof(Box), hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("build kernels", e);
    out->numRecords = (int32_t)(lastIndex + lastFlag);
    if ((e = hipMalloc(reinterpret_cast<void **>(&out->records), std::max<size_t>(1, (size_t)out->numRecords) * 64)) != hipSuccess) return fail("hipMalloc", e);
    if (out->numRecords > 0)
      k_emit<<<gridOf(numInternal), kBuildBlock, 0, stream>>>(aabbMin, aabbMax, out->order, numInternal, left, right, first, last, isRecord, recordIndex,
                                                             nodeBox, out->records, flags);
    out->rootRef = n > kLbvhLeafMax ? 0 : (int32_t)(0x80000000u | ((uint32_t)n << 24));  // node 0 spans everything: record 0, or one leaf
  } else {
    if ((e = hipMalloc(reinterpret_cast<void **>(&out->records), 64)) != hipSuccess) return fail("hipMalloc", e);
    out->rootRef = (int32_t)(0x80000000u | (1u << 24));
    float mn[3], mx[3];
    if ((e = hipMemcpyAsync(mn, aabbMin, 12, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
    if ((e = hipMemcpyAsync(mx, aabbMax, 12, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("copy", e);
    std::memcpy(rootBox.mn, mn, 12); std::memcpy(rootBox.mx, mx, 12);
  }
  if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the record kernel", e);
  if ((e = hipEventRecord(ev.b, stream)) != hipSuccess) return fail("hipEventRecord", e);
  uint32_t hostFlags[2] = {0, 0};
  if ((e = hipMemcpyAsync(hostFlags, flags, sizeof hostFlags, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("build kernels", e);
  out->boxesFinite = hostFlags[0] == 0u;
  for (int k = 0; k < 3; ++k) out->boxesFinite = out->boxesFinite && std::isfinite(rootBox.mn[k]) && std::isfinite(rootBox.mx[k]);
  if ((e = hipGetLastError()) != hipSuccess) return fail("kernel launch", e);
  if (hipEventElapsedTime(&out->buildMs, ev.a, ev.b) != hipSuccess) out->buildMs = 0.0f;
  std::memcpy(out->rootMin, rootBox.mn, 12);
  std::memcpy(out->rootMax, rootBox.mx, 12);
  return true;
}

bool buildPloc(hipStream_t stream, const float *aabbMin, const float *aabbMax, int32_t n, DeviceBvh *out, std::string &error, bool *unsuitable) {
  *out = DeviceBvh{};
  *unsuitable = false;
  if (n < 1 || n > (1 << 24)) { error = "the device BVH build takes 1 .. 2^24 objects"; return false; }
  auto fail = [&](const char *what, hipError_t e) {
    error = std::string("device BVH build (clustering): ") + what + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string());
    if (out->records) (void)hipFree(out->records);
    if (out->order) (void)hipFree(out->order);
    *out = DeviceBvh{};
    return false;
  };
  Scratch sc;
  const uint32_t un = (uint32_t)n, numNodes = 2u * un - 1u, numInternal = un - 1u;
  uint32_t *bounds = nullptr, *vals = nullptr, *order = nullptr, *clusterA = nullptr, *clusterB = nullptr, *clusterOut = nullptr, *partner = nullptr,
           *keep = nullptr, *pos = nullptr, *child0 = nullptr, *child1 = nullptr, *objects = nullptr, *records = nullptr, *firstObject = nullptr,
           *recordIndex = nullptr, *arrived = nullptr, *words = nullptr;
  uint64_t *keys = nullptr, *keysSorted = nullptr;
  int32_t *parent = nullptr;
  ploc::Box *nodeBox = nullptr;
  if (!sc.alloc(&bounds, 8) || !sc.alloc(&keys, un) || !sc.alloc(&keysSorted, un) || !sc.alloc(&vals, un) || !sc.alloc(&order, un) ||
      !sc.alloc(&clusterA, un) || !sc.alloc(&clusterB, un) || !sc.alloc(&clusterOut, un) || !sc.alloc(&partner, un) || !sc.alloc(&keep, un) ||
      !sc.alloc(&pos, un) || !sc.alloc(&child0, numInternal) || !sc.alloc(&child1, numInternal) || !sc.alloc(&objects, numNodes) ||
      !sc.alloc(&records, numNodes) || !sc.alloc(&firstObject, numNodes) || !sc.alloc(&recordIndex, numNodes) || !sc.alloc(&arrived, numInternal) ||
      !sc.alloc(&parent, numNodes) || !sc.alloc(&nodeBox, numNodes) || !sc.alloc(&words, 8))
    return fail("out of device memory", hipSuccess);
  uint32_t *made = words, *countOut = words + 1, *maxDepth = words + 2, *nonFinite = words + 3;
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void **>(&out->order), (size_t)un * sizeof(uint32_t))) != hipSuccess) return fail("hipMalloc", e);
  struct Events {
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  if ((e = hipEventCreate(&ev.a)) != hipSuccess || (e = hipEventCreate(&ev.b)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventRecord(ev.a, stream)) != hipSuccess) return fail("hipEventRecord", e);
  // ---- Morton order of the objects (as for the linear BVH)
  k_init_bounds<<<1, 64, 0, stream>>>(bounds);
  k_centre_bounds<<<gridOf(n), kBuildBlock, 0, stream>>>(aabbMin, aabbMax, n, bounds);
  k_morton<<<gridOf(n), kBuildBlock, 0, stream>>>(aabbMin, aabbMax, n, bounds, keys, vals);
  if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the bounds / Morton-code kernels", e);
  size_t sortBytes = 0, scanBytes = 0;
  if ((e = rocprim::radix_sort_pairs(nullptr, sortBytes, keys, keysSorted, vals, order, (size_t)un, 0, kMortonBits, stream)) != hipSuccess)
    return fail("radix sort (size query)", e);
  if ((e = rocprim::exclusive_scan(nullptr, scanBytes, keep, pos, 0u, (size_t)un, rocprim::plus<uint32_t>(), stream)) != hipSuccess)
    return fail("scan (size query)", e);
  unsigned char *temp = nullptr;
  if (!sc.alloc(&temp, std::max(sortBytes, scanBytes))) return fail("out of device memory", hipSuccess);
  if ((e = rocprim::radix_sort_pairs(temp, sortBytes, keys, keysSorted, vals, order, (size_t)un, 0, kMortonBits, stream)) != hipSuccess)
    return fail("radix sort", e);
  // ---- clustering rounds
  if ((e = hipMemsetAsync(words, 0, 8 * sizeof(uint32_t), stream)) != hipSuccess) return fail("hipMemsetAsync", e);
  k_ploc_init<<<gridOf(n), kBuildBlock, 0, stream>>>(aabbMin, aabbMax, order, n, nodeBox, clusterA, parent, objects, records);
  if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the set-up kernel", e);
  uint32_t count = un, rounds = 0;
  uint32_t *cur = clusterA, *next = clusterB;
  int radius = kPlocRadius;
  if (const char *x = std::getenv("RT_PLOC_RADIUS")) radius = std::min(256, std::max(1, std::atoi(x)));  // experiments
  uint32_t topTarget = kPlocTopClusters;
  if (const char *x = std::getenv("RT_PLOC_TOP")) topTarget = (uint32_t)std::min(8192, std::max(1, std::atoi(x)));  // experiments; 1 = clustering to the root
  while (count > topTarget) {
    if (++rounds > kPlocMaxRounds) { *unsuitable = true; return fail("the objects pair up too slowly", hipSuccess); }
    k_ploc_nearest<<<gridOf(count), kBuildBlock, 0, stream>>>(count, radius, cur, nodeBox, partner);
    k_ploc_merge<<<gridOf(count), kBuildBlock, 0, stream>>>(count, cur, partner, nodeBox, un, child0, child1, parent, clusterOut, keep, made);
    if ((e = hipGetLastError()) != hipSuccess) return fail("launch of a clustering round", e);
    size_t bytes = scanBytes;
    if ((e = rocprim::exclusive_scan(temp, bytes, keep, pos, 0u, (size_t)count, rocprim::plus<uint32_t>(), stream)) != hipSuccess) return fail("scan", e);
    k_ploc_compact<<<gridOf(count), kBuildBlock, 0, stream>>>(count, keep, pos, clusterOut, next, countOut);
    uint32_t left = 0;
    if ((e = hipMemcpyAsync(&left, countOut, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("clustering round", e);
    if (left == 0u || left >= count) return fail("a clustering round merged nothing", hipSuccess);  // cannot happen: the best pair over all is mutual
    count = left;
    std::swap(cur, next);
  }
  // ---- layout: object counts and record counts bottom-up (below the clusters that are left) ...
  if ((e = hipMemsetAsync(arrived, 0, (size_t)std::max(1u, numInternal) * sizeof(uint32_t), stream)) != hipSuccess) return fail("hipMemsetAsync", e);
  if (numInternal > 0) k_ploc_count<<<gridOf(n), kBuildBlock, 0, stream>>>(un, child0, child1, parent, arrived, objects, records);
  if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the count kernel", e);
  uint32_t root = 0;
  if (count == 1u) {
    if ((e = hipMemcpyAsync(&root, cur, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
  } else {
    // ... the levels above them top-down, on the host (rt_top_tree.h): a few thousand boxes out, as many nodes back
    ploc::Box *topBox = nullptr;
    uint32_t *topObjects = nullptr, *topRecords = nullptr;
    int32_t *topParents = nullptr;
    if (!sc.alloc(&topBox, count) || !sc.alloc(&topObjects, count) || !sc.alloc(&topRecords, count) || !sc.alloc(&topParents, count))
      return fail("out of device memory", hipSuccess);
    k_ploc_gather_roots<<<gridOf(count), kBuildBlock, 0, stream>>>(count, cur, nodeBox, objects, records, topBox, topObjects, topRecords);
    if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the gather kernel", e);
    std::vector<uint32_t> ids(count), hObjects(count), hRecords(count);
    std::vector<ploc::Box> hBox(count);
    uint32_t madeSoFar = 0;
    if ((e = hipMemcpyAsync(ids.data(), cur, (size_t)count * 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(hBox.data(), topBox, (size_t)count * sizeof(ploc::Box), hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(hObjects.data(), topObjects, (size_t)count * 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(hRecords.data(), topRecords, (size_t)count * 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(&madeSoFar, made, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess)
      return fail("copy", e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("clustering", e);
    if (madeSoFar + (count - 1u) != numInternal) return fail("number of internal nodes before the top", hipSuccess);
    for (uint32_t k = 0; k < count; ++k)
      if (ids[k] >= numNodes) return fail("cluster id out of range", hipSuccess);
    const uint32_t firstNew = un + madeSoFar;
    const ploc::TopTree top = ploc::buildTopTree(ids, hBox, hObjects, hRecords, firstNew, kLbvhLeafMax);
    const size_t m1 = (size_t)count - 1;
    if (top.child0.size() != m1 || top.parentOfItem.size() != count) return fail("top tree size", hipSuccess);
    // new nodes occupy the ids firstNew .. 2n - 2: contiguous in every per-node array
    if ((e = hipMemcpyAsync(child0 + madeSoFar, top.child0.data(), m1 * 4, hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(child1 + madeSoFar, top.child1.data(), m1 * 4, hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(objects + firstNew, top.objects.data(), m1 * 4, hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(records + firstNew, top.records.data(), m1 * 4, hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(parent + firstNew, top.parent.data(), m1 * 4, hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(nodeBox + firstNew, top.box.data(), m1 * sizeof(ploc::Box), hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(topParents, top.parentOfItem.data(), (size_t)count * 4, hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(made, &numInternal, 4, hipMemcpyHostToDevice, stream)) != hipSuccess)
      return fail("copy", e);
    k_ploc_set_parents<<<gridOf(count), kBuildBlock, 0, stream>>>(count, cur, topParents, numNodes, parent);
    if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the parent kernel", e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("top of the tree", e);  // the host vectors above go out of scope
    root = top.root;
  }
  k_ploc_place<<<gridOf(numNodes), kBuildBlock, 0, stream>>>(un, child0, child1, parent, objects, records, order, firstObject, recordIndex, out->order, maxDepth);
  if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the layout kernels", e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("layout kernels", e);
  if (root >= numNodes) return fail("no root", hipSuccess);
  uint32_t numRecords = 0, rootObjects = 0, hostWords[4] = {0, 0, 0, 0};
  ploc::Box rootBox{};
  if ((e = hipMemcpyAsync(&numRecords, records + root, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
  if ((e = hipMemcpyAsync(&rootObjects, objects + root, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
  if ((e = hipMemcpyAsync(&rootBox, nodeBox + root, sizeof rootBox, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
  if ((e = hipMemcpyAsync(hostWords, words, sizeof hostWords, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("copy", e);
  if (hostWords[0] != numInternal || rootObjects != un) return fail("the tree does not hold every object once", hipSuccess);
  if (hostWords[2] > kPlocMaxDepth) { *unsuitable = true; return fail("tree deeper than the traversal stack", hipSuccess); }
  if (numRecords > numInternal) return fail("number of records", hipSuccess);
  out->numRecords = (int32_t)numRecords;
  if ((e = hipMalloc(reinterpret_cast<void **>(&out->records), std::max<size_t>(1, (size_t)numRecords) * 64)) != hipSuccess) return fail("hipMalloc", e);
  if (numRecords > 0) {
    k_ploc_emit<<<gridOf(numInternal), kBuildBlock, 0, stream>>>(un, child0, child1, objects, records, firstObject, recordIndex, nodeBox, numRecords, out->records, nonFinite);
    if ((e = hipGetLastError()) != hipSuccess) return fail("launch of the record kernel", e);
  }
  out->rootRef = numRecords > 0 ? 0 : leafRefHost(0, n);  // the root is record 0 (pre-order), or one leaf
  if ((e = hipEventRecord(ev.b, stream)) != hipSuccess) return fail("hipEventRecord", e);
  uint32_t flag = 0;
  if ((e = hipMemcpyAsync(&flag, nonFinite, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail("copy", e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("record kernel", e);
  if (flag & 2u) return fail("record index out of range", hipSuccess);
  out->boxesFinite = flag == 0u;
  for (int k = 0; k < 3; ++k) out->boxesFinite = out->boxesFinite && std::isfinite(rootBox.mn[k]) && std::isfinite(rootBox.mx[k]);
  if (hipEventElapsedTime(&out->buildMs, ev.a, ev.b) != hipSuccess) out->buildMs = 0.0f;
  std::memcpy(out->rootMin, rootBox.mn, 12);
  std::memcpy(out->rootMax, rootBox.mx, 12);
  return true;
}

}  // namespace rtd
