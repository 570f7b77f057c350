// Device-side BVH build (used when a scene arrives without a host-built tree, RtSceneDesc.nodes == NULL).
//
// The reference builds its tree on the host with binned SAH (src/acceleration/BVH.cpp:72-281); the host library restates
// that decision for decision and the kernels then return the reference's hits bit for bit. This builder is the other
// route: a linear BVH (Morton order + the radix-tree construction of Karras, "Maximizing Parallelism in the
// Construction of BVHs, Octrees, and k-d Trees", HPG 2012) built entirely on the GPU in a few milliseconds. It is a
// DIFFERENT tree — other leaves, another object order — so it sits behind that flag: closest hits have the same t,
// but which of two objects hit at exactly the same t is reported may differ, and so may the rare hits that the
// reference's mixed-unit culling (box distances in units of |d|, t in units of 1) drops or keeps depending on the order
// of the visit, and hits on zero-area triangles (their set-up divides by a zero cross product: with infinite edge vectors
// no barycentric comparison rejects anything, so they are hit wherever a ray meets their plane — if the boxes on the way
// to their leaf let the ray through, which is the tree's doing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace rtd {

constexpr int kLbvhLeafMax = 4;  // objects per leaf, the reference's MIN_LEAF_SIZE (src/acceleration/BVH.cpp:233)

struct DeviceBvh {
  float4 *records = nullptr;   // numRecords 64-byte inner records (layout of rt_dev_scene.h), device memory, caller frees
  int32_t numRecords = 0;
  int32_t rootRef = 0;         // packed reference of the root
  float rootMin[3] = {0, 0, 0}, rootMax[3] = {0, 0, 0};
  uint32_t *order = nullptr;   // order[i] = input index of the object at position i of the tree's object array, device memory, caller frees
  bool boxesFinite = true;     // every box coordinate in the records is finite
  float buildMs = 0.0f;        // device time of the whole build (HIP events on `stream`)
};

// aabbMin / aabbMax: device arrays of n * 3 floats (object boxes). n >= 1 and n <= 2^24 (leaf references hold 24-bit object
// positions). Returns false with a message on failure; nothing is left allocated then.
bool buildLbvh(hipStream_t stream, const float *aabbMin, const float *aabbMax, int32_t n, DeviceBvh *out, std::string &error);

// The same interface, another tree: the objects in Morton order are clustered bottom-up by surface area (parallel
// locally-ordered clustering, rt_ploc.h), which gives about the quality of the reference's binned-SAH build
// (src/acceleration/BVH.cpp:72-140) in some tens of milliseconds. Inputs it is unsuitable for (objects that pair up too
// slowly, a tree deeper than the traversal stack) are reported with *unsuitable = true: the caller then builds the linear BVH.
bool buildPloc(hipStream_t stream, const float *aabbMin, const float *aabbMax, int32_t n, DeviceBvh *out, std::string &error, bool *unsuitable);

}  // namespace rtd
