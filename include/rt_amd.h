/*
 * rt_amd.h — C ABI of the MI355X render-loop library (libraytracer_amd.so).
 *
 * The reference (RichardXiao13/Raytracer) has no plugin/FFI interface; its seam is the
 * C++ call `PNG *Scene::render(int numThreads, int seed)` (src/scene/raytracer.h:50,
 * src/scene/raytracer.cpp:287-300) and, below it, `Scene::findClosestObject`
 * (src/scene/raytracer.h:51, src/scene/raytracer.cpp:187-196). This header is what a
 * binding for that seam binds: the host flattens its scene graph into the plain
 * arrays of RtSceneDesc, hands them over once (rt_scene_create uploads them to HBM),
 * and then calls rt_render / rt_trace_* any number of times.
 *
 * Conventions: plain pointers and sizes, opaque handles, int status codes
 * (0 = RT_OK), no exceptions cross the boundary, the caller owns every host buffer
 * it passes in. Errors: the failing call returns non-zero (or NULL) and
 * rt_last_error() returns a thread-local message.
 *
 * The CPU restatement used by the tests (oracle/rt_oracle.cpp) implements the same
 * entry points with an `rto_` prefix over the same structs.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID 1   /* bad argument / unsupported scene (message says which) */
#define RT_ERR_DEVICE 2    /* HIP runtime error */
#define RT_ERR_NOMEM 3     /* host memory ran out inside the library (device memory that runs out is RT_ERR_DEVICE) */

/* ---- scene description (host arrays; copied during rt_scene_create) ---------------- */

/* One flattened BVH node, byte-for-byte the reference's BVH::FlattenedNode
 * (src/acceleration/BVH.h:49-61): pre-order DFS array, left child = index + 1,
 * right child = rightOrStart; leaf iff numObjects > 0, then rightOrStart is the
 * first object index (src/acceleration/BVH.cpp:411-428). */
typedef struct RtBvhNode {
  float aabbMin[3];
  float aabbMax[3];
  int32_t rightOrStart;
  int32_t numObjects;
} RtBvhNode;

enum { RT_PRIM_TRIANGLE = 0, RT_PRIM_SPHERE = 1 };

/* MaterialType order of src/scene/Material.h:8-14. RT_MAT_DEFAULT is the
 * default-constructed material (one Lambertian lobe, src/scene/Material.cpp:4-11). */
enum { RT_MAT_DEFAULT = 0, RT_MAT_GLASS = 1, RT_MAT_MIRROR = 2, RT_MAT_PLASTIC = 3, RT_MAT_METAL = 4 };

typedef struct RtMaterial { /* src/scene/Material.h:41-55 */
  int32_t type;
  float Kd, Ks, eta, Kr, Kt, Ka, roughness;
} RtMaterial;

typedef struct RtPlane { /* src/scene/Object.h:123-143; not part of the BVH */
  float normal[3];     /* normalised, possibly flipped toward the camera by the parser */
  float point[3];
  float texTopLeft[3];
  float texZoom;
  float texShift[3];
  float color[4];
  int32_t material;
  int32_t texture;     /* index into textures, -1 = none */
} RtPlane;

enum { RT_LIGHT_DISTANT = 0, RT_LIGHT_POINT = 1, RT_LIGHT_ENVIRONMENT = 2 };

typedef struct RtLight { /* src/scene/Object.h:54-105 */
  int32_t type;
  float color[4];
  float v[3];          /* distant: direction; point: center; environment: center */
  float radius;        /* environment only */
  int32_t texture;     /* environment luminance map, -1 = constant colour */
} RtLight;

typedef struct RtTexture { /* linear float RGBA, row-major, what PNG::readFromFile leaves in image_ (src/image/PNG.cpp:125-147) */
  int32_t width, height;
  const float *rgba;
} RtTexture;

typedef struct RtCamera { /* src/scene/Camera.h:18-21 */
  float eye[3], forward[3], right[3], up[3];
} RtCamera;

typedef struct RtOptions { /* Scene width_/height_ + SceneOptions, src/scene/raytracer.h:34-42 */
  int32_t width, height;
  float bias, exposure;
  int32_t maxBounces, numRays;
  int32_t fisheye;
  float focus, lens;
} RtOptions;

typedef struct RtSceneDesc {
  /* BVH over `prims`, exactly as BVH::BVH leaves it (src/acceleration/BVH.cpp:161-182). */
  const RtBvhNode *nodes;
  int32_t numNodes;
  /* Scene::objects in BVH (post-build) order. */
  int32_t numPrims;
  const int32_t *primKind;      /* [numPrims] RT_PRIM_* */
  const float *primHit;         /* [numPrims*12] triangle: p1, normal, e1, e2 (src/scene/Object.h:162-165);
                                   sphere: center, r, then 8 unused floats */
  const float *primNormals;     /* [numPrims*9] triangle n1,n2,n3 (unused for spheres) */
  const float *primTexCoords;   /* [numPrims*9] triangle t1,t2,t3; may be NULL when no triangle is textured */
  const float *primColor;       /* [numPrims*4] Object::color */
  const int32_t *primMaterial;  /* [numPrims] index into materials */
  const int32_t *primTexture;   /* [numPrims] index into textures, -1 = none */
  const int32_t *primOrigIndex; /* [numPrims] insertion index before the BVH reordered objects; reported as hit id */
  int32_t numPlanes;
  const RtPlane *planes;
  int32_t numMaterials;
  const RtMaterial *materials;
  int32_t numLights;
  const RtLight *lights;
  int32_t numTextures;
  const RtTexture *textures;
  RtCamera camera;
  RtOptions options;
  /* Object bounding boxes, [numPrims*3] each: Object::aabbMin / aabbMax (src/scene/Object.h:48-49). Optional (may be
   * NULL) when `nodes` is given. With nodes == NULL and numPrims > 0 they are required: the library then builds a BVH of
   * its own on the device (a linear BVH: Morton order + radix tree, leaves of up to 4 objects) instead of taking the
   * host-built tree of BVH::BVH (src/acceleration/BVH.cpp:72-281), and `prims` may be in any order. That tree is not the
   * reference's: closest hits keep their t, but which of several objects hit at exactly the same t is reported, and the
   * rare hits the reference's visit-order-dependent culling decides, may differ. */
  const float *primAabbMin;
  const float *primAabbMax;
} RtSceneDesc;

/* ---- rendering ---------------------------------------------------------------------- */

enum {
  RT_RNG_COUNTER = 0,     /* per-(pixel,sample) counter-based stream: independent of partition / GPU count */
  RT_RNG_REF_MT19937 = 1  /* oracle only: one std::mt19937 consumed in the reference's -t1 order */
};

typedef struct RtRenderParams {
  int32_t rngMode;        /* RT_RNG_* (the HIP library only accepts RT_RNG_COUNTER) */
  uint32_t seed;
  /* image-tile partition: tiles of tileW x tileH pixels, row-major tile index i is rendered
   * iff i % partCount == partIndex. partCount <= 1 renders the whole image. */
  int32_t partIndex, partCount;
  int32_t tileW, tileH;
  int32_t samplesPerPass; /* camera samples per pixel traced per wavefront pass; 0 = library default: as many as fit
                           * in half of the free device memory, at most 64 GiB of path state (184 B per path with 2 lights) */
  int32_t collectCounters;/* non-zero: count traversal work (inner expansions, primitive tests); slower */
  int32_t skipExposure;   /* non-zero: leave out Scene::expose (src/scene/raytracer.cpp:303-312) */
  /* ---- extensions of the sample loop `for (i = 0; i < numRays; ++i)` (src/scene/raytracer.cpp:41-52); all zero = the
   * reference's loop. Sample i of a pixel is the same camera sample in every mode (the random stream is keyed on it). */
  /* Progressive rendering: this call traces samples [firstSample, firstSample + sampleCount) of every pixel
   * (sampleCount 0 = up to options.numRays) and, with keepAccumulator, adds them to the per-pixel sums the previous
   * call(s) on this scene left (same image size and tile partition required). The frame written is the estimate from
   * all samples accumulated so far (alpha = valid samples / samples taken); once all numRays samples are in, it is
   * bit for bit the frame of a single call. */
  int32_t firstSample, sampleCount, keepAccumulator;
  /* Adaptive sampling (adaptiveThreshold > 0): every adaptiveBatch samples (0 = 16), and not before adaptiveMinSamples
   * (0 = one batch), a pixel stops taking samples when the standard error of the mean luminance of its valid samples,
   * sqrt(var / n), is at most adaptiveThreshold * (mean + 0.05), or when it has no valid sample at all. Fewer camera
   * samples than the reference takes; the estimator of a pixel is the reference's over the samples it did take. */
  float adaptiveThreshold;
  int32_t adaptiveBatch, adaptiveMinSamples;
} RtRenderParams;

typedef struct RtStats {
  uint64_t cameraSamples;
  uint64_t raysClosest;    /* Scene::findClosestObject-equivalent queries from the path loop + point lights */
  uint64_t raysAny;        /* Scene::findAnyObject-equivalent queries (distant / environment lights) */
  uint64_t innerExpansions;/* only with collectCounters: inner-node expansions (two child boxes each), all queries */
  uint64_t primTests;      /* only with collectCounters: triangle/sphere tests, all queries */
  uint64_t planeTests;     /* only with collectCounters */
  uint64_t raysExtend;     /* queries traced by the path-extension kernel (one per live path per bounce) */
  uint64_t raysShadow;     /* queries traced by the light-visibility kernel (one per light per hit) */
  uint64_t innerExtend, primExtend;   /* the two counters above, split by kernel (only with collectCounters) */
  uint64_t innerShadow, primShadow;
  double msTotal;          /* render phase, device-side (HIP events on the library's stream) */
  /* Device time by kernel family. A pass of fewer than 100 Mi paths traces a bounce's visibility queries and the next
   * bounce's rays in ONE launch, timed under msShadow: msExtend then only holds the camera rays. */
  double msRaygen, msExtend, msShade, msShadow, msDirect, msAccumulate, msResolve;
  uint32_t launchesExtend, launchesShadow;
  uint32_t passes;
  /* only with collectCounters: average fraction of a wavefront's 64 lanes that do work in the inner-node loop and in
   * the primitive-test loop, per traversal kernel (1.0 = no divergence loss) */
  double laneUseInnerExtend, laneUsePrimExtend, laneUseInnerShadow, laneUsePrimShadow;
  /* light-visibility queries the reference traces (src/scene/raytracer.cpp:198-206) but the HIP library answers
   * without touching the BVH: the light's unshadowed contribution was already exactly zero, or (raysPlaneAnswered, a
   * subset) one of the scene's planes alone decides the yes/no answer; not included in raysClosest / raysAny /
   * raysShadow, which count traversed queries only */
  uint64_t raysSkipped;
  uint64_t raysPlaneAnswered;
  /* launches (counted in launchesShadow) that traced a bounce's visibility queries AND the next bounce's rays; the time
   * of those rays is then part of msShadow, their number still part of raysExtend */
  uint32_t launchesFused;
} RtStats;

typedef struct RtScene RtScene;

/* Selects `device` (hipSetDevice), builds device copies of the scene. Returns NULL on error.
 * Replaces: the scene state Scene::render reads (src/scene/raytracer.cpp:287-300) — objects,
 * planes, lights, camera, options and the BVH built at :288. */
RtScene *rt_scene_create(const RtSceneDesc *desc, int device);
void rt_scene_destroy(RtScene *scene);

/* Changes options that do not alter geometry (numRays, maxBounces, bias, exposure, fisheye, focus, lens,
 * width, height) and the camera, without re-uploading the scene. */
int rt_scene_set_view(RtScene *scene, const RtCamera *camera, const RtOptions *options);

/* Primary-ray closest hits for the pixel rectangle [x0,x0+w) x [y0,y0+h), un-jittered pixel centres
 * (the numRays == 1 form of src/scene/raytracer.cpp:44-47 followed by Scene::findClosestObject, :187-196).
 * primId[j] = original insertion index of the hit object, numPrims + k for plane k, -1 for a miss;
 * t[j] = IntersectionInfo::t (+inf on a miss). Row-major over the rectangle; host or device buffers. */
int rt_trace_primary(RtScene *scene, int32_t x0, int32_t y0, int32_t w, int32_t h, int32_t *primId, float *t);

/* rt_trace_primary for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc). It has no counterpart in the reference. Same arguments, same results bit for bit; the queue ticket and the
 * device-side staging of ids and t live in a static device array of the library (32 MiB + 1 KiB per GPU, held from
 * library load to process exit) instead of memory the scene allocates, and the call makes no hipMalloc / hipFree. Like
 * rt_trace_primary it is synchronous and runs on the library's own stream (it returns once primId and t are written), so
 * it is not something to record into a HIP graph. Limits: w * h <= rt_trace_primary_static_max_pixels() (4 Mi pixels;
 * larger rectangles are refused with RT_ERR_INVALID); all scenes on one GPU share that GPU's array, so calls that
 * overlap in time on one GPU race (calls on different GPUs do not). The array's device address is looked up on the
 * first call on each GPU; a GPU on which that fails makes this entry point return RT_ERR_DEVICE and nothing else. */
int rt_trace_primary_static(RtScene *scene, int32_t x0, int32_t y0, int32_t w, int32_t h, int32_t *primId, float *t);
int64_t rt_trace_primary_static_max_pixels(void);

/* Scene::findClosestObject (src/scene/raytracer.cpp:187-196) for n arbitrary rays.
 * origins/directions: [n*3] host floats, directions un-normalised as the reference passes them.
 * Outputs (host or device memory, any may be NULL): primId[n], t[n], point[n*3], normal[n*3]. The device-side staging of
 * the trace entry points belongs to the scene and only grows: a repeated call of the same size allocates nothing. */
int rt_trace_closest(RtScene *scene, int64_t n, const float *origins, const float *directions,
                     int32_t *primId, float *t, float *point, float *normal);

/* Scene::findAnyObject != nullptr (src/scene/raytracer.cpp:174-185) for n rays; occluded[n] = 0/1. */
int rt_trace_any(RtScene *scene, int64_t n, const float *origins, const float *directions, uint8_t *occluded);

/* The whole of Scene::render's pixel loop (src/scene/raytracer.cpp:253-285 with the worker chosen at :290-299):
 * writes the clipped (and, unless skipExposure, exposed) linear float RGBA framebuffer, row-major
 * width*height*4 floats, i.e. the contents of the PNG object Scene::render returns. Pixels outside this
 * call's tile partition are written as 0. rgbaOut may be a host pointer or a device pointer of the same device
 * (the copy is made with hipMemcpyDefault), or NULL (result stays on the device, see rt_framebuffer_device).
 * stats may be NULL. */
int rt_render(RtScene *scene, const RtRenderParams *params, float *rgbaOut, RtStats *stats);

/* Device pointer of the float RGBA framebuffer of the last rt_render on this scene (width*height*4 floats). */
const float *rt_framebuffer_device(RtScene *scene);

/* Multi-GPU gather: adds the framebuffer of `src` (last rt_render there) to the framebuffer of `dst`, on dst's device;
 * the copy is device to device (xGMI between GPUs of a node). Meant for scenes that rendered disjoint tile partitions
 * of the same image: pixels outside a partition are exactly 0, so the sum is the gathered frame, bit for bit. This is
 * the join of the reference's worker threads (src/scene/raytracer.cpp:276-279), which write into one shared image. */
int rt_framebuffer_add(RtScene *dst, RtScene *src);

/* The same for several sources at once: each source's own pixels (its partition only, 16 bytes per pixel, kept packed
 * by its rt_render) are copied device to device into staging buffers that `dst` keeps between calls and added into
 * dst's framebuffer. Peer access between the devices is enabled on first use, every source's copy runs on a stream of
 * its own (a different xGMI link per peer), and the call returns after one synchronisation. */
int rt_framebuffer_gather(RtScene *dst, RtScene *const *srcs, int32_t numSrcs);

/* Copies the float RGBA framebuffer of the last rt_render (after any rt_framebuffer_add) to rgbaOut
 * (width*height*4 floats, host or device memory). */
int rt_framebuffer_read(RtScene *scene, float *rgbaOut);

/* PNG::saveToFile's pixel conversion (src/image/PNG.cpp:149-157): sRGB transfer + round(c*255), on the
 * device, from the framebuffer of the last rt_render; rgba8Out: width*height*4 bytes (host). */
int rt_resolve_rgba8(RtScene *scene, uint8_t *rgba8Out);

/* Device time (HIP events on the scene's stream) of the traversal kernel of the last rt_trace_primary / rt_trace_closest /
 * rt_trace_any call on this scene, in milliseconds; copies not included. */
float rt_last_trace_kernel_ms(RtScene *scene);

/* Device time (ms) of the BVH build rt_scene_create ran for this scene because RtSceneDesc.nodes was NULL; 0 when the
 * scene came with a host-built tree. Replaces: the "BVH created" timing the reference prints after BVH::BVH
 * (src/scene/raytracer.cpp:288). */
float rt_scene_bvh_build_ms(RtScene *scene);

const char *rt_last_error(void);
const char *rt_version(void);
int rt_device_count(void); /* GPUs this process can use (0 when there is none) */

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
