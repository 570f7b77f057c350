/*
 * rt_host.h — C ABI of the host-side scene library (librt_host.so, plain C++17, no GPU code).
 *
 * It holds what stays on the CPU in the reference and feeds the render loop: the .sdml / .txt / OBJ
 * front-ends (src/parser/parser.cpp, src/parser/ParserTree.cpp), the scene graph records
 * (src/scene/Object.cpp constructors), the binned-SAH BVH build (src/acceleration/BVH.cpp:72-281)
 * and PNG pixels in/out (src/image/PNG.cpp:125-165). Its product is an RtSceneDesc (rt_amd.h) that
 * rt_scene_create uploads.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RthScene RthScene;

/* readFromFile (src/parser/parser.cpp:362-378). NULL on error (see rth_last_error). */
RthScene *rth_scene_load(const char *path);
/* Empty scene for programmatic construction; default camera/options (src/scene/Camera.h:7-12, raytracer.h:34-42). */
RthScene *rth_scene_new(int32_t width, int32_t height);
void rth_scene_free(RthScene *scene);

/* Programmatic construction (same constructors the parsers use). Materials: returns a material index. */
int32_t rth_scene_add_material(RthScene *scene, const RtMaterial *m); /* type RT_MAT_DEFAULT -> default Lambertian */
int32_t rth_scene_add_texture(RthScene *scene, int32_t width, int32_t height, const float *linearRgba);
int32_t rth_scene_add_texture_file(RthScene *scene, const char *pngPath);
int rth_scene_add_sphere(RthScene *scene, const float center[3], float r, const float color[4], int32_t material, int32_t texture);
int rth_scene_add_plane(RthScene *scene, const float normal[3], float D, const float color[4], int32_t material,
                        const float texTopLeft[3], float texZoom, const float texShift[3], int32_t texture, int32_t flipTowardCamera);
/* Indexed mesh with per-vertex normals, run through loadOBJ's recentre + uniform scale (src/parser/parser.cpp:93-94,
 * 124-133): vertices [numVerts*3], normals [numVerts*3] or NULL, indices [numTris*3] (0-based). */
int rth_scene_add_mesh(RthScene *scene, int64_t numVerts, const float *vertices, const float *normals, int64_t numTris,
                       const int32_t *indices, const float center[3], float scale, const float color[4], int32_t material);
int rth_scene_add_light(RthScene *scene, const RtLight *light); /* environment: v is overwritten with worldCenter() */
int rth_scene_set_camera(RthScene *scene, const float eye[3], const float forward[3], const float up[3]);
int rth_scene_set_options(RthScene *scene, const RtOptions *options);

/* BVH::BVH (src/acceleration/BVH.cpp:161-182) + flattening into the RtSceneDesc arrays. */
int rth_scene_build(RthScene *scene);
/* The RtSceneDesc arrays WITHOUT a tree (nodes = NULL, objects in the order the scene holds them, object boxes filled in):
 * what rt_scene_create takes to build its own BVH on the GPU instead of BVH::BVH on the host. */
int rth_scene_flatten_without_bvh(RthScene *scene);
/* Valid after rth_scene_build / rth_scene_flatten_without_bvh, until the scene is freed or rebuilt. */
const RtSceneDesc *rth_scene_desc(RthScene *scene);
const char *rth_scene_filename(RthScene *scene);
/* Lines the reference prints to stdout while loading/building ("Scanned ...", "BVH created with ..."). */
int32_t rth_scene_log_count(RthScene *scene);
const char *rth_scene_log_line(RthScene *scene, int32_t i);

/* PNG pixels. rth_png_read_linear: what PNG::readFromFile stores (÷255 then sRGB->linear on r,g,b);
 * caller frees with rth_free. rth_png_write_rgba8 / rth_to_srgb8: PNG::saveToFile (src/image/PNG.cpp:149-165). */
int rth_png_read_linear(const char *path, int32_t *width, int32_t *height, float **rgba);
int rth_png_read_rgba8(const char *path, int32_t *width, int32_t *height, uint8_t **rgba8);
int rth_png_write_rgba8(const char *path, int32_t width, int32_t height, const uint8_t *rgba8);
void rth_to_srgb8(const float *rgba, int64_t numPixels, uint8_t *rgba8);
void rth_free(void *p);

/* Host worker threads for parsing, triangle set-up and the BVH build: what `-t N` of the reference's command line
 * (src/main.cpp:14-25) means here. n <= 0 restores the default (hardware threads, at most 16). The thread count never
 * changes a result. */
void rth_set_threads(int32_t n);
int32_t rth_get_threads(void);

const char *rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
