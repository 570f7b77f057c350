// C ABI of the host scene library (include/rt_host.h).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "rt_host.h"
#include "rth_scene.h"
#include "rth_threads.h"

using namespace rth;

struct RthScene {
  std::unique_ptr<Scene> scene;
  FlatScene flat;
  bool built = false;
};

namespace {
thread_local std::string g_error;
int fail(const std::string &msg) { g_error = msg; return RT_ERR_INVALID; }
// No exception may cross the C boundary: every entry point that allocates or parses runs its body through this.
template <class R, class F>
R guarded(R onError, F &&body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    g_error = "out of memory";
  } catch (const std::exception &e) {
    g_error = std::string("exception: ") + e.what();
  } catch (...) {
    g_error = "unknown exception";
  }
  return onError;
}
Vec3 v3(const float *p) { return Vec3(p[0], p[1], p[2]); }
Color c4(const float *p) { return Color(p[0], p[1], p[2], p[3]); }
}  // namespace

extern "C" {

void rth_set_threads(int32_t n) { rth::setHostThreads(n); }
int32_t rth_get_threads(void) { return rth::hostThreads(); }

const char *rth_last_error(void) { return g_error.c_str(); }

RthScene *rth_scene_load(const char *path) {
  return guarded<RthScene *>(nullptr, [&]() -> RthScene * {
    std::string err;
    std::unique_ptr<Scene> s = readFromFile(path ? path : "", err);
    if (!s) { g_error = err; return nullptr; }
    auto *h = new RthScene();
    h->scene = std::move(s);
    return h;
  });
}

RthScene *rth_scene_new(int32_t width, int32_t height) {
  return guarded<RthScene *>(nullptr, [&]() -> RthScene * {
    auto *h = new RthScene();
    h->scene = std::make_unique<Scene>();
    h->scene->width = width;
    h->scene->height = height;
    return h;
  });
}

void rth_scene_free(RthScene *scene) { delete scene; }

int32_t rth_scene_add_material(RthScene *h, const RtMaterial *m) {
  return guarded<int32_t>(-1, [&]() -> int32_t {
    if (!h || !m) { fail("null argument"); return -1; }
    if (m->type == RT_MAT_DEFAULT) return h->scene->addMaterialDefault();
    if (m->type < RT_MAT_DEFAULT || m->type > RT_MAT_METAL) { fail("bad material type"); return -1; }
    return h->scene->addMaterial(m->Kd, m->Ks, m->eta, m->Kr, m->Kt, m->Ka, m->roughness, m->type);
  });
}

int32_t rth_scene_add_texture(RthScene *h, int32_t width, int32_t height, const float *rgba) {
  return guarded<int32_t>(-1, [&]() -> int32_t {
    if (!h || !rgba || width <= 0 || height <= 0) { fail("bad texture"); return -1; }
    auto img = std::make_shared<Image>();
    img->width = width; img->height = height;
    img->rgba.assign(rgba, rgba + (size_t)width * height * 4);
    h->scene->textures.push_back(img);
    h->scene->texturePaths.push_back("<memory:" + std::to_string(h->scene->textures.size()) + ">");
    return (int32_t)h->scene->textures.size() - 1;
  });
}

int32_t rth_scene_add_texture_file(RthScene *h, const char *pngPath) {
  return guarded<int32_t>(-1, [&]() -> int32_t {
    if (!h || !pngPath) { fail("null argument"); return -1; }
    return h->scene->loadTexture(pngPath);
  });
}

int rth_scene_add_sphere(RthScene *h, const float center[3], float r, const float color[4], int32_t material, int32_t texture) {
  return guarded<int>(RT_ERR_NOMEM, [&]() -> int {
    if (!h) return fail("null scene");
    if (material < 0 || material >= (int)h->scene->materials.size()) return fail("bad material index");
    h->scene->addPrim(makeSphere(v3(center), r, c4(color), material, texture));
    h->built = false;
    return RT_OK;
  });
}

int rth_scene_add_plane(RthScene *h, const float normal[3], float D, const float color[4], int32_t material,
                        const float texTopLeft[3], float texZoom, const float texShift[3], int32_t texture,
                        int32_t flipTowardCamera) {
  return guarded<int>(RT_ERR_NOMEM, [&]() -> int {
    if (!h) return fail("null scene");
    if (material < 0 || material >= (int)h->scene->materials.size()) return fail("bad material index");
    Plane p = makePlane(v3(normal), D, c4(color), material, texTopLeft ? v3(texTopLeft) : Vec3(), texZoom,
                        texShift ? v3(texShift) : Vec3(), texture);
    if (flipTowardCamera && dot(h->scene->camera.forward, p.normal) > 0) p.normal = -p.normal;
    h->scene->planes.push_back(p);
    return RT_OK;
  });
}

int rth_scene_add_mesh(RthScene *h, int64_t numVerts, const float *vertices, const float *normals, int64_t numTris,
                       const int32_t *indices, const float center[3], float scale, const float color[4], int32_t material) {
  return guarded<int>(RT_ERR_NOMEM, [&]() -> int {
    if (!h || !vertices || !indices || numVerts <= 0) return fail("bad mesh");
    if (material < 0 || material >= (int)h->scene->materials.size()) return fail("bad material index");
    const float inf = std::numeric_limits<float>::infinity();
    Vec3 extMin(inf, inf, inf), extMax(-inf, -inf, -inf);
    for (int64_t i = 0; i < numVerts; ++i) {
      const float *p = vertices + i * 3;
      extMin = Vec3(std::min(extMin.x, p[0]), std::min(extMin.y, p[1]), std::min(extMin.z, p[2]));
      extMax = Vec3(std::max(extMax.x, p[0]), std::max(extMax.y, p[1]), std::max(extMax.z, p[2]));
    }
    Vec3 shift = v3(center) - (extMax + extMin) / 2;
    float scaleFactor = scale / maxDimension(extMax - extMin);
    Color col = c4(color);
    for (int64_t t = 0; t < numTris * 3; ++t)
      if (indices[t] < 0 || indices[t] >= numVerts) return fail("mesh index out of range");
    std::vector<Prim> tris((size_t)numTris);
    parallelChunks(numTris, 1 << 12, [&](int64_t b, int64_t e, int) {  // triangle set-up is independent per triangle
      for (int64_t t = b; t < e; ++t) {
        int32_t i = indices[t * 3], j = indices[t * 3 + 1], k = indices[t * 3 + 2];
        Prim tri = makeTriangle(scaleFactor * v3(vertices + 3 * (int64_t)i) + shift, scaleFactor * v3(vertices + 3 * (int64_t)j) + shift,
                                scaleFactor * v3(vertices + 3 * (int64_t)k) + shift, col, material);
        if (normals) {
          tri.n1 = v3(normals + 3 * (int64_t)i); tri.n2 = v3(normals + 3 * (int64_t)j); tri.n3 = v3(normals + 3 * (int64_t)k);
        }
        tris[(size_t)t] = tri;
      }
    });
    h->scene->prims.reserve(h->scene->prims.size() + (size_t)numTris);
    for (const Prim &tri : tris) h->scene->addPrim(tri);  // in order: the centroid sum is a float sum
    h->built = false;
    return RT_OK;
  });
}

int rth_scene_add_light(RthScene *h, const RtLight *light) {
  return guarded<int>(RT_ERR_NOMEM, [&]() -> int {
    if (!h || !light) return fail("null argument");
    Light l;
    l.type = light->type;
    l.color = c4(light->color);
    l.v = v3(light->v);
    l.radius = light->radius;
    l.texture = light->texture;
    if (l.type == RT_LIGHT_ENVIRONMENT) l.v = h->scene->worldCenter();
    else if (l.type != RT_LIGHT_DISTANT && l.type != RT_LIGHT_POINT) return fail("bad light type");
    h->scene->lights.push_back(l);
    return RT_OK;
  });
}

int rth_scene_set_camera(RthScene *h, const float eye[3], const float forward[3], const float up[3]) {
  if (!h) return fail("null scene");
  Camera c;
  c.eye = v3(eye);
  c.setForward(v3(forward));
  c.setUp(v3(up));
  h->scene->camera = c;
  return RT_OK;
}

int rth_scene_set_options(RthScene *h, const RtOptions *o) {
  if (!h || !o) return fail("null argument");
  h->scene->width = o->width; h->scene->height = o->height;
  h->scene->options.bias = o->bias; h->scene->options.exposure = o->exposure;
  h->scene->options.maxBounces = o->maxBounces; h->scene->options.numRays = o->numRays;
  h->scene->options.fisheye = o->fisheye != 0; h->scene->options.focus = o->focus; h->scene->options.lens = o->lens;
  if (h->built) {
    h->flat.desc.options = *o;
  }
  return RT_OK;
}

int rth_scene_build(RthScene *h) {
  return guarded<int>(RT_ERR_NOMEM, [&]() -> int {
    if (!h) return fail("null scene");
    if (!h->built) {
      if (h->scene->bvhBuilt) {
        // objects were added after a previous build: the builder's result depends on the order it
        // starts from, so go back to insertion order first
        std::sort(h->scene->prims.begin(), h->scene->prims.end(),
                  [](const Prim &a, const Prim &b) { return a.origIndex < b.origIndex; });
      }
      buildBVH(*h->scene);
    }
    std::string err;
    if (!flatten(*h->scene, h->flat, err)) return fail(err);
    h->built = true;
    return RT_OK;
  });
}

int rth_scene_flatten_without_bvh(RthScene *h) {
  return guarded<int>(RT_ERR_NOMEM, [&]() -> int {
    if (!h) return fail("null scene");
    std::string err;
    if (!flatten(*h->scene, h->flat, err, /*requireBvh*/ false)) return fail(err);
    h->built = true;
    return RT_OK;
  });
}

const RtSceneDesc *rth_scene_desc(RthScene *h) {
  if (!h || !h->built) { fail("scene not built"); return nullptr; }
  return &h->flat.desc;
}

const char *rth_scene_filename(RthScene *h) { return h ? h->scene->filename.c_str() : ""; }
int32_t rth_scene_log_count(RthScene *h) { return h ? (int32_t)h->scene->log.size() : 0; }
const char *rth_scene_log_line(RthScene *h, int32_t i) {
  if (!h || i < 0 || i >= (int32_t)h->scene->log.size()) return "";
  return h->scene->log[i].c_str();
}

int rth_png_read_linear(const char *path, int32_t *width, int32_t *height, float **rgba) {
  return guarded<int>(RT_ERR_NOMEM, [&]() -> int {
    Image img;
    std::string err;
    if (!readPNG(path, img, err)) return fail(err);
    *width = img.width; *height = img.height;
    *rgba = (float *)std::malloc(img.rgba.size() * sizeof(float));
    if (!*rgba) return RT_ERR_NOMEM;
    std::memcpy(*rgba, img.rgba.data(), img.rgba.size() * sizeof(float));
    return RT_OK;
  });
}

int rth_png_read_rgba8(const char *path, int32_t *width, int32_t *height, uint8_t **rgba8) {
  return guarded<int>(RT_ERR_NOMEM, [&]() -> int {
    std::vector<uint8_t> bytes;
    std::string err;
    int w, h;
    if (!decodePNG8(path, w, h, bytes, err)) return fail(err);
    *width = w; *height = h;
    *rgba8 = (uint8_t *)std::malloc(bytes.size());
    if (!*rgba8) return RT_ERR_NOMEM;
    std::memcpy(*rgba8, bytes.data(), bytes.size());
    return RT_OK;
  });
}

int rth_png_write_rgba8(const char *path, int32_t width, int32_t height, const uint8_t *rgba8) {
  return guarded<int>(RT_ERR_NOMEM, [&]() -> int {
    std::string err;
    if (!writePNG8(path, width, height, rgba8, err)) return fail(err);
    return RT_OK;
  });
}

void rth_to_srgb8(const float *rgba, int64_t numPixels, uint8_t *rgba8) { toSRGB8(rgba, (size_t)numPixels, rgba8); }
void rth_free(void *p) { std::free(p); }

}  // extern "C"
