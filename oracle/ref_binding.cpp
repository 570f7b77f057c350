// TEST INFRASTRUCTURE ONLY. Compiled in the build container only, against the unmodified reference headers and
// objects under /root/reference/src, into oracle/_ref/ref_binding.
//
// This is the reference-side binding of include/rt_amd.h written out and compiled: the code a maintainer of
// RichardXiao13/Raytracer adds so that Scene::render (src/scene/raytracer.cpp:287-300) and Scene::findClosestObject
// (:187-196) run behind the C ABI. It uses only members the reference really has:
//
//   Scene::objects, planes, lights, bvh, width_, height_        private (src/scene/raytracer.h:127-136): the binding is a
//                                                               member function of Scene in the reference tree
//   Scene::camera, options                                      public  (raytracer.h:107-108)
//   BVH::nodes                                                  private raw pointer (src/acceleration/BVH.h:78); the node
//                                                               count is not stored anywhere: it follows from walking the
//                                                               pre-order array (left = i + 1, right = nodes[i].right)
//   Object::color, material, textureMap                         public  (src/scene/Object.h:45-47)
//   Triangle::p1, normal, e1, e2, t1..t3, n1..n3                public  (Object.h:162-174)
//   Sphere::center, r / Plane::normal, point, texture*          public  (Object.h:119-120, 137-141)
//   DistantLight::direction, PointLight::center,
//   EnvironmentLight::center, radius, luminanceMap, Light::color public (Object.h:61-105)
//   Material::Kd .. roughness, type                             public  (src/scene/Material.h:46-54)
//   PNG::getPixel(row, col), width(), height()                  public  (src/image/PNG.h:47-55)
//
// In the reference tree the two private reaches cost one line each (`friend class Scene;` in BVH, and the function being
// a Scene member). This file cannot edit the reference, so it opens them with `#define private public` around the
// includes — the only liberty it takes. Everything else is what the maintainer would write.
//
// The program then proves the binding: it loads an implementation of the ABI with dlopen (the CPU restatement
// librt_oracle.so with prefix rto_ in the build container, libraytracer_amd.so with prefix rt_ on a GPU box), and
// requires (1) primary-ray hits through the ABI == Scene::findClosestObject of the same reference build, id and t bits,
// for every pixel, and (2) with an implementation that can replay the reference's random stream (RT_RNG_REF_MT19937,
// the restatement only) the frame through the ABI == Scene::render(1), bit for bit.
//
//   ref_binding <scene file> <library.so> <prefix> [render]
#include "macros.h"

#define private public
#define protected public
#include "scene/raytracer.h"
#include "scene/Object.h"
#include "scene/Material.h"
#include "acceleration/BVH.h"
#include "parser/parser.h"
#include "image/PNG.h"
#undef private
#undef protected

#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "rt_amd.h"

namespace {

// ---------------------------------------------------------------------------------------------- the binding proper
void put3(float *dst, const Vector3D &v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
void put4(float *dst, const RGBAColor &c) { dst[0] = c.r; dst[1] = c.g; dst[2] = c.b; dst[3] = c.a; }

// Host arrays behind an RtSceneDesc; they only have to live until rt_scene_create returns (it copies everything).
struct FlatScene {
  std::vector<RtBvhNode> nodes;
  std::vector<int32_t> kind, mat, tex, orig;
  std::vector<float> hit, nrm, uv, col;
  std::vector<RtPlane> planes;
  std::vector<RtMaterial> materials;
  std::vector<RtLight> lights;
  std::vector<std::vector<float>> texels;
  std::vector<RtTexture> textures;
  std::map<const Material *, int32_t> materialIndex;
  std::map<PNG *, int32_t> textureIndex;
  RtSceneDesc desc{};

  int32_t materialOf(const std::shared_ptr<Material> &m) {
    auto it = materialIndex.find(m.get());
    if (it != materialIndex.end()) return it->second;
    RtMaterial r{};
    r.type = static_cast<int32_t>(m->type);  // MaterialType order == RT_MAT_* order (Material.h:8-14)
    r.Kd = m->Kd; r.Ks = m->Ks; r.eta = m->eta; r.Kr = m->Kr; r.Kt = m->Kt; r.Ka = m->Ka; r.roughness = m->roughness;
    materials.push_back(r);
    return materialIndex[m.get()] = (int32_t)materials.size() - 1;
  }
  int32_t textureOf(const std::shared_ptr<PNG> &t) {
    if (!t) return -1;
    auto it = textureIndex.find(t.get());
    if (it != textureIndex.end()) return it->second;
    const int w = t->width(), h = t->height();
    std::vector<float> px((size_t)w * h * 4);
    for (int row = 0; row < h; ++row)
      for (int col = 0; col < w; ++col) put4(&px[((size_t)row * w + col) * 4], t->getPixel(row, col));
    texels.push_back(std::move(px));
    return textureIndex[t.get()] = (int32_t)texels.size() - 1;
  }
};

// Scene (after `bvh = std::make_unique<BVH>(objects, 1)`, raytracer.cpp:288) -> RtSceneDesc.
// `insertion` maps an object to the index it had before the BVH build reordered Scene::objects: the id the ABI reports.
void flattenScene(Scene &scene, const std::unordered_map<const Object *, int> &insertion, FlatScene &f) {
  // 1. BVH: RtBvhNode is byte for byte BVH::FlattenedNode (BVH.h:49-61)
  static_assert(sizeof(BVH::FlattenedNode) == sizeof(RtBvhNode), "node layout");
  int numNodes = 0;
  if (scene.bvh && scene.bvh->nodes && !scene.objects.empty()) {
    std::vector<int> walk{0};
    while (!walk.empty()) {
      int i = walk.back();
      walk.pop_back();
      numNodes = std::max(numNodes, i + 1);
      if (!scene.bvh->nodes[i].isLeaf()) { walk.push_back(i + 1); walk.push_back(scene.bvh->nodes[i].right); }
    }
    f.nodes.resize(numNodes);
    std::memcpy(f.nodes.data(), scene.bvh->nodes, (size_t)numNodes * sizeof(RtBvhNode));
  }
  // 2. objects in post-build order -> structure of arrays
  const size_t n = scene.objects.size();
  f.kind.assign(n, 0); f.mat.assign(n, 0); f.tex.assign(n, -1); f.orig.assign(n, 0);
  f.hit.assign(12 * n, 0.0f); f.nrm.assign(9 * n, 0.0f); f.uv.assign(9 * n, 0.0f); f.col.assign(4 * n, 0.0f);
  for (size_t i = 0; i < n; ++i) {
    const Object *o = scene.objects[i].get();
    if (auto *t = dynamic_cast<const Triangle *>(o)) {
      f.kind[i] = RT_PRIM_TRIANGLE;
      put3(&f.hit[12 * i + 0], t->p1); put3(&f.hit[12 * i + 3], t->normal);
      put3(&f.hit[12 * i + 6], t->e1); put3(&f.hit[12 * i + 9], t->e2);
      put3(&f.nrm[9 * i], t->n1); put3(&f.nrm[9 * i + 3], t->n2); put3(&f.nrm[9 * i + 6], t->n3);
      put3(&f.uv[9 * i], t->t1); put3(&f.uv[9 * i + 3], t->t2); put3(&f.uv[9 * i + 6], t->t3);
    } else if (auto *s = dynamic_cast<const Sphere *>(o)) {
      f.kind[i] = RT_PRIM_SPHERE;
      put3(&f.hit[12 * i], s->center);
      f.hit[12 * i + 3] = s->r;
    }
    put4(&f.col[4 * i], o->color);
    f.mat[i] = f.materialOf(o->material);
    f.tex[i] = f.textureOf(o->textureMap);
    f.orig[i] = insertion.at(o);
  }
  // 3. planes, lights
  for (auto &up : scene.planes) {
    const Plane *p = up.get();
    RtPlane q{};
    put3(q.normal, p->normal); put3(q.point, p->point); put3(q.texTopLeft, p->textureTopLeft);
    q.texZoom = p->textureZoom; put3(q.texShift, p->textureShift); put4(q.color, p->color);
    q.material = f.materialOf(p->material);
    q.texture = f.textureOf(p->textureMap);
    f.planes.push_back(q);
  }
  for (Light *l : scene.lights) {
    RtLight q{};
    q.texture = -1;
    put4(q.color, l->color);
    if (auto *d = dynamic_cast<DistantLight *>(l)) { q.type = RT_LIGHT_DISTANT; put3(q.v, d->direction); }
    else if (auto *p = dynamic_cast<PointLight *>(l)) { q.type = RT_LIGHT_POINT; put3(q.v, p->center); }
    else if (auto *e = dynamic_cast<EnvironmentLight *>(l)) {
      q.type = RT_LIGHT_ENVIRONMENT; put3(q.v, e->center); q.radius = e->radius; q.texture = f.textureOf(e->luminanceMap);
    }
    f.lights.push_back(q);
  }
  for (size_t i = 0; i < f.texels.size(); ++i) {
    RtTexture t{};
    for (auto &kv : f.textureIndex)
      if (kv.second == (int32_t)i) { t.width = kv.first->width(); t.height = kv.first->height(); }
    t.rgba = f.texels[i].data();
    f.textures.push_back(t);
  }
  RtSceneDesc &d = f.desc;
  d.nodes = f.nodes.data(); d.numNodes = numNodes;
  d.numPrims = (int32_t)n; d.primKind = f.kind.data(); d.primHit = f.hit.data(); d.primNormals = f.nrm.data();
  d.primTexCoords = f.uv.data(); d.primColor = f.col.data(); d.primMaterial = f.mat.data(); d.primTexture = f.tex.data();
  d.primOrigIndex = f.orig.data();
  d.numPlanes = (int32_t)f.planes.size(); d.planes = f.planes.data();
  d.numMaterials = (int32_t)f.materials.size(); d.materials = f.materials.data();
  d.numLights = (int32_t)f.lights.size(); d.lights = f.lights.data();
  d.numTextures = (int32_t)f.textures.size(); d.textures = f.textures.data();
  put3(d.camera.eye, scene.camera.eye); put3(d.camera.forward, scene.camera.forward);
  put3(d.camera.right, scene.camera.right); put3(d.camera.up, scene.camera.up);
  d.options.width = scene.width_; d.options.height = scene.height_;
  d.options.bias = scene.options.bias; d.options.exposure = scene.options.exposure;
  d.options.maxBounces = scene.options.maxBounces; d.options.numRays = scene.options.numRays;
  d.options.fisheye = scene.options.fisheye ? 1 : 0; d.options.focus = scene.options.focus; d.options.lens = scene.options.lens;
}

// The entry points of include/rt_amd.h this binding calls, resolved from whichever implementation is loaded.
struct Abi {
  void *lib = nullptr;
  const char *(*last_error)() = nullptr;
  RtScene *(*scene_create)(const RtSceneDesc *, int) = nullptr;
  void (*scene_destroy)(RtScene *) = nullptr;
  int (*trace_primary)(RtScene *, int32_t, int32_t, int32_t, int32_t, int32_t *, float *) = nullptr;
  int (*render)(RtScene *, const RtRenderParams *, float *, RtStats *) = nullptr;
  void (*set_threads)(RtScene *, int) = nullptr;  // restatement only
  bool load(const char *path, const std::string &prefix) {
    lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { std::fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return false; }
    auto sym = [&](const char *name) { return dlsym(lib, (prefix + name).c_str()); };
    last_error = reinterpret_cast<decltype(last_error)>(sym("last_error"));
    scene_create = reinterpret_cast<decltype(scene_create)>(sym("scene_create"));
    scene_destroy = reinterpret_cast<decltype(scene_destroy)>(sym("scene_destroy"));
    trace_primary = reinterpret_cast<decltype(trace_primary)>(sym("trace_primary"));
    render = reinterpret_cast<decltype(render)>(sym("render"));
    set_threads = reinterpret_cast<decltype(set_threads)>(sym("set_threads"));
    return last_error && scene_create && scene_destroy && trace_primary && render;
  }
};

// Scene::render behind the ABI (replacement body of raytracer.cpp:287-300). Returns the PNG the reference returns.
PNG *renderThroughAbi(Scene &scene, const Abi &abi, RtScene *dev, int rngMode, int seed) {
  PNG *img = new PNG(scene.width_, scene.height_);
  RtRenderParams p{};
  p.rngMode = rngMode; p.seed = (uint32_t)seed; p.partCount = 1; p.tileW = p.tileH = 32;
  std::vector<float> frame((size_t)scene.width_ * scene.height_ * 4);
  if (abi.render(dev, &p, frame.data(), nullptr) != RT_OK) {
    std::fprintf(stderr, "render: %s\n", abi.last_error());
    delete img;
    return nullptr;
  }
  for (int row = 0; row < scene.height_; ++row)
    for (int col = 0; col < scene.width_; ++col) {
      const float *px = &frame[((size_t)row * scene.width_ + col) * 4];
      img->getPixel(row, col) = RGBAColor(px[0], px[1], px[2], px[3]);
    }
  return img;
}

uint32_t bits(float x) { uint32_t u; std::memcpy(&u, &x, 4); return u; }

}  // namespace

int main(int argc, char **argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <scene> <library.so> <prefix> [render]\n", argv[0]);
    return 64;
  }
  const bool wantRender = argc > 4 && std::string(argv[4]) == "render";
  Abi abi;
  if (!abi.load(argv[2], argv[3])) { std::fprintf(stderr, "the library does not export the ABI with prefix %s\n", argv[3]); return 2; }

  std::unique_ptr<Scene> scene = readFromFile(argv[1]);
  if (!scene) return 1;
  std::unordered_map<const Object *, int> insertion;
  for (size_t i = 0; i < scene->objects.size(); ++i) insertion[scene->objects[i].get()] = (int)i;
  scene->bvh = std::make_unique<BVH>(scene->objects, 1);  // raytracer.cpp:288

  FlatScene flat;
  flattenScene(*scene, insertion, flat);
  RtScene *dev = abi.scene_create(&flat.desc, 0);
  if (!dev) { std::fprintf(stderr, "scene_create: %s\n", abi.last_error()); return 3; }
  if (abi.set_threads) abi.set_threads(dev, 1);

  // (1) Scene::findClosestObject vs rt_trace_primary, every pixel
  const int W = scene->width(), H = scene->height();
  std::vector<int32_t> ids((size_t)W * H);
  std::vector<float> ts((size_t)W * H);
  if (abi.trace_primary(dev, 0, 0, W, H, ids.data(), ts.data()) != RT_OK) { std::fprintf(stderr, "trace_primary: %s\n", abi.last_error()); return 3; }
  long long hitMismatch = 0;
  const Camera &cam = scene->camera;
  const int numObjects = (int)scene->objects.size();
  std::unordered_map<const Object *, int> planeIndex;
  for (size_t k = 0; k < scene->planes.size(); ++k) planeIndex[scene->planes[k].get()] = numObjects + (int)k;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float Sx = getRayScaleX(x, W, H), Sy = getRayScaleY(y, W, H);  // raytracer.cpp:44-47 without jitter
      IntersectionInfo info = scene->findClosestObject(cam.eye, cam.forward + Sx * cam.right + Sy * cam.up);
      int want = -1;
      if (info.obj) {
        auto it = insertion.find(info.obj);
        want = it != insertion.end() ? it->second : planeIndex.at(info.obj);
      }
      const size_t j = (size_t)y * W + x;
      if (ids[j] != want || (want >= 0 && bits(ts[j]) != bits(info.t))) ++hitMismatch;
    }
  std::printf("primary hits: %d x %d pixels, %lld mismatches\n", W, H, hitMismatch);

  // (2) Scene::render(1) vs rt_render with the reference's random stream (restatement only)
  long long pixelMismatch = -1;
  if (wantRender) {
    PNG *viaAbi = renderThroughAbi(*scene, abi, dev, RT_RNG_REF_MT19937, 56);
    if (!viaAbi) return 3;
    abi.scene_destroy(dev);
    dev = nullptr;
    std::unique_ptr<Scene> again = readFromFile(argv[1]);  // render() builds its own BVH and reorders the objects again
    PNG *ref = again->render(1);
    pixelMismatch = 0;
    for (int row = 0; row < H; ++row)
      for (int col = 0; col < W; ++col) {
        RGBAColor a = viaAbi->getPixel(row, col), b = ref->getPixel(row, col);
        if (bits(a.r) != bits(b.r) || bits(a.g) != bits(b.g) || bits(a.b) != bits(b.b) || bits(a.a) != bits(b.a)) ++pixelMismatch;
      }
    std::printf("render: %d x %d pixels, %lld differ from Scene::render(1)\n", W, H, pixelMismatch);
    delete viaAbi;
    delete ref;
  }
  if (dev) abi.scene_destroy(dev);
  const bool ok = hitMismatch == 0 && (!wantRender || pixelMismatch == 0);
  std::printf("%s\n", ok ? "BINDING OK" : "BINDING MISMATCH");
  return ok ? 0 : 5;
}
