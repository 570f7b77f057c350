// TEST INFRASTRUCTURE ONLY. Compiled in this container only, against the
// unmodified reference headers/objects under /root/reference/src, into
// oracle/_ref/ref_harness. It never travels as source of truth for the product;
// it dumps what the *reference itself* computes so that fixtures and the CPU
// restatement (oracle/rt_oracle.cpp) can be pinned to it.
//
// Modes (all outputs are little-endian raw dumps, layout documented per mode):
//   hits   <scene> <out>            primary-ray closest hits, un-jittered pixel centres
//   bvh    <scene> <out>            flattened BVH nodes + object permutation
//   render <scene> <out> <threads>  Scene::render float RGBA framebuffer
//   scene  <scene> <out>            parsed scene (objects/planes/lights/camera/options)
//   tex    <png>   <out>            PNG::readFromFile linear float RGBA
//   trace  <scene> <rays.bin> <out> closest hits for arbitrary (o,d) rays
#include "macros.h"

#define private public
#define protected public
#include "scene/raytracer.h"
#include "scene/Object.h"
#include "scene/Material.h"
#include "acceleration/BVH.h"
#include "parser/parser.h"
#include "image/PNG.h"
#include "acceleration/Profiler.h"
#undef private
#undef protected

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <unordered_map>

namespace {

struct Out {
  FILE *f;
  explicit Out(const char *path) : f(std::fopen(path, "wb")) {
    if (!f) { std::perror(path); std::exit(2); }
  }
  ~Out() { std::fclose(f); }
  void i32(int32_t v) { std::fwrite(&v, 4, 1, f); }
  void u32(uint32_t v) { std::fwrite(&v, 4, 1, f); }
  void f32(float v) { std::fwrite(&v, 4, 1, f); }
  void v3(const Vector3D &v) { f32(v.x); f32(v.y); f32(v.z); }
  void rgba(const RGBAColor &c) { f32(c.r); f32(c.g); f32(c.b); f32(c.a); }
  void raw(const void *p, size_t n) { std::fwrite(p, 1, n, f); }
};

uint32_t bits(float x) { uint32_t u; std::memcpy(&u, &x, 4); return u; }

std::unordered_map<const Object *, int> indexObjects(Scene *scene) {
  std::unordered_map<const Object *, int> idx;
  for (size_t i = 0; i < scene->objects.size(); ++i) idx[scene->objects[i].get()] = (int)i;
  for (size_t k = 0; k < scene->planes.size(); ++k)
    idx[scene->planes[k].get()] = (int)(scene->objects.size() + k);
  return idx;
}

// Same statement Scene::render(int,int) runs first (src/scene/raytracer.cpp:288).
void buildBVH(Scene *scene) { scene->bvh = std::make_unique<BVH>(scene->objects, 1); }

int materialKind(const Material &m) { return static_cast<int>(m.type); }

void dumpMaterial(Out &o, const Material &m) {
  o.f32(m.Kd); o.f32(m.Ks); o.f32(m.eta); o.f32(m.Kr); o.f32(m.Kt); o.f32(m.Ka);
  o.f32(m.roughness); o.i32(materialKind(m)); o.i32((int)m.bsdf.bdfs.size());
}

void dumpTexture(Out &o, const std::shared_ptr<PNG> &t) {
  if (!t) { o.i32(0); o.i32(0); return; }
  o.i32(t->width()); o.i32(t->height());
  // FNV-1a over the float bits: enough to tell whether the same texels were loaded.
  uint32_t h = 2166136261u;
  const unsigned char *p = reinterpret_cast<const unsigned char *>(t->image_);
  size_t n = (size_t)t->width() * t->height() * 16;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 16777619u; }
  o.u32(h);
}

int modeScene(const char *scenePath, const char *outPath) {
  std::unique_ptr<Scene> scene = readFromFile(scenePath);
  if (!scene) return 1;
  Out o(outPath);
  o.i32(scene->width()); o.i32(scene->height());
  o.f32(scene->options.bias); o.f32(scene->options.exposure);
  o.i32(scene->options.maxBounces); o.i32(scene->options.numRays);
  o.i32(scene->options.fisheye ? 1 : 0); o.f32(scene->options.focus); o.f32(scene->options.lens);
  o.v3(scene->camera.eye); o.v3(scene->camera.forward); o.v3(scene->camera.right); o.v3(scene->camera.up);
  o.i32((int)scene->objects.size());
  for (auto &up : scene->objects) {
    const Object *obj = up.get();
    if (auto *t = dynamic_cast<const Triangle *>(obj)) {
      o.i32(0);
      o.v3(t->p1); o.v3(t->normal); o.v3(t->e1); o.v3(t->e2);
      o.v3(t->n1); o.v3(t->n2); o.v3(t->n3);
      o.v3(t->t1); o.v3(t->t2); o.v3(t->t3);
    } else if (auto *s = dynamic_cast<const Sphere *>(obj)) {
      o.i32(1);
      o.v3(s->center); o.f32(s->r);
    } else {
      o.i32(-1);
    }
    o.v3(obj->aabbMin); o.v3(obj->aabbMax); o.v3(obj->centroid);
    o.rgba(obj->color);
    dumpMaterial(o, *obj->material);
    dumpTexture(o, obj->textureMap);
  }
  o.i32((int)scene->planes.size());
  for (auto &up : scene->planes) {
    const Plane *p = up.get();
    o.v3(p->normal); o.v3(p->point); o.v3(p->textureTopLeft); o.f32(p->textureZoom); o.v3(p->textureShift);
    o.rgba(p->color);
    dumpMaterial(o, *p->material);
    dumpTexture(o, p->textureMap);
  }
  o.i32((int)scene->lights.size());
  for (Light *l : scene->lights) {
    if (auto *d = dynamic_cast<DistantLight *>(l)) {
      o.i32(0); o.rgba(d->color); o.v3(d->direction);
    } else if (auto *p = dynamic_cast<PointLight *>(l)) {
      o.i32(1); o.rgba(p->color); o.v3(p->center);
    } else if (auto *e = dynamic_cast<EnvironmentLight *>(l)) {
      o.i32(2); o.rgba(e->color); o.v3(e->center); o.f32(e->radius);
      dumpTexture(o, e->luminanceMap);
    }
  }
  return 0;
}

int modeBVH(const char *scenePath, const char *outPath) {
  std::unique_ptr<Scene> scene = readFromFile(scenePath);
  if (!scene) return 1;
  auto idx = indexObjects(scene.get());
  buildBVH(scene.get());
  // Count nodes by walking the pre-order array: node i's subtree ends where the walk ends.
  BVH *bvh = scene->bvh.get();
  int numNodes = 0;
  if (bvh->nodes) {
    // pre-order DFS layout: left child = i+1, right child = nodes[i].right (src/acceleration/BVH.cpp:411-428)
    std::vector<int> stack{0};
    while (!stack.empty()) {
      int i = stack.back(); stack.pop_back();
      numNodes = std::max(numNodes, i + 1);
      if (!bvh->nodes[i].isLeaf()) { stack.push_back(i + 1); stack.push_back(bvh->nodes[i].right); }
    }
  }
  Out o(outPath);
  o.i32(numNodes); o.i32((int)scene->objects.size());
  static_assert(sizeof(BVH::FlattenedNode) == 32, "node size");
  o.raw(bvh->nodes, (size_t)numNodes * 32);
  for (auto &up : scene->objects) o.i32(idx[up.get()]);
  return 0;
}

int modeHits(const char *scenePath, const char *outPath) {
  std::unique_ptr<Scene> scene = readFromFile(scenePath);
  if (!scene) return 1;
  auto idx = indexObjects(scene.get());
  buildBVH(scene.get());
  std::unordered_map<const Object *, int> post;
  for (size_t i = 0; i < scene->objects.size(); ++i) post[scene->objects[i].get()] = (int)i;
  int W = scene->width(), H = scene->height();
  Out o(outPath);
  o.i32(W); o.i32(H); o.i32((int)scene->objects.size()); o.i32((int)scene->planes.size());
  const Camera &cam = scene->camera;
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) {
      // un-jittered form of src/scene/raytracer.cpp:44-47 (allowAntiAliasing == 0)
      float Sx = getRayScaleX(x + (0.5f - 0.5f) * 0, W, H);
      float Sy = getRayScaleY(y + (0.5f - 0.5f) * 0, W, H);
      IntersectionInfo info = scene->findClosestObject(cam.eye, cam.forward + Sx * cam.right + Sy * cam.up);
      if (info.obj == nullptr) {
        o.i32(-1); o.i32(-1); o.u32(bits(info.t));
      } else {
        auto it = post.find(info.obj);
        o.i32(idx[info.obj]); o.i32(it == post.end() ? idx[info.obj] : it->second); o.u32(bits(info.t));
      }
    }
  }
  return 0;
}

int modeTrace(const char *scenePath, const char *raysPath, const char *outPath) {
  std::unique_ptr<Scene> scene = readFromFile(scenePath);
  if (!scene) return 1;
  auto idx = indexObjects(scene.get());
  buildBVH(scene.get());
  FILE *rf = std::fopen(raysPath, "rb");
  if (!rf) { std::perror(raysPath); return 2; }
  Out o(outPath);
  float r[6];
  while (std::fread(r, 4, 6, rf) == 6) {
    IntersectionInfo info = scene->findClosestObject(Vector3D(r[0], r[1], r[2]), Vector3D(r[3], r[4], r[5]));
    if (info.obj == nullptr) { o.i32(-1); o.u32(bits(info.t)); o.v3(Vector3D(0, 0, 0)); o.v3(Vector3D(0, 0, 0)); }
    else { o.i32(idx[info.obj]); o.u32(bits(info.t)); o.v3(info.point); o.v3(info.normal); }
  }
  std::fclose(rf);
  return 0;
}

int modeRender(const char *scenePath, const char *outPath, int threads) {
  std::unique_ptr<Scene> scene = readFromFile(scenePath);
  if (!scene) return 1;
  PNG *img = scene->render(threads);
  Out o(outPath);
  o.i32(img->width()); o.i32(img->height());
  o.raw(img->image_, (size_t)img->width() * img->height() * sizeof(RGBAColor));
  // 8-bit RGBA exactly as PNG::saveToFile quantises it (src/image/PNG.cpp:149-157)
  for (unsigned i = 0; i < (unsigned)(img->width() * img->height()); ++i) {
    RGBAColor g = img->image_[i].toSRGB();
    unsigned char px[4] = {
      static_cast<unsigned char>(round(g.r * 255)), static_cast<unsigned char>(round(g.g * 255)),
      static_cast<unsigned char>(round(g.b * 255)), static_cast<unsigned char>(round(g.a * 255))};
    o.raw(px, 4);
  }
  printStats();
  delete img;
  return 0;
}

int modeTex(const char *pngPath, const char *outPath) {
  PNG img(pngPath);
  Out o(outPath);
  o.i32(img.width()); o.i32(img.height());
  o.raw(img.image_, (size_t)img.width() * img.height() * sizeof(RGBAColor));
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s hits|bvh|render|scene|tex|trace <in> <out> [arg]\n", argv[0]);
    return 64;
  }
  std::string mode = argv[1];
  if (mode == "hits") return modeHits(argv[2], argv[3]);
  if (mode == "bvh") return modeBVH(argv[2], argv[3]);
  if (mode == "scene") return modeScene(argv[2], argv[3]);
  if (mode == "tex") return modeTex(argv[2], argv[3]);
  if (mode == "render") return modeRender(argv[2], argv[3], argc > 4 ? std::atoi(argv[4]) : 1);
  if (mode == "trace" && argc > 4) return modeTrace(argv[2], argv[3], argv[4]);
  std::fprintf(stderr, "unknown mode %s\n", mode.c_str());
  return 64;
}
