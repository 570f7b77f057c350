#include "rth_scene.h"
#include "rth_threads.h"

#include <algorithm>
#include <cstring>

namespace rth {

int Scene::addMaterialDefault() {
  // Material::Material() (src/scene/Material.cpp:4-11)
  RtMaterial m{};
  m.type = RT_MAT_DEFAULT;
  m.Kd = 1.0f; m.Ks = 0.0f; m.eta = 1.0f; m.Kr = 1.0f; m.Kt = 0.0f; m.Ka = 0.0f; m.roughness = 0.0f;
  materials.push_back(m);
  materialHasLobes.push_back(true);
  return (int)materials.size() - 1;
}

int Scene::addMaterial(float Kd, float Ks, float eta, float Kr, float Kt, float Ka, float roughness, int type) {
  // Material::Material(8 args) (src/scene/Material.cpp:13-36): type Dialectric adds no lobe at all.
  RtMaterial m{};
  m.type = type;
  m.Kd = Kd; m.Ks = Ks; m.eta = eta; m.Kr = Kr; m.Kt = Kt; m.Ka = Ka; m.roughness = roughness;
  materials.push_back(m);
  materialHasLobes.push_back(type != RT_MAT_DEFAULT);
  return (int)materials.size() - 1;
}

int Scene::loadTexture(const std::string &path) {
  for (size_t i = 0; i < texturePaths.size(); ++i)
    if (texturePaths[i] == path) return (int)i;
  auto img = std::make_shared<Image>();
  std::string err;
  if (!readPNG(path, *img, err)) {
    // The reference prints the decoder error and carries on with an empty image (src/image/PNG.cpp:129-132).
    log.push_back("PNG decoder error: " + err);
    img->width = img->height = 0;
  }
  textures.push_back(img);
  texturePaths.push_back(path);
  return (int)textures.size() - 1;
}

Prim makeTriangle(const Vec3 &p1, const Vec3 &p2, const Vec3 &p3, const Color &color, int material,
                  const Vec3 &t1, const Vec3 &t2, const Vec3 &t3, int texture) {
  Prim t;
  t.kind = RT_PRIM_TRIANGLE;
  t.color = color; t.material = material; t.texture = texture;
  t.p1 = p1; t.t1 = t1; t.t2 = t2; t.t3 = t3;
  t.aabbMin.x = std::min(std::min(p1.x, p2.x), p3.x);
  t.aabbMin.y = std::min(std::min(p1.y, p2.y), p3.y);
  t.aabbMin.z = std::min(std::min(p1.z, p2.z), p3.z);
  t.aabbMax.x = std::max(std::max(p1.x, p2.x), p3.x);
  t.aabbMax.y = std::max(std::max(p1.y, p2.y), p3.y);
  t.aabbMax.z = std::max(std::max(p1.z, p2.z), p3.z);
  // ONE_THIRD expands to `1.0 / 3.0` without parentheses (src/macros.h:36): (sum * 1.0) / 3.0 in double.
  t.centroid.x = (float)((double)(p1.x + p2.x + p3.x) * 1.0 / 3.0);
  t.centroid.y = (float)((double)(p1.y + p2.y + p3.y) * 1.0 / 3.0);
  t.centroid.z = (float)((double)(p1.z + p2.z + p3.z) * 1.0 / 3.0);

  Vec3 p3p1 = p3 - p1;
  Vec3 p2p1 = p2 - p1;
  t.normal = normalized(cross(p2p1, p3p1));
  Vec3 a1 = cross(p3p1, t.normal);
  Vec3 a2 = cross(p2p1, t.normal);
  // `1.0 / dot(..) * a1`: reciprocal in double, narrowed to float by operator*(float, Vector3D)
  t.e1 = (float)(1.0 / (double)dot(a1, p2p1)) * a1;
  t.e2 = (float)(1.0 / (double)dot(a2, p3p1)) * a2;
  return t;
}

Prim makeSphere(const Vec3 &center, float r, const Color &color, int material, int texture) {
  Prim s;
  s.kind = RT_PRIM_SPHERE;
  s.color = color; s.material = material; s.texture = texture;
  s.center = center; s.r = r;
  s.aabbMin = Vec3(center.x - r, center.y - r, center.z - r);
  s.aabbMax = Vec3(center.x + r, center.y + r, center.z + r);
  s.centroid = center;
  return s;
}

Plane makePlane(const Vec3 &normal, float D, const Color &color, int material, const Vec3 &texTopLeft,
                float texZoom, const Vec3 &texShift, int texture) {
  Plane p;
  p.color = color; p.material = material; p.texture = texture;
  p.normal = normalized(normal);
  p.texTopLeft = texTopLeft; p.texZoom = texZoom; p.texShift = texShift;
  // The constructor body tests the un-normalised parameter, and only one component of
  // `point` is ever assigned; the others are whatever the allocator left (zero here).
  if (normal.x != 0) p.point.x = -D / normal.x;
  else if (normal.y != 0) p.point.y = -D / normal.y;
  else p.point.z = -D / normal.z;
  return p;
}

static void put3(std::vector<float> &v, size_t at, const Vec3 &a) { v[at] = a.x; v[at + 1] = a.y; v[at + 2] = a.z; }

bool flatten(const Scene &scene, FlatScene &out, std::string &error, bool requireBvh) {
  if (requireBvh && !scene.bvhBuilt) { error = "flatten: buildBVH() has not run"; return false; }
  size_t n = scene.prims.size();
  out.primKind.resize(n); out.primMaterial.resize(n); out.primTexture.resize(n); out.primOrigIndex.resize(n);
  out.primHit.assign(n * 12, 0.0f); out.primNormals.assign(n * 9, 0.0f); out.primColor.resize(n * 4);
  out.primAabbMin.resize(n * 3); out.primAabbMax.resize(n * 3);
  bool anyTexturedTriangle = false;
  for (const Prim &p : scene.prims) anyTexturedTriangle |= (p.kind == RT_PRIM_TRIANGLE && p.texture >= 0);
  if (anyTexturedTriangle) out.primTexCoords.assign(n * 9, 0.0f); else out.primTexCoords.clear();
  parallelChunks((int64_t)n, 1 << 14, [&](int64_t begin, int64_t end, int) {
   for (size_t i = (size_t)begin; i < (size_t)end; ++i) {
    const Prim &p = scene.prims[i];
    out.primKind[i] = p.kind; out.primMaterial[i] = p.material; out.primTexture[i] = p.texture;
    out.primOrigIndex[i] = p.origIndex;
    put3(out.primAabbMin, i * 3, p.aabbMin); put3(out.primAabbMax, i * 3, p.aabbMax);
    if (p.kind == RT_PRIM_TRIANGLE) {
      put3(out.primHit, i * 12 + 0, p.p1); put3(out.primHit, i * 12 + 3, p.normal);
      put3(out.primHit, i * 12 + 6, p.e1); put3(out.primHit, i * 12 + 9, p.e2);
      put3(out.primNormals, i * 9 + 0, p.n1); put3(out.primNormals, i * 9 + 3, p.n2); put3(out.primNormals, i * 9 + 6, p.n3);
      if (anyTexturedTriangle) {
        put3(out.primTexCoords, i * 9 + 0, p.t1); put3(out.primTexCoords, i * 9 + 3, p.t2); put3(out.primTexCoords, i * 9 + 6, p.t3);
      }
    } else {
      put3(out.primHit, i * 12 + 0, p.center); out.primHit[i * 12 + 3] = p.r;
    }
    out.primColor[i * 4 + 0] = p.color.r; out.primColor[i * 4 + 1] = p.color.g;
    out.primColor[i * 4 + 2] = p.color.b; out.primColor[i * 4 + 3] = p.color.a;
   }
  });
  out.planes.clear();
  for (const Plane &p : scene.planes) {
    RtPlane q{};
    std::memcpy(q.normal, &p.normal, 12); std::memcpy(q.point, &p.point, 12);
    std::memcpy(q.texTopLeft, &p.texTopLeft, 12); std::memcpy(q.texShift, &p.texShift, 12);
    q.texZoom = p.texZoom;
    q.color[0] = p.color.r; q.color[1] = p.color.g; q.color[2] = p.color.b; q.color[3] = p.color.a;
    q.material = p.material; q.texture = p.texture;
    out.planes.push_back(q);
  }
  out.lights.clear();
  for (const Light &l : scene.lights) {
    RtLight q{};
    q.type = l.type;
    q.color[0] = l.color.r; q.color[1] = l.color.g; q.color[2] = l.color.b; q.color[3] = l.color.a;
    std::memcpy(q.v, &l.v, 12);
    q.radius = l.radius; q.texture = l.texture;
    out.lights.push_back(q);
  }
  out.textures.clear();
  for (const auto &img : scene.textures) {
    RtTexture t{};
    t.width = img->width; t.height = img->height; t.rgba = img->rgba.data();
    out.textures.push_back(t);
  }
  out.materials = scene.materials;
  for (size_t i = 0; i < scene.materials.size(); ++i) {
    if (!scene.materialHasLobes[i]) {
      // A material without lobes makes the reference index bdfs[-1] (src/bsdf/BDF.cpp:134-141); refuse it
      // only if something actually uses it.
      bool used = false;
      for (const Prim &p : scene.prims) used |= (p.material == (int)i);
      for (const Plane &p : scene.planes) used |= (p.material == (int)i);
      if (used) { error = "material " + std::to_string(i) + " has no BSDF lobes (unsupported input)"; return false; }
    }
  }
  RtSceneDesc &d = out.desc;
  d.nodes = requireBvh ? reinterpret_cast<const RtBvhNode *>(scene.nodes.data()) : nullptr;
  d.numNodes = requireBvh ? (int32_t)scene.nodes.size() : 0;
  d.numPrims = (int32_t)n;
  d.primKind = out.primKind.data(); d.primHit = out.primHit.data(); d.primNormals = out.primNormals.data();
  d.primTexCoords = out.primTexCoords.empty() ? nullptr : out.primTexCoords.data();
  d.primColor = out.primColor.data(); d.primMaterial = out.primMaterial.data();
  d.primTexture = out.primTexture.data(); d.primOrigIndex = out.primOrigIndex.data();
  d.primAabbMin = out.primAabbMin.data(); d.primAabbMax = out.primAabbMax.data();
  d.numPlanes = (int32_t)out.planes.size(); d.planes = out.planes.data();
  d.numMaterials = (int32_t)out.materials.size(); d.materials = out.materials.data();
  d.numLights = (int32_t)out.lights.size(); d.lights = out.lights.data();
  d.numTextures = (int32_t)out.textures.size(); d.textures = out.textures.data();
  std::memcpy(d.camera.eye, &scene.camera.eye, 12); std::memcpy(d.camera.forward, &scene.camera.forward, 12);
  std::memcpy(d.camera.right, &scene.camera.right, 12); std::memcpy(d.camera.up, &scene.camera.up, 12);
  d.options.width = scene.width; d.options.height = scene.height;
  d.options.bias = scene.options.bias; d.options.exposure = scene.options.exposure;
  d.options.maxBounces = scene.options.maxBounces; d.options.numRays = scene.options.numRays;
  d.options.fisheye = scene.options.fisheye ? 1 : 0;
  d.options.focus = scene.options.focus; d.options.lens = scene.options.lens;
  return true;
}

}  // namespace rth

// ---- host worker threads (rth_threads.h)
namespace rth {
namespace {
std::atomic<int> g_hostThreads{0};
}
int hostThreads() {
  int n = g_hostThreads.load();
  if (n > 0) return n;
  unsigned hw = std::thread::hardware_concurrency();
  return (int)std::max(1u, std::min(16u, hw ? hw : 1u));
}
void setHostThreads(int n) { g_hostThreads.store(n > 0 ? std::min(n, 256) : 0); }
}  // namespace rth
