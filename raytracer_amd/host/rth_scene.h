// Host scene graph: what the reference keeps in Scene / Object / Light / Material
// (src/scene/raytracer.h:44-119, src/scene/Object.h, src/scene/Material.h), reduced to
// plain records that can be flattened into an RtSceneDesc.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "rth_math.h"

namespace rth {

struct Image {  // linear float RGBA, row-major (PNG::image_, src/image/PNG.h:70-73)
  int width = 0, height = 0;
  std::vector<float> rgba;
};

struct Prim {  // Triangle or Sphere: an Object that lives in the BVH
  int kind = RT_PRIM_TRIANGLE;
  // triangle (src/scene/Object.h:162-173)
  Vec3 p1, normal, e1, e2, t1, t2, t3, n1, n2, n3;
  // sphere (src/scene/Object.h:119-120)
  Vec3 center;
  float r = 0.0f;
  // Object (src/scene/Object.h:45-50)
  Color color;
  int material = 0;
  int texture = -1;
  Vec3 aabbMin, aabbMax, centroid;
  int origIndex = 0;
};

struct Plane {
  Vec3 normal, point, texTopLeft, texShift;
  float texZoom = 1.0f;
  Color color;
  int material = 0;
  int texture = -1;
};

struct Light {
  int type = RT_LIGHT_POINT;
  Color color;
  Vec3 v;  // direction or center
  float radius = 0.0f;
  int texture = -1;
};

struct Camera {  // src/scene/Camera.h:7-21, src/scene/Camera.cpp:7-16
  Vec3 eye{0, 0, 0}, forward{0, 0, -1}, right{1, 0, 0}, up{0, 1, 0};
  void setForward(const Vec3 &f) {
    forward = f;
    right = normalized(cross(forward, up));
    up = normalized(cross(right, forward));
  }
  void setUp(const Vec3 &u) {
    right = normalized(cross(forward, u));
    up = normalized(cross(right, forward));
  }
};

struct Options {  // src/scene/raytracer.h:34-42
  float bias = 1e-4f;
  float exposure = -1.0f;
  int maxBounces = 4;
  int numRays = 1;
  bool fisheye = false;
  float focus = -1.0f;
  float lens = 0.0f;
};

struct BvhNode {  // == RtBvhNode == reference FlattenedNode
  float aabbMin[3], aabbMax[3];
  int32_t rightOrStart, numObjects;
};

struct Scene {
  int width = 0, height = 0;
  std::string filename;
  Camera camera;
  Options options;
  std::vector<Prim> prims;  // insertion order until buildBVH() reorders it
  std::vector<Plane> planes;
  std::vector<Light> lights;
  std::vector<RtMaterial> materials;
  std::vector<bool> materialHasLobes;  // false for the 8-argument constructor with type Dialectric
  std::vector<std::shared_ptr<Image>> textures;
  std::vector<std::string> texturePaths;
  Vec3 centroidSum;  // Scene::addObject, src/scene/raytracer.cpp:314-317
  std::vector<BvhNode> nodes;
  bool bvhBuilt = false;
  std::vector<std::string> log;  // lines the reference prints to stdout while loading/building

  void addPrim(Prim p) {
    centroidSum = centroidSum + p.centroid;  // Vector3D::operator+= is component-wise +=
    p.origIndex = (int)prims.size();
    prims.push_back(p);
  }
  // Scene::worldCenter, src/scene/raytracer.h:88-90
  Vec3 worldCenter() const { return centroidSum / (float)prims.size(); }
  int addMaterialDefault();
  int addMaterial(float Kd, float Ks, float eta, float Kr, float Kt, float Ka, float roughness, int type);
  int loadTexture(const std::string &path);  // cached by path; -1 and a message on failure
};

// Triangle::Triangle (src/scene/Object.cpp:160-193); normals n1..n3 are left zero like a zero-filled heap.
Prim makeTriangle(const Vec3 &p1, const Vec3 &p2, const Vec3 &p3, const Color &color, int material,
                  const Vec3 &t1 = Vec3(), const Vec3 &t2 = Vec3(), const Vec3 &t3 = Vec3(), int texture = -1);
// Sphere::Sphere (src/scene/Object.cpp:57-73)
Prim makeSphere(const Vec3 &center, float r, const Color &color, int material, int texture = -1);
// Plane::Plane (src/scene/Object.cpp:111-131)
Plane makePlane(const Vec3 &normal, float D, const Color &color, int material, const Vec3 &texTopLeft = Vec3(),
                float texZoom = 1.0f, const Vec3 &texShift = Vec3(), int texture = -1);

// BVH::BVH (src/acceleration/BVH.cpp:161-182): binned SAH build that reorders scene.prims in place and
// fills scene.nodes with the pre-order flattened array.
void buildBVH(Scene &scene);

// loaders (src/parser/parser.cpp, src/parser/ParserTree.cpp); return nullptr and set `error` on failure
std::unique_ptr<Scene> readFromFile(const std::string &filename, std::string &error);
bool loadOBJ(const Vec3 &center, float scale, const std::string &filename, Scene &scene, const Color &color,
             int material, std::string &error);

// PNG codec (pixels, not file bytes, match the reference's LodePNG-based src/image/PNG.cpp)
bool readPNG(const std::string &filename, Image &out, std::string &error);        // 8-bit RGBA/255 -> toLinear
bool decodePNG8(const std::string &filename, int &w, int &h, std::vector<uint8_t> &rgba8, std::string &error);
bool writePNG8(const std::string &filename, int w, int h, const uint8_t *rgba8, std::string &error);
void toSRGB8(const float *rgba, size_t numPixels, uint8_t *rgba8);                // PNG::saveToFile conversion

// Scene -> RtSceneDesc; the arrays live in the returned object.
struct FlatScene {
  RtSceneDesc desc{};
  std::vector<int32_t> primKind, primMaterial, primTexture, primOrigIndex;
  std::vector<float> primAabbMin, primAabbMax;  // 3 per object: Object::aabbMin / aabbMax
  std::vector<float> primHit, primNormals, primTexCoords, primColor;
  std::vector<RtPlane> planes;
  std::vector<RtLight> lights;
  std::vector<RtTexture> textures;
  std::vector<RtMaterial> materials;
};
// requireBvh = false: objects in whatever order the scene holds them, no nodes (the device library builds its own tree)
bool flatten(const Scene &scene, FlatScene &out, std::string &error, bool requireBvh = true);

}  // namespace rth
