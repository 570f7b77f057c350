// Scene front-ends: the keyword-per-line .txt format (src/parser/parser.cpp:144-360), the tag-tree
// .sdml format (src/parser/ParserTree.cpp) and the Wavefront OBJ loader (src/parser/parser.cpp:47-142).
// They run once per render on the host; what matters is that they produce the same floats, the same
// object order and the same quirks (normal flips toward the camera, lights parsed last, OBJ recentring).
#include <cerrno>
#include <climits>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "rth_scene.h"
#include "rth_threads.h"

namespace rth {
namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();

// parser.cpp:16-31 — empty items are dropped
std::vector<std::string> split(const std::string &s, char delim) {
  std::vector<std::string> elems;
  std::istringstream iss(s);
  std::string item;
  while (std::getline(iss, item, delim))
    if (!item.empty()) elems.push_back(item);
  return elems;
}

bool endsWith(const std::string &value, const std::string &ending) {
  return ending.size() <= value.size() && std::equal(ending.rbegin(), ending.rend(), value.rbegin());
}

std::string trimCopy(std::string s) {
  auto notSpace = [](unsigned char ch) { return !std::isspace(ch); };
  s.erase(std::find_if(s.rbegin(), s.rend(), notSpace).base(), s.end());
  s.erase(s.begin(), std::find_if(s.begin(), s.end(), notSpace));
  return s;
}

// The named presets of src/scene/Material.h:57-64, created on first use per scene.
struct MaterialBook {
  Scene &scene;
  std::unordered_map<std::string, int> named;
  explicit MaterialBook(Scene &s) : scene(s) {}
  int get(const std::string &name) {
    auto it = named.find(name);
    if (it != named.end()) return it->second;
    int id;
    if (name == "glass") id = scene.addMaterial(0.0f, 1.0f, 1.5f, 1.0f, 1.0f, 0.0f, 0.0f, RT_MAT_GLASS);
    else if (name == "plastic") id = scene.addMaterial(0.5f, 0.5f, 1.3f, 1.0f, 0.0f, 0.0f, 0.1f, RT_MAT_PLASTIC);
    else if (name == "copper") id = scene.addMaterial(0.0f, 1.0f, 0.23883f, 0.9553f, 0.0f, 3.415658f, 0.01f, RT_MAT_METAL);
    else if (name == "gold") id = scene.addMaterial(0.0f, 1.0f, 0.18104f, 0.99f, 0.0f, 3.068099f, 0.01f, RT_MAT_METAL);
    else if (name == "mirror") id = scene.addMaterial(0.0f, 1.0f, 0.0f, 0.9f, 0.0f, 0.0f, 0.0f, RT_MAT_MIRROR);
    else id = scene.addMaterialDefault();
    named[name] = id;
    return id;
  }
};

// src/scene/Material.h:67-74
const std::unordered_map<std::string, Color> &materialColors() {
  static const std::unordered_map<std::string, Color> m = {
      {"copper", Color(0.4793201831f, 0.1714411007f, 0.03310476657f)},
      {"pale copper", Color(0.7011018919f, 0.2541520943f, 0.1356333297f)},
      {"copper red", Color(0.5972017884f, 0.152926152f, 0.08228270713f)},
      {"copper penny", Color(0.4178850708f, 0.1589608351f, 0.1412632911f)},
      {"gold", Color(1.0f, 0.6795424696330938f, 0.0f)},
      {"metallic gold", Color(0.6583748172794485f, 0.4286904966139066f, 0.0382043715953465f)}};
  return m;
}

int materialTypeByName(const std::string &s) {  // src/scene/Material.h:16-22
  if (s == "dialectric") return RT_MAT_DEFAULT;
  if (s == "glass") return RT_MAT_GLASS;
  if (s == "mirror") return RT_MAT_MIRROR;
  if (s == "plastic") return RT_MAT_PLASTIC;
  if (s == "metal") return RT_MAT_METAL;
  throw std::out_of_range("unknown material type '" + s + "'");
}

// Orient a triangle's normal against the camera when it is in front of it and faces along forward
// (parser.cpp:250-257, ParserTree.cpp:438-445), then use it as all three shading normals.
void orientTriangle(Prim &t, const Camera &camera) {
  if (dot(camera.forward, t.centroid - camera.eye) > 0 && dot(camera.forward, t.normal) > 0) t.normal = -t.normal;
  t.n1 = t.normal; t.n2 = t.normal; t.n3 = t.normal;
}

// ------------------------------------------------------------------ .txt
std::unique_ptr<Scene> readTxt(std::istream &in, std::string &error) {
  std::string line;
  std::getline(in, line);
  std::vector<std::string> info = split(line, ' ');
  if (info.size() != 4) {
    error = "Supplied PNG info doesn't have the correct number of arguments. Expected 4. Got " +
            std::to_string(info.size()) + ".";
    return nullptr;
  }
  if (info[0] != "png") { error = "Expected PNG image type. Got " + info[0] + "."; return nullptr; }
  auto scene = std::make_unique<Scene>();
  scene->width = std::stoi(info.at(1));
  scene->height = std::stoi(info.at(2));
  scene->filename = info.at(3);
  std::vector<Vec3> points;
  int currentMaterial = scene->addMaterialDefault();
  Color currentColor(1, 1, 1, 1);
  int currentTexture = -1;
  Camera camera;
  Options options;
  auto f = [&](size_t i) { return std::stof(info.at(i)); };

  while (std::getline(in, line)) {
    info = split(line, ' ');
    if (info.empty()) continue;
    const std::string keyword = info.at(0);
    if (keyword == "sphere") {
      scene->addPrim(makeSphere(Vec3(f(1), f(2), f(3)), f(4), currentColor, currentMaterial, currentTexture));
    } else if (keyword == "sun") {
      Light l; l.type = RT_LIGHT_DISTANT; l.v = Vec3(f(1), f(2), f(3)); l.color = currentColor;
      scene->lights.push_back(l);
    } else if (keyword == "color") {
      currentColor = Color(f(1), f(2), f(3), 1);
    } else if (keyword == "plane") {
      float A = f(1), B = f(2), C = f(3), D = f(4);
      scene->planes.push_back(makePlane(Vec3(A, B, C), D, currentColor, currentMaterial));
    } else if (keyword == "bulb") {
      Light l; l.type = RT_LIGHT_POINT; l.v = Vec3(f(1), f(2), f(3)); l.color = currentColor;
      scene->lights.push_back(l);
    } else if (keyword == "environment") {
      Light l; l.type = RT_LIGHT_ENVIRONMENT; l.radius = f(1); l.v = scene->worldCenter();
      if (info.size() < 3) {
        l.color = currentColor;
      } else {
        (void)f(2);  // scale: parsed, stored, never read by the render loop
        l.color = Color();
        l.texture = scene->loadTexture(info.at(3));
      }
      scene->lights.push_back(l);
    } else if (keyword == "xyz") {
      points.emplace_back(f(1), f(2), f(3));
    } else if (keyword == "trif") {
      int idx[3];
      for (int k = 0; k < 3; ++k) {
        idx[k] = std::stoi(info.at(k + 1)) - 1;
        if (idx[k] < 0) idx[k] += (int)points.size() + 1;
      }
      Prim t = makeTriangle(points.at(idx[0]), points.at(idx[1]), points.at(idx[2]), currentColor, currentMaterial);
      orientTriangle(t, camera);
      scene->addPrim(t);
    } else if (keyword == "expose") {
      options.exposure = f(1);
    } else if (keyword == "bounces") {
      options.maxBounces = std::stoi(info.at(1));
    } else if (keyword == "aa") {
      options.numRays = std::stoi(info.at(1));
    } else if (keyword == "roughness" || keyword == "ior") {
      // These mutate fields of a material whose lobes were already built from the constructor
      // arguments (parser.cpp:268-270, 288-290); the render loop never reads the fields again.
      (void)f(1);
    } else if (keyword == "eye") {
      camera.eye = Vec3(f(1), f(2), f(3));
    } else if (keyword == "forward") {
      camera.setForward(Vec3(f(1), f(2), f(3)));
    } else if (keyword == "up") {
      camera.setUp(Vec3(f(1), f(2), f(3)));
    } else if (keyword == "fisheye") {
      options.fisheye = true;
    } else if (keyword == "dof") {
      options.focus = f(1);
      options.lens = f(2);
    } else if (keyword == "glass") {
      currentColor = Color(0, 0, 0, 0);
      currentMaterial = scene->addMaterial(0.0f, 1.0f, 1.5f, 1.0f, 1.0f, 0.0f, 0.0f, RT_MAT_GLASS);
    } else if (keyword == "plastic") {
      currentMaterial = scene->addMaterial(0.5f, 0.5f, 1.3f, 1.0f, 0.0f, 0.0f, 0.1f, RT_MAT_PLASTIC);
    } else if (keyword == "none") {
      currentTexture = -1;
      currentMaterial = scene->addMaterialDefault();
    } else if (keyword == "copper") {
      currentColor = Color(0.7011018919f, 0.2541520943f, 0.1356333297f);
      currentMaterial = scene->addMaterial(0.0f, 1.0f, 0.23883f, 0.9553f, 0.0f, 3.415658f, 0.01f, RT_MAT_METAL);
    } else if (keyword == "gold") {
      currentColor = Color(0.6583748172794485f, 0.4286904966139066f, 0.0382043715953465f);
      currentMaterial = scene->addMaterial(0.0f, 1.0f, 0.18104f, 0.99f, 0.0f, 3.068099f, 0.01f, RT_MAT_METAL);
    } else if (keyword == "mirror") {
      currentColor = Color(0, 0, 0, 0);
      currentMaterial = scene->addMaterial(0.0f, 1.0f, 0.0f, 0.9f, 0.0f, 0.0f, 0.0f, RT_MAT_MIRROR);
    } else if (keyword == "diffuse") {
      currentMaterial = scene->addMaterialDefault();
    } else if (keyword == "reflective") {
      // 8-argument constructor with type Dialectric: a material without any lobe (parser.cpp:337-339)
      currentMaterial = scene->addMaterial(0.0f, 1.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, RT_MAT_DEFAULT);
    } else if (keyword == "texture") {
      currentTexture = scene->loadTexture(info.at(1));
    } else if (keyword == "obj") {
      std::string objError;
      if (!loadOBJ(Vec3(f(1), f(2), f(3)), f(4), info.at(5), *scene, currentColor, currentMaterial, objError))
        scene->log.push_back(objError);  // the reference ignores the failure and carries on
    }
    // every other keyword (refractive, gi, shininess, transparency, ...) is silently ignored
  }
  scene->camera = camera;
  scene->options = options;
  return scene;
}

// ------------------------------------------------------------------ .sdml
enum class Tag { Scene, Camera, Light, Shape, Material, Texture, Wavefront, Unknown };
const char *const kTagNames[] = {"Scene", "Camera", "Light", "Shape", "Material", "Texture", "Wavefront"};

Tag getTagType(const std::string &content) {  // ParserTree.cpp:42-48: content starts with the tag name
  for (int i = 0; i < 7; ++i)
    if (content.find(kTagNames[i]) == 0) return static_cast<Tag>(i);
  return Tag::Unknown;
}

struct Node {
  Tag type = Tag::Unknown;
  std::string content;
  std::vector<std::unique_ptr<Node>> children;
};

std::string parseKeyword(const std::string &line, const std::string &keyword) {  // ParserTree.cpp:50-58
  size_t start = line.find(keyword + "=\"");
  if (start == std::string::npos) return "";
  start += keyword.size() + 2;
  size_t end = line.find('"', start);
  return line.substr(start, end - start);
}

using OptionMap = std::unordered_map<std::string, std::string>;

OptionMap parseOptions(const std::string &line) {  // ParserTree.cpp:60-79; the first occurrence of a key wins
  OptionMap options;
  size_t start = line.find("options={");
  if (start == std::string::npos) return options;
  start += 9;
  size_t end = line.find('}', start);
  std::string body = line.substr(start, end == std::string::npos ? std::string::npos : end - start);
  for (const std::string &option : split(body, ';')) {
    std::vector<std::string> kv = split(option, ':');
    if (kv.size() < 2) continue;
    options.insert({trimCopy(kv[0]), trimCopy(kv[1])});
  }
  return options;
}

Vec3 tupleToVec3(const std::string &s) {  // ParserTree.cpp:81-91: "(x, y, z)"; stof stops at the ')'
  std::vector<std::string> values = split(s.substr(1, s.size() - 1), ',');
  return Vec3(std::stof(trimCopy(values.at(0))), std::stof(trimCopy(values.at(1))), std::stof(trimCopy(values.at(2))));
}

Vec3 optVec3(const OptionMap &o, const std::string &key, const Vec3 &dflt) {
  auto it = o.find(key);
  return it != o.end() ? tupleToVec3(it->second) : dflt;
}
float optFloat(const OptionMap &o, const std::string &key, float dflt) {
  auto it = o.find(key);
  return it != o.end() ? std::stof(it->second) : dflt;
}
int optInt(const OptionMap &o, const std::string &key, int dflt) {
  auto it = o.find(key);
  return it != o.end() ? std::stoi(it->second) : dflt;
}
void overrideColor(const OptionMap &o, Color &color) {  // only r,g,b change; alpha keeps the material's
  auto it = o.find("color");
  if (it == o.end()) return;
  Vec3 t = tupleToVec3(it->second);
  color.r = t.x; color.g = t.y; color.b = t.z;
}

void buildTree(std::istream &in, Node *node) {  // ParserTree.cpp:530-561
  while (true) {
    std::string line;
    while (line.empty()) {
      if (!std::getline(in, line)) return;
      line = trimCopy(line);
    }
    size_t start = line.find('<');
    size_t end = line.find('>');
    if (start == std::string::npos || end == std::string::npos || end < start) continue;
    std::string content = line.substr(start + 1, end - start - 1);
    if (!content.empty() && content[0] == '/' && node->type == getTagType(content.substr(1))) return;
    auto child = std::make_unique<Node>();
    child->type = getTagType(content);
    if (end == 0 || line[end - 1] != '/') buildTree(in, child.get());
    else content.pop_back();
    child->content = content;
    node->children.push_back(std::move(child));
  }
}

struct SdmlLoader {
  Scene &scene;
  MaterialBook book;
  explicit SdmlLoader(Scene &s) : scene(s), book(s) {}

  int parseMaterial(const Node &node, Color &color) {  // ParserTree.cpp:359-409
    std::string name = parseKeyword(node.content, "name");
    if (name.empty()) {
      OptionMap o = parseOptions(node.content);
      float Kd = optFloat(o, "Kd", 1e-4f), Ks = optFloat(o, "Ks", -1.0f), eta = optFloat(o, "eta", -1.0f);
      float Kr = optFloat(o, "Kr", -1.0f), Kt = optFloat(o, "Kt", -1.0f), Ka = optFloat(o, "Ka", -1.0f);
      float roughness = optFloat(o, "roughness", -1.0f);
      auto it = o.find("type");
      int type = it != o.end() ? materialTypeByName(it->second) : RT_MAT_DEFAULT;
      return scene.addMaterial(Kd, Ks, eta, Kr, Kt, Ka, roughness, type);
    }
    if (name == "glass") { color = Color(0, 0, 0, 0); return book.get("glass"); }
    if (name == "plastic") return book.get("plastic");
    if (name.find("copper") != std::string::npos) { color = materialColors().at(name); return book.get("copper"); }
    if (name.find("gold") != std::string::npos) { color = materialColors().at(name); return book.get("gold"); }
    if (name == "mirror") { color = Color(0, 0, 0, 0); return book.get("mirror"); }
    return book.get("default");
  }

  void parseShape(const Node &node) {  // ParserTree.cpp:416-458
    OptionMap o = parseOptions(node.content);
    std::string type = parseKeyword(node.content, "type");
    Color color(1, 1, 1, 1);
    int material = book.get("default");
    int texture = -1;
    for (const auto &child : node.children) {
      if (child->type == Tag::Material) material = parseMaterial(*child, color);
      else if (child->type == Tag::Texture) texture = scene.loadTexture(parseKeyword(child->content, "path"));
    }
    if (type == "sphere") {
      Vec3 center = optVec3(o, "center", Vec3(0, 0, 0));
      float radius = optFloat(o, "radius", 1.0f);
      overrideColor(o, color);
      scene.addPrim(makeSphere(center, radius, color, material, texture));
    } else if (type == "triangle") {
      Vec3 p1 = optVec3(o, "p1", Vec3()), p2 = optVec3(o, "p2", Vec3()), p3 = optVec3(o, "p3", Vec3());
      Vec3 t1 = optVec3(o, "t1", Vec3()), t2 = optVec3(o, "t2", Vec3()), t3 = optVec3(o, "t3", Vec3());
      overrideColor(o, color);
      Prim t = makeTriangle(p1, p2, p3, color, material, t1, t2, t3, texture);
      orientTriangle(t, scene.camera);
      scene.addPrim(t);
    } else if (type == "plane") {
      Vec3 normal = optVec3(o, "normal", Vec3(0, 0, 0));
      float D = optFloat(o, "D", 1.0f);
      Vec3 topLeft = optVec3(o, "top-left", Vec3(0, 0, 0));
      float zoom = optFloat(o, "texture-zoom", 1.0f);
      Vec3 shift = optVec3(o, "texture-shift", Vec3(0, 0, 0));
      overrideColor(o, color);
      Plane p = makePlane(normal, D, color, material, topLeft, zoom, shift, texture);
      if (dot(scene.camera.forward, p.normal) > 0) p.normal = -p.normal;  // ParserTree.cpp:449-452
      scene.planes.push_back(p);
    }
  }

  void parseWavefront(const Node &node) {  // ParserTree.cpp:460-488
    OptionMap o = parseOptions(node.content);
    std::string path = parseKeyword(node.content, "path");
    Color color(1, 1, 1, 1);
    int material = -1;
    for (const auto &child : node.children)
      if (child->type == Tag::Material) material = parseMaterial(*child, color);
    if (material < 0) material = scene.addMaterialDefault();
    Vec3 center = optVec3(o, "center", Vec3(0, 0, 0));
    float scale = optFloat(o, "scale", 1.0f);
    overrideColor(o, color);
    std::string objError;
    if (!loadOBJ(center, scale, path, scene, color, material, objError)) scene.log.push_back(objError);
  }

  void parseLight(const Node &node) {  // ParserTree.cpp:335-357, 105-137
    OptionMap o = parseOptions(node.content);
    std::string type = parseKeyword(node.content, "type");
    Light l;
    Vec3 c = optVec3(o, "color", Vec3(1, 1, 1));
    if (type == "distant") {
      l.type = RT_LIGHT_DISTANT; l.v = optVec3(o, "direction", Vec3(0, 0, 0)); l.color = Color(c.x, c.y, c.z, 1);
    } else if (type == "point") {
      l.type = RT_LIGHT_POINT; l.v = optVec3(o, "center", Vec3(0, 0, 0)); l.color = Color(c.x, c.y, c.z, 1);
    } else if (type == "environment") {
      l.type = RT_LIGHT_ENVIRONMENT;
      l.v = scene.worldCenter();
      l.radius = optFloat(o, "radius", 1.0f);
      std::string path;
      if (node.content.find("path") != std::string::npos) path = parseKeyword(node.content, "path");
      if (path.empty()) {
        l.color = Color(c.x, c.y, c.z, 1);
      } else {
        (void)optFloat(o, "scale", 1.0f);
        l.color = Color();
        l.texture = scene.loadTexture(path);
      }
    } else {
      return;
    }
    scene.lights.push_back(l);
  }

  void parseCamera(const Node &node) {  // ParserTree.cpp:320-333
    OptionMap o = parseOptions(node.content);
    Camera camera;
    camera.eye = optVec3(o, "eye", Vec3(0, 0, 0));
    camera.setForward(optVec3(o, "forward", Vec3(0, 0, -1)));
    camera.setUp(optVec3(o, "up", Vec3(0, 1, 0)));
    scene.camera = camera;
  }

  bool parseScene(Node &root, std::string &error) {  // ParserTree.cpp:258-318
    OptionMap o = parseOptions(root.content);
    Options so;
    so.bias = optFloat(o, "bias", 1e-4f);
    so.exposure = optFloat(o, "exposure", -1.0f);
    so.maxBounces = optInt(o, "maxBounces", 4);
    so.numRays = optInt(o, "numRays", 1);
    so.fisheye = optInt(o, "fisheye", 0) != 0;
    so.focus = optFloat(o, "focus", -1.0f);
    so.lens = (float)optInt(o, "lens", 0);  // the reference parses `lens` with stoi
    if (!o.count("width")) { error = "Render image width not provided."; return false; }
    if (!o.count("height")) { error = "Render image height not provided."; return false; }
    if (!o.count("filename")) { error = "Render image file name not provided."; return false; }
    scene.width = std::stoi(o["width"]);
    scene.height = std::stoi(o["height"]);
    scene.filename = o["filename"];
    scene.options = so;
    // lights are parsed after everything else so that worldCenter() is final (ParserTree.cpp:306-317)
    for (int pass = 0; pass < 2; ++pass) {
      for (const auto &child : root.children) {
        bool isLight = child->type == Tag::Light;
        if (isLight != (pass == 1)) continue;
        switch (child->type) {
          case Tag::Camera: parseCamera(*child); break;
          case Tag::Light: parseLight(*child); break;
          case Tag::Shape: parseShape(*child); break;
          case Tag::Wavefront: parseWavefront(*child); break;
          default: break;
        }
      }
    }
    return true;
  }
};

std::unique_ptr<Scene> readSdml(std::istream &in, std::string &error) {
  std::string line;
  std::getline(in, line);
  size_t start = line.find('<');
  size_t end = line.find('>');
  if (start == std::string::npos || end == std::string::npos) { error = "sdml: first line is not a tag"; return nullptr; }
  Node root;
  root.content = line.substr(start + 1, end - start - 1);
  root.type = getTagType(root.content);
  buildTree(in, &root);
  auto scene = std::make_unique<Scene>();
  if (root.type != Tag::Scene) return scene;  // the reference dispatches on the root type and does nothing else
  SdmlLoader loader(*scene);
  if (!loader.parseScene(root, error)) return nullptr;
  return scene;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- OBJ meshes
// What the reference's loadOBJ accepts and produces (src/parser/parser.cpp:47-142), read differently: the file is
// loaded once, scanned for `v` / `vn` lines and then for `f` lines (faces may name vertices that come later in the file,
// as with the reference's two passes) — both scans by all host threads, on pieces of the text cut at line ends —
// numbers go through strtof / strtol (what std::stof / std::stoi call), and the triangles (double-precision constants
// per triangle) are set up by all host threads too.
// Kept behaviours: only ' ' separates fields (a tab is part of a field); empty fields are dropped; extra fields are
// ignored; a malformed or missing number is an error (the reference aborts on the uncaught exception); `a/b/c` and
// `a//c` take normals, without transforming or normalising them, only when the FIRST corner has three fields;
// polygons are fanned; a corner index beyond the vertex list skips the triangle with a message; index 0 or a
// negative index is an error (out-of-range `at()` in the reference); vertices are recentred on the bounding-box
// centre and scaled by the largest extent (parser.cpp:93-94).
namespace {

struct Field { const char *p; size_t n; };

// fields of the line [b, e), separated by single spaces, empty ones dropped
inline void fieldsOf(const char *b, const char *e, std::vector<Field> &out) {
  out.clear();
  while (b < e) {
    const char *q = static_cast<const char *>(std::memchr(b, ' ', (size_t)(e - b)));
    if (!q) q = e;
    if (q > b) out.push_back({b, (size_t)(q - b)});
    b = q + 1;
  }
}

inline bool isField(const Field &f, const char *word) { return std::strlen(word) == f.n && std::memcmp(f.p, word, f.n) == 0; }

// std::stof on the field: the buffer is NUL-terminated and a field ends at ' ' or '\n', where strtof stops too
float fieldFloat(const std::vector<Field> &f, size_t i) {
  if (i >= f.size()) throw std::out_of_range("vector::_M_range_check");
  char *end = nullptr;
  errno = 0;
  float v = std::strtof(f[i].p, &end);
  if (end == f[i].p || end > f[i].p + f[i].n) throw std::invalid_argument("stof");
  if (errno == ERANGE) throw std::out_of_range("stof");
  return v;
}

// one corner "a", "a/b", "a//c", "a/b/c": up to three integers, -1 for an empty one; returns how many there are
int cornerOf(const Field &f, int out[3]) {
  const char *b = f.p, *e = f.p + f.n;
  int n = 0;
  while (b < e) {  // std::getline(iss, item, '/') yields no trailing empty item
    const char *q = static_cast<const char *>(std::memchr(b, '/', (size_t)(e - b)));
    if (!q) q = e;
    int v = -1;
    if (q > b) {
      char *end = nullptr;
      errno = 0;
      long l = std::strtol(b, &end, 10);
      if (end == b || end > q) throw std::invalid_argument("stoi");
      if (errno == ERANGE || l < INT_MIN || l > INT_MAX) throw std::out_of_range("stoi");
      v = (int)l;
    }
    if (n < 3) out[n] = v;
    ++n;
    b = q + 1;
  }
  return n;
}

struct Face { int v[3]; int n[3]; bool withNormals; };

}  // namespace

bool loadOBJ(const Vec3 &center, float scale, const std::string &filename, Scene &scene, const Color &color,
             int material, std::string &error) {
  std::string text;
  {
    std::ifstream infile(filename, std::ios::binary);
    if (!infile) { error = "Couldn't open file " + filename; return false; }
    infile.seekg(0, std::ios::end);
    std::streamoff size = infile.tellg();
    infile.seekg(0, std::ios::beg);
    text.resize((size_t)std::max<std::streamoff>(0, size));
    infile.read(text.data(), (std::streamsize)text.size());
    text.resize((size_t)infile.gcount());
  }
  const char *base = text.c_str(), *stop = base + text.size();
  // The text is cut at line ends into pieces that the host threads scan independently; per-piece results are joined in
  // file order, and of the errors the pieces ran into the first one in file order is reported, so that the outcome is
  // what one scan from top to bottom gives.
  std::vector<const char *> cuts;
  {
    const size_t want = (size_t)std::max(1, hostThreads() * 4);
    const size_t per = std::max<size_t>((size_t)1 << 16, text.size() / want + 1);
    cuts.push_back(base);
    for (const char *b = base; b < stop;) {
      const char *q = b + std::min<size_t>(per, (size_t)(stop - b));
      const char *e = q < stop ? static_cast<const char *>(std::memchr(q, '\n', (size_t)(stop - q))) : nullptr;
      b = e ? e + 1 : stop;
      cuts.push_back(b);
    }
  }
  const int64_t numPieces = (int64_t)cuts.size() - 1;
  struct Failure { int kind = 0; std::string what; };  // 1 out_of_range, 2 invalid_argument, 3 other
  auto guarded = [](Failure &fail, auto &&body) {
    try { body(); }
    catch (const std::out_of_range &e) { fail.kind = 1; fail.what = e.what(); }
    catch (const std::invalid_argument &e) { fail.kind = 2; fail.what = e.what(); }
    catch (const std::exception &e) { fail.kind = 3; fail.what = e.what(); }
  };
  auto rethrowFirst = [](const std::vector<Failure> &fails) {
    for (const Failure &f : fails) {
      if (f.kind == 1) throw std::out_of_range(f.what);
      if (f.kind == 2) throw std::invalid_argument(f.what);
      if (f.kind == 3) throw std::runtime_error(f.what);
    }
  };
  auto eachLineOf = [&](int64_t piece, auto &&fn) {
    for (const char *b = cuts[(size_t)piece], *end = cuts[(size_t)piece + 1]; b < end;) {
      const char *e = static_cast<const char *>(std::memchr(b, '\n', (size_t)(end - b)));
      if (!e) e = end;
      fn(b, e);
      b = e + 1;
    }
  };
  // pass 1: `v` and `vn` lines
  struct VertexPiece { std::vector<Vec3> points, normals; Vec3 extMin{kInf, kInf, kInf}, extMax{-kInf, -kInf, -kInf}; };
  std::vector<VertexPiece> vpieces((size_t)numPieces);
  std::vector<Failure> fails((size_t)numPieces);
  parallelItems(numPieces, [&](int64_t piece, int) {
    VertexPiece &vp = vpieces[(size_t)piece];
    std::vector<Field> f;
    guarded(fails[(size_t)piece], [&]() {
      eachLineOf(piece, [&](const char *b, const char *e) {
        while (b < e && *b == ' ') ++b;
        if (b == e || *b != 'v') return;  // only `v` and `vn` lines matter here
        fieldsOf(b, e, f);
        if (f.empty()) return;
        if (isField(f[0], "v")) {
          float x = fieldFloat(f, 1), y = fieldFloat(f, 2), z = fieldFloat(f, 3);
          vp.extMin = Vec3(std::min(vp.extMin.x, x), std::min(vp.extMin.y, y), std::min(vp.extMin.z, z));
          vp.extMax = Vec3(std::max(vp.extMax.x, x), std::max(vp.extMax.y, y), std::max(vp.extMax.z, z));
          vp.points.emplace_back(x, y, z);
        } else if (isField(f[0], "vn")) {
          float x = fieldFloat(f, 1), y = fieldFloat(f, 2), z = fieldFloat(f, 3);
          vp.normals.emplace_back(x, y, z);
        }
      });
    });
  });
  rethrowFirst(fails);
  std::vector<Vec3> points, normals;
  Vec3 extMin(kInf, kInf, kInf), extMax(-kInf, -kInf, -kInf);
  {
    size_t np = 0, nn = 0;
    for (const VertexPiece &vp : vpieces) { np += vp.points.size(); nn += vp.normals.size(); }
    points.reserve(np); normals.reserve(nn);
    for (const VertexPiece &vp : vpieces) {
      points.insert(points.end(), vp.points.begin(), vp.points.end());
      normals.insert(normals.end(), vp.normals.begin(), vp.normals.end());
      // std::min / std::max as above, piece after piece: a NaN coordinate is passed over either way
      extMin = Vec3(std::min(extMin.x, vp.extMin.x), std::min(extMin.y, vp.extMin.y), std::min(extMin.z, vp.extMin.z));
      extMax = Vec3(std::max(extMax.x, vp.extMax.x), std::max(extMax.y, vp.extMax.y), std::max(extMax.z, vp.extMax.z));
    }
    vpieces.clear(); vpieces.shrink_to_fit();
  }
  const int numPoints = (int)points.size();
  // pass 2: `f` lines (a face may name vertices that come later in the file, hence after pass 1)
  struct FacePiece { std::vector<Face> faces; std::vector<std::string> skipped; };
  std::vector<FacePiece> fpieces((size_t)numPieces);
  std::fill(fails.begin(), fails.end(), Failure{});
  parallelItems(numPieces, [&](int64_t piece, int) {
    FacePiece &fp = fpieces[(size_t)piece];
    std::vector<Field> f;
    guarded(fails[(size_t)piece], [&]() {
      eachLineOf(piece, [&](const char *b, const char *e) {
        while (b < e && *b == ' ') ++b;
        if (b == e || *b != 'f') return;
        fieldsOf(b, e, f);
        if (f.empty() || !isField(f[0], "f")) return;
        if (f.size() < 2) throw std::out_of_range("vector::_M_range_check");
        int c1[3] = {-1, -1, -1}, c2[3], c3[3];
        const int n1 = cornerOf(f[1], c1);
        if (n1 < 1) throw std::out_of_range("vector::_M_range_check");
        for (size_t idx = 2; idx + 1 < f.size(); ++idx) {  // triangle fan around the first corner
          c2[0] = c2[1] = c2[2] = c3[0] = c3[1] = c3[2] = -1;
          const int n2 = cornerOf(f[idx], c2), n3 = cornerOf(f[idx + 1], c3);
          if (n2 < 1 || n3 < 1) throw std::out_of_range("vector::_M_range_check");
          const int i = c1[0] - 1, j = c2[0] - 1, k = c3[0] - 1;
          if (i >= numPoints || j >= numPoints || k >= numPoints) {
            fp.skipped.push_back("Indices out of range: " + std::to_string(i) + ' ' + std::to_string(j) + ' ' + std::to_string(k));
            continue;
          }
          if (i < 0 || j < 0 || k < 0) throw std::out_of_range("vector::_M_range_check");
          Face face{{i, j, k}, {0, 0, 0}, n1 == 3};
          if (face.withNormals) {
            if (n2 < 3 || n3 < 3) throw std::out_of_range("vector::_M_range_check");
            const int a = c1[2] - 1, bb = c2[2] - 1, c = c3[2] - 1, nn = (int)normals.size();
            if (a < 0 || bb < 0 || c < 0 || a >= nn || bb >= nn || c >= nn) throw std::out_of_range("vector::_M_range_check");
            face.n[0] = a; face.n[1] = bb; face.n[2] = c;
          }
          fp.faces.push_back(face);
        }
      });
    });
  });
  rethrowFirst(fails);
  std::vector<Face> faces;
  std::vector<std::string> skipped;
  {
    size_t nf = 0;
    for (const FacePiece &fp : fpieces) nf += fp.faces.size();
    faces.reserve(nf);
    for (FacePiece &fp : fpieces) {
      faces.insert(faces.end(), fp.faces.begin(), fp.faces.end());
      for (std::string &m : fp.skipped) skipped.push_back(std::move(m));
    }
    fpieces.clear(); fpieces.shrink_to_fit();
  }
  for (std::string &m : skipped) scene.log.push_back(std::move(m));
  // parser.cpp:93-94
  const Vec3 shift = center - (extMax + extMin) / 2;
  const float scaleFactor = scale / maxDimension(extMax - extMin);
  // Scene::addObject for every triangle, in file order (src/scene/raytracer.cpp:314-317): the records are built in place
  // by all host threads, the running centroid sum — a float sum, so its order matters — is added up afterwards
  const size_t first = scene.prims.size();
  scene.prims.resize(first + faces.size());
  parallelChunks((int64_t)faces.size(), 1 << 12, [&](int64_t b, int64_t e, int) {
    for (int64_t t = b; t < e; ++t) {
      const Face &fc = faces[(size_t)t];
      Prim &p = scene.prims[first + (size_t)t];
      p = makeTriangle(scaleFactor * points[(size_t)fc.v[0]] + shift, scaleFactor * points[(size_t)fc.v[1]] + shift,
                       scaleFactor * points[(size_t)fc.v[2]] + shift, color, material);
      if (fc.withNormals) {
        p.n1 = normals[(size_t)fc.n[0]]; p.n2 = normals[(size_t)fc.n[1]]; p.n3 = normals[(size_t)fc.n[2]];
      }
      p.origIndex = (int)(first + (size_t)t);
    }
  });
  for (size_t t = first; t < scene.prims.size(); ++t) scene.centroidSum = scene.centroidSum + scene.prims[t].centroid;
  scene.log.push_back("Scanned " + std::to_string(points.size()) + " points and " + std::to_string(faces.size()) + " objects");
  return true;
}

std::unique_ptr<Scene> readFromFile(const std::string &filename, std::string &error) {  // parser.cpp:362-378
  std::ifstream infile(filename);
  if (!infile) { error = "Couldn't open file " + filename; return nullptr; }
  try {
    if (endsWith(filename, ".txt")) return readTxt(infile, error);
    if (endsWith(filename, ".sdml")) return readSdml(infile, error);
  } catch (const std::exception &e) {
    // The reference lets std::stof/stoi/at() exceptions escape and aborts; report them instead.
    error = std::string("malformed scene file: ") + e.what();
    return nullptr;
  }
  error = "Unrecognized file format " + filename;
  return nullptr;
}

}  // namespace rth
