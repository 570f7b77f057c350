// TEST INFRASTRUCTURE ONLY. Linked into the oracle/_ref/ build of the reference.
//
// Zero-filling global operator new. The reference reads heap memory it never
// initialised: PNG::PNG(const std::string&) (src/image/PNG.h:48-50) calls
// readFromFile, which delete[]s the uninitialised image_ (src/image/PNG.cpp:134),
// and Vector3D's default constructor leaves its fields indeterminate
// (src/vector/vector3d.h:7; e.g. Plane::point, Triangle::n1..n3). With a zeroing
// allocator those reads are deterministic (0 / nullptr), which is what makes
// fixtures generated from the reference reproducible.
#include <cstdlib>
#include <new>

static void *zalloc(std::size_t n) {
  void *p = std::calloc(1, n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}

void *operator new(std::size_t n) { return zalloc(n); }
void *operator new[](std::size_t n) { return zalloc(n); }
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }
