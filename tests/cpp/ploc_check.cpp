// CPU check of the clustering steps the GPU BVH build runs (raytracer_amd/csrc/rt_ploc.h): the same per-element functions,
// driven by plain loops instead of kernels, on inputs chosen to stress the indexing (one object, two, runs of identical
// boxes, NaN boxes, random soups). Built with the address and undefined-behaviour sanitizers by the test that runs it.
// Prints "ok" and exits 0, or says what broke.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "rt_ploc.h"
#include "rt_top_tree.h"

using namespace rtd::ploc;

namespace {

struct Tree {
  uint32_t n = 0, root = 0, rounds = 0;
  std::vector<Box> box;
  std::vector<uint32_t> child0, child1, objects, records;
  std::vector<int32_t> parent;
};

bool fail(const char *what, long a = 0, long b = 0) {
  std::printf("FAILED: %s (%ld, %ld)\n", what, a, b);
  return false;
}

bool build(const std::vector<Box> &leaves, int radius, uint32_t maxRounds, uint32_t topTarget, Tree &t) {
  const uint32_t n = (uint32_t)leaves.size();
  t.n = n;
  t.box.assign(leaves.begin(), leaves.end());
  t.box.resize(2 * (size_t)n - 1);
  t.child0.assign(n > 1 ? n - 1 : 1, 0xdeadbeefu);
  t.child1 = t.child0;
  t.parent.assign(2 * (size_t)n - 1, kNoParent);
  std::vector<uint32_t> cur(n), next(n), out(n), partner(n), keep(n);
  for (uint32_t i = 0; i < n; ++i) cur[i] = i;
  uint32_t count = n, made = 0;
  t.rounds = 0;
  while (count > topTarget) {
    if (++t.rounds > maxRounds) return fail("too many rounds", t.rounds, count);
    for (uint32_t i = 0; i < count; ++i) nearest(i, count, radius, cur.data(), t.box.data(), partner.data());
    for (uint32_t i = 0; i < count; ++i)
      if (partner[i] >= count || partner[i] == i) return fail("partner out of range", i, partner[i]);
    for (uint32_t i = 0; i < count; ++i)
      keep[i] = merge(i, cur.data(), partner.data(), t.box.data(), n, t.child0.data(), t.child1.data(), t.parent.data(), out.data(), [&]() {
        const uint32_t id = n + made++;
        if (id > 2 * n - 2) { std::printf("FAILED: node id %u beyond 2n - 2\n", id); std::exit(1); }
        return id;
      });
    uint32_t m = 0;
    for (uint32_t i = 0; i < count; ++i)
      if (keep[i]) next[m++] = out[i];
    if (m >= count) return fail("a round merged nothing", t.rounds, count);
    count = m;
    cur.swap(next);
  }
  // counts below the clusters that are left (a forest), then the levels above them top-down
  t.objects.assign(2 * (size_t)n - 1, 0);
  t.records.assign(2 * (size_t)n - 1, 0);
  for (uint32_t i = 0; i < n; ++i) t.objects[i] = 1;
  std::vector<uint32_t> arrived(n > 1 ? n - 1 : 1, 0);
  for (uint32_t i = 0; i < n; ++i)
    countUp(i, n, 4, t.child0.data(), t.child1.data(), t.parent.data(), t.objects.data(), t.records.data(),
            [&](uint32_t k) { return arrived[k]++ == 1u; });
  {
    std::vector<uint32_t> ids(cur.begin(), cur.begin() + count), obj(count), rec(count);
    std::vector<Box> bx(count);
    for (uint32_t k = 0; k < count; ++k) { bx[k] = t.box[ids[k]]; obj[k] = t.objects[ids[k]]; rec[k] = t.records[ids[k]]; }
    const TopTree top = buildTopTree(ids, bx, obj, rec, n + made, 4);
    if (top.child0.size() != count - 1) return fail("top tree size", (long)top.child0.size(), count - 1);
    for (uint32_t k = 0; k + 1 < count; ++k) {
      const uint32_t id = n + made + k;
      if (id > 2 * n - 2) return fail("top node id", id);
      t.child0[id - n] = top.child0[k]; t.child1[id - n] = top.child1[k];
      t.box[id] = top.box[k]; t.objects[id] = top.objects[k]; t.records[id] = top.records[k]; t.parent[id] = top.parent[k];
    }
    for (uint32_t k = 0; k < count; ++k)
      if (count > 1) t.parent[ids[k]] = top.parentOfItem[k];
    made += count - 1;
    if (made != n - 1) return fail("number of internal nodes", made, n - 1);
    t.root = top.root;
  }
  if (t.objects[t.root] != n) return fail("objects under the root", t.objects[t.root], n);
  return true;
}

bool check(const std::vector<Box> &leaves, int radius, uint32_t maxRounds, const char *name, uint32_t topTarget = 1) {
  Tree t;
  if (!build(leaves, radius, maxRounds, topTarget, t)) { std::printf("  in case %s\n", name); return false; }
  const uint32_t n = t.n;
  // every object gets its own place in the depth-first order; every record its own index below records[root]
  std::vector<char> seen(n, 0), recSeen(std::max<uint32_t>(1, t.records[t.root]), 0);
  uint32_t maxDepth = 0;
  for (uint32_t v = 0; v < 2 * n - 1; ++v) {
    const Place p = place(v, n, t.child0.data(), t.child1.data(), t.parent.data(), t.objects.data(), t.records.data());
    maxDepth = std::max(maxDepth, p.depth);
    if (v < n) {
      if (p.firstObject >= n || seen[p.firstObject]) return fail("object position", v, p.firstObject);
      seen[p.firstObject] = 1;
    } else {
      if (p.firstObject + t.objects[v] > n) return fail("subtree range", v, p.firstObject);
      if (t.records[v] > 0) {  // a record node
        if (p.recordIndex >= t.records[t.root] || recSeen[p.recordIndex]) return fail("record index", v, p.recordIndex);
        recSeen[p.recordIndex] = 1;
      }
      // a node's box holds its children's (NaN coordinates aside)
      for (uint32_t c : {t.child0[v - n], t.child1[v - n]})
        for (int k = 0; k < 3; ++k) {
          if (t.box[c].mn[k] == t.box[c].mn[k] && !(t.box[v].mn[k] <= t.box[c].mn[k])) return fail("box min", v, c);
          if (t.box[c].mx[k] == t.box[c].mx[k] && !(t.box[v].mx[k] >= t.box[c].mx[k])) return fail("box max", v, c);
        }
      // the objects of a subtree are contiguous: the left child starts where the node starts, the right one behind it
      const Place pl = place(t.child0[v - n], n, t.child0.data(), t.child1.data(), t.parent.data(), t.objects.data(), t.records.data());
      const Place pr = place(t.child1[v - n], n, t.child0.data(), t.child1.data(), t.parent.data(), t.objects.data(), t.records.data());
      if (pl.firstObject != p.firstObject || pr.firstObject != p.firstObject + t.objects[t.child0[v - n]]) return fail("child ranges", v);
    }
  }
  for (uint32_t i = 0; i < n; ++i)
    if (!seen[i]) return fail("object position unused", i);
  for (uint32_t i = 0; i < t.records[t.root]; ++i)
    if (!recSeen[i]) return fail("record index unused", i);
  std::printf("  %-28s n %7u  rounds %3u  records %7u  depth %u\n", name, n, t.rounds, t.records[t.root], maxDepth);
  return true;
}

Box boxAt(float x, float y, float z, float s) { return Box{{x, y, z}, {x + s, y + s, z + s}}; }

}  // namespace

int main() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  bool ok = true;
  for (int radius : {1, 4, 16}) {
    std::printf("radius %d\n", radius);
    ok &= check({boxAt(0, 0, 0, 1)}, radius, 8, "one object");
    ok &= check({boxAt(0, 0, 0, 1), boxAt(2, 0, 0, 1)}, radius, 8, "two objects");
    ok &= check({boxAt(0, 0, 0, 1), boxAt(2, 0, 0, 1), boxAt(4, 0, 0, 1), boxAt(9, 0, 0, 1), boxAt(9.5f, 0, 0, 1)}, radius, 16, "five objects");
    {
      std::vector<Box> same(5000, boxAt(1, 2, 3, 0.5f));  // identical boxes: pairs 0-1, 2-3, ... every round
      ok &= check(same, radius, 40, "5000 identical boxes");
    }
    {
      std::vector<Box> soup;
      for (int i = 0; i < 20000; ++i) soup.push_back(boxAt(U(rng) * 10, U(rng) * 10, U(rng) * 10, U(rng) * 0.1f));
      ok &= check(soup, radius, 400, "random soup, unsorted");
      std::sort(soup.begin(), soup.end(), [](const Box &a, const Box &b) { return a.mn[0] < b.mn[0]; });
      ok &= check(soup, radius, 400, "random soup, sorted by x");
    }
    {
      std::vector<Box> line;  // boxes growing geometrically along a line: the worst case for pairing
      float x = 0, s = 1e-3f;
      for (int i = 0; i < 3000; ++i) { line.push_back(boxAt(x, 0, 0, s)); x += s * 1.5f; s *= 1.01f; }
      ok &= check(line, radius, 3000, "geometric line");
    }
    {
      std::vector<Box> bad;
      const float nan = std::nanf("");
      for (int i = 0; i < 3000; ++i) {
        Box b = boxAt(U(rng), U(rng), U(rng), 0.01f);
        if (i % 7 == 0) b = Box{{nan, nan, nan}, {nan, nan, nan}};
        if (i % 11 == 0) b.mx[1] = INFINITY;
        bad.push_back(b);
      }
      ok &= check(bad, radius, 3000, "NaN and infinite boxes");
    }
  }
  // clustering stopped early, the levels above the remaining clusters built top-down (what the GPU build does)
  std::printf("with a top-down top\n");
  for (uint32_t top : {2u, 7u, 64u, 2048u, 100000u}) {
    std::vector<Box> soup;
    for (int i = 0; i < 20000; ++i) soup.push_back(boxAt(U(rng) * 10, U(rng) * 10, U(rng) * 10, U(rng) * 0.1f));
    std::sort(soup.begin(), soup.end(), [](const Box &a, const Box &b) { return a.mn[0] < b.mn[0]; });
    ok &= check(soup, 16, 400, "random soup", top);
    std::vector<Box> same(3000, boxAt(1, 2, 3, 0.5f));
    ok &= check(same, 16, 40, "3000 identical boxes", top);
    std::vector<Box> bad;
    const float nan = std::nanf("");
    for (int i = 0; i < 3000; ++i) {
      Box b = boxAt(U(rng), U(rng), U(rng), 0.01f);
      if (i % 7 == 0) b = Box{{nan, nan, nan}, {nan, nan, nan}};
      if (i % 11 == 0) b.mx[1] = INFINITY;
      bad.push_back(b);
    }
    ok &= check(bad, 16, 3000, "NaN and infinite boxes", top);
    ok &= check({boxAt(0, 0, 0, 1)}, 16, 8, "one object", top);
    ok &= check({boxAt(0, 0, 0, 1), boxAt(2, 0, 0, 1), boxAt(4, 0, 0, 1)}, 16, 8, "three objects", top);
  }
  std::printf(ok ? "ok\n" : "FAILED\n");
  return ok ? 0 : 1;
}
