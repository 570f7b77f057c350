// Checks of the host library's parallel loops (raytracer_amd/host/rth_threads.h), built and run by
// tests/test_host_and_oracle.py: every index is handed out exactly once, chunk numbers follow the data, and what a
// worker throws arrives on the calling thread after all workers have been joined.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "rth_threads.h"

static int failures = 0;
#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); ++failures; } \
  } while (0)

int main() {
  for (int threads : {1, 2, 8, 16}) {
    rth::setHostThreads(threads);
    CHECK(rth::hostThreads() == threads);

    // items: each exactly once, worker numbers below the thread count
    const int64_t n = 10007;
    std::vector<std::atomic<int>> seen((size_t)n);
    for (auto &s : seen) s = 0;
    std::atomic<int> badWorker{0};
    rth::parallelItems(n, [&](int64_t i, int w) { seen[(size_t)i]++; if (w < 0 || w >= threads) badWorker++; });
    int wrong = 0;
    for (auto &s : seen) wrong += s != 1;
    CHECK(wrong == 0);
    CHECK(badWorker == 0);

    // chunks: disjoint, in index order, together [0, n)
    std::vector<int64_t> begin((size_t)threads, -1), end((size_t)threads, -1);
    rth::parallelChunks(n, 16, [&](int64_t b, int64_t e, int c) { begin[(size_t)c] = b; end[(size_t)c] = e; });
    int64_t at = 0;
    for (int c = 0; c < threads && begin[(size_t)c] >= 0; ++c) { CHECK(begin[(size_t)c] == at); CHECK(end[(size_t)c] > at); at = end[(size_t)c]; }
    CHECK(at == n);

    // a worker's exception is the caller's, and no item starts after it was seen
    bool got = false;
    std::atomic<int64_t> started{0};
    try {
      rth::parallelItems(1 << 20, [&](int64_t i, int) { ++started; if (i == 100) throw std::runtime_error("item 100"); });
    } catch (const std::runtime_error &e) {
      got = std::strcmp(e.what(), "item 100") == 0;
    }
    CHECK(got);
    CHECK(started < (1 << 20));
    got = false;
    try {
      rth::parallelChunks(n, 16, [&](int64_t b, int64_t, int) { if (b == 0) throw std::length_error("first chunk"); });
    } catch (const std::length_error &e) {
      got = std::strcmp(e.what(), "first chunk") == 0;
    }
    CHECK(got);
    got = false;
    try {
      rth::parallelChunks(n, 16, [&](int64_t, int64_t e, int) { if (e == n) throw std::length_error("last chunk"); });
    } catch (const std::length_error &e) {
      got = std::strcmp(e.what(), "last chunk") == 0;
    }
    CHECK(got);
  }
  rth::setHostThreads(0);
  if (failures == 0) std::printf("ok\n");
  return failures != 0;
}
