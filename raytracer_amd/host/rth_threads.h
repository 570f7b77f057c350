// Host-side worker threads: what `-t N` of the command line controls here (parsing, triangle set-up, BVH build).
// Every parallel loop in the host library splits work whose result does not depend on the split, so any thread count
// gives the same bytes.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

namespace rth {

int hostThreads();          // current setting (default: hardware threads, at most 16)
void setHostThreads(int n); // n <= 0 restores the default

// fn(begin, end, chunk) over [0, n) in contiguous chunks of at least `grain` items, one chunk per worker; `chunk`
// counts the chunks in index order (0 .. at most hostThreads() - 1), so per-chunk partial results can be combined in
// the order of the data.
template <class F>
void parallelChunks(int64_t n, int64_t grain, F fn) {
  int workers = (int)std::min<int64_t>(hostThreads(), std::max<int64_t>(1, n / std::max<int64_t>(1, grain)));
  if (workers <= 1) { fn((int64_t)0, n, 0); return; }
  std::vector<std::thread> pool;
  int64_t per = (n + workers - 1) / workers;
  for (int w = 0; w < workers; ++w) {
    int64_t b = w * per, e = std::min(n, b + per);
    if (b >= e) break;
    pool.emplace_back([=]() { fn(b, e, w); });
  }
  for (auto &t : pool) t.join();
}

// fn(i, worker) for i in [0, n), items handed out one at a time (uneven work).
template <class F>
void parallelItems(int64_t n, F fn) {
  int workers = (int)std::min<int64_t>(hostThreads(), n);
  if (workers <= 1) { for (int64_t i = 0; i < n; ++i) fn(i, 0); return; }
  std::atomic<int64_t> next{0};
  std::vector<std::thread> pool;
  for (int w = 0; w < workers; ++w)
    pool.emplace_back([&, w]() { for (int64_t i; (i = next.fetch_add(1)) < n;) fn(i, w); });
  for (auto &t : pool) t.join();
}

}  // namespace rth
