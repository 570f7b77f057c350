// Host-side worker threads: what `-t N` of the command line controls here (parsing, triangle set-up, BVH build).
// Every parallel loop in the host library splits work whose result does not depend on the split, so any thread count
// gives the same bytes.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace rth {

int hostThreads();          // current setting (default: hardware threads, at most 16)
void setHostThreads(int n); // n <= 0 restores the default

// What a worker's function threw, kept for the calling thread (the first one wins).
struct WorkerFailure {
  std::mutex lock;
  std::exception_ptr first;
  void record() noexcept {
    std::lock_guard<std::mutex> g(lock);
    if (!first) first = std::current_exception();
  }
  void rethrow() { if (first) std::rethrow_exception(first); }
};

// Both loops below keep two promises. The calling thread is one of the workers, and where the system has no further
// thread to give, it (and the threads that did start) do all the work. An exception inside fn, on whichever thread,
// is thrown again on the calling thread after every worker has been joined.

// fn(begin, end, chunk) over [0, n) in contiguous chunks of at least `grain` items, one chunk per worker; `chunk`
// counts the chunks in index order (0 .. at most hostThreads() - 1), so per-chunk partial results can be combined in
// the order of the data.
template <class F>
void parallelChunks(int64_t n, int64_t grain, F fn) {
  int workers = (int)std::min<int64_t>(hostThreads(), std::max<int64_t>(1, n / std::max<int64_t>(1, grain)));
  if (workers <= 1) { fn((int64_t)0, n, 0); return; }
  WorkerFailure failure;
  auto run = [&](int64_t b, int64_t e, int w) noexcept {
    try { fn(b, e, w); } catch (...) { failure.record(); }
  };
  std::vector<std::thread> pool;
  pool.reserve((size_t)workers);
  const int64_t per = (n + workers - 1) / workers;
  for (int w = 1; w < workers; ++w) {
    const int64_t b = w * per, e = std::min(n, b + per);
    if (b >= e) break;
    try {
      pool.emplace_back([=, &run]() { run(b, e, w); });
    } catch (const std::system_error &) {
      run(b, e, w);
    }
  }
  run(0, std::min(n, per), 0);
  for (auto &t : pool) t.join();
  failure.rethrow();
}

// fn(i, worker) for i in [0, n), items handed out one at a time (uneven work).
template <class F>
void parallelItems(int64_t n, F fn) {
  int workers = (int)std::min<int64_t>(hostThreads(), n);
  if (workers <= 1) { for (int64_t i = 0; i < n; ++i) fn(i, 0); return; }
  std::atomic<int64_t> next{0};
  WorkerFailure failure;
  auto run = [&](int w) noexcept {
    try {
      for (int64_t i; (i = next.fetch_add(1)) < n;) fn(i, w);
    } catch (...) {
      failure.record();
      next.store(n);  // nobody starts another item
    }
  };
  std::vector<std::thread> pool;
  pool.reserve((size_t)workers);
  for (int w = 1; w < workers; ++w) {
    try {
      pool.emplace_back([&run, w]() { run(w); });
    } catch (const std::system_error &) {
      break;
    }
  }
  run(0);
  for (auto &t : pool) t.join();
  failure.rethrow();
}

}  // namespace rth
