// How many L1 (vector cache) accesses per cycle can one CU serve when every lane of a 16-byte load touches a different
// cache line that is already in L1? This is the access shape of the traversal kernels (one 64-byte node record per
// lane, fetched as four 16-byte loads): the L1 looks up one line per lane and instruction. Every workgroup reads
// random 64-byte records of its own small table (8 KiB, L1-resident after the first touches), 8 waves per SIMD, loads
// independent of each other. Printed: lane-level 16-byte loads per nanosecond chip-wide and per CU per shader clock.
// Run it under `rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE` to see what the counter makes of it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int LOADS_PER_RECORD>
__global__ __launch_bounds__(256) void k_l1(const float4 *table, int recordsPerBlock, int iters, float *sink, long long *clocks) {
  const float4 *mine = table + (size_t)blockIdx.x * recordsPerBlock * 4;
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99u;
  float acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const float4 *r = mine + (size_t)(x % (uint32_t)recordsPerBlock) * 4;
#pragma unroll
    for (int k = 0; k < LOADS_PER_RECORD; ++k) { float4 v = r[k]; acc += v.x; }
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int L>
void run(const float4 *table, int recordsPerBlock, float *sink, long long *clocks, long long *host, int blocks) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k_l1<L><<<blocks, 256>>>(table, recordsPerBlock, iters, sink, clocks);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  (void)hipMemcpy(host, clocks, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double ghz = 0;
  for (int i = 0; i < blocks; ++i) ghz += (double)host[2 * i] / (double)host[2 * i + 1] * 0.1;
  ghz /= blocks;
  const double laneLoads = (double)blocks * 256.0 * iters * L;
  std::printf("%d x 16 B per 64-B record: %8.3f ms  %8.1f lane-loads/ns chip  %.3f lane-loads/clk/CU at %.2f GHz\n", L, ms, laneLoads / (ms * 1e6),
              laneLoads / (ms * 1e-3) / 256.0 / (ghz * 1e9), ghz);
}

int main() {
  const int blocks = 256 * 8, recordsPerBlock = 128;  // 8 KiB per workgroup
  float4 *table; float *sink; long long *clocks;
  (void)hipMalloc(&table, (size_t)blocks * recordsPerBlock * 64);
  (void)hipMemset(table, 0, (size_t)blocks * recordsPerBlock * 64);
  (void)hipMalloc(&sink, sizeof(float) * blocks * 256);
  (void)hipMalloc(&clocks, sizeof(long long) * 2 * blocks);
  long long *host = (long long *)std::malloc(sizeof(long long) * 2 * blocks);
  run<1>(table, recordsPerBlock, sink, clocks, host, blocks);
  run<4>(table, recordsPerBlock, sink, clocks, host, blocks);
  return 0;
}
