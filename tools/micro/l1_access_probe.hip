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
    for (int k = 0; k < LOADS_PER_RECORD; ++k) { float4 v = r[k]; acc += v.x + v.y + v.z + v.w; }  // all four words used: a 16-byte load
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

// One load of BYTES (4, 8 or 16) per lane and iteration, every lane a different L1-resident record: does the L1 charge a
// scattered access by the lane or by the byte?
template <int BYTES>
__global__ __launch_bounds__(256) void k_l1_size(const float4 *table, int recordsPerBlock, int iters, float *sink, long long *clocks) {
  const float4 *mine = table + (size_t)blockIdx.x * recordsPerBlock * 4;
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99u;
  float acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const float4 *r = mine + (size_t)(x % (uint32_t)recordsPerBlock) * 4;
    if (BYTES == 4) acc += *reinterpret_cast<const float *>(r);
    else if (BYTES == 8) { float2 v = *reinterpret_cast<const float2 *>(r); acc += v.x + v.y; }
    else { float4 v = *r; acc += v.x + v.y + v.z + v.w; }
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

// Random 64-byte records out of LDS (four ds_read_b128 per lane and iteration): the rate a tree top kept in LDS would be read at.
__global__ __launch_bounds__(256) void k_lds(int records, int iters, float *sink, long long *clocks) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < records * 4; i += 256) lds[i] = make_float4((float)i, 1, 2, 3);
  __syncthreads();
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99u;
  float acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const float4 *r = lds + (size_t)(x % (uint32_t)records) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { float4 v = r[k]; acc += v.x + v.y + v.z + v.w; }
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

// The same gather done by quads: in instruction j the four lanes of a quad read the four 16-byte pieces of the record
// that quad lane j wants (one 64-byte line segment per quad and instruction instead of four). Same number of lane-level
// 16-byte loads as k_l1<4>; the question is whether the L1 serves a quad that touches one line faster.
__global__ __launch_bounds__(256) void k_l1_quad(const float4 *table, int recordsPerBlock, int iters, float *sink, long long *clocks) {
  const float4 *mine = table + (size_t)blockIdx.x * recordsPerBlock * 4;
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99u;
  float acc = 0;
  const int q = threadIdx.x & 3, base = __lane_id() & ~3;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t rec = x % (uint32_t)recordsPerBlock;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t rj = (uint32_t)__shfl((int)rec, base + j);
      float4 v = mine[(size_t)rj * 4 + q];
      acc += v.x + v.y + v.z + v.w;
    }
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
    if (L == 0) k_l1_quad<<<blocks, 256>>>(table, recordsPerBlock, iters, sink, clocks);
    else if (L == -4) k_l1_size<4><<<blocks, 256>>>(table, recordsPerBlock, iters, sink, clocks);
    else if (L == -8) k_l1_size<8><<<blocks, 256>>>(table, recordsPerBlock, iters, sink, clocks);
    else if (L == -16) k_l1_size<16><<<blocks, 256>>>(table, recordsPerBlock, iters, sink, clocks);
    else if (L == -64) k_lds<<<blocks, 256, 16384>>>(256, iters, sink, clocks);
    else k_l1<(L <= 0 ? 4 : L)><<<blocks, 256>>>(table, recordsPerBlock, iters, sink, clocks);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  (void)hipMemcpy(host, clocks, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double ghz = 0;
  for (int i = 0; i < blocks; ++i) ghz += (double)host[2 * i] / (double)host[2 * i + 1] * 0.1;
  ghz /= blocks;
  const double laneLoads = (double)blocks * 256.0 * iters * (L == 0 || L == -64 ? 4 : L < 0 ? 1 : L);
  if (L == 0) std::printf("quad-cooperative, ");
  if (L == -64) std::printf("LDS (16 KiB per workgroup), ");
  if (L < 0 && L != -64) std::printf("one %d-byte load per lane, ", -L);
  std::printf("%d x 16 B per 64-B record: %8.3f ms  %8.1f lane-loads/ns chip  %.3f lane-loads/clk/CU at %.2f GHz\n", L, ms, laneLoads / (ms * 1e6),
              laneLoads / (ms * 1e-3) / 256.0 / (ghz * 1e9), ghz);
}

// Every lane reads a random 64-byte record (four 16-byte loads) of ONE table shared by all workgroups: 2 MiB stays in every
// L2 but not in the 32-KiB L1s, 64 MiB stays in the Infinity Cache but not in the L2s, 4 GiB is HBM. What a scattered record
// fetch costs a CU when it misses the L1, by where it is found.
__global__ __launch_bounds__(256) void k_shared(const float4 *table, uint32_t records, int iters, float *sink, long long *clocks) {
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99u;
  float acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const float4 *r = table + (size_t)(x % records) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { float4 v = r[k]; acc += v.x + v.y + v.z + v.w; }
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

// The same record fetch done by quads through LDS: in instruction j the four lanes of a quad load the four 16-byte pieces
// of the record quad lane j wants, straight into LDS (global_load_lds_dwordx4: lane L's 16 bytes land at base + 16 L, so
// a record arrives in one piece); then every lane reads its own 64 bytes back with four ds_read_b128.
__global__ __launch_bounds__(256) void k_shared_quad_lds(const float4 *table, uint32_t records, int iters, float *sink, long long *clocks) {
  __shared__ float4 buf[4 * 256];  // [instruction j][thread]
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99u;
  float acc = 0;
  const int q = threadIdx.x & 3, base = __lane_id() & ~3, wave = threadIdx.x >> 6;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t rec = x % records;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t rj = (uint32_t)__shfl((int)rec, base + j);
      const float4 *src = table + (size_t)rj * 4 + q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)&buf[j * 256 + wave * 64], 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): the four loads have written LDS
    const float4 *mine = &buf[q * 256 + wave * 64 + base];  // instruction q served this lane; its quad's 64 bytes
#pragma unroll
    for (int k = 0; k < 4; ++k) { float4 v = mine[k]; acc += v.x + v.y + v.z + v.w; }
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

void runShared(size_t bytes, int wavesPerSimd) {
  const int blocks = 256 * wavesPerSimd, iters = 400;
  float4 *table; float *sink; long long *clocks;
  (void)hipMalloc(&table, bytes);
  (void)hipMemset(table, 0, bytes);
  (void)hipMalloc(&sink, sizeof(float) * blocks * 256);
  (void)hipMalloc(&clocks, sizeof(long long) * 2 * blocks);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k_shared<<<blocks, 256>>>(table, (uint32_t)(bytes / 64), iters, sink, clocks);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double records = (double)blocks * 256.0 * iters;
  float ms2 = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k_shared_quad_lds<<<blocks, 256>>>(table, (uint32_t)(bytes / 64), iters, sink, clocks);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms2, e0, e1);
  }
  std::printf("  by quads through LDS: %8.3f ms  %7.1f records/ns chip  %.3f records/clk/CU\n", ms2, records / (ms2 * 1e6), records / (ms2 * 1e-3) / 256.0 / 2.4e9);
  std::printf("shared table %6zu MiB, %d waves/SIMD: %8.3f ms  %7.1f records/ns chip  %.3f records/clk/CU (at 2.4 GHz)  %.0f GB/s\n", bytes >> 20,
              wavesPerSimd, ms, records / (ms * 1e6), records / (ms * 1e-3) / 256.0 / 2.4e9, records * 64 / (ms * 1e6));
  (void)hipFree(table); (void)hipFree(sink); (void)hipFree(clocks);
}

int main() {
  for (int w : {8}) {
    runShared((size_t)2 << 20, w);
    runShared((size_t)64 << 20, w);
    runShared((size_t)4 << 30, w);
  }
  const int blocks = 256 * 8;
  // records per workgroup: 128 = 8 KiB (stays in L1), 2048 = 128 KiB (misses the 32-KiB L1, stays in L2: 256 MiB in
  // all would not, so only a quarter of the workgroups get a table of their own and the others share them)
  for (int recordsPerBlock : {128, 2048}) {
    const int tables = recordsPerBlock == 128 ? blocks : blocks / 8;
    float4 *table; float *sink; long long *clocks;
    (void)hipMalloc(&table, (size_t)blocks * recordsPerBlock * 64);
    (void)hipMemset(table, 0, (size_t)blocks * recordsPerBlock * 64);
    (void)hipMalloc(&sink, sizeof(float) * blocks * 256);
    (void)hipMalloc(&clocks, sizeof(long long) * 2 * blocks);
    long long *host = (long long *)std::malloc(sizeof(long long) * 2 * blocks);
    std::printf("-- %d KiB of records per workgroup\n", recordsPerBlock * 64 / 1024);
    (void)tables;
    run<1>(table, recordsPerBlock, sink, clocks, host, blocks);
    run<4>(table, recordsPerBlock, sink, clocks, host, blocks);
    run<0>(table, recordsPerBlock, sink, clocks, host, blocks);
    run<-4>(table, recordsPerBlock, sink, clocks, host, blocks);
    run<-8>(table, recordsPerBlock, sink, clocks, host, blocks);
    run<-16>(table, recordsPerBlock, sink, clocks, host, blocks);
    if (recordsPerBlock == 128) run<-64>(table, recordsPerBlock, sink, clocks, host, blocks);
    (void)hipFree(table); (void)hipFree(sink); (void)hipFree(clocks); std::free(host);
  }
  return 0;
}
