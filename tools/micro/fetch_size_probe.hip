// Calibrates rocprofv3's FETCH_SIZE for the access shape of the traversal kernels: every lane reads one 64-byte
// record (four dwordx4 loads) at a random index of a table far larger than the 256 MiB Infinity Cache, so nearly every
// record comes from HBM and the true byte count is known: records x 64 B. A second kernel streams the same table with
// coalesced 16-byte loads (the shape the 2x rule was stated for). Run under
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_size_probe
// and compare FETCH_SIZE (KiB) per kernel with the "true bytes" this program prints.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(256) void k_gather64(const float4 *table, uint32_t numRecords, uint32_t perThread, float *sink) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = tid * 2654435761u + 12345u;
  float acc = 0;
  for (uint32_t i = 0; i < perThread; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;  // xorshift32: a different record per lane and step
    const float4 *r = table + (size_t)(x % numRecords) * 4;
    float4 a = r[0], b = r[1], c = r[2], d = r[3];
    acc += a.x + b.y + c.z + d.w;
  }
  sink[tid] = acc;
}

__global__ __launch_bounds__(256) void k_stream16(const float4 *table, size_t numVec, float *sink) {
  size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0;
  for (size_t i = tid; i < numVec; i += stride) { float4 v = table[i]; acc += v.x + v.w; }
  sink[tid] = acc;
}

int main() {
  const size_t tableBytes = (size_t)4 << 30;  // 4 GiB, 16x the Infinity Cache
  const uint32_t numRecords = (uint32_t)(tableBytes / 64);
  float4 *table; float *sink;
  if (hipMalloc(&table, tableBytes) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
  hipMemset(table, 0, tableBytes);
  const int blocks = 256 * 8, threads = 256;
  hipMalloc(&sink, sizeof(float) * blocks * threads);
  const uint32_t perThread = 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k_gather64<<<blocks, threads>>>(table, numRecords, perThread, sink);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * threads * perThread * 64.0;
    std::printf("k_gather64: true bytes %.0f (%.1f KiB), %.3f ms, %.1f GB/s\n", bytes, bytes / 1024, ms, bytes / ms / 1e6);
    hipEventRecord(e0);
    k_stream16<<<blocks, threads>>>(table, tableBytes / 16, sink);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    std::printf("k_stream16: true bytes %.0f (%.1f KiB), %.3f ms, %.1f GB/s\n", (double)tableBytes, (double)tableBytes / 1024, ms, tableBytes / ms / 1e6);
  }
  return 0;
}
