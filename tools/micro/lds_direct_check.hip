// Where does global_load_lds_dwordx4 put a lane's 16 bytes? Checks on known data that lane L of a wavefront lands at
// LDS base + 16 L, so that four quad lanes reading consecutive 16-byte pieces of one record leave the record contiguous.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(const float4 *table, const int *recOfLane, float4 *out) {
  __shared__ float4 buf[4 * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 3, base = lane & ~3;
  const int rec = recOfLane[threadIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rj = __shfl(rec, base + j);
    const float4 *src = table + (size_t)rj * 4 + q;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)&buf[j * 256 + wave * 64], 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float4 *mine = &buf[q * 256 + wave * 64 + base];
  for (int p = 0; p < 4; ++p) out[threadIdx.x * 4 + p] = mine[p];
}
int main() {
  const int R = 1000;
  std::vector<float4> t(R * 4);
  for (int i = 0; i < R * 4; ++i) t[i] = make_float4(i * 4 + 0, i * 4 + 1, i * 4 + 2, i * 4 + 3);
  std::vector<int> rec(256);
  for (int i = 0; i < 256; ++i) rec[i] = (i * 37 + 11) % R;
  float4 *dt, *dout; int *drec;
  hipMalloc(&dt, t.size() * 16); hipMalloc(&dout, 256 * 4 * 16); hipMalloc(&drec, 256 * 4);
  hipMemcpy(dt, t.data(), t.size() * 16, hipMemcpyHostToDevice); hipMemcpy(drec, rec.data(), 256 * 4, hipMemcpyHostToDevice);
  k<<<1, 256>>>(dt, drec, dout);
  std::vector<float4> out(256 * 4);
  hipMemcpy(out.data(), dout, out.size() * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i)
    for (int p = 0; p < 4; ++p) {
      float want = (rec[i] * 4 + p) * 4;
      if (out[i * 4 + p].x != want || out[i * 4 + p].w != want + 3) { if (bad < 5) printf("thread %d piece %d: got %g want %g\n", i, p, out[i * 4 + p].x, want); ++bad; }
    }
  printf("%s (%d mismatches)\n", bad ? "LAYOUT DIFFERS" : "layout as assumed: lane L -> base + 16 L", bad);
  return bad != 0;
}
