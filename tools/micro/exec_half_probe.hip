// Does a wave64 VALU instruction cost less when one 32-lane half of EXEC is empty? Times a long chain of FMAs under
// three lane masks: all 64 lanes, the low 32 lanes, every second lane (32 lanes spread over both halves).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(float *out, int iters, unsigned long long mask) {
  unsigned lane = threadIdx.x & 63u;
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1, e = a + 2, f = a + 3;
  if ((mask >> lane) & 1ull) {
    for (int i = 0; i < iters; ++i) {
      a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c;
      a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f;
}

int main() {
  float *out;
  const int blocks = 256 * 8, threads = 256, iters = 20000;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  const unsigned long long masks[3] = {~0ull, 0xffffffffull, 0x5555555555555555ull};
  const char *names[3] = {"all 64 lanes", "low 32 lanes", "every second lane"};
  for (int r = 0; r < 2; ++r)
    for (int m = 0; m < 3; ++m) {
      hipEventRecord(t0);
      k<<<blocks, threads>>>(out, iters, masks[m]);
      hipEventRecord(t1);
      hipEventSynchronize(t1);
      float ms = 0;
      hipEventElapsedTime(&ms, t0, t1);
      if (r) std::printf("%-18s %.3f ms\n", names[m], ms);
    }
  return 0;
}
