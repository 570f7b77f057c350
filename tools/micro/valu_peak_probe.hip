// What is the vector-ALU issue roof of an MI355X for the instruction kinds the traversal kernels are made of?
// Every case runs the same shape: 256 CUs x 8 workgroups x 256 threads (8 wavefronts per SIMD), 8 independent
// dependency chains per lane, so the measured rate is the issue rate of the SIMDs and not a latency. Printed per
// case: wave64 instructions per nanosecond chip-wide, and per SIMD per shader clock at the clock measured inside the
// kernel (s_memtime ticks per s_memrealtime 100 MHz tick).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

enum Op { OP_FMA, OP_MUL, OP_ADD, OP_MIN, OP_MAX, OP_CMP_SEL, OP_SLAB, OP_RCP, OP_DIV, OP_SQRT, OP_COUNT };
static const char *kNames[OP_COUNT] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_min_f32", "v_max_f32", "v_cmp+v_cndmask",
                                       "slab mix (sub,mul,min,max)", "v_rcp_f32", "IEEE a/b (div_scale..fixup)", "IEEE sqrtf"};
// VALU instructions each `step` of a chain compiles to (checked against the ISA: see the count printed by the build recipe)
static const double kInstrPerStep[OP_COUNT] = {1, 1, 1, 1, 1, 2, 4, 1, 0, 0};

// One chain step. The single-instruction cases are written as inline assembly so that the count of issued
// instructions is exactly what the table above says (the compiler would fold a repeated min or max).
template <int OP>
__device__ __forceinline__ float step(float x, float b, float c) {
  float r;
  if (OP == OP_FMA) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(b), "v"(c)); return r; }
  if (OP == OP_MUL) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(b)); return r; }
  if (OP == OP_ADD) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c)); return r; }
  if (OP == OP_MIN) { asm volatile("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(b)); return r; }
  if (OP == OP_MAX) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c)); return r; }
  if (OP == OP_CMP_SEL) {
    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %3, vcc" : "=v"(r) : "v"(x), "v"(b), "v"(c) : "vcc");
    return r;
  }
  if (OP == OP_SLAB) {
    asm volatile("v_sub_f32 %0, %1, %3\n\tv_mul_f32 %0, %0, %2\n\tv_min_f32 %0, %0, %2\n\tv_max_f32 %0, %0, %3"
                 : "=&v"(r) : "v"(x), "v"(b), "v"(c));
    return r;
  }
  if (OP == OP_RCP) { asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
  if (OP == OP_DIV) return b / x;
  return __builtin_sqrtf(x) + c;
}

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, long long *clocks, int iters, float b, float c) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 1.0f + 0.001f * (float)(threadIdx.x + j);
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = step<OP>(v[j], b, c);
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP>
void run(float *out, long long *clocks, long long *hostClocks, int blocks, double instrPerStepOverride) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, clocks, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipMemcpy(hostClocks, clocks, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double ghz = 0;
  for (int i = 0; i < blocks; ++i) ghz += (double)hostClocks[2 * i] / (double)hostClocks[2 * i + 1] * 0.1;
  ghz /= blocks;
  const double per = instrPerStepOverride > 0 ? instrPerStepOverride : kInstrPerStep[OP];
  const double steps = (double)blocks * 4.0 /*waves*/ * iters * 32.0;  // chain steps per wavefront: 4 unrolls x 8 chains
  const double simds = 1024.0;
  if (per > 0) {
    double instr = steps * per;
    std::printf("%-30s %8.3f ms  %8.1f wave-instr/ns chip  %.3f instr/clk/SIMD at %.2f GHz\n", kNames[OP], ms, instr / (ms * 1e6),
                instr / (ms * 1e-3) / simds / (ghz * 1e9), ghz);
  } else {
    std::printf("%-30s %8.3f ms  %8.2f steps/ns chip   %.2f clk per step per SIMD at %.2f GHz\n", kNames[OP], ms, steps / (ms * 1e6),
                (ms * 1e-3) * simds * (ghz * 1e9) / steps, ghz);
  }
}

int main() {
  const int blocks = 256 * 8;
  float *out; long long *clocks;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&clocks, sizeof(long long) * 2 * blocks);
  long long *host = (long long *)std::malloc(sizeof(long long) * 2 * blocks);
  run<OP_FMA>(out, clocks, host, blocks, 0); run<OP_MUL>(out, clocks, host, blocks, 0); run<OP_ADD>(out, clocks, host, blocks, 0);
  run<OP_MIN>(out, clocks, host, blocks, 0); run<OP_MAX>(out, clocks, host, blocks, 0); run<OP_CMP_SEL>(out, clocks, host, blocks, 0);
  run<OP_SLAB>(out, clocks, host, blocks, 0); run<OP_RCP>(out, clocks, host, blocks, 0); run<OP_DIV>(out, clocks, host, blocks, 0);
  run<OP_SQRT>(out, clocks, host, blocks, 0);
  return 0;
}
