// Development aid: sustained rate of v_mfma_f32_32x32x2_f32 with 8 accumulators, one wave per SIMD, and the
// in-kernel clock (s_memtime ticks per s_memrealtime 100 MHz tick).  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f + 0.5f, b = 1.0f - threadIdx.x * 1e-3f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* clk;
  const int blocks = 256;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    const int iters = 200000;
    hipEventRecord(e0);
    k<<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512]; hipMemcpy(h, clk, blocks * 16, hipMemcpyDeviceToHost);
    double flop = (double)blocks * 4 * iters * 8 * 4096.0;
    printf("rep %d: %.2f ms  %.1f TFLOP/s  clock %.0f MHz\n", rep, ms, flop / ms / 1e9, (double)h[0] / h[1] * 100.0);
  }
  return 0;
}
