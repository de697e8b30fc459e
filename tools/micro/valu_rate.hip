// VALU issue-rate probe: how many cycles per wave64 v_fma_f32 does one SIMD sustain with 1, 2, 3, 4 waves resident?
// (decides whether more, smaller waves (lane-per-component layout) or fewer instructions per wave (MFMA layout) pays)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void chain(float* out, int iters, float a) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      x0 = __builtin_fmaf(x0, a, 1.0f); x1 = __builtin_fmaf(x1, a, 1.0f); x2 = __builtin_fmaf(x2, a, 1.0f); x3 = __builtin_fmaf(x3, a, 1.0f);
      x4 = __builtin_fmaf(x4, a, 1.0f); x5 = __builtin_fmaf(x5, a, 1.0f); x6 = __builtin_fmaf(x6, a, 1.0f); x7 = __builtin_fmaf(x7, a, 1.0f);
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(64) void dep(float* out, int iters, float a) {
  float x0 = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 128; ++j) x0 = __builtin_fmaf(x0, a, 1.0f);
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0;
}
int main() {
  float* d; hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int kind = 0; kind < 2; ++kind)
    for (int waves : {256, 625, 1024, 1250, 2048, 2500, 3072, 4096, 8192}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(chain, dim3(waves), dim3(64), 0, 0, d, iters, 0.999f);
        else hipLaunchKernelGGL(dep, dim3(waves), dim3(64), 0, 0, d, iters, 0.999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)iters * 128;
      printf("%s waves=%5d  %.3f ms  -> %.2f ns per wave-instruction (%.2f cycles at 2.4 GHz); per-SIMD at %.2f waves/SIMD: %.2f cycles/inst\n",
             kind == 0 ? "indep8" : "dep   ", waves, ms, ms * 1e6 / insts, ms * 1e6 / insts * 2.4, waves / 1024.0,
             ms * 1e6 / insts * 2.4 / (waves > 1024 ? waves / 1024.0 : 1.0));
    }
  return 0;
}
