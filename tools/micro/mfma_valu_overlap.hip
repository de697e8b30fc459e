// Does VALU work hide under fp32 MFMAs on gfx950?  One MFMA stream (v_mfma_f32_16x16x4_f32, 8 independent accumulators)
// and one VALU stream (independent v_fma_f32), alone, interleaved in ONE wave, and in TWO waves of the same SIMD.
// Inline asm throughout so that nothing is reordered; time = s_memtime of one wave around the loop.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA(i) "v_mfma_f32_16x16x4_f32 %" #i ", %8, %9, %" #i "\n"
#define FMA2(a, b) "v_fma_f32 %" #a ", %" #a ", %12, %" #a "\n v_fma_f32 %" #b ", %" #b ", %12, %" #b "\n"

// MODE 0: 8 MFMAs per iteration; 1: 16 VALU per iteration; 2: 8 x (1 MFMA + 2 VALU) interleaved in the wave
// MODE 3: waves 0-3 of the block run mode 0, waves 4-7 run mode 1 (block of 512: one of each per SIMD)
// MODE 6: 8 MFMAs then 16 VALU per iteration, in bursts, in the wave
// MODE 4 / 5: the same block shape with one of the two groups idle (baselines for mode 3)
template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, float* sink, int iters, float a) {
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  float x = threadIdx.x, y = a;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
  const int wave = threadIdx.x >> 6;
  const int mode = MODE == 3 ? (wave < 4 ? 0 : 1) : MODE == 4 ? (wave < 4 ? 9 : 1) : MODE == 5 ? (wave < 4 ? 0 : 9) : MODE;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (mode == 9) break;
    if (mode == 0) {
      asm volatile(MFMA(0) MFMA(1) MFMA(2) MFMA(3) MFMA(4) MFMA(5) MFMA(6) MFMA(7)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                   : "v"(x), "v"(y), "v"(v0), "v"(v1), "v"(a));
    } else if (mode == 1) {
      asm volatile(FMA2(0, 1) FMA2(2, 3) FMA2(0, 1) FMA2(2, 3) FMA2(0, 1) FMA2(2, 3) FMA2(0, 1) FMA2(2, 3)
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                   : "v"(x), "v"(y), "v"(x), "v"(y), "v"(a));
    } else if (mode == 6) {
      asm volatile(MFMA(0) MFMA(1) MFMA(2) MFMA(3) MFMA(4) MFMA(5) MFMA(6) MFMA(7)
                   "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                   : "v"(x), "v"(y), "v"(v0), "v"(v1), "v"(a));
    } else {
      asm volatile(MFMA(0) "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   MFMA(1) "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   MFMA(2) "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   MFMA(3) "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   MFMA(4) "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   MFMA(5) "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   MFMA(6) "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   MFMA(7) "v_fma_f32 %10, %10, %12, %10\n v_fma_f32 %11, %11, %12, %11\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                   : "v"(x), "v"(y), "v"(v0), "v"(v1), "v"(a));
    }
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0] + v0 + v1 + v2 + v3;
}

template <int MODE>
void run(const char* name, int threads, int iters, unsigned long long* d_out, float* d_sink) {
  const int blocks = 256;
  hipMemset(d_out, 0, blocks * 8 * sizeof(unsigned long long));
  for (int rep = 0; rep < 2; ++rep) probe<MODE><<<blocks, threads>>>(d_out, d_sink, iters, 1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), d_out, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
  std::vector<double> a, b;
  for (int i = 0; i < blocks; ++i)
    for (int w = 0; w < threads / 64; ++w) (w < 4 ? a : b).push_back((double)h[i * 8 + w] / iters);
  std::sort(a.begin(), a.end());
  printf("%-58s waves 0-3: %7.1f cycles per iteration (8 MFMA and / or 16 VALU)", name, a[a.size() / 2]);
  if (!b.empty()) { std::sort(b.begin(), b.end()); printf("   waves 4-7: %7.1f", b[b.size() / 2]); }
  printf("\n");
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 256 * 8 * sizeof(unsigned long long));
  hipMalloc(&d_sink, 256 * 512 * sizeof(float));
  const int iters = 20000;
  run<0>("MFMA only, 1 wave / SIMD", 256, iters, d_out, d_sink);
  run<1>("VALU only, 1 wave / SIMD", 256, iters, d_out, d_sink);
  run<2>("1 MFMA : 2 VALU interleaved, 1 wave / SIMD", 256, iters, d_out, d_sink);
  run<0>("MFMA only, 2 waves / SIMD", 512, iters, d_out, d_sink);
  run<3>("MFMA wave + VALU wave on each SIMD", 512, iters, d_out, d_sink);
  run<2>("1 MFMA : 2 VALU interleaved, 2 waves / SIMD", 512, iters, d_out, d_sink);
  run<6>("8 MFMA then 16 VALU (bursts), 1 wave / SIMD", 256, iters, d_out, d_sink);
  run<6>("8 MFMA then 16 VALU (bursts), 2 waves / SIMD", 512, iters, d_out, d_sink);
  run<4>("waves 0-3 idle, waves 4-7 VALU", 512, iters, d_out, d_sink);
  run<5>("waves 0-3 MFMA, waves 4-7 idle", 512, iters, d_out, d_sink);
  run<1>("VALU only, 2 waves / SIMD", 512, iters, d_out, d_sink);
  return 0;
}
