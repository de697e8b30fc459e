// Ensemble CRPS of (optionally linearly read-out) posterior samples: the evaluation metric of the reference's
// training_utils.evaluate / evaluate_horizon (training_utils.py:147-176, :247-264), which stacks mc_itr decoder outputs
// (T', B, obs, M) and calls properscoring.crps_ensemble element by element in a triple Python loop.
//
// Here the M decoder passes are ONE solver launch over a batch of M * B latents, and this kernel scores the result
// without materialising x_hat (M x 320 MB at the bench shape): a workgroup owns one (time, patient) row, stages the row's M
// latent vectors (M * D floats) and the readout matrix in LDS, every thread owns one observed component, forms its M
// ensemble values x_m = W[o] . h_m + b[o] into an LDS column and evaluates
//     CRPS = 1/M sum_m |x_m - y|  -  1/M^2 sum_{i<j} |x_i - x_j|          (properscoring's equal-weight estimator)
// with the pairwise form: at M = 50 the 1225 pairs cost what a 64-wide sorting network would, with no cancellation.
// Deterministic (fixed summation order, no atomics).
#include <hip/hip_runtime.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"

namespace hode {

constexpr int kCrpsThreads = 128;

struct CrpsArgs {
  const float* __restrict__ h;
  const float* __restrict__ w;
  const float* __restrict__ b;
  const float* __restrict__ truth;
  float* __restrict__ crps;
  float* __restrict__ crps_sum;
  long long ts, ms, ps;
  int B, M, Dv, obs;
};

__global__ __launch_bounds__(kCrpsThreads) void crps_kernel(CrpsArgs a) {
  extern __shared__ float lds[];
  float* hm = lds;                                        // [M][Dv] member vectors of this row
  float* wT = hm + a.M * a.Dv;                            // [Dv][128] readout, component-minor (conflict-free)
  float* vals = wT + (a.w ? a.Dv * kCrpsThreads : 0);     // [M][128] ensemble values, one column per thread
  __shared__ float red[kCrpsThreads / 64];
  const int tid = threadIdx.x;
  const long long row = blockIdx.x;
  const int t = (int)(row / a.B), b = (int)(row % a.B);
  const float* hrow = a.h + t * a.ts + b * a.ps;
  for (int idx = tid; idx < a.M * a.Dv; idx += kCrpsThreads) {
    const int m = idx / a.Dv, d = idx - m * a.Dv;
    hm[idx] = hrow[m * a.ms + d];
  }
  if (a.w) {
    for (int idx = tid; idx < a.obs * a.Dv; idx += kCrpsThreads) {
      const int o = idx / a.Dv, d = idx - o * a.Dv;
      wT[d * kCrpsThreads + o] = a.w[idx];
    }
  }
  __syncthreads();
  const bool active = tid < a.obs;
  const int o = active ? tid : 0;
  const float y = a.truth[row * a.obs + o];
  const float bias = (a.w && a.b) ? a.b[o] : 0.f;
  float s1 = 0.f;
  for (int m = 0; m < a.M; ++m) {
    float v;
    if (a.w) {
      v = bias;
      for (int d = 0; d < a.Dv; ++d) v = __builtin_fmaf(wT[d * kCrpsThreads + o], hm[m * a.Dv + d], v);
    } else {
      v = hm[m * a.Dv + o];
    }
    vals[m * kCrpsThreads + tid] = v;
    s1 += __builtin_fabsf(v - y);
  }
  // each thread reads back only its own column: no barrier needed
  float s2 = 0.f;
  for (int i = 1; i < a.M; ++i) {
    const float xi = vals[i * kCrpsThreads + tid];
    float acc = 0.f;
    for (int j = 0; j < i; ++j) acc += __builtin_fabsf(xi - vals[j * kCrpsThreads + tid]);
    s2 += acc;
  }
  const float inv = 1.0f / (float)a.M;
  const float c = active ? (s1 * inv - s2 * inv * inv) : 0.f;
  if (a.crps && active) a.crps[row * a.obs + o] = c;
  if (a.crps_sum) {
    const float w = wave_sum(c);
    if ((tid & 63) == 0) red[tid >> 6] = w;
    __syncthreads();
    if (tid == 0) a.crps_sum[row] = red[0] + red[1];
  }
}

}  // namespace hode

extern "C" int hode_ensemble_crps(const hode_crps_desc* d, void* stream) {
  if (!d) return hode::fail(HODE_E_NULL, "desc is NULL");
  if (d->struct_size != sizeof(hode_crps_desc))
    return hode::fail(HODE_E_SIZE, "struct_size %u != %zu", d->struct_size, sizeof(hode_crps_desc));
  if (d->n_times <= 0 || d->batch <= 0 || d->n_members <= 0 || d->obs_dim <= 0 || d->latent_dim <= 0)
    return hode::fail(HODE_E_SIZE, "non-positive dimension");
  if (d->obs_dim > hode::kCrpsThreads || d->n_members > 128 || d->latent_dim > 128)
    return hode::fail(HODE_E_UNSUPPORTED, "obs_dim %d / n_members %d / latent_dim %d beyond 128", d->obs_dim, d->n_members,
                      d->latent_dim);
  if (!d->w && d->latent_dim < d->obs_dim)
    return hode::fail(HODE_E_SIZE, "identity readout needs latent_dim >= obs_dim");
  if (!d->h || !d->truth || (!d->crps && !d->crps_sum)) return hode::fail(HODE_E_NULL, "h / truth / an output is NULL");
  if ((long long)d->n_times * d->batch > 0x7fffffffLL) return hode::fail(HODE_E_SIZE, "n_times * batch exceeds 2^31");
  hode::CrpsArgs a{};
  a.h = d->h; a.w = d->w; a.b = d->b; a.truth = d->truth; a.crps = d->crps; a.crps_sum = d->crps_sum;
  a.ts = d->time_stride; a.ms = d->member_stride; a.ps = d->patient_stride;
  a.B = d->batch; a.M = d->n_members; a.Dv = d->latent_dim; a.obs = d->obs_dim;
  const size_t lds = sizeof(float) * ((size_t)a.M * a.Dv + (a.w ? (size_t)a.Dv * hode::kCrpsThreads : 0) +
                                      (size_t)a.M * hode::kCrpsThreads);
  if (lds > 160 * 1024) return hode::fail(HODE_E_UNSUPPORTED, "needs %zu B of LDS", lds);
  if (lds > 64 * 1024)
    if (int e = hode::hip_fail(hipFuncSetAttribute((const void*)hode::crps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds), "crps LDS attribute")) return e;
  hipLaunchKernelGGL(hode::crps_kernel, dim3((unsigned)((long long)d->n_times * d->batch)), dim3(hode::kCrpsThreads), lds,
                     (hipStream_t)stream, a);
  return hode::hip_fail(hipGetLastError(), "crps launch");
}
