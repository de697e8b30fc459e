// Monte-Carlo KL term of the ELBO against the Exponential(rate) prior, forward and gradients in one pass, gfx950.
//
// Reference: VariationalInference.mc_kl (model.py:1198-1214) with encoder.reparameterize / log_density (model.py:18-31)
// and ExponentialPrior.log_density (model.py:41-45), called with sample_size = 100 at model.py:1190:
//     for s in range(S):  z = eps_s * sigma + mu;  z[z <= 0] = epsilon;  mc_s = log N(z; mu, sigma) - log Exp(z; rate)
//     kl[b] = mean_s sum_d mc_s[b][d]
// As eager ops that is ~40 element-wise kernels over (S, B, D) tensors per training step (0.6 ms of 12.6 ms at the bench
// shape).  Here one thread owns one (b, d) element, walks the S noise draws (coalesced across threads), and emits the
// element's mean contribution and its analytic derivatives (the in-place clamp blocks the gradient through z exactly as
// autograd's index_put_ does):
//     z > 0:   term = -eps^2/2 - log sigma - c - log rate + rate z     d/dmu = rate             d/dlogvar = -1/2 + rate eps sigma / 2
//     z <= 0:  term = -(e-mu)^2/(2 sigma^2) - log sigma - c - log rate + rate e   d/dmu = (e-mu)/sigma^2   d/dlogvar = (e-mu)^2/(2 sigma^2) - 1/2
#include <hip/hip_runtime.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"

namespace hode {

struct McKlArgs {
  const float* __restrict__ mu;
  const float* __restrict__ log_var;
  const float* __restrict__ noise;  // [S][rows]
  float* __restrict__ kl;           // [rows]
  float* __restrict__ grad_mu;      // [rows] or nullptr
  float* __restrict__ grad_lv;      // [rows] or nullptr
  long long rows;
  int S;
  float rate, eps_clamp;
};

__global__ __launch_bounds__(256) void mc_kl_exp_kernel(McKlArgs a) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= a.rows) return;
  const float mu = a.mu[i], lv = a.log_var[i];
  const float sigma = exp_f32(0.5f * lv);
  const float inv_var = __builtin_amdgcn_rcpf(sigma * sigma);
  const float e = a.eps_clamp;
  // clamped-sample constants
  const float dm = e - mu;
  const float q_cl = -0.5f * dm * dm * inv_var;
  const float gmu_cl = dm * inv_var;
  const float glv_cl = 0.5f * dm * dm * inv_var;
  float s_term = 0.f, s_gmu = 0.f, s_glv = 0.f;
  for (int s = 0; s < a.S; ++s) {
    const float eps = a.noise[(size_t)s * a.rows + i];
    const float z = __builtin_fmaf(eps, sigma, mu);
    const bool pos = z > 0.0f;
    s_term += pos ? __builtin_fmaf(a.rate, z, -0.5f * eps * eps) : __builtin_fmaf(a.rate, e, q_cl);
    s_gmu += pos ? a.rate : gmu_cl;
    s_glv += pos ? 0.5f * a.rate * eps * sigma : glv_cl;
  }
  const float inv_s = 1.0f / (float)a.S;
  const float common = -0.5f * lv - 0.9189385332046727f - log_f32(a.rate);  // -log sigma - log sqrt(2 pi) - log rate
  a.kl[i] = __builtin_fmaf(s_term, inv_s, common);
  if (a.grad_mu) a.grad_mu[i] = s_gmu * inv_s;
  if (a.grad_lv) a.grad_lv[i] = __builtin_fmaf(s_glv, inv_s, -0.5f);
}

}  // namespace hode

extern "C" int hode_mc_kl_exponential(const hode_mckl_desc* d, void* stream) {
  if (!d) return hode::fail(HODE_E_NULL, "desc is NULL");
  if (d->struct_size != sizeof(hode_mckl_desc))
    return hode::fail(HODE_E_SIZE, "struct_size %u != %zu", d->struct_size, sizeof(hode_mckl_desc));
  if (d->rows <= 0 || d->n_samples <= 0) return hode::fail(HODE_E_SIZE, "rows / n_samples must be positive");
  if (!(d->rate > 0)) return hode::fail(HODE_E_SIZE, "rate must be positive");
  if (!d->mu || !d->log_var || !d->noise || !d->kl) return hode::fail(HODE_E_NULL, "mu / log_var / noise / kl must be non-NULL");
  hode::McKlArgs a{};
  a.mu = d->mu; a.log_var = d->log_var; a.noise = d->noise; a.kl = d->kl; a.grad_mu = d->grad_mu; a.grad_lv = d->grad_log_var;
  a.rows = d->rows; a.S = d->n_samples; a.rate = d->rate; a.eps_clamp = d->clamp_value;
  const long long blocks = (d->rows + 255) / 256;
  if (blocks > 0x7fffffffLL) return hode::fail(HODE_E_SIZE, "rows too large");
  hipLaunchKernelGGL(hode::mc_kl_exp_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return hode::hip_fail(hipGetLastError(), "mc_kl launch");
}
