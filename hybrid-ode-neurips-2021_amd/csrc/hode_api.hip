// C-ABI entry points of libhode.so (declared in include/hode.h): argument checks, variant selection, launches.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "hode_host.hpp"
#include "hode_roche.hpp"

namespace hode {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

// out[j] += sum over waves of partials[w][j].  One wave per output element: lane l adds rows l, l+64, ... in order,
// then a fixed-shape butterfly folds the 64 lane sums -- the summation tree depends only on (n_waves), so the
// result is bit-reproducible run to run (no float atomics).
__global__ __launch_bounds__(64) void fold_partials_kernel(const float* __restrict__ partials, int n_waves, int P,
                                                           int n_w, int n_b, float* __restrict__ gw,
                                                           float* __restrict__ gb, float* __restrict__ gth, int need_th) {
  const int j = blockIdx.x;
  const int lane = threadIdx.x;
  float s = 0.f;
  for (int w = lane; w < n_waves; w += 64) s += partials[(size_t)w * P + j];
  s = wave_sum(s);
  if (lane != 0) return;
  if (j < n_w) {
    if (gw) gw[j] += s;
  } else if (j < n_w + n_b) {
    if (gb) gb[j - n_w] += s;
  } else if (need_th && gth) {
    gth[j - n_w - n_b] += s;
  }
}

// patients per wave: as many waves as it takes to put one on (almost) every SIMD, then whole rounds of 1024
int patients_per_wave(int B, int lpp) {
  const int cap = 64 / lpp;
  if (const char* env = getenv("HODE_PPW")) {  // tuning / test override
    const int v = atoi(env);
    if (v >= 1 && v <= cap) return v;
  }
  const long long simds = 1024;
  const long long rounds = (B + simds * cap - 1) / (simds * cap);
  long long ppw = (B + simds * rounds - 1) / (simds * rounds);
  if (ppw < 1) ppw = 1;
  if (ppw > cap) ppw = cap;
  return (int)ppw;
}
int n_waves_for(int B, int lpp) {
  const int ppw = patients_per_wave(B, lpp);
  return (B + ppw - 1) / ppw;
}

// LPP = 4 (a patient per DPP quad, 16 patients per wave) fills the chip at the 10k-patient shape; LPP = 1 has
// the lowest total instruction count and wins once every SIMD has >= 2 waves without splitting patients
// (256 CUs x 4 SIMDs x 2 waves x 64 lanes = 131072 patients).
int choose_lpp(const hode_solve_desc* d) {
  const int M = d->latent_dim - 4;
  const bool can4 = M > 0 && M % 4 == 0;
  if (d->lanes_per_patient == 1) return 1;
  if (d->lanes_per_patient == 4) return can4 ? 4 : 1;
  if (!can4) return 1;
  return d->batch >= 131072 ? 1 : 4;
}

int n_partials(const hode_solve_desc* d) {
  const int M = d->latent_dim - 4;
  return M * d->latent_dim + M + kNTheta;
}


int launch_fold_partials(const float* partials, int n_waves, int P, int n_w, int n_b, float* gw, float* gb, float* gth,
                         int need_th, hipStream_t s) {
  hipLaunchKernelGGL(fold_partials_kernel, dim3(P), dim3(64), 0, s, partials, n_waves, P, n_w, n_b, gw, gb, gth, need_th);
  return hip_fail(hipGetLastError(), "fold_partials launch");
}

}  // namespace hode

namespace {

using hode::choose_lpp;
using hode::n_partials;
using hode::n_waves_for;

using hode::RkArgs;
using hode::RkLaunch;

RkArgs make_args(const hode_solve_desc* d) {
  RkArgs a{};
  a.t = d->t; a.y0 = d->y0; a.dosage = d->dosage; a.dose_times = d->dose_times; a.theta = d->theta;
  a.w1 = d->w1; a.b1 = d->b1; a.h = d->h; a.grad_h = d->grad_h; a.grad_y0 = d->grad_y0;
  a.partials = (float*)d->workspace; a.status = d->status;
  a.B = d->batch; a.T = d->n_times; a.K = d->n_dose; a.perturb = d->perturb;
  a.ppw = hode::patients_per_wave(d->batch, choose_lpp(d));
  return a;
}

// lanes_per_patient == 16 (or HODE_RK_LAYOUT=m) selects the MFMA layout (hode_rk_mf.hip) where it exists for the dimension
bool use_mf(const hode_solve_desc* d) {
  if (!hode::mf_supported(d)) return false;
  if (d->lanes_per_patient == 16) return true;
  if (d->lanes_per_patient != 0) return false;
  if (const char* env = getenv("HODE_RK_LAYOUT")) return env[0] == 'm';
  // measured at 10 000 patients (T=100, D=12, rk4): MFMA layout fwd 118 us / bwd 313 us vs quad layout 104 / 339 -- a
  // wash (the 3 dependent 16x16x4 MFMAs + hazard nops cost as much latency as the 24 fmas they replace), so the quad
  // layout stays the default and the MFMA layout is opt-in
  return false;
}

// lanes_per_patient == 48 / HODE_RK_LAYOUT=s select the wave-specialised split layout (hode_rk_split.hip); it is also the
// default for the dimensions it is built for (HODE_RK_LAYOUT=q forces the quad layout)
bool use_split(const hode_solve_desc* d, bool bwd) {
  if (!hode::split_supported(d)) return false;
  if (bwd && d->n_times < 2) return false;
  if (d->lanes_per_patient == 48) return true;
  if (d->lanes_per_patient != 0) return false;
  if (const char* env = getenv("HODE_RK_LAYOUT")) return env[0] == 's';
  // default wherever it exists: measured at 10 000 patients (T=100, D=12, rk4) fwd 70 us / bwd 161 us vs 104 / 339 us
  // for the quad layout; it also issues fewer wave-instructions per patient (4.1 vs 7.3 per rhs), so it keeps winning
  // once every SIMD is busy
  return true;
}

int dispatch_dim(const hode_solve_desc* d, bool bwd, hipStream_t s) {
  if (use_split(d, bwd)) return bwd ? hode::split_rk_bwd(d, s) : hode::split_rk_fwd(d, s);
  if (use_mf(d)) return hode::mf_rk(d, bwd, s);
  RkLaunch L;
  L.method = d->method;
  L.lpp = choose_lpp(d);
  L.ablate = d->rhs_kind == HODE_RHS_ROCHE_ABLATE;
  L.bwd = bwd;
  L.need_th = d->need_theta_grad != 0;
  const RkArgs a = make_args(d);
  switch (d->latent_dim) {
    case 4: return hode::rk_dispatch_d4(L, a, s);
    case 6: return hode::rk_dispatch_d6(L, a, s);
    case 8: return hode::rk_dispatch_d8(L, a, s);
    case 12: return hode::rk_dispatch_d12(L, a, s);
    case 20: return hode::rk_dispatch_d20(L, a, s);
  }
  return hode::fail(HODE_E_UNSUPPORTED, "latent_dim %d has no compiled kernel (have 4, 6, 8, 12, 20)", d->latent_dim);
}

int check_rk(const hode_solve_desc* d, bool bwd) {
  if (!d) return hode::fail(HODE_E_NULL, "descriptor is NULL");
  if (d->struct_size != sizeof(hode_solve_desc))
    return hode::fail(HODE_E_SIZE, "struct_size %u != %zu (ABI mismatch)", d->struct_size, sizeof(hode_solve_desc));
  if (d->rhs_kind != HODE_RHS_ROCHE && d->rhs_kind != HODE_RHS_ROCHE_ABLATE)
    return hode::fail(HODE_E_UNSUPPORTED, "rhs_kind %d is not handled by the fixed-grid Roche kernels", d->rhs_kind);
  if (d->method < HODE_METHOD_EULER || d->method > HODE_METHOD_RK4_38)
    return hode::fail(HODE_E_UNSUPPORTED, "unknown fixed-grid method %d", d->method);
  if (d->batch <= 0 || d->n_times <= 0 || d->latent_dim < 4 || d->n_dose < 0)
    return hode::fail(HODE_E_SIZE, "bad sizes: batch=%d n_times=%d latent_dim=%d n_dose=%d", d->batch, d->n_times,
                      d->latent_dim, d->n_dose);
  if (!d->t || !d->y0 || !d->dosage || !d->theta || !d->h || (d->n_dose > 0 && !d->dose_times))
    return hode::fail(HODE_E_NULL, "t / y0 / dosage / dose_times / theta / h must be non-NULL");
  if (d->latent_dim > 4 && (!d->w1 || !d->b1)) return hode::fail(HODE_E_NULL, "w1 / b1 required when latent_dim > 4");
  if (bwd && (!d->grad_h || !d->grad_y0)) return hode::fail(HODE_E_NULL, "grad_h / grad_y0 required by the backward");
  if (d->latent_dim % 4 == 0) {
    uintptr_t m = (uintptr_t)d->y0 | (uintptr_t)d->h;
    if (bwd) m |= (uintptr_t)d->grad_h | (uintptr_t)d->grad_y0;
    if (m & 15) return hode::fail(HODE_E_ALIGN, "y0 / h / grad_h / grad_y0 must be 16-byte aligned");
  }
  return 0;
}

}  // namespace

extern "C" size_t hode_dopri5_workspace_bytes(const hode_solve_desc* d);  // hode_dopri5.hip

extern "C" int hode_version(void) { return HODE_ABI_VERSION; }

extern "C" const char* hode_last_error_string(void) { return hode::g_err; }

extern "C" size_t hode_workspace_bytes(const hode_solve_desc* d, int which) {
  if (!d || d->struct_size != sizeof(hode_solve_desc)) return 0;
  if (d->rhs_kind == HODE_RHS_NEURAL && (which == HODE_WS_RK_FWD || which == HODE_WS_RK_BWD))
    return hode::neural_workspace_bytes(d, which == HODE_WS_RK_BWD);
  if (d->rhs_kind == HODE_RHS_ROCHE_REAL && (which == HODE_WS_RK_FWD || which == HODE_WS_RK_BWD))
    return hode::real_workspace_bytes(d, which == HODE_WS_RK_BWD);
  switch (which) {
    case HODE_WS_RK_FWD:  // only the split layout's tape (HODE_FLAG_TAPE): the buffer hode_rk_bwd will be handed again
      return ((d->flags & HODE_FLAG_TAPE) && use_split(d, false) && use_split(d, true)) ? hode::split_workspace_bytes(d) : 0;
    case HODE_WS_RK_BWD:
      if (use_split(d, true)) return hode::split_workspace_bytes(d);
      if (use_mf(d)) return hode::mf_workspace_bytes(d);
      return (size_t)n_waves_for(d->batch, choose_lpp(d)) * n_partials(d) * sizeof(float);
    case HODE_WS_DOPRI5_FWD:
    case HODE_WS_DOPRI5_BWD: return hode_dopri5_workspace_bytes(d);
    default: return 0;
  }
}

extern "C" int hode_rk_fwd(const hode_solve_desc* d, void* stream) {
  if (d && d->struct_size == sizeof(hode_solve_desc) && d->rhs_kind == HODE_RHS_NEURAL)
    return hode::neural_rk(d, false, (hipStream_t)stream);
  if (d && d->struct_size == sizeof(hode_solve_desc) && d->rhs_kind == HODE_RHS_ROCHE_REAL)
    return hode::real_rk(d, false, (hipStream_t)stream);
  if (int e = check_rk(d, false)) return e;
  const size_t need = hode_workspace_bytes(d, HODE_WS_RK_FWD);
  if (need && (!d->workspace || d->workspace_bytes < need))
    return hode::fail(HODE_E_WORKSPACE, "workspace %zu B < required %zu B (HODE_FLAG_TAPE)", d->workspace_bytes, need);
  return dispatch_dim(d, false, (hipStream_t)stream);
}

extern "C" int hode_rk_bwd(const hode_solve_desc* d, void* stream) {
  if (d && d->struct_size == sizeof(hode_solve_desc) &&
      (d->rhs_kind == HODE_RHS_NEURAL || d->rhs_kind == HODE_RHS_ROCHE_REAL)) {
    if (d->flags & HODE_FLAG_OVERWRITE_GRADS)
      return hode::fail(HODE_E_UNSUPPORTED, "HODE_FLAG_OVERWRITE_GRADS is only implemented for the ROCHE rhs kinds");
    return d->rhs_kind == HODE_RHS_NEURAL ? hode::neural_rk(d, true, (hipStream_t)stream)
                                          : hode::real_rk(d, true, (hipStream_t)stream);
  }
  if (int e = check_rk(d, true)) return e;
  const size_t need = hode_workspace_bytes(d, HODE_WS_RK_BWD);
  if (!d->workspace || d->workspace_bytes < need)
    return hode::fail(HODE_E_WORKSPACE, "workspace %zu B < required %zu B", d->workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  if ((d->flags & HODE_FLAG_OVERWRITE_GRADS) && !use_split(d, true)) {
    // the split layout's fold stores directly; the other layouts accumulate, so clear the outputs first
    const size_t M = d->latent_dim - 4;
    if (d->grad_w1) if (int e = hode::hip_fail(hipMemsetAsync(d->grad_w1, 0, M * d->latent_dim * sizeof(float), s), "grad_w1 clear")) return e;
    if (d->grad_b1) if (int e = hode::hip_fail(hipMemsetAsync(d->grad_b1, 0, M * sizeof(float), s), "grad_b1 clear")) return e;
    if (d->grad_theta) if (int e = hode::hip_fail(hipMemsetAsync(d->grad_theta, 0, hode::kNTheta * sizeof(float), s), "grad_theta clear")) return e;
  }
  if (int e = dispatch_dim(d, true, s)) return e;
  if (use_split(d, true) || use_mf(d)) return 0;  // these layouts fold their own partials
  if (d->flags & HODE_FLAG_SKIP_FOLD) return 0;
  const int M = d->latent_dim - 4;
  const int P = n_partials(d);
  const int nw = n_waves_for(d->batch, choose_lpp(d));
  return hode::launch_fold_partials((const float*)d->workspace, nw, P, M * d->latent_dim, M, d->grad_w1, d->grad_b1,
                                    d->grad_theta, d->need_theta_grad, s);
}
