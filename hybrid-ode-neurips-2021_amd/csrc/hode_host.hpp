// Host-side declarations shared by the translation units of libhode.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hode.h"

namespace hode {

// records a thread-local message (returned by hode_last_error_string) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// kernel arguments of the fixed-grid Roche kernels (device pointers + sizes), see hode_rk_kernels.hpp
struct RkArgs {
  const float* __restrict__ t;
  const float* __restrict__ y0;
  const float* __restrict__ dosage;
  const float* __restrict__ dose_times;
  const float* __restrict__ theta;
  const float* __restrict__ w1;
  const float* __restrict__ b1;
  float* __restrict__ h;
  const float* __restrict__ grad_h;
  float* __restrict__ grad_y0;
  float* __restrict__ partials;  // [n_waves][P] per-wave parameter-gradient partials (backward)
  int* __restrict__ status;
  int B, T, K, perturb, ppw;
};

struct RkLaunch {
  int method;   // HODE_METHOD_*
  int lpp;      // lanes per patient: 1 or 4
  bool ablate, bwd, need_th;
};

// one entry per compiled latent dimension (hode_rk_dim.hip is compiled once per -DHODE_DIM=<D>)
int rk_dispatch_d4(const RkLaunch&, const RkArgs&, hipStream_t);
int rk_dispatch_d6(const RkLaunch&, const RkArgs&, hipStream_t);
int rk_dispatch_d8(const RkLaunch&, const RkArgs&, hipStream_t);
int rk_dispatch_d12(const RkLaunch&, const RkArgs&, hipStream_t);
int rk_dispatch_d20(const RkLaunch&, const RkArgs&, hipStream_t);

// dopri5: one entry per compiled latent dimension (hode_dopri5_dim.hip, -DHODE_DIM=<D>)
struct DpArgs;
struct DpLaunch;
int dp_dispatch_d4(const DpLaunch&, const DpArgs&, hipStream_t);
int dp_dispatch_d6(const DpLaunch&, const DpArgs&, hipStream_t);
int dp_dispatch_d8(const DpLaunch&, const DpArgs&, hipStream_t);
int dp_dispatch_d12(const DpLaunch&, const DpArgs&, hipStream_t);

// neural rhs (hode_neural.hip)
size_t neural_workspace_bytes(const hode_solve_desc* d, bool bwd);
int neural_rk(const hode_solve_desc* d, bool bwd, hipStream_t s);
struct NeuralArgs;
int launch_neural_mf(const hode_solve_desc* d, const NeuralArgs& a, bool bwd, hipStream_t s);  // hode_neural_mf.hip
size_t neural_mf_partial_bytes(const hode_solve_desc* d);

// adaptive solve of the neural rhs on the matrix cores (hode_neural_dopri5.hip)
size_t neural_dopri5_workspace_bytes(const hode_solve_desc* d);
int neural_dopri5_tape_offsets(const hode_solve_desc* d, size_t* out5);
int neural_dopri5(const hode_solve_desc* d, bool bwd, hipStream_t s);

// real-data rhs (hode_real.hip)
size_t real_workspace_bytes(const hode_solve_desc* d, bool bwd);
int real_rk(const hode_solve_desc* d, bool bwd, hipStream_t s);
struct RealArgs;
bool real_mf_supported(const hode_solve_desc* d);                                              // hode_real_mf.hip
int launch_real_mf(const hode_solve_desc* d, const RealArgs& a, bool bwd, hipStream_t s);
size_t real_mf_partial_bytes(const hode_solve_desc* d);

// MFMA-layout Roche kernels (hode_rk_mf.hip)
bool mf_supported(const hode_solve_desc* d);
size_t mf_workspace_bytes(const hode_solve_desc* d);
int mf_rk(const hode_solve_desc* d, bool bwd, hipStream_t s);

// wave-specialised split layout (hode_rk_split.hip)
bool split_supported(const hode_solve_desc* d);
int split_rk_fwd(const hode_solve_desc* d, hipStream_t s);
size_t split_workspace_bytes(const hode_solve_desc* d);
int split_rk_bwd(const hode_solve_desc* d, hipStream_t s);

// shared host helpers (hode_api.hip)
int hip_fail(hipError_t e, const char* what);
int patients_per_wave(int B, int lpp);
int n_waves_for(int B, int lpp);
int choose_lpp(const hode_solve_desc* d);
int n_partials(const hode_solve_desc* d);
// out += fixed-order sum over waves of partials[w][0..P): [w1 | b1 | theta]
int launch_fold_partials(const float* partials, int n_waves, int P, int n_w, int n_b, float* gw, float* gb, float* gth,
                         int need_th, hipStream_t s);

}  // namespace hode
