// Fixed-grid solve of the pure neural latent ODE  dy/dt = tanh(W2 tanh(W1 [y, Dose(t)] + b1) + b2)  and its discrete
// adjoint, gfx950.  Replaces torchdiffeq.odeint(NeuralODE, ...) (reference model.py:969-1026, call site :1116) for
// method in {euler, midpoint, rk4}; CPU restatement: oracle/rhs.py::NeuralRHS + oracle/solvers.py.
//
// First implementation (correctness + coverage; the Roche kernels carry the benchmark): one patient per lane, the
// (10D x (D+1)) and (D x 10D) weight matrices are read through wave-uniform addresses (scalar loads / broadcasts).
// The parameter gradient of an MLP is an outer product summed over patients -- a GEMM -- so the backward does not
// accumulate it in registers: for every (step, stage) it writes the four operands patient-minor
//     A1T[inst][10D][B] hidden activations      U1T[inst][10D][B] hidden pre-activation cotangents
//     YET[inst][D+1][B] layer-1 input [y, Dose]  U2T[inst][D][B]   output pre-activation cotangents
// into the caller's workspace, and the host contracts them with batched BLAS GEMMs (hode/neural.py):
//     grad_W1 = sum_inst U1T YET^T, grad_b1 = sum U1T, grad_W2 = sum_inst U2T A1T^T, grad_b2 = sum U2T.
// Dose(t) = dosage * #{k : tau_k == t} is an impulse that only exists when a stage time hits a dose time exactly
// (reference model.py:1017), so stage times are formed with non-contracted fp32 ops like the reference's tensors.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"
#include "hode_neural_args.hpp"

namespace hode {

__global__ void transpose_w2_kernel(const float* __restrict__ w2, float* __restrict__ w2t, int D, int HD) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D * HD) {
    const int c = i / HD, n = i - c * HD;
    w2t[n * D + c] = w2[i];
  }
}

// k = f(ye), ye = [y, dose].  If A1 != nullptr the hidden activations are written to A1[n * strideB] (tape).
template <int D>
HODE_DEV void neural_rhs(const NeuralArgs& a, const float (&ye)[D + 1], float (&k)[D], float* __restrict__ A1, size_t strideB) {
  constexpr int HD = 10 * D;
  float z2[D];
#pragma unroll
  for (int c = 0; c < D; ++c) z2[c] = a.b2[c];
  for (int n = 0; n < HD; ++n) {
    const float* wr = a.w1 + (size_t)n * (D + 1);
    float z = a.b1[n];
#pragma unroll
    for (int i = 0; i <= D; ++i) z = __builtin_fmaf(wr[i], ye[i], z);
    const float an = tanh_f32(z);
    if (A1) A1[(size_t)n * strideB] = an;
    const float* wc = a.w2t + (size_t)n * D;
#pragma unroll
    for (int c = 0; c < D; ++c) z2[c] = __builtin_fmaf(wc[c], an, z2[c]);
  }
#pragma unroll
  for (int c = 0; c < D; ++c) k[c] = tanh_f32(z2[c]);
}

template <int D, int METHOD>
__global__ __launch_bounds__(64) void neural_fwd_kernel(NeuralArgs a) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.B) return;
  const float dosage = a.dosage[p];
  float y[D];
#pragma unroll
  for (int i = 0; i < D; ++i) y[i] = a.y0[(size_t)p * D + i];
  const size_t row = (size_t)a.B * D;
  float* hp = a.h + (size_t)p * D;
#pragma unroll
  for (int i = 0; i < D; ++i) hp[i] = y[i];
  for (int n = 0; n + 1 < a.T; ++n) {
    const NStageTimes st(a.t, n, a.perturb, METHOD);
    const float dt = st.dt;
    float ye[D + 1], k1[D];
#pragma unroll
    for (int i = 0; i < D; ++i) ye[i] = y[i];
    ye[D] = neural_dose(a, p, dosage, st.t_first);
    neural_rhs<D>(a, ye, k1, nullptr, 0);
    if constexpr (METHOD == HODE_METHOD_EULER) {
#pragma unroll
      for (int i = 0; i < D; ++i) y[i] = __builtin_fmaf(dt, k1[i], y[i]);
    } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
      float k2[D];
      const float half = 0.5f * dt;
#pragma unroll
      for (int i = 0; i < D; ++i) ye[i] = __builtin_fmaf(k1[i], half, y[i]);
      ye[D] = neural_dose(a, p, dosage, st.ta);
      neural_rhs<D>(a, ye, k2, nullptr, 0);
#pragma unroll
      for (int i = 0; i < D; ++i) y[i] = __builtin_fmaf(dt, k2[i], y[i]);
    } else {
      float k2[D], k3[D], k4[D];
#pragma unroll
      for (int i = 0; i < D; ++i) ye[i] = __builtin_fmaf(dt * k1[i], kThird, y[i]);
      ye[D] = neural_dose(a, p, dosage, st.ta);
      neural_rhs<D>(a, ye, k2, nullptr, 0);
#pragma unroll
      for (int i = 0; i < D; ++i) ye[i] = __builtin_fmaf(dt, __builtin_fmaf(-k1[i], kThird, k2[i]), y[i]);
      ye[D] = neural_dose(a, p, dosage, st.tb);
      neural_rhs<D>(a, ye, k3, nullptr, 0);
#pragma unroll
      for (int i = 0; i < D; ++i) ye[i] = __builtin_fmaf(dt, (k1[i] - k2[i]) + k3[i], y[i]);
      ye[D] = neural_dose(a, p, dosage, st.t_last);
      neural_rhs<D>(a, ye, k4, nullptr, 0);
      const float w = dt * 0.125f;
#pragma unroll
      for (int i = 0; i < D; ++i) y[i] = __builtin_fmaf((k1[i] + 3.0f * (k2[i] + k3[i])) + k4[i], w, y[i]);
    }
    hp += row;
#pragma unroll
    for (int i = 0; i < D; ++i) hp[i] = y[i];
  }
}

// VJP of the rhs at the stage whose inputs/activations were taped as instance `inst`:
//   g = cotangent of k.  Writes U2T, U1T; returns a = (df/dy)^T g.
template <int D>
HODE_DEV void neural_vjp(const NeuralArgs& a, int p, size_t inst, const float (&kout)[D], const float (&g)[D], float (&av)[D]) {
  constexpr int HD = 10 * D;
  const size_t B = a.B;
  float u2[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    u2[c] = g[c] * __builtin_fmaf(-kout[c], kout[c], 1.0f);
    a.u2t[(inst * D + c) * B + p] = u2[c];
  }
  float acc[D + 1];
#pragma unroll
  for (int i = 0; i <= D; ++i) acc[i] = 0.f;
  const float* A1 = a.a1t + inst * HD * B + p;
  float* U1 = a.u1t + inst * HD * B + p;
  for (int n = 0; n < HD; ++n) {
    const float an = A1[(size_t)n * B];
    const float* wc = a.w2t + (size_t)n * D;
    float da = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) da = __builtin_fmaf(wc[c], u2[c], da);
    const float u1 = da * __builtin_fmaf(-an, an, 1.0f);
    U1[(size_t)n * B] = u1;
    const float* wr = a.w1 + (size_t)n * (D + 1);
#pragma unroll
    for (int i = 0; i <= D; ++i) acc[i] = __builtin_fmaf(wr[i], u1, acc[i]);
  }
#pragma unroll
  for (int i = 0; i < D; ++i) av[i] = acc[i];
}

template <int D, int METHOD>
__global__ __launch_bounds__(64) void neural_bwd_kernel(NeuralArgs a) {
  constexpr int HD = 10 * D;
  constexpr int NS = METHOD == HODE_METHOD_EULER ? 1 : (METHOD == HODE_METHOD_MIDPOINT ? 2 : 4);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.B) return;
  const size_t B = a.B;
  const float dosage = a.dosage[p];
  const size_t row = B * D;
  float lam[D];
#pragma unroll
  for (int i = 0; i < D; ++i) lam[i] = a.grad_h[(size_t)(a.T - 1) * row + (size_t)p * D + i];

  auto tape_in = [&](size_t inst, const float (&ye)[D + 1]) {
#pragma unroll
    for (int i = 0; i <= D; ++i) a.yet[(inst * (D + 1) + i) * B + p] = ye[i];
  };

  for (int n = a.T - 2; n >= 0; --n) {
    const NStageTimes st(a.t, n, a.perturb, METHOD);
    const float dt = st.dt;
    const size_t i0 = (size_t)n * NS;
    float y[D];
#pragma unroll
    for (int i = 0; i < D; ++i) y[i] = a.h[(size_t)n * row + (size_t)p * D + i];
    float ye[D + 1], k1[D], av[D], g[D];
#pragma unroll
    for (int i = 0; i < D; ++i) ye[i] = y[i];
    ye[D] = neural_dose(a, p, dosage, st.t_first);
    tape_in(i0, ye);
    neural_rhs<D>(a, ye, k1, a.a1t + i0 * HD * B + p, B);
    if constexpr (METHOD == HODE_METHOD_EULER) {
#pragma unroll
      for (int i = 0; i < D; ++i) g[i] = dt * lam[i];
      neural_vjp<D>(a, p, i0, k1, g, av);
#pragma unroll
      for (int i = 0; i < D; ++i) lam[i] += av[i];
    } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
      float k2[D];
      const float half = 0.5f * dt;
#pragma unroll
      for (int i = 0; i < D; ++i) ye[i] = __builtin_fmaf(k1[i], half, y[i]);
      ye[D] = neural_dose(a, p, dosage, st.ta);
      tape_in(i0 + 1, ye);
      neural_rhs<D>(a, ye, k2, a.a1t + (i0 + 1) * HD * B + p, B);
#pragma unroll
      for (int i = 0; i < D; ++i) g[i] = dt * lam[i];
      neural_vjp<D>(a, p, i0 + 1, k2, g, av);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        lam[i] += av[i];
        g[i] = half * av[i];
      }
      neural_vjp<D>(a, p, i0, k1, g, av);
#pragma unroll
      for (int i = 0; i < D; ++i) lam[i] += av[i];
    } else {
      float k2[D], k3[D], k4[D];
#pragma unroll
      for (int i = 0; i < D; ++i) ye[i] = __builtin_fmaf(dt * k1[i], kThird, y[i]);
      ye[D] = neural_dose(a, p, dosage, st.ta);
      tape_in(i0 + 1, ye);
      neural_rhs<D>(a, ye, k2, a.a1t + (i0 + 1) * HD * B + p, B);
#pragma unroll
      for (int i = 0; i < D; ++i) ye[i] = __builtin_fmaf(dt, __builtin_fmaf(-k1[i], kThird, k2[i]), y[i]);
      ye[D] = neural_dose(a, p, dosage, st.tb);
      tape_in(i0 + 2, ye);
      neural_rhs<D>(a, ye, k3, a.a1t + (i0 + 2) * HD * B + p, B);
#pragma unroll
      for (int i = 0; i < D; ++i) ye[i] = __builtin_fmaf(dt, (k1[i] - k2[i]) + k3[i], y[i]);
      ye[D] = neural_dose(a, p, dosage, st.t_last);
      tape_in(i0 + 3, ye);
      neural_rhs<D>(a, ye, k4, a.a1t + (i0 + 3) * HD * B + p, B);

      const float w1 = dt * 0.125f, w3 = dt * 0.375f;
      float g1[D], g2[D];
#pragma unroll
      for (int i = 0; i < D; ++i) g[i] = w1 * lam[i];
      neural_vjp<D>(a, p, i0 + 3, k4, g, av);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const float da = dt * av[i];
        g1[i] = __builtin_fmaf(w1, lam[i], da);
        g2[i] = __builtin_fmaf(w3, lam[i], -da);
        g[i] = __builtin_fmaf(w3, lam[i], da);
        lam[i] += av[i];
      }
      neural_vjp<D>(a, p, i0 + 2, k3, g, av);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const float da = dt * av[i];
        g2[i] += da;
        g1[i] = __builtin_fmaf(-kThird, da, g1[i]);
        lam[i] += av[i];
      }
      neural_vjp<D>(a, p, i0 + 1, k2, g2, av);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        g1[i] = __builtin_fmaf(kThird, dt * av[i], g1[i]);
        lam[i] += av[i];
      }
      neural_vjp<D>(a, p, i0, k1, g1, av);
#pragma unroll
      for (int i = 0; i < D; ++i) lam[i] += av[i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) lam[i] += a.grad_h[(size_t)n * row + (size_t)p * D + i];
  }
#pragma unroll
  for (int i = 0; i < D; ++i) a.grad_y0[(size_t)p * D + i] = lam[i];
}

}  // namespace hode

// ====================================================================================================== host
namespace {

using hode::NeuralArgs;

size_t al256(size_t x) { return (x + 255) / 256 * 256; }

int n_stages(int method) { return method == HODE_METHOD_EULER ? 1 : (method == HODE_METHOD_MIDPOINT ? 2 : 4); }

struct NeuralLayout {
  size_t w2t, a1t, u1t, yet, u2t, total;
};

// the matrix-core backward accumulates the weight gradients on chip when the caller hands it the accumulators (grad_w1 set);
// HODE_NEURAL_LAYOUT=t (lane-per-patient kernels) and callers without accumulators get the operand tapes
bool neural_onchip(const hode_solve_desc* d) {
  const char* env = getenv("HODE_NEURAL_LAYOUT");
  return !(env && env[0] == 't') && d->grad_w1 != nullptr;
}

NeuralLayout neural_layout(const hode_solve_desc* d, bool bwd) {
  const size_t D = d->latent_dim, HD = 10 * D, B = d->batch;
  const size_t inst = (size_t)(d->n_times > 0 ? d->n_times - 1 : 0) * n_stages(d->method);
  NeuralLayout L;
  size_t off = 0;
  L.w2t = off; off = al256(off + HD * D * 4);
  if (bwd && neural_onchip(d)) {  // one block of gradient partials per wave in the a1t slot, no tapes
    L.a1t = off; off = al256(off + hode::neural_mf_partial_bytes(d));
    L.u1t = L.yet = L.u2t = off;
    L.total = off;
    return L;
  }
  L.a1t = off; if (bwd) off = al256(off + inst * HD * B * 4);
  L.u1t = off; if (bwd) off = al256(off + inst * HD * B * 4);
  L.yet = off; if (bwd) off = al256(off + inst * (D + 1) * B * 4);
  L.u2t = off; if (bwd) off = al256(off + inst * D * B * 4);
  L.total = off;
  return L;
}

int check_neural(const hode_solve_desc* d, bool bwd) {
  if (d->method < HODE_METHOD_EULER || d->method > HODE_METHOD_RK4_38)
    return hode::fail(HODE_E_UNSUPPORTED, "neural rhs: unknown fixed-grid method %d", d->method);
  if (d->batch <= 0 || d->n_times <= 0 || d->n_dose < 0) return hode::fail(HODE_E_SIZE, "bad sizes");
  if (d->latent_dim < 4 || d->latent_dim > 14 || (d->latent_dim & 1))
    return hode::fail(HODE_E_UNSUPPORTED, "neural rhs: latent_dim %d has no compiled kernel (have 4, 6, 8, 10, 12, 14)", d->latent_dim);
  if (d->hidden_dim != 10 * d->latent_dim)
    return hode::fail(HODE_E_SIZE, "neural rhs: hidden_dim %d != 10 * latent_dim (reference model.py:991-996)", d->hidden_dim);
  if (!d->t || !d->y0 || !d->dosage || !d->h || !d->w1 || !d->b1 || !d->w2 || !d->b2 || (d->n_dose > 0 && !d->dose_times))
    return hode::fail(HODE_E_NULL, "t / y0 / dosage / dose_times / h / w1 / b1 / w2 / b2 must be non-NULL");
  if (bwd && (!d->grad_h || !d->grad_y0)) return hode::fail(HODE_E_NULL, "grad_h / grad_y0 required by the backward");
  if (bwd && d->grad_w1 && (!d->grad_b1 || !d->grad_w2 || !d->grad_b2))
    return hode::fail(HODE_E_NULL, "grad_w1 given: grad_b1 / grad_w2 / grad_b2 are required too (on-chip weight gradients)");
  const NeuralLayout L = neural_layout(d, bwd);
  if (!d->workspace || d->workspace_bytes < L.total)
    return hode::fail(HODE_E_WORKSPACE, "workspace %zu B < required %zu B", d->workspace_bytes, L.total);
  return 0;
}

template <int D>
int launch_neural(const hode_solve_desc* d, const NeuralArgs& a, bool bwd, hipStream_t s) {
  const dim3 grid((d->batch + 63) / 64), block(64);
#define HODE_NEURAL_LAUNCH(M)                                                                     \
  if (bwd) hipLaunchKernelGGL((hode::neural_bwd_kernel<D, M>), grid, block, 0, s, a);             \
  else hipLaunchKernelGGL((hode::neural_fwd_kernel<D, M>), grid, block, 0, s, a);
  switch (d->method) {
    case HODE_METHOD_EULER: HODE_NEURAL_LAUNCH(HODE_METHOD_EULER) break;
    case HODE_METHOD_MIDPOINT: HODE_NEURAL_LAUNCH(HODE_METHOD_MIDPOINT) break;
    default: HODE_NEURAL_LAUNCH(HODE_METHOD_RK4_38) break;
  }
  return hode::hip_fail(hipGetLastError(), "neural kernel launch");
}

}  // namespace

namespace hode {

size_t neural_workspace_bytes(const hode_solve_desc* d, bool bwd) { return neural_layout(d, bwd).total; }

// byte offsets of the four tapes inside the backward workspace (for the caller's GEMMs): a1t, u1t, yet, u2t
void neural_tape_offsets(const hode_solve_desc* d, size_t out[4]) {
  const NeuralLayout L = neural_layout(d, true);
  out[0] = L.a1t; out[1] = L.u1t; out[2] = L.yet; out[3] = L.u2t;
}

int neural_rk(const hode_solve_desc* d, bool bwd, hipStream_t s) {
  if (int e = check_neural(d, bwd)) return e;
  const NeuralLayout L = neural_layout(d, bwd);
  char* ws = (char*)d->workspace;
  NeuralArgs a{};
  a.t = d->t; a.y0 = d->y0; a.dosage = d->dosage; a.dose_times = d->dose_times; a.w1 = d->w1; a.b1 = d->b1; a.b2 = d->b2;
  a.w2t = (const float*)(ws + L.w2t);
  a.h = d->h; a.grad_h = d->grad_h; a.grad_y0 = d->grad_y0;
  a.a1t = (float*)(ws + L.a1t); a.u1t = (float*)(ws + L.u1t); a.yet = (float*)(ws + L.yet); a.u2t = (float*)(ws + L.u2t);
  a.B = d->batch; a.T = d->n_times; a.K = d->n_dose; a.perturb = d->perturb;
  a.w2 = d->w2;
  {
    // default: the matrix-core kernels (hode_neural_mf.hip); HODE_NEURAL_LAYOUT=t selects the one-patient-per-lane ones
    const char* env = getenv("HODE_NEURAL_LAYOUT");
    if (!(env && env[0] == 't')) return launch_neural_mf(d, a, bwd, s);
  }
  const int D = d->latent_dim, HD = 10 * D;
  if (D != 6 && D != 8 && D != 12)
    return hode::fail(HODE_E_UNSUPPORTED, "neural rhs: the lane-per-patient layout (HODE_NEURAL_LAYOUT=t) is compiled for 6, 8, 12 only");
  hipLaunchKernelGGL(hode::transpose_w2_kernel, dim3((D * HD + 255) / 256), dim3(256), 0, s, d->w2, (float*)(ws + L.w2t), D, HD);
  if (int e = hode::hip_fail(hipGetLastError(), "transpose_w2 launch")) return e;
  switch (D) {
    case 6: return launch_neural<6>(d, a, bwd, s);
    case 8: return launch_neural<8>(d, a, bwd, s);
    default: return launch_neural<12>(d, a, bwd, s);
  }
}

}  // namespace hode

extern "C" int hode_neural_tape_offsets(const hode_solve_desc* d, size_t* out4) {
  if (!d || !out4 || d->struct_size != sizeof(hode_solve_desc)) return hode::fail(HODE_E_NULL, "descriptor / out4");
  if (neural_onchip(d))
    return hode::fail(HODE_E_UNSUPPORTED, "grad_w1 is set: the backward accumulates the weight gradients itself, there are no tapes");
  hode::neural_tape_offsets(d, out4);
  return 0;
}
