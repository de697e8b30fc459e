// Fixed-grid solve of the real-data hybrid ODE (two small MLPs + GRU-ODE block) and its discrete adjoint, gfx950.
// Replaces torchdiffeq.odeint(RocheODEReal, ...) as called by DecoderReal.forward (reference model.py:837) with
// func = RocheODEReal.forward (model.py:613-645) and dose_at_time (model.py:653-657); real.sh runs it with
// method = midpoint and options["perturb"] = True.  CPU restatement: oracle/rhs.py::RocheRealRHS.
//
// State y = [x1, x2, x3, Dose2, h(M)], M = D - 4:
//   dx1 = tanh(w12 . tanh(W11 y[0:3] + b11) + b12)       dx2 = tanh(w22 . tanh(W21 y[0:2] + b21) + b22)
//   dx3 = y[1] k_immunity                                  dx4 = kel Dose(t) - kel2 y[3]
//   r = sigma(W_r h), z = sigma(W_z h), u = tanh(W_h (r*h)), dh = (1 - z)(u - h)
//   Dose(t) = sum_{k=1..Ta} a[k-1] 1[t >= k] exp(kel (k - t))     (every grid point is a potential dose at time k)
// First implementation (coverage; see hode_neural.hip for the same structure): one patient per lane, weights through
// wave-uniform loads, and the weight gradients -- sums of outer products over patients -- are NOT accumulated in
// registers: the backward tapes their GEMM operands patient-minor and the host contracts them (hode/real.py).
// The three scalars (k_immunity, kel, kel2) are accumulated per lane and folded in a fixed order.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"
#include "hode_real_args.hpp"

namespace hode {

// k = f(t, y).  With tp != nullptr the hidden activations are taped (tp points at this patient's column of the instance).
template <int D>
HODE_DEV void real_rhs(const RealArgs& a, const RealW& w, const RealTape& tl, float dose, const float (&y)[D], float (&k)[D],
                       float* __restrict__ tp) {
  constexpr int M = D - 4;
  const size_t B = a.B;
  const float kim = a.theta[0], kel = a.theta[1], kel2 = a.theta[2];
  float s1 = w.b12[0], s2 = w.b22[0];
  for (int j = 0; j < a.H; ++j) {
    const float z1 = __builtin_fmaf(w.W11[3 * j + 2], y[2], __builtin_fmaf(w.W11[3 * j + 1], y[1], __builtin_fmaf(w.W11[3 * j], y[0], w.b11[j])));
    const float a1 = tanh_f32(z1);
    s1 = __builtin_fmaf(w.w12[j], a1, s1);
    const float z2 = __builtin_fmaf(w.W21[2 * j + 1], y[1], __builtin_fmaf(w.W21[2 * j], y[0], w.b21[j]));
    const float a2 = tanh_f32(z2);
    s2 = __builtin_fmaf(w.w22[j], a2, s2);
    if (tp) {
      tp[(size_t)(tl.a11() + j) * B] = a1;
      tp[(size_t)(tl.a21() + j) * B] = a2;
    }
  }
  k[0] = tanh_f32(s1);
  k[1] = tanh_f32(s2);
  k[2] = y[1] * kim;
  k[3] = kel * dose - kel2 * y[3];
  if constexpr (M > 0) {
    float r[M], z[M], rh[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      float ar = 0.f, az = 0.f;
#pragma unroll
      for (int c = 0; c < M; ++c) {
        ar = __builtin_fmaf(w.Whr[i * M + c], y[4 + c], ar);
        az = __builtin_fmaf(w.Whz[i * M + c], y[4 + c], az);
      }
      r[i] = sigm(ar);
      z[i] = sigm(az);
    }
#pragma unroll
    for (int i = 0; i < M; ++i) rh[i] = r[i] * y[4 + i];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      float au = 0.f;
#pragma unroll
      for (int c = 0; c < M; ++c) au = __builtin_fmaf(w.Whh[i * M + c], rh[c], au);
      const float u = tanh_f32(au);
      k[4 + i] = (1.0f - z[i]) * (u - y[4 + i]);
    }
  }
}

// a = (df/dy)^T g at the taped stage `tp`; writes the cotangent tapes; accumulates the three scalar gradients
template <int D>
HODE_DEV void real_vjp(const RealArgs& a, const RealW& w, const RealTape& tl, DoseK dose, const float (&y)[D],
                       const float (&kout)[D], const float (&g)[D], float (&av)[D], float* __restrict__ tp, float (&dth)[3]) {
  constexpr int M = D - 4;
  const size_t B = a.B;
  const float kim = a.theta[0], kel = a.theta[1], kel2 = a.theta[2];
#pragma unroll
  for (int i = 0; i < D; ++i) av[i] = 0.f;
  tp[(size_t)(tl.y3() + 0) * B] = y[0];
  tp[(size_t)(tl.y3() + 1) * B] = y[1];
  tp[(size_t)(tl.y3() + 2) * B] = y[2];
  const float u12 = g[0] * __builtin_fmaf(-kout[0], kout[0], 1.0f);
  const float u22 = g[1] * __builtin_fmaf(-kout[1], kout[1], 1.0f);
  tp[(size_t)tl.u12() * B] = u12;
  tp[(size_t)tl.u22() * B] = u22;
  for (int j = 0; j < a.H; ++j) {
    const float a1 = tp[(size_t)(tl.a11() + j) * B];
    const float a2 = tp[(size_t)(tl.a21() + j) * B];
    const float u1 = w.w12[j] * u12 * __builtin_fmaf(-a1, a1, 1.0f);
    const float u2 = w.w22[j] * u22 * __builtin_fmaf(-a2, a2, 1.0f);
    tp[(size_t)(tl.u11() + j) * B] = u1;
    tp[(size_t)(tl.u21() + j) * B] = u2;
    av[0] = __builtin_fmaf(w.W11[3 * j], u1, __builtin_fmaf(w.W21[2 * j], u2, av[0]));
    av[1] = __builtin_fmaf(w.W11[3 * j + 1], u1, __builtin_fmaf(w.W21[2 * j + 1], u2, av[1]));
    av[2] = __builtin_fmaf(w.W11[3 * j + 2], u1, av[2]);
  }
  av[1] = __builtin_fmaf(g[2], kim, av[1]);
  av[3] = __builtin_fmaf(-kel2, g[3], av[3]);
  dth[0] = __builtin_fmaf(g[2], y[1], dth[0]);
  dth[1] = __builtin_fmaf(g[3], __builtin_fmaf(kel, dose.dk, dose.v), dth[1]);
  dth[2] = __builtin_fmaf(-g[3], y[3], dth[2]);
  if constexpr (M > 0) {
    // recompute the gate values (cheap next to the tape traffic they would cost)
    float r[M], z[M], rh[M], u[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      float ar = 0.f, az = 0.f;
#pragma unroll
      for (int c = 0; c < M; ++c) {
        ar = __builtin_fmaf(w.Whr[i * M + c], y[4 + c], ar);
        az = __builtin_fmaf(w.Whz[i * M + c], y[4 + c], az);
      }
      r[i] = sigm(ar);
      z[i] = sigm(az);
      rh[i] = r[i] * y[4 + i];
    }
    float uh[M], uz[M], drh[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      float au = 0.f;
#pragma unroll
      for (int c = 0; c < M; ++c) au = __builtin_fmaf(w.Whh[i * M + c], rh[c], au);
      u[i] = tanh_f32(au);
      const float gd = g[4 + i];
      uz[i] = -gd * (u[i] - y[4 + i]) * z[i] * (1.0f - z[i]);
      uh[i] = gd * (1.0f - z[i]) * __builtin_fmaf(-u[i], u[i], 1.0f);
      av[4 + i] = -gd * (1.0f - z[i]);
      drh[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int c = 0; c < M; ++c) drh[c] = __builtin_fmaf(w.Whh[i * M + c], uh[i], drh[c]);
    float ur[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      ur[i] = drh[i] * y[4 + i] * r[i] * (1.0f - r[i]);
      av[4 + i] = __builtin_fmaf(drh[i], r[i], av[4 + i]);
    }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int c = 0; c < M; ++c)
        av[4 + c] = __builtin_fmaf(w.Whr[i * M + c], ur[i], __builtin_fmaf(w.Whz[i * M + c], uz[i], av[4 + c]));
#pragma unroll
    for (int i = 0; i < M; ++i) {
      tp[(size_t)(tl.hh() + i) * B] = y[4 + i];
      tp[(size_t)(tl.rh() + i) * B] = rh[i];
      tp[(size_t)(tl.ur() + i) * B] = ur[i];
      tp[(size_t)(tl.uz() + i) * B] = uz[i];
      tp[(size_t)(tl.uh() + i) * B] = uh[i];
    }
  }
}

// forward (BWD = false) or discrete adjoint (BWD = true), one patient per lane.  NS stages per step.
template <int D, int METHOD, bool BWD>
__global__ __launch_bounds__(64) void real_kernel(RealArgs a) {
  constexpr int M = D - 4;
  constexpr int NS = METHOD == HODE_METHOD_EULER ? 1 : (METHOD == HODE_METHOD_MIDPOINT ? 2 : 4);
  constexpr float c13 = (float)(1.0 / 3.0);
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = gid < a.B;
  const int p = live ? gid : a.B - 1;
  const RealW w(a.wflat, a.H, M);
  const RealTape tl{a.H, M};
  const float kel = a.theta[1];
  const size_t row = (size_t)a.B * D;
  const size_t B = a.B;
  const size_t tstride = (size_t)tl.rows() * B;
  real_dose_table(a, p, kel);  // each thread reads back only its own column: no barrier needed

  if constexpr (!BWD) {
    float y[D];
#pragma unroll
    for (int i = 0; i < D; ++i) y[i] = a.y0[(size_t)p * D + i];
    if (live) {
#pragma unroll
      for (int i = 0; i < D; ++i) a.h[(size_t)p * D + i] = y[i];
    }
    for (int n = 0; n + 1 < a.T; ++n) {
      const RStageTimes st(a.t, n, a.perturb, METHOD);
      const float dt = st.dt;
      float k1[D], Y[D];
      real_rhs<D>(a, w, tl, real_dose(a, p, st.t_first, kel).v, y, k1, nullptr);
      if constexpr (METHOD == HODE_METHOD_EULER) {
#pragma unroll
        for (int i = 0; i < D; ++i) y[i] = __builtin_fmaf(dt, k1[i], y[i]);
      } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
        float k2[D];
        const float half = 0.5f * dt;
#pragma unroll
        for (int i = 0; i < D; ++i) Y[i] = __builtin_fmaf(k1[i], half, y[i]);
        real_rhs<D>(a, w, tl, real_dose(a, p, st.ta, kel).v, Y, k2, nullptr);
#pragma unroll
        for (int i = 0; i < D; ++i) y[i] = __builtin_fmaf(dt, k2[i], y[i]);
      } else {
        float k2[D], k3[D], k4[D];
#pragma unroll
        for (int i = 0; i < D; ++i) Y[i] = __builtin_fmaf(dt * k1[i], c13, y[i]);
        real_rhs<D>(a, w, tl, real_dose(a, p, st.ta, kel).v, Y, k2, nullptr);
#pragma unroll
        for (int i = 0; i < D; ++i) Y[i] = __builtin_fmaf(dt, __builtin_fmaf(-k1[i], c13, k2[i]), y[i]);
        real_rhs<D>(a, w, tl, real_dose(a, p, st.tb, kel).v, Y, k3, nullptr);
#pragma unroll
        for (int i = 0; i < D; ++i) Y[i] = __builtin_fmaf(dt, (k1[i] - k2[i]) + k3[i], y[i]);
        real_rhs<D>(a, w, tl, real_dose(a, p, st.t_last, kel).v, Y, k4, nullptr);
        const float ww = dt * 0.125f;
#pragma unroll
        for (int i = 0; i < D; ++i) y[i] = __builtin_fmaf((k1[i] + 3.0f * (k2[i] + k3[i])) + k4[i], ww, y[i]);
      }
      if (live) {
#pragma unroll
        for (int i = 0; i < D; ++i) a.h[(size_t)(n + 1) * row + (size_t)p * D + i] = y[i];
      }
    }
  } else {
    const float lv = 1.0f;
    float lam[D], dth[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < D; ++i) lam[i] = a.grad_h[(size_t)(a.T - 1) * row + (size_t)p * D + i];
    // idle lanes of the last wave skip the sweep entirely (no cross-lane traffic inside it): they must neither write
    // the tape columns of the patient they shadow nor feed anything into the scalar-gradient sums below
    for (int n = live ? a.T - 2 : -1; n >= 0; --n) {
      const RStageTimes st(a.t, n, a.perturb, METHOD);
      const float dt = st.dt;
      float* tp0 = a.tape + (size_t)n * NS * tstride + p;
      float y[D], k1[D], av[D], g[D];
#pragma unroll
      for (int i = 0; i < D; ++i) y[i] = a.h[(size_t)n * row + (size_t)p * D + i];
      const DoseK d1 = real_dose(a, p, st.t_first, kel);
      real_rhs<D>(a, w, tl, d1.v, y, k1, tp0);
      if constexpr (METHOD == HODE_METHOD_EULER) {
#pragma unroll
        for (int i = 0; i < D; ++i) g[i] = dt * lam[i];
        real_vjp<D>(a, w, tl, d1, y, k1, g, av, tp0, dth);
#pragma unroll
        for (int i = 0; i < D; ++i) lam[i] += av[i];
      } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
        float Y2[D], k2[D];
        const float half = 0.5f * dt;
#pragma unroll
        for (int i = 0; i < D; ++i) Y2[i] = __builtin_fmaf(k1[i], half, y[i]);
        const DoseK d2 = real_dose(a, p, st.ta, kel);
        real_rhs<D>(a, w, tl, d2.v, Y2, k2, tp0 + tstride);
#pragma unroll
        for (int i = 0; i < D; ++i) g[i] = dt * lam[i];
        real_vjp<D>(a, w, tl, d2, Y2, k2, g, av, tp0 + tstride, dth);
#pragma unroll
        for (int i = 0; i < D; ++i) {
          lam[i] += av[i];
          g[i] = half * av[i];
        }
        real_vjp<D>(a, w, tl, d1, y, k1, g, av, tp0, dth);
#pragma unroll
        for (int i = 0; i < D; ++i) lam[i] += av[i];
      } else {
        float Y2[D], Y3[D], Y4[D], k2[D], k3[D], k4[D];
#pragma unroll
        for (int i = 0; i < D; ++i) Y2[i] = __builtin_fmaf(dt * k1[i], c13, y[i]);
        const DoseK d2 = real_dose(a, p, st.ta, kel);
        real_rhs<D>(a, w, tl, d2.v, Y2, k2, tp0 + tstride);
#pragma unroll
        for (int i = 0; i < D; ++i) Y3[i] = __builtin_fmaf(dt, __builtin_fmaf(-k1[i], c13, k2[i]), y[i]);
        const DoseK d3 = real_dose(a, p, st.tb, kel);
        real_rhs<D>(a, w, tl, d3.v, Y3, k3, tp0 + 2 * tstride);
#pragma unroll
        for (int i = 0; i < D; ++i) Y4[i] = __builtin_fmaf(dt, (k1[i] - k2[i]) + k3[i], y[i]);
        const DoseK d4 = real_dose(a, p, st.t_last, kel);
        real_rhs<D>(a, w, tl, d4.v, Y4, k4, tp0 + 3 * tstride);
        const float w1 = dt * 0.125f, w3 = dt * 0.375f;
        float g1[D], g2[D];
#pragma unroll
        for (int i = 0; i < D; ++i) g[i] = w1 * lam[i];
        real_vjp<D>(a, w, tl, d4, Y4, k4, g, av, tp0 + 3 * tstride, dth);
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const float da = dt * av[i];
          g1[i] = __builtin_fmaf(w1, lam[i], da);
          g2[i] = __builtin_fmaf(w3, lam[i], -da);
          g[i] = __builtin_fmaf(w3, lam[i], da);
          lam[i] += av[i];
        }
        real_vjp<D>(a, w, tl, d3, Y3, k3, g, av, tp0 + 2 * tstride, dth);
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const float da = dt * av[i];
          g2[i] += da;
          g1[i] = __builtin_fmaf(-c13, da, g1[i]);
          lam[i] += av[i];
        }
        real_vjp<D>(a, w, tl, d2, Y2, k2, g2, av, tp0 + tstride, dth);
#pragma unroll
        for (int i = 0; i < D; ++i) {
          g1[i] = __builtin_fmaf(c13, dt * av[i], g1[i]);
          lam[i] += av[i];
        }
        real_vjp<D>(a, w, tl, d1, y, k1, g1, av, tp0, dth);
#pragma unroll
        for (int i = 0; i < D; ++i) lam[i] += av[i];
      }
#pragma unroll
      for (int i = 0; i < D; ++i) lam[i] = __builtin_fmaf(lv, a.grad_h[(size_t)n * row + (size_t)p * D + i], lam[i]);
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < D; ++i) a.grad_y0[(size_t)p * D + i] = lam[i];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float v = wave_sum(dth[j]);
      if ((threadIdx.x & 63) == 0) a.partials[(size_t)(gid >> 6) * 3 + j] = v;
    }
  }
}

}  // namespace hode

// ====================================================================================================== host
namespace {

using hode::RealArgs;

size_t ral256(size_t x) { return (x + 255) / 256 * 256; }
int real_stages(int method) { return method == HODE_METHOD_EULER ? 1 : (method == HODE_METHOD_MIDPOINT ? 2 : 4); }
int real_rows(const hode_solve_desc* d) { return 5 + 4 * d->hidden_dim + 5 * (d->latent_dim - 4); }

struct RealLayout {
  size_t tape, partials, dose, total;
};
// [ tape (bwd) | theta partials (bwd) | dose table (both) ]; the tape stays first: hode/real.py views it from offset 0
// the matrix-core backward accumulates the weight gradients on chip when the caller hands it the flat accumulator (grad_w1)
bool real_onchip(const hode_solve_desc* d) {
  const char* env = getenv("HODE_REAL_LAYOUT");
  return hode::real_mf_supported(d) && !(env && env[0] == 't') && d->grad_w1 != nullptr;
}

RealLayout real_layout(const hode_solve_desc* d, bool bwd) {
  RealLayout L{0, 0, 0, 0};
  size_t off = 0;
  if (bwd) {
    const size_t inst = (size_t)(d->n_times - 1) * real_stages(d->method);
    L.tape = off;
    off = ral256(off + (real_onchip(d) ? hode::real_mf_partial_bytes(d) : inst * real_rows(d) * (size_t)d->batch * 4));
    L.partials = off; off = ral256(off + (size_t)((d->batch + 15) / 16) * 3 * 4);  // one row per wave of either layout
  }
  L.dose = off; off = ral256(off + (size_t)2 * (d->n_action_times + 1) * (size_t)d->batch * 4);
  L.total = off;
  return L;
}

int check_real(const hode_solve_desc* d, bool bwd) {
  if (d->method < HODE_METHOD_EULER || d->method > HODE_METHOD_RK4_38)
    return hode::fail(HODE_E_UNSUPPORTED, "real rhs: unknown fixed-grid method %d", d->method);
  if (d->batch <= 0 || d->n_times <= 0 || d->hidden_dim <= 0 || d->n_action_times <= 0)
    return hode::fail(HODE_E_SIZE, "bad sizes: batch=%d n_times=%d hidden=%d n_action_times=%d", d->batch, d->n_times,
                      d->hidden_dim, d->n_action_times);
  if (d->latent_dim != 4 && d->latent_dim != 20)
    return hode::fail(HODE_E_UNSUPPORTED, "real rhs: latent_dim %d has no compiled kernel (have 4, 20)", d->latent_dim);
  if (!d->t || !d->y0 || !d->dosage || !d->theta || !d->w1 || !d->h)
    return hode::fail(HODE_E_NULL, "t / y0 / dosage (dose table) / theta / w1 (flat weights) / h must be non-NULL");
  if (bwd && (!d->grad_h || !d->grad_y0 || !d->grad_theta)) return hode::fail(HODE_E_NULL, "grad_h / grad_y0 / grad_theta required");
  const RealLayout L = real_layout(d, bwd);
  if (!d->workspace || d->workspace_bytes < L.total)
    return hode::fail(HODE_E_WORKSPACE, "workspace %zu B < required %zu B", d->workspace_bytes, L.total);
  return 0;
}

template <int D, bool BWD>
int launch_real(const hode_solve_desc* d, const RealArgs& a, hipStream_t s) {
  const dim3 grid((d->batch + 63) / 64), block(64);
  switch (d->method) {
    case HODE_METHOD_EULER: hipLaunchKernelGGL((hode::real_kernel<D, HODE_METHOD_EULER, BWD>), grid, block, 0, s, a); break;
    case HODE_METHOD_MIDPOINT: hipLaunchKernelGGL((hode::real_kernel<D, HODE_METHOD_MIDPOINT, BWD>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((hode::real_kernel<D, HODE_METHOD_RK4_38, BWD>), grid, block, 0, s, a); break;
  }
  return hode::hip_fail(hipGetLastError(), "real kernel launch");
}

}  // namespace

namespace hode {

size_t real_workspace_bytes(const hode_solve_desc* d, bool bwd) { return real_layout(d, bwd).total; }

int real_rk(const hode_solve_desc* d, bool bwd, hipStream_t s) {
  if (int e = check_real(d, bwd)) return e;
  const RealLayout L = real_layout(d, bwd);
  char* ws = (char*)d->workspace;
  RealArgs a{};
  a.t = d->t; a.y0 = d->y0; a.act = d->dosage; a.theta = d->theta; a.wflat = d->w1; a.h = d->h;
  a.grad_h = d->grad_h; a.grad_y0 = d->grad_y0;
  a.tape = bwd ? (float*)(ws + L.tape) : nullptr;
  a.partials = bwd ? (float*)(ws + L.partials) : nullptr;
  a.dose_tab = (float*)(ws + L.dose);
  a.B = d->batch; a.T = d->n_times; a.Ta = d->n_action_times; a.H = d->hidden_dim; a.perturb = d->perturb;
  // default where it exists (D = 20, hidden <= 64): the matrix-core kernels of hode_real_mf.hip, 16 patients per wave;
  // HODE_REAL_LAYOUT=t forces the one-patient-per-lane kernels of this file
  const char* env = getenv("HODE_REAL_LAYOUT");
  const bool mf = real_mf_supported(d) && !(env && env[0] == 't');
  int e;
  if (mf) e = launch_real_mf(d, a, bwd, s);
  else if (d->latent_dim == 4) e = bwd ? launch_real<4, true>(d, a, s) : launch_real<4, false>(d, a, s);
  else e = bwd ? launch_real<20, true>(d, a, s) : launch_real<20, false>(d, a, s);
  if (e || !bwd) return e;
  const int nw = mf ? (d->batch + 15) / 16 : (d->batch + 63) / 64;
  return launch_fold_partials(a.partials, nw, 3, 0, 0, nullptr, nullptr, d->grad_theta, 1, s);
}

}  // namespace hode
