// NeuralODE rhs on the matrix cores: dy/dt = tanh(W2 tanh(W1 [y, Dose(t)] + b1) + b2) (reference model.py:969-1026) inside
// the fixed-grid euler / midpoint / rk4(3/8) loop and its discrete adjoint, gfx950.  Same C-ABI contract, tape format and
// arithmetic (up to summation order) as the one-patient-per-lane kernels in hode_neural.hip, which stay as the fallback
// (HODE_NEURAL_LAYOUT=t).
//
// A wave owns 16 patients for the whole time loop; the state never leaves registers.  With v_mfma_f32_16x16x4_f32
// (A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15], C/D rows 4 (lane >> 4) + reg, column lane & 15) and
// g = lane >> 4, n = lane & 15 (the patient):
//   * a vector over at most 16 "rows" (the input e = [y, Dose, 0..], an output, a cotangent) is ONE accumulator tile:
//     lane (g, n) holds rows 4g + r in register r;
//   * the contraction index of every product is ordered so that k-chunk r consists of the rows {4g + r : g = 0..3}: then
//     the B fragment of chunk r IS register r of the tile the previous product (or the element-wise step) left -- no
//     cross-lane traffic, no LDS, anywhere in the loop;
//   * the four weight operands (W1, W2, W2^T, W1^T, zero padded to 16-row tiles) are gathered once per launch into
//     that fragment order and stay in registers: 4 x 4 x HT floats per lane, HT = ceil(10 D / 16) hidden tiles.
// Per rhs evaluation: 4 HT MFMAs for the hidden layer, 4 HT for the output layer (4 partial accumulators: a single
// dependent chain would serialise on the MFMA latency), tanh on 4 HT + 4 values per lane; the VJP costs the same again.
// The weight gradients are outer products summed over patients: as in hode_neural.hip the backward tapes their operands
// patient-minor for the caller's BLAS GEMMs (hode/neural.py) -- same offsets, same layout.
#include <hip/hip_runtime.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"
#include "hode_neural_args.hpp"
#include "hode_neural_mf.hpp"

namespace hode {

template <int D, int METHOD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void neural_mf_fwd_kernel(NeuralArgs a) {
  const int lane = threadIdx.x;
  NeuralMf<D> nn;
  nn.load(a, lane);
  const int g = nn.g;
  const int pr = blockIdx.x * 16 + nn.n;
  const bool live = pr < a.B;
  const int p = live ? pr : a.B - 1;
  const float dosage = a.dosage[p];
  const size_t row = (size_t)a.B * D;
  v4 y = mf_load_rows<D>(a.y0 + (size_t)p * D, g);
  float* hp = a.h + (size_t)p * D;
  mf_store_rows<D>(hp, g, y, live);
  v4 a1[NeuralMf<D>::HT];
  for (int nstep = 0; nstep + 1 < a.T; ++nstep) {
    const NStageTimes st(a.t, nstep, a.perturb, METHOD);
    const float dt = st.dt;
    const v4 k1 = nn.rhs(mf_with_dose<D>(y, neural_dose(a, p, dosage, st.t_first), g), a1);
    if constexpr (METHOD == HODE_METHOD_EULER) {
      y = y + dt * k1;
    } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
      const v4 Y = y + (0.5f * dt) * k1;
      const v4 k2 = nn.rhs(mf_with_dose<D>(Y, neural_dose(a, p, dosage, st.ta), g), a1);
      y = y + dt * k2;
    } else {
      v4 Y = y + (dt * k1) * kThird;
      const v4 k2 = nn.rhs(mf_with_dose<D>(Y, neural_dose(a, p, dosage, st.ta), g), a1);
      Y = y + dt * (k2 - k1 * kThird);
      const v4 k3 = nn.rhs(mf_with_dose<D>(Y, neural_dose(a, p, dosage, st.tb), g), a1);
      Y = y + dt * ((k1 - k2) + k3);
      const v4 k4 = nn.rhs(mf_with_dose<D>(Y, neural_dose(a, p, dosage, st.t_last), g), a1);
      y = y + ((k1 + 3.0f * (k2 + k3)) + k4) * (dt * 0.125f);
    }
    hp += row;
    mf_store_rows<D>(hp, g, y, live);
  }
}

// ONCHIP: the weight gradients are accumulated by the wave on the matrix cores (NeuralGradAcc, hode_neural_mf.hpp) and
// leave as one partial block per wave in a.a1t (folded by neural_grad_fold_kernel); otherwise their operands are taped
// patient-minor for the caller's GEMMs (the contract of hode_neural_tape_offsets, kept for the lane-per-patient kernels).
template <int D, int METHOD, bool ONCHIP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void neural_mf_bwd_kernel(NeuralArgs a) {
  constexpr int HD = 10 * D;
  constexpr int HT = NeuralMf<D>::HT;
  __shared__ __attribute__((aligned(16))) float lds[ONCHIP ? NeuralGradAcc<D>::kLdsFloats : 4];
  NeuralGradAcc<D> acc;
  if constexpr (ONCHIP) acc.init(lds);
  constexpr int NS = METHOD == HODE_METHOD_EULER ? 1 : (METHOD == HODE_METHOD_MIDPOINT ? 2 : 4);
  const int lane = threadIdx.x;
  NeuralMf<D> nn;
  nn.load(a, lane);
  const int g = nn.g;
  const int pr = blockIdx.x * 16 + nn.n;
  const bool live = pr < a.B;
  const int p = live ? pr : a.B - 1;
  const float lv = live ? 1.0f : 0.0f;
  const size_t B = a.B;
  const float dosage = a.dosage[p];
  const size_t row = B * D;
  v4 lam = lv * mf_load_rows<D>(a.grad_h + (size_t)(a.T - 1) * row + (size_t)p * D, g);

  // tapes: operand rows patient-minor, [inst][rows][B]
  auto tape_hidden = [&](float* base, size_t inst, const v4 (&v)[HT]) {
    if (!live) return;
    float* dst = base + inst * HD * B + p;
#pragma unroll
    for (int i = 0; i < HT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rw = 16 * i + 4 * g + r;
        if (rw < HD) dst[(size_t)rw * B] = v[i][r];
      }
  };
  auto tape_rows = [&](float* base, size_t inst, int nrows, const v4& v) {
    if (!live) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rw = 4 * g + r;
      if (rw < nrows) base[(inst * nrows + rw) * B + p] = v[r];
    }
  };

  for (int nstep = a.T - 2; nstep >= 0; --nstep) {
    const NStageTimes st(a.t, nstep, a.perturb, METHOD);
    const float dt = st.dt;
    const size_t i0 = (size_t)nstep * NS;
    const v4 y = mf_load_rows<D>(a.h + (size_t)nstep * row + (size_t)p * D, g);
    v4 e[NS], k[NS], a1[ONCHIP ? 1 : NS][HT];  // ONCHIP: only the last stage's activations stay, the VJPs recompute theirs
    // ---- recompute the stages (inputs and hidden activations go to the tape as they are formed)
    e[0] = mf_with_dose<D>(y, neural_dose(a, p, dosage, st.t_first), g);
    k[0] = nn.rhs(e[0], a1[0]);
    if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
      e[1] = mf_with_dose<D>(y + (0.5f * dt) * k[0], neural_dose(a, p, dosage, st.ta), g);
      k[1] = nn.rhs(e[1], a1[ONCHIP ? 0 : 1]);
    } else if constexpr (METHOD == HODE_METHOD_RK4_38) {
      e[1] = mf_with_dose<D>(y + (dt * k[0]) * kThird, neural_dose(a, p, dosage, st.ta), g);
      k[1] = nn.rhs(e[1], a1[ONCHIP ? 0 : 1]);
      e[2] = mf_with_dose<D>(y + dt * (k[1] - k[0] * kThird), neural_dose(a, p, dosage, st.tb), g);
      k[2] = nn.rhs(e[2], a1[ONCHIP ? 0 : 2]);
      e[3] = mf_with_dose<D>(y + dt * ((k[0] - k[1]) + k[2]), neural_dose(a, p, dosage, st.t_last), g);
      k[3] = nn.rhs(e[3], a1[ONCHIP ? 0 : 3]);
    }
    if constexpr (!ONCHIP) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        tape_rows(a.yet, i0 + s, D + 1, e[s]);
        tape_hidden(a.a1t, i0 + s, a1[s]);
      }
    }
    // ---- adjoint of the stages
    auto vjp = [&](int s, const v4& gk) {
      v4 u2, u1[HT];
      if constexpr (ONCHIP) {
        if (s != NS - 1) nn.hidden(e[s], a1[0]);  // 384 registers of activations would not fit next to the accumulators
      }
      v4 av = nn.vjp(a1[ONCHIP ? 0 : s], k[s], gk, u2, u1);
      if constexpr (ONCHIP) {
        acc.add(u1, e[s], u2, a1[0], g, nn.n);
      } else {
        tape_rows(a.u2t, i0 + s, D, u2);
        tape_hidden(a.u1t, i0 + s, u1);
      }
      if (g == NeuralMf<D>::GD) av[NeuralMf<D>::RD] = 0.f;  // the Dose input is not a state
      return av;
    };
    if constexpr (METHOD == HODE_METHOD_EULER) {
      lam = lam + vjp(0, dt * lam);
    } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
      const v4 a1v = vjp(1, dt * lam);
      lam = lam + a1v;
      lam = lam + vjp(0, (0.5f * dt) * a1v);
    } else {
      const float w1 = dt * 0.125f, w3 = dt * 0.375f;
      const v4 a3 = vjp(3, w1 * lam);
      v4 da = dt * a3;
      v4 g1 = w1 * lam + da;
      v4 g2 = w3 * lam - da;
      const v4 gg = w3 * lam + da;
      lam = lam + a3;
      const v4 a2 = vjp(2, gg);
      da = dt * a2;
      g2 = g2 + da;
      g1 = g1 - kThird * da;
      lam = lam + a2;
      const v4 a1v = vjp(1, g2);
      g1 = g1 + kThird * (dt * a1v);
      lam = lam + a1v;
      lam = lam + vjp(0, g1);
    }
    lam = lam + lv * mf_load_rows<D>(a.grad_h + (size_t)nstep * row + (size_t)p * D, g);
  }
  mf_store_rows<D>(a.grad_y0 + (size_t)p * D, g, lam, live);
  if constexpr (ONCHIP) acc.store(a.a1t + (size_t)blockIdx.x * NeuralGradAcc<D>::NP, lane);
}

template <int D>
int launch_neural_mf_d(const hode_solve_desc* d, const NeuralArgs& a, bool bwd, hipStream_t s) {
  const dim3 grid((d->batch + 15) / 16), block(64);
  const bool onchip = bwd && d->grad_w1 != nullptr;
#define HODE_NEURAL_MF_LAUNCH(M)                                                                   \
  if (bwd && onchip) hipLaunchKernelGGL((neural_mf_bwd_kernel<D, M, true>), grid, block, 0, s, a);  \
  else if (bwd) hipLaunchKernelGGL((neural_mf_bwd_kernel<D, M, false>), grid, block, 0, s, a);      \
  else hipLaunchKernelGGL((neural_mf_fwd_kernel<D, M>), grid, block, 0, s, a);
  switch (d->method) {
    case HODE_METHOD_EULER: HODE_NEURAL_MF_LAUNCH(HODE_METHOD_EULER) break;
    case HODE_METHOD_MIDPOINT: HODE_NEURAL_MF_LAUNCH(HODE_METHOD_MIDPOINT) break;
    default: HODE_NEURAL_MF_LAUNCH(HODE_METHOD_RK4_38) break;
  }
  if (onchip)
    hipLaunchKernelGGL((neural_grad_fold_kernel<D>), dim3(NeuralGradAcc<D>::NP), block, 0, s, a.a1t, (int)grid.x, d->grad_w1,
                       d->grad_b1, d->grad_w2, d->grad_b2);
  return hip_fail(hipGetLastError(), "neural MFMA kernel launch");
}

// bytes of per-wave gradient partials the on-chip backward needs (it uses the a1t slot of the workspace for them)
size_t neural_mf_partial_bytes(const hode_solve_desc* d) {
  const size_t nw = (d->batch + 15) / 16;
  switch (d->latent_dim) {
    case 4: return nw * NeuralGradAcc<4>::NP * sizeof(float);
    case 6: return nw * NeuralGradAcc<6>::NP * sizeof(float);
    case 8: return nw * NeuralGradAcc<8>::NP * sizeof(float);
    case 10: return nw * NeuralGradAcc<10>::NP * sizeof(float);
    case 14: return nw * NeuralGradAcc<14>::NP * sizeof(float);
    default: return nw * NeuralGradAcc<12>::NP * sizeof(float);
  }
}

int launch_neural_mf(const hode_solve_desc* d, const NeuralArgs& a, bool bwd, hipStream_t s) {
  switch (d->latent_dim) {   // check_neural admits 4, 6, ..., 14: [y, Dose, 1] fits one 16-row tile
    case 4: return launch_neural_mf_d<4>(d, a, bwd, s);
    case 6: return launch_neural_mf_d<6>(d, a, bwd, s);
    case 8: return launch_neural_mf_d<8>(d, a, bwd, s);
    case 10: return launch_neural_mf_d<10>(d, a, bwd, s);
    case 14: return launch_neural_mf_d<14>(d, a, bwd, s);
    default: return launch_neural_mf_d<12>(d, a, bwd, s);
  }
}

}  // namespace hode
