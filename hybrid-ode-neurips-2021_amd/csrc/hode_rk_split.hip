// "Split" layout of the fixed-grid Roche solve: wave-specialised expert / learned pipelines, gfx950 (D in {8, 12}).
//
// Same arithmetic and C-ABI contract as hode_rk_kernels.hpp.  Why it exists: the quad-layout kernels are
// bound by VALU ISSUE on ONE wave -- a SIMD retires one dependent wave-instruction per ~4-5 cycles and does not overlap
// the VALU work of several waves -- while at 10 000 patients 40 % of the SIMDs have no wave at all.  So the work of a
// patient group is split over waves that run CONCURRENTLY ON DIFFERENT SIMDs of one CU:
//   * the expert block (Disease, ImmuneReact, Immunity, Dose2) is AUTONOMOUS: it never reads the learned latents
//     (reference model.py:527-544).  Wave 0 of a workgroup integrates it alone, one patient per lane (48 patients),
//     ONE STEP AHEAD, and publishes the 4 stage states of every step in an LDS ring ([2][4 stages][48][4] floats);
//   * waves 1..3 integrate the learned latents of 16 patients each in the quad layout (a patient per DPP quad, each lane
//     owns (D-4)/4 rows of tanh(W y + b) AND only the matching components of the state -- stage algebra on 1-2
//     components instead of D), reading the expert stage states from LDS (one ds_read_b128, quad-broadcast).
// One __syncthreads per step hands the ring over.  A workgroup = 4 waves = the 4 SIMDs of a CU; 209 workgroups at the
// bench shape.  Each wave issues ~50 VALU instructions per rhs instead of ~117.
#include <hip/hip_runtime.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"
#include "hode_roche.hpp"

namespace hode {

constexpr int kSplitPatients = 48;  // per workgroup: wave 0 holds all 48 (one per lane), waves 1..3 hold 16 each

struct SplitArgs {
  const float* __restrict__ t;
  const float* __restrict__ y0;
  const float* __restrict__ dosage;
  const float* __restrict__ dose_times;
  const float* __restrict__ theta;
  const float* __restrict__ w1;
  const float* __restrict__ b1;
  float* __restrict__ h;
  int* __restrict__ status;
  int B, T, K, perturb;
};

struct SpTimes {
  float t0, t1, dt, ts[4];
  HODE_DEV SpTimes(const float* __restrict__ t, int n, int perturb, int method) {
    t0 = t[n];
    t1 = t[n + 1];
    dt = t1 - t0;
    ts[0] = perturb ? nextafter_up(t0) : t0;
    ts[3] = perturb ? nextafter_down(t1) : t1;
    if (method == HODE_METHOD_RK4_38) {
      ts[1] = add_rn(t0, mul_rn(dt, (float)(1.0 / 3.0)));
      ts[2] = add_rn(t0, mul_rn(dt, (float)(2.0 / 3.0)));
    } else {
      ts[1] = add_rn(t0, mul_rn(0.5f, dt));
      ts[2] = ts[1];
    }
  }
};

template <int METHOD>
constexpr int sp_stages() { return METHOD == HODE_METHOD_EULER ? 1 : (METHOD == HODE_METHOD_MIDPOINT ? 2 : 4); }

// stage state i (0-based) of the scheme from y and the earlier stage derivatives, one component
template <int METHOD>
HODE_DEV float sp_stage_state(int i, float y, float dt, float k1, float k2, float k3) {
  constexpr float c13 = (float)(1.0 / 3.0);
  if constexpr (METHOD == HODE_METHOD_MIDPOINT) return i == 0 ? y : __builtin_fmaf(k1, 0.5f * dt, y);
  else if constexpr (METHOD == HODE_METHOD_RK4_38) {
    if (i == 0) return y;
    if (i == 1) return __builtin_fmaf(dt * k1, c13, y);
    if (i == 2) return __builtin_fmaf(dt, __builtin_fmaf(-k1, c13, k2), y);
    return __builtin_fmaf(dt, (k1 - k2) + k3, y);
  } else return y;
}
template <int METHOD>
HODE_DEV float sp_advance(float y, float dt, float k1, float k2, float k3, float k4) {
  if constexpr (METHOD == HODE_METHOD_EULER) return __builtin_fmaf(dt, k1, y);
  else if constexpr (METHOD == HODE_METHOD_MIDPOINT) return __builtin_fmaf(dt, k2, y);
  else return __builtin_fmaf((k1 + 3.0f * (k2 + k3)) + k4, dt * 0.125f, y);
}

template <int D, int METHOD, bool ABLATE, bool HILL2, bool K1>
HODE_DEV void split_fwd_body(const SplitArgs& a) {
  constexpr int NS = sp_stages<METHOD>();
  constexpr int M = D - 4;
  constexpr int MR = M / 4;
  __shared__ __attribute__((aligned(16))) float ring[2][4][kSplitPatients][4];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int b0 = blockIdx.x * kSplitPatients;
  const RocheTheta th = load_theta(a.theta, ABLATE);
  const size_t row = (size_t)a.B * D;

  if (wave == 0) {
    // ------------------------------------------------------------------ expert pipeline: one patient per lane
    const int slot = lane;
    const bool live = slot < kSplitPatients && b0 + slot < a.B;
    const int p = min(b0 + (slot < kSplitPatients ? slot : 0), a.B - 1);
    DoseSched<K1> ds;
    ds.dosage = a.dosage[p];
    ds.K = a.K;
    ds.taus = a.dose_times + (size_t)p * a.K;
    ds.tau0 = K1 ? ds.taus[0] : 0.f;
    MlSlice<4, 1> none;
    float own1[1];
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = a.y0[(size_t)p * D + i];
    if (live) *reinterpret_cast<float4*>(a.h + (size_t)p * D) = make_float4(y[0], y[1], y[2], y[3]);
    for (int it = 0; it < a.T; ++it) {
      if (it + 1 < a.T) {
        const SpTimes st(a.t, it, a.perturb, METHOD);
        float k[4][4] = {};
        float Y[4];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int c = 0; c < 4; ++c) Y[c] = sp_stage_state<METHOD>(s, y[c], st.dt, k[0][c], k[1][c], k[2][c]);
          if (slot < kSplitPatients) *reinterpret_cast<float4*>(&ring[it & 1][s][slot][0]) = make_float4(Y[0], Y[1], Y[2], Y[3]);
          roche_rhs<4, 1, ABLATE, HILL2>(th, none, ds.at(st.ts[s], th.kel).v, Y, k[s], own1);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = sp_advance<METHOD>(y[c], st.dt, k[0][c], k[1][c], k[2][c], k[3][c]);
        if (live) *reinterpret_cast<float4*>(a.h + (size_t)(it + 1) * row + (size_t)p * D) = make_float4(y[0], y[1], y[2], y[3]);
      }
      __syncthreads();
    }
    if (a.status) {
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 4; ++c) bad |= !__builtin_isfinite(y[c]);
      if (bad && live) atomicOr(a.status, HODE_STATUS_NONFINITE);
    }
  } else {
    // ------------------------------------------------------------------ learned pipeline: a patient per DPP quad
    const int q = lane & 3;
    const int slot = (wave - 1) * 16 + (lane >> 2);
    const bool live = b0 + slot < a.B;
    const int p = min(b0 + slot, a.B - 1);
    float w[MR][D], bias[MR], yo[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      const int rowi = q * MR + r;
#pragma unroll
      for (int i = 0; i < D; ++i) w[r][i] = a.w1[rowi * D + i];
      bias[r] = a.b1[rowi];
      yo[r] = a.y0[(size_t)p * D + 4 + rowi];
    }
    auto store_own = [&](float* dst) {
      if (!live) return;
      if constexpr (MR == 2) *reinterpret_cast<float2*>(dst + 4 + 2 * q) = make_float2(yo[0], yo[1]);
      else dst[4 + q] = yo[0];
    };
    store_own(a.h + (size_t)p * D);
    __syncthreads();  // iteration 0: the expert wave fills ring[0]
    for (int it = 1; it < a.T; ++it) {
      const int n = it - 1;
      const float dt = a.t[n + 1] - a.t[n];
      float k[4][MR] = {};
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float Yo[MR], Y[D];
#pragma unroll
        for (int r = 0; r < MR; ++r) Yo[r] = sp_stage_state<METHOD>(s, yo[r], dt, k[0][r], k[1][r], k[2][r]);
        const float4 e = *reinterpret_cast<const float4*>(&ring[n & 1][s][slot][0]);
        Y[0] = e.x; Y[1] = e.y; Y[2] = e.z; Y[3] = e.w;
#pragma unroll
        for (int r = 0; r < MR; ++r) {
          Y[4 + 0 * MR + r] = quad_bcast<0>(Yo[r]);
          Y[4 + 1 * MR + r] = quad_bcast<1>(Yo[r]);
          Y[4 + 2 * MR + r] = quad_bcast<2>(Yo[r]);
          Y[4 + 3 * MR + r] = quad_bcast<3>(Yo[r]);
        }
#pragma unroll
        for (int r = 0; r < MR; ++r) {
          // two partial sums per row: shorter dependent fma chains
          float z0 = bias[r], z1 = 0.f;
#pragma unroll
          for (int i = 0; i < D; i += 2) {
            z0 = __builtin_fmaf(w[r][i], Y[i], z0);
            z1 = __builtin_fmaf(w[r][i + 1], Y[i + 1], z1);
          }
          k[s][r] = tanh_f32(z0 + z1);
        }
      }
#pragma unroll
      for (int r = 0; r < MR; ++r) yo[r] = sp_advance<METHOD>(yo[r], dt, k[0][r], k[1][r], k[2][r], k[3][r]);
      store_own(a.h + (size_t)(n + 1) * row + (size_t)p * D);
      __syncthreads();
    }
    if (a.status) {
      bool bad = false;
#pragma unroll
      for (int r = 0; r < MR; ++r) bad |= !__builtin_isfinite(yo[r]);
      if (bad && live) atomicOr(a.status, HODE_STATUS_NONFINITE);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// The adjoint system reverses the dependency: the LEARNED cotangents are autonomous (the expert rhs never reads learned
// latents, so no expert term enters d/dy_learned), while the expert cotangent needs c_s = sum_j W[j][0..3] u_j from
// the learned block at every stage.  Per iteration k (ML handles step m_k = T-2-k):
//   wave 0 (expert):  (b) adjoint of step m_{k-1}: reads its own stage states / doses and the learned block's c_s from
//                     the rings of iteration k-1;  (a) recomputes the stage states of step m_{k+1} into the rings
//   waves 1..3 (ML):  recompute + adjoint of step m_k (stage states from the ring), publish c_s
//   one __syncthreads.  Rings are double buffered by iteration parity; (b) runs before (a) because (a) overwrites the
//   buffer (b) reads.
struct SplitBwdArgs {
  const float* __restrict__ t;
  const float* __restrict__ dosage;
  const float* __restrict__ dose_times;
  const float* __restrict__ theta;
  const float* __restrict__ w1;
  const float* __restrict__ b1;
  const float* __restrict__ h;
  const float* __restrict__ grad_h;
  float* __restrict__ grad_y0;
  float* __restrict__ part_ml;   // [3 * nblk][M*D + M]
  float* __restrict__ part_th;   // [nblk][kNTheta]
  int B, T, K, perturb;
};

template <int D, int METHOD, bool ABLATE, bool HILL2, bool NEED_TH, bool K1>
HODE_DEV void split_bwd_body(const SplitBwdArgs& a) {
  constexpr int NS = sp_stages<METHOD>();
  constexpr int M = D - 4;
  constexpr int MR = M / 4;
  constexpr float c13 = (float)(1.0 / 3.0);
  __shared__ __attribute__((aligned(16))) float yring[2][4][kSplitPatients][4];   // expert stage states
  __shared__ __attribute__((aligned(16))) float cring[2][4][kSplitPatients][4];   // learned block -> expert cotangent
  __shared__ __attribute__((aligned(16))) float dring[2][4][kSplitPatients][2];   // Dose(t_s), dDose/dkel
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int b0 = blockIdx.x * kSplitPatients;
  const RocheTheta th = load_theta(a.theta, ABLATE);
  const size_t row = (size_t)a.B * D;
  const int T = a.T;

  if (wave == 0) {
    // ================================================================== expert wave
    const int slot = lane < kSplitPatients ? lane : 0;
    const bool mine = lane < kSplitPatients;
    const bool live = mine && b0 + slot < a.B;
    const int p = min(b0 + slot, a.B - 1);
    const float lv = live ? 1.0f : 0.0f;
    DoseSched<K1> ds;
    ds.dosage = a.dosage[p];
    ds.K = a.K;
    ds.taus = a.dose_times + (size_t)p * a.K;
    ds.tau0 = K1 ? ds.taus[0] : 0.f;
    const float ln_ec50 = log_f32(th.ec50);
    MlSlice<4, 1> none;
    MlColSlice<4, 1> nonec;
    GradAcc<4, 1> acc;
    acc.zero();
    float own1[1] = {0.f};

    // (a): stage states + doses of step m into ring buffer `par`
    auto recompute = [&](int m, int par) {
      const SpTimes st(a.t, m, a.perturb, METHOD);
      float y[4], k[4][4] = {}, Y[4];
      const float4 hv = *reinterpret_cast<const float4*>(a.h + (size_t)m * row + (size_t)p * D);
      y[0] = hv.x; y[1] = hv.y; y[2] = hv.z; y[3] = hv.w;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Y[c] = sp_stage_state<METHOD>(s, y[c], st.dt, k[0][c], k[1][c], k[2][c]);
        const DoseVal dv = ds.at(st.ts[s], th.kel);
        if (mine) {
          *reinterpret_cast<float4*>(&yring[par][s][slot][0]) = make_float4(Y[0], Y[1], Y[2], Y[3]);
          *reinterpret_cast<float2*>(&dring[par][s][slot][0]) = make_float2(dv.v, dv.dk);
        }
        if (s + 1 < NS) roche_rhs<4, 1, ABLATE, HILL2>(th, none, dv.v, Y, k[s], own1);
      }
    };

    float lam[4];
    {
      const float4 g4 = *reinterpret_cast<const float4*>(a.grad_h + (size_t)(T - 1) * row + (size_t)p * D);
      lam[0] = lv * g4.x; lam[1] = lv * g4.y; lam[2] = lv * g4.z; lam[3] = lv * g4.w;
    }
    if (T >= 2) recompute(T - 2, 0);
    __syncthreads();
    for (int k = 0; k < T; ++k) {
      if (k >= 1) {
        // ---- (b) adjoint of step m = T-1-k
        const int m = T - 1 - k;
        const int par = (k - 1) & 1;
        const float dt = a.t[m + 1] - a.t[m];
        float Y[4][4], cs[4][4];
        DoseVal dv[4];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float4 yv = *reinterpret_cast<const float4*>(&yring[par][s][slot][0]);
          const float4 cv = *reinterpret_cast<const float4*>(&cring[par][s][slot][0]);
          const float2 d2 = *reinterpret_cast<const float2*>(&dring[par][s][slot][0]);
          Y[s][0] = yv.x; Y[s][1] = yv.y; Y[s][2] = yv.z; Y[s][3] = yv.w;
          cs[s][0] = lv * cv.x; cs[s][1] = lv * cv.y; cs[s][2] = lv * cv.z; cs[s][3] = lv * cv.w;
          dv[s].v = d2.x; dv[s].dk = d2.y;
        }
        float g[4], av[4];
        auto vjp = [&](int s, const float (&gs)[4]) {
          roche_vjp<4, 1, ABLATE, HILL2, NEED_TH>(th, none, nonec, ln_ec50, dv[s], Y[s], own1, gs, 0, av, acc);
#pragma unroll
          for (int c = 0; c < 4; ++c) av[c] += cs[s][c];
        };
        if constexpr (METHOD == HODE_METHOD_EULER) {
#pragma unroll
          for (int c = 0; c < 4; ++c) g[c] = dt * lam[c];
          vjp(0, g);
#pragma unroll
          for (int c = 0; c < 4; ++c) lam[c] += av[c];
        } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
          const float half = 0.5f * dt;
#pragma unroll
          for (int c = 0; c < 4; ++c) g[c] = dt * lam[c];
          vjp(1, g);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            lam[c] += av[c];
            g[c] = half * av[c];
          }
          vjp(0, g);
#pragma unroll
          for (int c = 0; c < 4; ++c) lam[c] += av[c];
        } else {
          const float w1 = dt * 0.125f, w3 = dt * 0.375f;
          float g1[4], g2[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) g[c] = w1 * lam[c];
          vjp(3, g);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float da = dt * av[c];
            g1[c] = __builtin_fmaf(w1, lam[c], da);
            g2[c] = __builtin_fmaf(w3, lam[c], -da);
            g[c] = __builtin_fmaf(w3, lam[c], da);
            lam[c] += av[c];
          }
          vjp(2, g);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float da = dt * av[c];
            g2[c] += da;
            g1[c] = __builtin_fmaf(-c13, da, g1[c]);
            lam[c] += av[c];
          }
          vjp(1, g2);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            g1[c] = __builtin_fmaf(c13, dt * av[c], g1[c]);
            lam[c] += av[c];
          }
          vjp(0, g1);
#pragma unroll
          for (int c = 0; c < 4; ++c) lam[c] += av[c];
        }
        const float4 g4 = *reinterpret_cast<const float4*>(a.grad_h + (size_t)m * row + (size_t)p * D);
        lam[0] = __builtin_fmaf(lv, g4.x, lam[0]);
        lam[1] = __builtin_fmaf(lv, g4.y, lam[1]);
        lam[2] = __builtin_fmaf(lv, g4.z, lam[2]);
        lam[3] = __builtin_fmaf(lv, g4.w, lam[3]);
      }
      // ---- (a) stage states of step T-3-k for the learned waves' next iteration
      if (T - 3 - k >= 0) recompute(T - 3 - k, (k + 1) & 1);
      __syncthreads();
    }
    if (live) *reinterpret_cast<float4*>(a.grad_y0 + (size_t)p * D) = make_float4(lam[0], lam[1], lam[2], lam[3]);
#pragma unroll
    for (int i = 0; i < kNTheta; ++i) {
      const float v = wave_sum(NEED_TH ? acc.dth[i] : 0.f);
      if (lane == 0) a.part_th[(size_t)blockIdx.x * kNTheta + i] = v;
    }
  } else {
    // ================================================================== learned waves (quad layout, own components)
    const int q = lane & 3;
    const int slot = (wave - 1) * 16 + (lane >> 2);
    const bool live = b0 + slot < a.B;
    const int p = min(b0 + slot, a.B - 1);
    const float lv = live ? 1.0f : 0.0f;
    float w[MR][D], bias[MR], wt[M][MR], wc[M], dw[MR][D], db[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      const int rowi = q * MR + r;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        w[r][i] = a.w1[rowi * D + i];
        dw[r][i] = 0.f;
      }
      bias[r] = a.b1[rowi];
      db[r] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      wc[j] = a.w1[j * D + q];
#pragma unroll
      for (int r = 0; r < MR; ++r) wt[j][r] = a.w1[j * D + 4 + q * MR + r];
    }
    auto load_own = [&](const float* src, float (&v)[MR]) {
      if constexpr (MR == 2) {
        const float2 x = *reinterpret_cast<const float2*>(src + 4 + 2 * q);
        v[0] = x.x; v[1] = x.y;
      } else v[0] = src[4 + q];
    };
    float lam[MR];
    load_own(a.grad_h + (size_t)(T - 1) * row + (size_t)p * D, lam);
#pragma unroll
    for (int r = 0; r < MR; ++r) lam[r] *= lv;
    // the operands of iteration k+1 are fetched during iteration k (the state is needed by the very first instruction of
    // an iteration: an un-hidden HBM round trip would cost a quarter of it)
    float yo_nx[MR] = {}, gh_nx[MR] = {};
    {
      const int m0 = T >= 2 ? T - 2 : 0;
      load_own(a.h + (size_t)m0 * row + (size_t)p * D, yo_nx);
      load_own(a.grad_h + (size_t)m0 * row + (size_t)p * D, gh_nx);
    }
    __syncthreads();  // the expert wave's prologue fills ring 0
    for (int k = 0; k < T; ++k) {
      if (k <= T - 2) {
        const int m = T - 2 - k;
        const int par = k & 1;
        const float dt = a.t[m + 1] - a.t[m];
        float yo[MR], gh[MR];
#pragma unroll
        for (int r = 0; r < MR; ++r) {
          yo[r] = yo_nx[r];
          gh[r] = gh_nx[r];
        }
        {
          const int mn = m >= 1 ? m - 1 : 0;  // clamped: the last prefetch is simply unused
          load_own(a.h + (size_t)mn * row + (size_t)p * D, yo_nx);
          load_own(a.grad_h + (size_t)mn * row + (size_t)p * D, gh_nx);
        }
        // ---- recompute the learned stage derivatives
        float Y[4][D], so[4][MR] = {};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          float Yo[MR];
#pragma unroll
          for (int r = 0; r < MR; ++r) Yo[r] = sp_stage_state<METHOD>(s, yo[r], dt, so[0][r], so[1][r], so[2][r]);
          const float4 e = *reinterpret_cast<const float4*>(&yring[par][s][slot][0]);
          Y[s][0] = e.x; Y[s][1] = e.y; Y[s][2] = e.z; Y[s][3] = e.w;
#pragma unroll
          for (int r = 0; r < MR; ++r) {
            Y[s][4 + 0 * MR + r] = quad_bcast<0>(Yo[r]);
            Y[s][4 + 1 * MR + r] = quad_bcast<1>(Yo[r]);
            Y[s][4 + 2 * MR + r] = quad_bcast<2>(Yo[r]);
            Y[s][4 + 3 * MR + r] = quad_bcast<3>(Yo[r]);
          }
#pragma unroll
          for (int r = 0; r < MR; ++r) {
            float z0 = bias[r], z1 = 0.f;
#pragma unroll
            for (int i = 0; i < D; i += 2) {
              z0 = __builtin_fmaf(w[r][i], Y[s][i], z0);
              z1 = __builtin_fmaf(w[r][i + 1], Y[s][i + 1], z1);
            }
            so[s][r] = tanh_f32(z0 + z1);
          }
        }
        // ---- adjoint of the stages
        float g[MR], av[MR];
        auto vjp = [&](int s, const float (&gs)[MR]) {
          float u[MR], uf[M];
#pragma unroll
          for (int r = 0; r < MR; ++r) {
            u[r] = gs[r] * __builtin_fmaf(-so[s][r], so[s][r], 1.0f);
            db[r] += u[r];
#pragma unroll
            for (int i = 0; i < D; ++i) dw[r][i] = __builtin_fmaf(u[r], Y[s][i], dw[r][i]);
            uf[0 * MR + r] = quad_bcast<0>(u[r]);
            uf[1 * MR + r] = quad_bcast<1>(u[r]);
            uf[2 * MR + r] = quad_bcast<2>(u[r]);
            uf[3 * MR + r] = quad_bcast<3>(u[r]);
          }
          float cq = 0.f;
#pragma unroll
          for (int r = 0; r < MR; ++r) av[r] = 0.f;
#pragma unroll
          for (int j = 0; j < M; ++j) {
            cq = __builtin_fmaf(wc[j], uf[j], cq);
#pragma unroll
            for (int r = 0; r < MR; ++r) av[r] = __builtin_fmaf(wt[j][r], uf[j], av[r]);
          }
          cring[par][s][slot][q] = cq;
        };
        if constexpr (METHOD == HODE_METHOD_EULER) {
#pragma unroll
          for (int r = 0; r < MR; ++r) g[r] = dt * lam[r];
          vjp(0, g);
#pragma unroll
          for (int r = 0; r < MR; ++r) lam[r] += av[r];
        } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
          const float half = 0.5f * dt;
#pragma unroll
          for (int r = 0; r < MR; ++r) g[r] = dt * lam[r];
          vjp(1, g);
#pragma unroll
          for (int r = 0; r < MR; ++r) {
            lam[r] += av[r];
            g[r] = half * av[r];
          }
          vjp(0, g);
#pragma unroll
          for (int r = 0; r < MR; ++r) lam[r] += av[r];
        } else {
          const float w1 = dt * 0.125f, w3 = dt * 0.375f;
          float g1[MR], g2[MR];
#pragma unroll
          for (int r = 0; r < MR; ++r) g[r] = w1 * lam[r];
          vjp(3, g);
#pragma unroll
          for (int r = 0; r < MR; ++r) {
            const float da = dt * av[r];
            g1[r] = __builtin_fmaf(w1, lam[r], da);
            g2[r] = __builtin_fmaf(w3, lam[r], -da);
            g[r] = __builtin_fmaf(w3, lam[r], da);
            lam[r] += av[r];
          }
          vjp(2, g);
#pragma unroll
          for (int r = 0; r < MR; ++r) {
            const float da = dt * av[r];
            g2[r] += da;
            g1[r] = __builtin_fmaf(-c13, da, g1[r]);
            lam[r] += av[r];
          }
          vjp(1, g2);
#pragma unroll
          for (int r = 0; r < MR; ++r) {
            g1[r] = __builtin_fmaf(c13, dt * av[r], g1[r]);
            lam[r] += av[r];
          }
          vjp(0, g1);
#pragma unroll
          for (int r = 0; r < MR; ++r) lam[r] += av[r];
        }
#pragma unroll
        for (int r = 0; r < MR; ++r) lam[r] = __builtin_fmaf(lv, gh[r], lam[r]);
      }
      __syncthreads();
    }
    if (live) {
      float* dst = a.grad_y0 + (size_t)p * D;
      if constexpr (MR == 2) *reinterpret_cast<float2*>(dst + 4 + 2 * q) = make_float2(lam[0], lam[1]);
      else dst[4 + q] = lam[0];
    }
    float* out = a.part_ml + ((size_t)blockIdx.x * 3 + (wave - 1)) * (M * D + M);
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const float v = wave_sum_stride4(dw[r][i]);
        if (lane < 4) out[(lane * MR + r) * D + i] = v;
      }
      const float vb = wave_sum_stride4(db[r]);
      if (lane < 4) out[M * D + lane * MR + r] = vb;
    }
  }
}

// One launch folds both partial arrays of the split backward in a fixed order (deterministic): block j < P_ml sums column j
// of part_ml over its 3*nblk rows into grad_w1 / grad_b1, the remaining kNTheta blocks sum part_th into grad_theta.
// overwrite != 0 stores instead of accumulating (saves the caller's memset of the accumulators).
__global__ __launch_bounds__(64) void split_fold_kernel(const float* __restrict__ part_ml, const float* __restrict__ part_th,
                                                        int nblk, int P_ml, int n_w, float* __restrict__ gw,
                                                        float* __restrict__ gb, float* __restrict__ gth, int need_th,
                                                        int overwrite) {
  const int j = blockIdx.x;
  const int lane = threadIdx.x;
  const bool ml = j < P_ml;
  const float* src = ml ? part_ml + j : part_th + (j - P_ml);
  const int rows = ml ? 3 * nblk : nblk;
  const int stride = ml ? P_ml : kNTheta;
  float s = 0.f;
  for (int base = 0; base < rows; base += 64 * 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int w = base + lane + 64 * q;
      v[q] = w < rows ? src[(size_t)w * stride] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  s = wave_sum(s);
  if (lane != 0) return;
  float* dst = nullptr;
  if (ml) dst = j < n_w ? (gw ? gw + j : nullptr) : (gb ? gb + (j - n_w) : nullptr);
  else if (gth) dst = gth + (j - P_ml);
  if (!dst) return;
  if (!ml && !need_th) s = 0.f;
  *dst = overwrite ? s : *dst + s;
}

template <int D, int METHOD, bool ABLATE, bool NEED_TH>
__global__ __launch_bounds__(256) void split_bwd_kernel(SplitBwdArgs a) {
  const bool hill2 = ABLATE || (a.theta[0] == 2.0f && a.theta[1] == 2.0f);
  if (hill2 && a.K == 1) split_bwd_body<D, METHOD, ABLATE, true, NEED_TH, true>(a);
  else if (hill2) split_bwd_body<D, METHOD, ABLATE, true, NEED_TH, false>(a);
  else split_bwd_body<D, METHOD, ABLATE, false, NEED_TH, false>(a);
}

template <int D, int METHOD, bool ABLATE>
__global__ __launch_bounds__(256) void split_fwd_kernel(SplitArgs a) {
  const bool hill2 = ABLATE || (a.theta[0] == 2.0f && a.theta[1] == 2.0f);
  if (hill2 && a.K == 1) split_fwd_body<D, METHOD, ABLATE, true, true>(a);
  else if (hill2) split_fwd_body<D, METHOD, ABLATE, true, false>(a);
  else split_fwd_body<D, METHOD, ABLATE, false, false>(a);
}

}  // namespace hode

namespace {

template <int D, bool ABLATE>
int split_method(const hode_solve_desc* d, const hode::SplitArgs& a, hipStream_t s) {
  const dim3 grid((d->batch + hode::kSplitPatients - 1) / hode::kSplitPatients), block(256);
  switch (d->method) {
    case HODE_METHOD_EULER: hipLaunchKernelGGL((hode::split_fwd_kernel<D, HODE_METHOD_EULER, ABLATE>), grid, block, 0, s, a); break;
    case HODE_METHOD_MIDPOINT: hipLaunchKernelGGL((hode::split_fwd_kernel<D, HODE_METHOD_MIDPOINT, ABLATE>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((hode::split_fwd_kernel<D, HODE_METHOD_RK4_38, ABLATE>), grid, block, 0, s, a); break;
  }
  return hode::hip_fail(hipGetLastError(), "split kernel launch");
}

}  // namespace

namespace {

template <int D, bool ABLATE>
int split_bwd_method(const hode_solve_desc* d, const hode::SplitBwdArgs& a, hipStream_t s) {
  const dim3 grid((d->batch + hode::kSplitPatients - 1) / hode::kSplitPatients), block(256);
#define HODE_SPLIT_BWD(M)                                                                                        \
  if (d->need_theta_grad) hipLaunchKernelGGL((hode::split_bwd_kernel<D, M, ABLATE, true>), grid, block, 0, s, a);  \
  else hipLaunchKernelGGL((hode::split_bwd_kernel<D, M, ABLATE, false>), grid, block, 0, s, a);
  switch (d->method) {
    case HODE_METHOD_EULER: HODE_SPLIT_BWD(HODE_METHOD_EULER) break;
    case HODE_METHOD_MIDPOINT: HODE_SPLIT_BWD(HODE_METHOD_MIDPOINT) break;
    default: HODE_SPLIT_BWD(HODE_METHOD_RK4_38) break;
  }
  return hode::hip_fail(hipGetLastError(), "split bwd kernel launch");
}

}  // namespace

namespace hode {

bool split_supported(const hode_solve_desc* d) { return d->latent_dim == 8 || d->latent_dim == 12; }

static size_t al256s(size_t x) { return (x + 255) / 256 * 256; }

size_t split_workspace_bytes(const hode_solve_desc* d) {
  const size_t nblk = (d->batch + kSplitPatients - 1) / kSplitPatients;
  const size_t M = d->latent_dim - 4;
  return al256s(nblk * 3 * (M * d->latent_dim + M) * sizeof(float)) + al256s(nblk * kNTheta * sizeof(float));
}

int split_rk_bwd(const hode_solve_desc* d, hipStream_t s) {
  const int nblk = (d->batch + kSplitPatients - 1) / kSplitPatients;
  const int M = d->latent_dim - 4;
  const int Pml = M * d->latent_dim + M;
  char* ws = (char*)d->workspace;
  SplitBwdArgs a{};
  a.t = d->t; a.dosage = d->dosage; a.dose_times = d->dose_times; a.theta = d->theta; a.w1 = d->w1; a.b1 = d->b1;
  a.h = d->h; a.grad_h = d->grad_h; a.grad_y0 = d->grad_y0;
  a.part_ml = (float*)ws;
  a.part_th = (float*)(ws + al256s((size_t)nblk * 3 * Pml * sizeof(float)));
  a.B = d->batch; a.T = d->n_times; a.K = d->n_dose; a.perturb = d->perturb;
  const bool abl = d->rhs_kind == HODE_RHS_ROCHE_ABLATE;
  int e;
  if (d->latent_dim == 8) e = abl ? split_bwd_method<8, true>(d, a, s) : split_bwd_method<8, false>(d, a, s);
  else e = abl ? split_bwd_method<12, true>(d, a, s) : split_bwd_method<12, false>(d, a, s);
  if (e || (d->flags & HODE_FLAG_SKIP_FOLD)) return e;
  const bool th_out = d->grad_theta != nullptr && (d->need_theta_grad || (d->flags & HODE_FLAG_OVERWRITE_GRADS));
  hipLaunchKernelGGL(split_fold_kernel, dim3(Pml + (th_out ? kNTheta : 0)), dim3(64), 0, s, a.part_ml, a.part_th, nblk, Pml,
                     M * d->latent_dim, d->grad_w1, d->grad_b1, d->grad_theta, d->need_theta_grad,
                     (d->flags & HODE_FLAG_OVERWRITE_GRADS) ? 1 : 0);
  return hip_fail(hipGetLastError(), "split_fold launch");
}

int split_rk_fwd(const hode_solve_desc* d, hipStream_t s) {
  SplitArgs a{};
  a.t = d->t; a.y0 = d->y0; a.dosage = d->dosage; a.dose_times = d->dose_times; a.theta = d->theta; a.w1 = d->w1; a.b1 = d->b1;
  a.h = d->h; a.status = d->status;
  a.B = d->batch; a.T = d->n_times; a.K = d->n_dose; a.perturb = d->perturb;
  const bool abl = d->rhs_kind == HODE_RHS_ROCHE_ABLATE;
  if (d->latent_dim == 8) return abl ? split_method<8, true>(d, a, s) : split_method<8, false>(d, a, s);
  return abl ? split_method<12, true>(d, a, s) : split_method<12, false>(d, a, s);
}

}  // namespace hode
