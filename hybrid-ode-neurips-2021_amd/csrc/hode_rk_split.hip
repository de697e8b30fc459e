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

// stage state i (0-based) of the scheme from y and the earlier stage derivatives; V = float (one component) or f2 (a
// packed pair of components, same operations in the same order: the results are identical)
template <int METHOD, class V>
HODE_DEV V sp_stage_state(int i, V y, float dt, V k1, V k2, V k3) {
  constexpr float c13 = (float)(1.0 / 3.0);
  if constexpr (METHOD == HODE_METHOD_MIDPOINT) return i == 0 ? y : vfma(k1, vsplat<V>(0.5f * dt), y);
  else if constexpr (METHOD == HODE_METHOD_RK4_38) {
    if (i == 0) return y;
    if (i == 1) return vfma(dt * k1, vsplat<V>(c13), y);
    if (i == 2) return vfma(vsplat<V>(dt), vfma(-k1, vsplat<V>(c13), k2), y);
    return vfma(vsplat<V>(dt), (k1 - k2) + k3, y);
  } else return y;
}
template <int METHOD, class V>
HODE_DEV V sp_advance(V y, float dt, V k1, V k2, V k3, V k4) {
  if constexpr (METHOD == HODE_METHOD_EULER) return vfma(vsplat<V>(dt), k1, y);
  else if constexpr (METHOD == HODE_METHOD_MIDPOINT) return vfma(vsplat<V>(dt), k2, y);
  else return vfma((k1 + 3.0f * (k2 + k3)) + k4, vsplat<V>(dt * 0.125f), y);
}

template <int MR> struct OwnSel { typedef float type; };
template <> struct OwnSel<2> { typedef f2 type; };

// The learned block as one lane of a patient's DPP quad sees it: MR = (D-4)/4 rows of tanh(W y + b), the weights held as
// (even column, odd column) pairs so that a row is D/2 v_pk_fma_f32 -- lane .x sums the even columns starting from the
// bias, lane .y the odd ones, one add joins them (the same two partial sums the scalar kernels form).
template <int D>
struct MlRows {
  static constexpr int M = D - 4, MR = M / 4, DP = D / 2, MP = M / 2;
  typedef typename OwnSel<MR>::type Own;
  f2 wp[MR][DP];
  float bias[MR];
  HODE_DEV void load(const float* __restrict__ W, const float* __restrict__ b, int q) {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      const int rowi = q * MR + r;
#pragma unroll
      for (int ip = 0; ip < DP; ++ip) wp[r][ip] = pair2(W[rowi * D + 2 * ip], W[rowi * D + 2 * ip + 1]);
      bias[r] = b[rowi];
    }
  }
  // all-gather of the quad's own components into the pairs Y2[2..DP) (component 4 + j lives in lane j / MR)
  static HODE_DEV void gather(Own v, f2* __restrict__ out) {
    if constexpr (MR == 2) {
      out[0] = pair2(quad_bcast<0>(v.x), quad_bcast<0>(v.y));
      out[1] = pair2(quad_bcast<1>(v.x), quad_bcast<1>(v.y));
      out[2] = pair2(quad_bcast<2>(v.x), quad_bcast<2>(v.y));
      out[3] = pair2(quad_bcast<3>(v.x), quad_bcast<3>(v.y));
    } else {
      out[0] = pair2(quad_bcast<0>(v), quad_bcast<1>(v));
      out[1] = pair2(quad_bcast<2>(v), quad_bcast<3>(v));
    }
  }
  HODE_DEV Own rhs(const f2 (&Y2)[DP]) const {
    float z[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      f2 acc = pair2(bias[r], 0.f);
#pragma unroll
      for (int ip = 0; ip < DP; ++ip) acc = vfma(wp[r][ip], Y2[ip], acc);
      z[r] = hsum(acc);
    }
    if constexpr (MR == 2) return tanh_f32(pair2(z[0], z[1]));
    else return tanh_f32(z[0]);
  }
  static HODE_DEV Own load_own(const float* __restrict__ src, int q) {
    if constexpr (MR == 2) {
      const float2 x = *reinterpret_cast<const float2*>(src + 4 + 2 * q);
      return pair2(x.x, x.y);
    } else return src[4 + q];
  }
  static HODE_DEV void store_own(float* __restrict__ dst, int q, Own v) {
    if constexpr (MR == 2) *reinterpret_cast<float2*>(dst + 4 + 2 * q) = make_float2(v.x, v.y);
    else dst[4 + q] = v;
  }
};

template <int D, int METHOD, bool ABLATE, bool HILL2, bool K1>
HODE_DEV void split_fwd_body(const SplitArgs& a) {
  constexpr int NS = sp_stages<METHOD>();
  typedef MlRows<D> Ml;
  typedef typename Ml::Own Own;
  constexpr int DP = Ml::DP;
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
    // the state as two packed pairs (Disease, ImmuneReact), (Immunity, Dose2): the stage algebra is 2 instructions wide
    f2 ya, yb;
    {
      const float4 v = *reinterpret_cast<const float4*>(a.y0 + (size_t)p * D);
      ya = pair2(v.x, v.y);
      yb = pair2(v.z, v.w);
      if (live) *reinterpret_cast<float4*>(a.h + (size_t)p * D) = v;
    }
    for (int it = 0; it < a.T; ++it) {
      if (it + 1 < a.T) {
        const SpTimes st(a.t, it, a.perturb, METHOD);
        f2 ka[4], kb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ka[s] = kb[s] = splat2(0.f);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const f2 Ya = sp_stage_state<METHOD>(s, ya, st.dt, ka[0], ka[1], ka[2]);
          const f2 Yb = sp_stage_state<METHOD>(s, yb, st.dt, kb[0], kb[1], kb[2]);
          if (slot < kSplitPatients) *reinterpret_cast<float4*>(&ring[it & 1][s][slot][0]) = make_float4(Ya.x, Ya.y, Yb.x, Yb.y);
          const float Y[4] = {Ya.x, Ya.y, Yb.x, Yb.y};
          float k[4];
          roche_rhs<4, 1, ABLATE, HILL2>(th, none, ds.at(st.ts[s], th.kel).v, Y, k, own1);
          ka[s] = pair2(k[0], k[1]);
          kb[s] = pair2(k[2], k[3]);
        }
        ya = sp_advance<METHOD>(ya, st.dt, ka[0], ka[1], ka[2], ka[3]);
        yb = sp_advance<METHOD>(yb, st.dt, kb[0], kb[1], kb[2], kb[3]);
        if (live) *reinterpret_cast<float4*>(a.h + (size_t)(it + 1) * row + (size_t)p * D) = make_float4(ya.x, ya.y, yb.x, yb.y);
      }
      __syncthreads();
    }
    if (a.status) {
      const bool bad = !(vfinite(ya) && vfinite(yb));
      if (bad && live) atomicOr(a.status, HODE_STATUS_NONFINITE);
    }
  } else {
    // ------------------------------------------------------------------ learned pipeline: a patient per DPP quad
    const int q = lane & 3;
    const int slot = (wave - 1) * 16 + (lane >> 2);
    const bool live = b0 + slot < a.B;
    const int p = min(b0 + slot, a.B - 1);
    Ml ml;
    ml.load(a.w1, a.b1, q);
    Own yo = Ml::load_own(a.y0 + (size_t)p * D, q);
    if (live) Ml::store_own(a.h + (size_t)p * D, q, yo);
    __syncthreads();  // iteration 0: the expert wave fills ring[0]
    for (int it = 1; it < a.T; ++it) {
      const int n = it - 1;
      const float dt = a.t[n + 1] - a.t[n];
      Own k[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) k[s] = vsplat<Own>(0.f);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const Own Yo = sp_stage_state<METHOD>(s, yo, dt, k[0], k[1], k[2]);
        const float4 e = *reinterpret_cast<const float4*>(&ring[n & 1][s][slot][0]);
        f2 Y2[DP];
        Y2[0] = pair2(e.x, e.y);
        Y2[1] = pair2(e.z, e.w);
        Ml::gather(Yo, Y2 + 2);
        k[s] = ml.rhs(Y2);
      }
      yo = sp_advance<METHOD>(yo, dt, k[0], k[1], k[2], k[3]);
      if (live) Ml::store_own(a.h + (size_t)(n + 1) * row + (size_t)p * D, q, yo);
      __syncthreads();
    }
    if (a.status) {
      if (!vfinite(yo) && live) atomicOr(a.status, HODE_STATUS_NONFINITE);
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

// The cotangent algebra of one step of the scheme, shared by the expert wave (V = f2, two pairs) and the learned waves
// (V = Own).  vjp(s, g) must return (df/dY)^T g at stage s.  lam is updated in place.
template <int METHOD, class V, class F>
HODE_DEV void sp_adjoint_step(V& lam, float dt, F&& vjp) {
  constexpr float c13 = (float)(1.0 / 3.0);
  if constexpr (METHOD == HODE_METHOD_EULER) {
    lam += vjp(0, dt * lam);
  } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
    const V a1 = vjp(1, dt * lam);
    lam += a1;
    lam += vjp(0, (0.5f * dt) * a1);
  } else {
    const float w1 = dt * 0.125f, w3 = dt * 0.375f;
    const V a3 = vjp(3, w1 * lam);
    V da = dt * a3;
    V g1 = vfma(w1, lam, da);
    V g2 = vfma(w3, lam, -da);
    const V g = vfma(w3, lam, da);
    lam += a3;
    const V a2 = vjp(2, g);
    da = dt * a2;
    g2 += da;
    g1 = vfma(-c13, da, g1);
    lam += a2;
    const V a1 = vjp(1, g2);
    g1 = vfma(c13, dt * a1, g1);
    lam += a1;
    lam += vjp(0, g1);
  }
}

struct F4 {  // the expert wave's 4 components as two packed pairs
  f2 a, b;
  HODE_DEV F4& operator+=(const F4& o) { a += o.a; b += o.b; return *this; }
  HODE_DEV F4 operator-() const { return F4{-a, -b}; }
};
HODE_DEV F4 operator*(float s, const F4& v) { return F4{s * v.a, s * v.b}; }
HODE_DEV F4 vfma(float s, const F4& x, const F4& y) { return F4{vfma(s, x.a, y.a), vfma(s, x.b, y.b)}; }

template <int D, int METHOD, bool ABLATE, bool HILL2, bool NEED_TH, bool K1>
HODE_DEV void split_bwd_body(const SplitBwdArgs& a) {
  constexpr int NS = sp_stages<METHOD>();
  typedef MlRows<D> Ml;
  typedef typename Ml::Own Own;
  constexpr int M = Ml::M, MR = Ml::MR, DP = Ml::DP, MP = Ml::MP;
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
      const float4 hv = *reinterpret_cast<const float4*>(a.h + (size_t)m * row + (size_t)p * D);
      const f2 ya = pair2(hv.x, hv.y), yb = pair2(hv.z, hv.w);
      f2 ka[4], kb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) ka[s] = kb[s] = splat2(0.f);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const f2 Ya = sp_stage_state<METHOD>(s, ya, st.dt, ka[0], ka[1], ka[2]);
        const f2 Yb = sp_stage_state<METHOD>(s, yb, st.dt, kb[0], kb[1], kb[2]);
        const DoseVal dv = ds.at(st.ts[s], th.kel);
        if (mine) {
          *reinterpret_cast<float4*>(&yring[par][s][slot][0]) = make_float4(Ya.x, Ya.y, Yb.x, Yb.y);
          *reinterpret_cast<float2*>(&dring[par][s][slot][0]) = make_float2(dv.v, dv.dk);
        }
        if (s + 1 < NS) {
          const float Y[4] = {Ya.x, Ya.y, Yb.x, Yb.y};
          float k[4];
          roche_rhs<4, 1, ABLATE, HILL2>(th, none, dv.v, Y, k, own1);
          ka[s] = pair2(k[0], k[1]);
          kb[s] = pair2(k[2], k[3]);
        }
      }
    };

    F4 lam;
    {
      const float4 g4 = *reinterpret_cast<const float4*>(a.grad_h + (size_t)(T - 1) * row + (size_t)p * D);
      lam.a = lv * pair2(g4.x, g4.y);
      lam.b = lv * pair2(g4.z, g4.w);
    }
    if (T >= 2) recompute(T - 2, 0);
    __syncthreads();
    for (int k = 0; k < T; ++k) {
      if (k >= 1) {
        // ---- (b) adjoint of step m = T-1-k
        const int m = T - 1 - k;
        const int par = (k - 1) & 1;
        const float dt = a.t[m + 1] - a.t[m];
        float Y[4][4];
        F4 cs[4];
        DoseVal dv[4];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float4 yv = *reinterpret_cast<const float4*>(&yring[par][s][slot][0]);
          const float4 cv = *reinterpret_cast<const float4*>(&cring[par][s][slot][0]);
          const float2 d2 = *reinterpret_cast<const float2*>(&dring[par][s][slot][0]);
          Y[s][0] = yv.x; Y[s][1] = yv.y; Y[s][2] = yv.z; Y[s][3] = yv.w;
          cs[s].a = lv * pair2(cv.x, cv.y);
          cs[s].b = lv * pair2(cv.z, cv.w);
          dv[s].v = d2.x; dv[s].dk = d2.y;
        }
        sp_adjoint_step<METHOD>(lam, dt, [&](int s, const F4& gs) {
          const float g[4] = {gs.a.x, gs.a.y, gs.b.x, gs.b.y};
          float av[4];
          roche_vjp<4, 1, ABLATE, HILL2, NEED_TH>(th, none, nonec, ln_ec50, dv[s], Y[s], own1, g, 0, av, acc);
          F4 r = cs[s];
          r.a += pair2(av[0], av[1]);
          r.b += pair2(av[2], av[3]);
          return r;
        });
        const float4 g4 = *reinterpret_cast<const float4*>(a.grad_h + (size_t)m * row + (size_t)p * D);
        lam.a = vfma(lv, pair2(g4.x, g4.y), lam.a);
        lam.b = vfma(lv, pair2(g4.z, g4.w), lam.b);
      }
      // ---- (a) stage states of step T-3-k for the learned waves' next iteration
      if (T - 3 - k >= 0) recompute(T - 3 - k, (k + 1) & 1);
      __syncthreads();
    }
    if (live) *reinterpret_cast<float4*>(a.grad_y0 + (size_t)p * D) = make_float4(lam.a.x, lam.a.y, lam.b.x, lam.b.y);
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
    Ml ml;
    ml.load(a.w1, a.b1, q);
    // transposed operands of a = W^T u.  wc2: column q of W (the expert component this lane reports to the expert wave),
    // paired over the rows j;  MR == 2: wtp[j] = (W[j][own0], W[j][own1]) (row-paired, the result is the Own pair);
    // MR == 1: wt2[jp] = (W[2jp][own], W[2jp+1][own]).
    f2 wc2[MP], wtp[MR == 2 ? M : MP];
#pragma unroll
    for (int jp = 0; jp < MP; ++jp) wc2[jp] = pair2(a.w1[(2 * jp) * D + q], a.w1[(2 * jp + 1) * D + q]);
    if constexpr (MR == 2) {
#pragma unroll
      for (int j = 0; j < M; ++j) wtp[j] = pair2(a.w1[j * D + 4 + 2 * q], a.w1[j * D + 4 + 2 * q + 1]);
    } else {
#pragma unroll
      for (int jp = 0; jp < MP; ++jp) wtp[jp] = pair2(a.w1[(2 * jp) * D + 4 + q], a.w1[(2 * jp + 1) * D + 4 + q]);
    }
    f2 dw2[MR][DP];
    Own db = vsplat<Own>(0.f);
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
      for (int ip = 0; ip < DP; ++ip) dw2[r][ip] = splat2(0.f);
    Own lam = lv * Ml::load_own(a.grad_h + (size_t)(T - 1) * row + (size_t)p * D, q);
    // the operands of iteration k+1 are fetched during iteration k (the state is needed by the very first instruction of
    // an iteration: an un-hidden HBM round trip would cost a quarter of it)
    Own yo_nx, gh_nx;
    {
      const int m0 = T >= 2 ? T - 2 : 0;
      yo_nx = Ml::load_own(a.h + (size_t)m0 * row + (size_t)p * D, q);
      gh_nx = Ml::load_own(a.grad_h + (size_t)m0 * row + (size_t)p * D, q);
    }
    __syncthreads();  // the expert wave's prologue fills ring 0
    for (int k = 0; k < T; ++k) {
      if (k <= T - 2) {
        const int m = T - 2 - k;
        const int par = k & 1;
        const float dt = a.t[m + 1] - a.t[m];
        const Own yo = yo_nx, gh = gh_nx;
        {
          const int mn = m >= 1 ? m - 1 : 0;  // clamped: the last prefetch is simply unused
          yo_nx = Ml::load_own(a.h + (size_t)mn * row + (size_t)p * D, q);
          gh_nx = Ml::load_own(a.grad_h + (size_t)mn * row + (size_t)p * D, q);
        }
        // ---- recompute the learned stage derivatives
        f2 Y2[4][DP];
        Own so[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) so[s] = vsplat<Own>(0.f);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const Own Yo = sp_stage_state<METHOD>(s, yo, dt, so[0], so[1], so[2]);
          const float4 e = *reinterpret_cast<const float4*>(&yring[par][s][slot][0]);
          Y2[s][0] = pair2(e.x, e.y);
          Y2[s][1] = pair2(e.z, e.w);
          Ml::gather(Yo, Y2[s] + 2);
          so[s] = ml.rhs(Y2[s]);
        }
        // ---- adjoint of the stages
        sp_adjoint_step<METHOD>(lam, dt, [&](int s, Own gs) {
          const Own u = gs * vfma(-so[s], so[s], vsplat<Own>(1.0f));
          db += u;
          f2 uf2[MP];
          Ml::gather(u, uf2);
          if constexpr (MR == 2) {
#pragma unroll
            for (int ip = 0; ip < DP; ++ip) {
              dw2[0][ip] = vfma(u.x, Y2[s][ip], dw2[0][ip]);
              dw2[1][ip] = vfma(u.y, Y2[s][ip], dw2[1][ip]);
            }
          } else {
#pragma unroll
            for (int ip = 0; ip < DP; ++ip) dw2[0][ip] = vfma(u, Y2[s][ip], dw2[0][ip]);
          }
          f2 c2 = splat2(0.f);
#pragma unroll
          for (int jp = 0; jp < MP; ++jp) c2 = vfma(wc2[jp], uf2[jp], c2);
          cring[par][s][slot][q] = hsum(c2);
          if constexpr (MR == 2) {
            f2 av = splat2(0.f);
#pragma unroll
            for (int jp = 0; jp < MP; ++jp) {
              av = vfma(wtp[2 * jp], uf2[jp].x, av);
              av = vfma(wtp[2 * jp + 1], uf2[jp].y, av);
            }
            return av;
          } else {
            f2 a2 = splat2(0.f);
#pragma unroll
            for (int jp = 0; jp < MP; ++jp) a2 = vfma(wtp[jp], uf2[jp], a2);
            return hsum(a2);
          }
        });
        lam = vfma(lv, gh, lam);
      }
      __syncthreads();
    }
    if (live) Ml::store_own(a.grad_y0 + (size_t)p * D, q, lam);
    float* out = a.part_ml + ((size_t)blockIdx.x * 3 + (wave - 1)) * (M * D + M);
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const float v = wave_sum_stride4((i & 1) ? dw2[r][i / 2].y : dw2[r][i / 2].x);
        if (lane < 4) out[(lane * MR + r) * D + i] = v;
      }
      float dbr;
      if constexpr (MR == 2) dbr = r == 0 ? db.x : db.y;
      else dbr = db;
      const float vb = wave_sum_stride4(dbr);
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
