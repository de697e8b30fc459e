// Right-hand side of the hybrid "Roche" ODE and its vector-Jacobian product, as register-resident device code.
//
// Reference arithmetic: RocheODE.forward (model.py:515-555), dose_at_time (model.py:509-513); restated for the
// CPU in oracle/rhs.py::RocheRHS.  State layout y = [Disease, ImmuneReact, Immunity, Dose2, learned...].
//
// Lane layout.  A patient is handled by LPP lanes (LPP = 1 or 4).  Every lane of a patient holds the FULL state
// (D registers) and evaluates the 4 expert components redundantly; the M = D-4 rows of the learned block
// tanh(W y + b) are split over the lanes (MR = M/LPP rows each, W rows kept in that lane's registers) and
// all-gathered with DPP quad broadcasts.  With LPP = 4 a wave covers 16 patients, which is what fills the chip
// at the 10k-patient shape; with LPP = 1 a wave covers 64 patients and the total instruction count is lowest.
#pragma once
#include "hode_common.hpp"

namespace hode {

struct RocheTheta {  // order = reference parameter creation order (model.py:468-485) = include/hode.h HODE_TH_*
  float hc, hp, ec50, emax, kdexa, kcir, kci, kprog, kid, kfb, koff, kim, kel, th1, th2;
};
constexpr int kNTheta = 15;

HODE_DEV RocheTheta load_theta(const float* __restrict__ th, bool ablate) {
  RocheTheta r;
  r.hc = th[0]; r.hp = th[1]; r.ec50 = th[2]; r.emax = th[3]; r.kdexa = th[4]; r.kcir = th[5]; r.kci = th[6];
  r.kprog = th[7]; r.kid = th[8]; r.kfb = th[9]; r.koff = th[10]; r.kim = th[11]; r.kel = th[12];
  r.th1 = ablate ? th[13] : 0.f;
  r.th2 = ablate ? th[14] : 0.f;
  return r;
}

// per-lane slice of ml_net.0: rows [q*MR, (q+1)*MR) of W (M x D, row-major as nn.Linear stores it) and of b
template <int D, int LPP>
struct MlSlice {
  static constexpr int M = D - 4;
  static constexpr int MR = (M / LPP) > 0 ? (M / LPP) : 1;
  float w[MR][D];
  float b[MR];
  HODE_DEV void load(const float* __restrict__ W, const float* __restrict__ bias, int q) {
    if constexpr (M > 0) {
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        const int row = q * MR + r;
#pragma unroll
        for (int i = 0; i < D; ++i) w[r][i] = W[row * D + i];
        b[r] = bias[row];
      }
    }
  }
};

// per-lane COLUMN slice of W for the transposed product a = W^T u of the VJP: columns [q*DC, (q+1)*DC).
// Only the quad layout needs it (with LPP = 1 the lane owns all of W already).
template <int D, int LPP>
struct MlColSlice {
  static constexpr int M = D - 4;
  static constexpr int DC = (LPP > 1 && M > 0) ? D / LPP : 1;
  static constexpr int MM = (LPP > 1 && M > 0) ? M : 1;
  float wt[MM][DC];
  HODE_DEV void load(const float* __restrict__ W, int q) {
    if constexpr (LPP > 1 && M > 0) {
      static_assert(D % LPP == 0, "quad layout needs D % 4 == 0");
#pragma unroll
      for (int j = 0; j < M; ++j)
#pragma unroll
        for (int c = 0; c < DC; ++c) wt[j][c] = W[j * D + q * DC + c];
    }
  }
};

// select v[q*MR + r] for a run-time quad position q out of compile-time indexed registers
template <int LPP, int MR, int N>
HODE_DEV float pick_own(const float (&v)[N], int base, int r, int q) {
  float out = v[base + r];
  if constexpr (LPP > 1) {
#pragma unroll
    for (int qq = 1; qq < LPP; ++qq) out = (q == qq) ? v[base + qq * MR + r] : out;
  }
  return out;
}

// all-gather the per-lane rows into the full learned block of k (k[4..D))
template <int D, int LPP>
HODE_DEV void gather_rows(const float (&own)[MlSlice<D, LPP>::MR], float (&k)[D]) {
  constexpr int MR = MlSlice<D, LPP>::MR;
  if constexpr (D > 4) {
    if constexpr (LPP == 1) {
#pragma unroll
      for (int r = 0; r < MR; ++r) k[4 + r] = own[r];
    } else {
      static_assert(LPP == 4, "only quad layout implemented");
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        k[4 + 0 * MR + r] = quad_bcast<0>(own[r]);
        k[4 + 1 * MR + r] = quad_bcast<1>(own[r]);
        k[4 + 2 * MR + r] = quad_bcast<2>(own[r]);
        k[4 + 3 * MR + r] = quad_bcast<3>(own[r]);
      }
    }
  }
}

// dose schedule of one patient: Dose(t) = dosage * sum_k 1[t >= tau_k] exp(kel (tau_k - t))   (model.py:509-513)
struct DoseVal {
  float v;   // Dose(t)
  float dk;  // d Dose / d kel
};
template <bool K1>
struct DoseSched {
  float dosage;
  float tau0;         // K == 1 (one dose per patient: every shipped synthetic configuration)
  const float* taus;  // [K] for this patient in global memory, walked when K != 1
  int K;
  HODE_DEV DoseVal at(float t, float kel) const {
    DoseVal r;
    if constexpr (K1) {
      const float d = tau0 - t;
      const float v = (t >= tau0) ? dosage * exp_f32(kel * d) : 0.0f;
      r.v = v;
      r.dk = d * v;
    } else {
      float s = 0.f, sk = 0.f;
      for (int k = 0; k < K; ++k) {
        const float tau = taus[k];
        const float d = tau - t;
        const float e = (t >= tau) ? exp_f32(kel * d) : 0.0f;
        s += e;
        sk = __builtin_fmaf(d, e, sk);
      }
      r.v = dosage * s;
      r.dk = dosage * sk;
    }
    return r;
  }
};

// x ** p with torch.pow semantics (negative base: finite for integer p, NaN otherwise); p == 2 is the shipped value
template <bool HILL2>
HODE_DEV float pow_hill(float x, float p) {
  if constexpr (HILL2) return x * x;
  else return powf(x, p);
}
// d/dx x**p = p x**(p-1)   (torch pow_backward_self: zero where p == 0)
template <bool HILL2>
HODE_DEV float dpow_dx(float x, float p) {
  if constexpr (HILL2) return 2.0f * x;
  else return p == 0.0f ? 0.0f : p * powf(x, p - 1.0f);
}
// d/dp x**p = x**p log x   (torch pow_backward_exponent: zero where x == 0 and p >= 0)
HODE_DEV float dpow_dp(float x, float p, float xp) { return (x == 0.0f && p >= 0.0f) ? 0.0f : xp * log_f32(x); }

// k = f(t, Y).  `own` receives this lane's tanh outputs (needed again by the VJP).
template <int D, int LPP, bool ABLATE, bool HILL2>
HODE_DEV void roche_rhs(const RocheTheta& th, const MlSlice<D, LPP>& ml, float dose, const float (&Y)[D], float (&k)[D],
                        float (&own)[MlSlice<D, LPP>::MR]) {
  constexpr int MR = MlSlice<D, LPP>::MR;
  const float dis = Y[0], ir = Y[1], imm = Y[2], d2 = Y[3];
  if constexpr (!ABLATE) {
    const float immp = pow_hill<HILL2>(imm, th.hc);
    const float irp = pow_hill<HILL2>(ir, th.hp);
    const float ecp = pow_hill<HILL2>(th.ec50, th.hp);
    k[0] = dis * th.kprog - dis * immp * th.kci - dis * ir * th.kcir;
    k[1] = dis * th.kid - ir * th.koff + dis * ir * th.kfb + div_f32(irp * th.emax, ecp + irp) - d2 * ir * th.kdexa;
    k[2] = ir * th.kim;
    k[3] = th.kel * dose - th.kel * d2;
  } else {  // model.py:545-549
    k[0] = ir;
    k[1] = -1.0f * dis * th.th1;
    k[2] = d2;
    k[3] = -1.0f * imm * th.th2;
  }
  if constexpr (D > 4) {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      float z = ml.b[r];
#pragma unroll
      for (int i = 0; i < D; ++i) z = __builtin_fmaf(ml.w[r][i], Y[i], z);
      own[r] = tanh_f32(z);
    }
    gather_rows<D, LPP>(own, k);
  }
}

// accumulators of the parameter gradient held by one lane
template <int D, int LPP>
struct GradAcc {
  static constexpr int MR = MlSlice<D, LPP>::MR;
  float dw[MR][D];
  float db[MR];
  float dth[kNTheta];
  HODE_DEV void zero() {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int i = 0; i < D; ++i) dw[r][i] = 0.f;
      db[r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < kNTheta; ++i) dth[i] = 0.f;
  }
};

// Vector-Jacobian product of the rhs at (t, Y): a = (df/dY)^T g, and parameter-gradient accumulation.
//   own_s : this lane's tanh outputs at Y (from roche_rhs), g : cotangent of k (full, all lanes), q : quad position.
// Quad layout: u is all-gathered (M DPP moves), every lane forms D/4 entries of W^T u from its column slice
// (M*D/4 fmas) and the entries are all-gathered again (D DPP moves) -- cheaper than a D-wide quad all-reduce.
template <int D, int LPP, bool ABLATE, bool HILL2, bool NEED_TH>
HODE_DEV void roche_vjp(const RocheTheta& th, const MlSlice<D, LPP>& ml, const MlColSlice<D, LPP>& mc, float ln_ec50,
                        DoseVal dose, const float (&Y)[D], const float (&own_s)[MlSlice<D, LPP>::MR],
                        const float (&g)[D], int q, float (&a)[D], GradAcc<D, LPP>& acc) {
  constexpr int MR = MlSlice<D, LPP>::MR;
  constexpr int M = D - 4;
  // ---- learned block: u_r = g_r (1 - s_r^2)
  if constexpr (D > 4) {
    float u[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      const float gr = pick_own<LPP, MR, D>(g, 4, r, q);
      u[r] = gr * __builtin_fmaf(-own_s[r], own_s[r], 1.0f);
      acc.db[r] += u[r];
#pragma unroll
      for (int i = 0; i < D; ++i) acc.dw[r][i] = __builtin_fmaf(u[r], Y[i], acc.dw[r][i]);
    }
    if constexpr (LPP == 1) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        float p = 0.f;
#pragma unroll
        for (int r = 0; r < MR; ++r) p = __builtin_fmaf(ml.w[r][i], u[r], p);
        a[i] = p;
      }
    } else {
      constexpr int DC = MlColSlice<D, LPP>::DC;
      float uf[M];
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        uf[0 * MR + r] = quad_bcast<0>(u[r]);
        uf[1 * MR + r] = quad_bcast<1>(u[r]);
        uf[2 * MR + r] = quad_bcast<2>(u[r]);
        uf[3 * MR + r] = quad_bcast<3>(u[r]);
      }
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < M; ++j) p = __builtin_fmaf(mc.wt[j][c], uf[j], p);
        a[0 * DC + c] = quad_bcast<0>(p);
        a[1 * DC + c] = quad_bcast<1>(p);
        a[2 * DC + c] = quad_bcast<2>(p);
        a[3 * DC + c] = quad_bcast<3>(p);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i) a[i] = 0.f;
  }
  // ---- expert block (evaluated redundantly by every lane of the patient)
  const float dis = Y[0], ir = Y[1], imm = Y[2], d2 = Y[3];
  const float g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
  if constexpr (!ABLATE) {
    const float immp = pow_hill<HILL2>(imm, th.hc);
    const float irp = pow_hill<HILL2>(ir, th.hp);
    const float ecp = pow_hill<HILL2>(th.ec50, th.hp);
    const float rden = __builtin_amdgcn_rcpf(ecp + irp);
    const float dirp = dpow_dx<HILL2>(ir, th.hp);
    const float g0d = g0 * dis, g1d = g1 * dis, g1i = g1 * ir;
    const float er2 = th.emax * rden * rden;  // emax / (E + P)^2
    a[0] += g0 * (th.kprog - immp * th.kci - ir * th.kcir) + g1 * (th.kid + ir * th.kfb);
    a[1] += g1 * (dis * th.kfb - th.koff + er2 * ecp * dirp - d2 * th.kdexa) - g0d * th.kcir + g2 * th.kim;
    a[2] -= g0d * th.kci * dpow_dx<HILL2>(imm, th.hc);
    a[3] -= g1i * th.kdexa + g3 * th.kel;
    if constexpr (NEED_TH) {
      acc.dth[0] -= g0d * th.kci * dpow_dp(imm, th.hc, immp);
      const float dP = dpow_dp(ir, th.hp, irp);
      const float dE = (th.ec50 == 0.0f && th.hp >= 0.0f) ? 0.0f : ecp * ln_ec50;
      acc.dth[1] += g1 * er2 * (dP * ecp - irp * dE);
      acc.dth[2] -= g1 * er2 * irp * dpow_dx<HILL2>(th.ec50, th.hp);
      acc.dth[3] += g1 * irp * rden;
      acc.dth[4] -= g1i * d2;
      acc.dth[5] -= g0d * ir;
      acc.dth[6] -= g0d * immp;
      acc.dth[7] += g0d;
      acc.dth[8] += g1d;
      acc.dth[9] += g1d * ir;
      acc.dth[10] -= g1i;
      acc.dth[11] += g2 * ir;
      acc.dth[12] += g3 * ((dose.v - d2) + th.kel * dose.dk);
    }
  } else {
    a[0] -= th.th1 * g1;
    a[1] += g0;
    a[2] -= th.th2 * g3;
    a[3] += g2;
    if constexpr (NEED_TH) {
      acc.dth[13] -= dis * g1;
      acc.dth[14] -= imm * g3;
    }
  }
}

// The parameter-gradient half of roche_vjp's expert block on its own (same expressions, same order): the split layout's
// adjoint runs it on a separate wave, off the cotangent chain (hode_rk_split.hip).
template <bool ABLATE, bool HILL2>
HODE_DEV void roche_theta_grad(const RocheTheta& th, float ln_ec50, DoseVal dose, const float (&Y)[4], const float (&g)[4],
                               float (&dth)[kNTheta]) {
  const float dis = Y[0], ir = Y[1], imm = Y[2], d2 = Y[3];
  const float g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
  if constexpr (!ABLATE) {
    const float immp = pow_hill<HILL2>(imm, th.hc);
    const float irp = pow_hill<HILL2>(ir, th.hp);
    const float ecp = pow_hill<HILL2>(th.ec50, th.hp);
    const float rden = __builtin_amdgcn_rcpf(ecp + irp);
    const float g0d = g0 * dis, g1d = g1 * dis, g1i = g1 * ir;
    const float er2 = th.emax * rden * rden;  // emax / (E + P)^2
    dth[0] -= g0d * th.kci * dpow_dp(imm, th.hc, immp);
    const float dP = dpow_dp(ir, th.hp, irp);
    const float dE = (th.ec50 == 0.0f && th.hp >= 0.0f) ? 0.0f : ecp * ln_ec50;
    dth[1] += g1 * er2 * (dP * ecp - irp * dE);
    dth[2] -= g1 * er2 * irp * dpow_dx<HILL2>(th.ec50, th.hp);
    dth[3] += g1 * irp * rden;
    dth[4] -= g1i * d2;
    dth[5] -= g0d * ir;
    dth[6] -= g0d * immp;
    dth[7] += g0d;
    dth[8] += g1d;
    dth[9] += g1d * ir;
    dth[10] -= g1i;
    dth[11] += g2 * ir;
    dth[12] += g3 * ((dose.v - d2) + th.kel * dose.dk);
  } else {
    dth[13] -= dis * g1;
    dth[14] -= imm * g3;
  }
}

}  // namespace hode
