// "MFMA layout" of the fixed-grid Roche solve and its discrete adjoint, gfx950 (D in {8, 12, 16}).
//
// Same arithmetic and C-ABI contract as hode_rk_kernels.hpp (reference model.py:515-555, :1116; oracle/rhs.py), different
// mapping to the hardware.  Measurements that drive it (tools/micro/valu_rate.hip, profiles/r01_pmc_summary.json):
// the quad-layout kernels are VALU-ISSUE bound (5-6 cycles per wave-instruction, VALU active 66-82 % of the wave's
// lifetime), a SIMD does not overlap VALU work of several waves, and at 10 000 patients there is at most one wave per
// SIMD -- so the only lever is FEWER VALU INSTRUCTIONS PER WAVE, and the matrix pipe is idle.
//
//   * a wave holds 16 patients; patient p lives in the 4 lanes {p, p+16, p+32, p+48}; lane (g = lane>>4, p = lane&15)
//     keeps components 4g .. 4g+3 of the state in 4 registers (g = 0: the expert block, g >= 1: learned latents).
//     Every per-component update (RK stage algebra, adjoint algebra) therefore costs 4 instructions, not D.
//   * the rhs is written as  f(y) = W_ext y + b + nonlinear corrections:  W_ext (16 x 16) stacks the LINEAR expert terms
//     (k_disprog, k_immune_disease, -k_immune_off, k_immunity, -kel) on top of ml_net's W; the product runs on the
//     matrix pipe as v_mfma_f32_16x16x4_f32 (exact fp32), whose 16x16 result layout (row = 4*(lane>>4) + reg,
//     col = lane&15) IS the state layout.  The B operand wants component 4s + g in lane group g for k-quad s: a 4x4
//     transpose across the patient's 4 lanes = two v_permlane32_swap + two v_permlane16_swap.
//   * the remaining nonlinear expert terms (products, Hill term, dose) involve only components 0..3, i.e. registers of
//     ONE lane (group 0): no cross-lane traffic; groups >= 1 apply tanh to their 4 rows.
//   * backward: W_ext^T u is the same MFMA with transposed weight fragments; the weight gradient dW_ext = sum over
//     patients of u y^T is an MFMA over the PATIENT axis (K = 16 patients of the wave), fed from two 1 KiB LDS images
//     that re-lay u and y patient-minor; it accumulates in 4 registers over all steps and stages, and the cross-patient
//     sum comes for free.  The gradients of the linear expert constants are read off dW_ext at the end.
#include <hip/hip_runtime.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"
#include "hode_roche.hpp"

namespace hode {

typedef float mf4 __attribute__((ext_vector_type(4)));
typedef unsigned mfu2 __attribute__((ext_vector_type(2)));

// Both swaps are written as inline asm: hipcc 7.2 miscompiles __builtin_amdgcn_permlane{16,32}_swap (the second result
// is replaced by the first -- every MFMA that follows reads the same register; and with wave-uniform inputs it emits the
// swap on SGPR operands).  hipcc pads nothing inside an asm statement, so the VALU-write -> permlane-read and
// permlane-write -> next-VALU-read wait states are inside the string.
HODE_DEV void swap32(float& a, float& b) {  // a[lanes 32..63] <-> b[lanes 0..31]
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
HODE_DEV void swap16(float& a, float& b) {  // odd 16-lane rows of a <-> even rows of b
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// v[s] of lane group g  <-  v[g] of lane group s   (4x4 transpose across the 4 lanes of a patient)
HODE_DEV void transpose4(float (&v)[4]) {
  swap32(v[0], v[2]);
  swap32(v[1], v[3]);
  swap16(v[0], v[1]);
  swap16(v[2], v[3]);
}

constexpr int kMfSlots = 9;   // nonlinear expert-constant gradient slots (see mf_vjp)
constexpr int kMfPartials = 256 + 16 + kMfSlots;

struct MfConst {
  float wA[4];    // forward A fragments:  W_ext[row = lane&15][col = 4s + (lane>>4)]
  float wT[4];    // backward A fragments: W_ext[row = 4s + (lane>>4)][col = lane&15]
  float bias[4];  // bias of the 4 components this lane owns
};

// entry (row, col) of the extended matrix: ml_net.0.weight below the linear expert coefficients
template <int D, bool ABLATE>
HODE_DEV float w_ext(const RocheTheta& th, const float* __restrict__ W, int row, int col) {
  if (row >= D || col >= D) return 0.f;
  if (row >= 4) return W[(row - 4) * D + col];
  if constexpr (ABLATE) {
    if (row == 0) return col == 1 ? 1.0f : 0.f;
    if (row == 1) return col == 0 ? -th.th1 : 0.f;
    if (row == 2) return col == 3 ? 1.0f : 0.f;
    return col == 2 ? -th.th2 : 0.f;
  } else {
    if (row == 0) return col == 0 ? th.kprog : 0.f;
    if (row == 1) return col == 0 ? th.kid : (col == 1 ? -th.koff : 0.f);
    if (row == 2) return col == 1 ? th.kim : 0.f;
    return col == 3 ? -th.kel : 0.f;
  }
}

template <int D, bool ABLATE>
HODE_DEV MfConst mf_load_const(const RocheTheta& th, const float* __restrict__ W, const float* __restrict__ b, int lane) {
  MfConst c;
  const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    c.wA[s] = w_ext<D, ABLATE>(th, W, lo, 4 * s + hi);
    c.wT[s] = w_ext<D, ABLATE>(th, W, 4 * s + hi, lo);
    const int comp = 4 * hi + s;
    c.bias[s] = (comp >= 4 && comp < D) ? b[comp - 4] : 0.f;
  }
  return c;
}

// k = f(t, y) for the 4 components of this lane.  `tz` receives tanh(z) (learned lanes) for the VJP.
template <int D, bool ABLATE, bool HILL2>
HODE_DEV void mf_rhs(const RocheTheta& th, const MfConst& c, bool expert_lane, float dose, const float (&y)[4], float (&k)[4]) {
  constexpr int NG = D / 4;
  float yt[4] = {y[0], y[1], y[2], y[3]};
  transpose4(yt);
  mf4 acc = {c.bias[0], c.bias[1], c.bias[2], c.bias[3]};
#pragma unroll
  for (int s = 0; s < NG; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.wA[s], yt[s], acc, 0, 0, 0);
  float e[4] = {acc[0], acc[1], acc[2], acc[3]};
  if constexpr (!ABLATE) {
    const float dis = y[0], ir = y[1], imm = y[2], d2 = y[3];
    const float p1 = dis * ir;
    const float immp = pow_hill<HILL2>(imm, th.hc);
    const float irp = pow_hill<HILL2>(ir, th.hp);
    const float ecp = pow_hill<HILL2>(th.ec50, th.hp);
    e[0] = e[0] - th.kcir * p1 - th.kci * (dis * immp);
    e[1] = e[1] + th.kfb * p1 + div_f32(irp * th.emax, ecp + irp) - th.kdexa * (d2 * ir);
    e[3] = __builtin_fmaf(th.kel, dose, e[3]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) k[r] = expert_lane ? e[r] : tanh_f32(acc[r]);
}

struct MfArgs {
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
  float* __restrict__ partials;  // [n_waves][kMfPartials]
  int* __restrict__ status;
  int B, T, K, perturb;
};

struct MfLane {
  int g, pc, p;
  bool live, expert;
  template <int D>
  HODE_DEV void init(int B) {
    const int lane = threadIdx.x & 63;
    g = lane >> 4;
    pc = lane & 15;
    const int pp = blockIdx.x * 16 + pc;
    live = pp < B && g < D / 4;
    p = pp < B ? pp : B - 1;
    expert = g == 0;
  }
};

template <bool K1>
HODE_DEV DoseSched<K1> mf_load_dose(const MfArgs& a, int p) {
  DoseSched<K1> ds;
  ds.dosage = a.dosage[p];
  ds.K = a.K;
  ds.taus = a.dose_times + (size_t)p * a.K;
  ds.tau0 = K1 ? ds.taus[0] : 0.f;
  return ds;
}

struct MfTimes {
  float t0, t1, dt, ta, tb, t_first, t_last;
  HODE_DEV MfTimes(const float* __restrict__ t, int n, int perturb, int method) {
    t0 = t[n];
    t1 = t[n + 1];
    dt = t1 - t0;
    t_first = perturb ? nextafter_up(t0) : t0;
    t_last = perturb ? nextafter_down(t1) : t1;
    if (method == HODE_METHOD_RK4_38) {
      ta = add_rn(t0, mul_rn(dt, (float)(1.0 / 3.0)));
      tb = add_rn(t0, mul_rn(dt, (float)(2.0 / 3.0)));
    } else {
      ta = add_rn(t0, mul_rn(0.5f, dt));
      tb = ta;
    }
  }
};

HODE_DEV void mf_load4(const float* __restrict__ p, float (&v)[4]) {
  const float4 x = *reinterpret_cast<const float4*>(p);
  v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
}
HODE_DEV void mf_store4(float* __restrict__ p, const float (&v)[4], bool live) {
  if (live) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

constexpr float kC13 = (float)(1.0 / 3.0);

// ------------------------------------------------------------------------------------------------ forward
template <int D, int METHOD, bool ABLATE, bool HILL2, bool K1>
HODE_DEV void mf_fwd_body(const MfArgs& a) {
  MfLane ln;
  ln.init<D>(a.B);
  const RocheTheta th = load_theta(a.theta, ABLATE);
  const MfConst c = mf_load_const<D, ABLATE>(th, a.w1, a.b1, threadIdx.x & 63);
  const DoseSched<K1> ds = mf_load_dose<K1>(a, ln.p);
  const size_t row = (size_t)a.B * D;
  const size_t off = (size_t)ln.p * D + 4 * (ln.g < D / 4 ? ln.g : 0);
  float y[4] = {0.f, 0.f, 0.f, 0.f};
  if (ln.g < D / 4) mf_load4(a.y0 + off, y);
  float* hp = a.h + off;
  mf_store4(hp, y, ln.live);
  for (int n = 0; n + 1 < a.T; ++n) {
    const MfTimes st(a.t, n, a.perturb, METHOD);
    const float dt = st.dt;
    float k1[4];
    mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, ds.at(st.t_first, th.kel).v, y, k1);
    if constexpr (METHOD == HODE_METHOD_EULER) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf(dt, k1[r], y[r]);
    } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
      float Y[4], k2[4];
      const float half = 0.5f * dt;
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[r] = __builtin_fmaf(k1[r], half, y[r]);
      mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, ds.at(st.ta, th.kel).v, Y, k2);
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf(dt, k2[r], y[r]);
    } else {
      float Y[4], k2[4], k3[4], k4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[r] = __builtin_fmaf(dt * k1[r], kC13, y[r]);
      mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, ds.at(st.ta, th.kel).v, Y, k2);
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[r] = __builtin_fmaf(dt, __builtin_fmaf(-k1[r], kC13, k2[r]), y[r]);
      mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, ds.at(st.tb, th.kel).v, Y, k3);
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[r] = __builtin_fmaf(dt, (k1[r] - k2[r]) + k3[r], y[r]);
      mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, ds.at(st.t_last, th.kel).v, Y, k4);
      const float w = dt * 0.125f;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf((k1[r] + 3.0f * (k2[r] + k3[r])) + k4[r], w, y[r]);
    }
    hp += row;
    mf_store4(hp, y, ln.live);
  }
  if (a.status) {
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) bad |= !__builtin_isfinite(y[r]);
    if (bad && ln.live) atomicOr(a.status, HODE_STATUS_NONFINITE);
  }
}

template <int D, int METHOD, bool ABLATE>
__global__ __launch_bounds__(64) void mf_fwd_kernel(MfArgs a) {
  const bool hill2 = ABLATE || (a.theta[0] == 2.0f && a.theta[1] == 2.0f);
  if (hill2 && a.K == 1) mf_fwd_body<D, METHOD, ABLATE, true, true>(a);
  else if (hill2) mf_fwd_body<D, METHOD, ABLATE, true, false>(a);
  else mf_fwd_body<D, METHOD, ABLATE, false, false>(a);
}

// ------------------------------------------------------------------------------------------------ backward
// LDS images: [comp 0..15][16 floats], patient p stored at position (p & 3) * 4 + (p >> 2), so that the MFMA fragment of
// lane (gk = lane>>4, c = lane&15) -- element (row c, patient 4s + gk) for k-quad s = 0..3 -- is ONE ds_read_b128.
HODE_DEV int mf_pos(int pc) { return (pc & 3) * 4 + (pc >> 2); }

struct MfAcc {
  mf4 dW;                // dW_ext[row 4*(lane>>4) + r][col lane&15]
  float db[4];           // bias gradient of the 4 owned components (summed over the wave's patients at the end)
  float slot[kMfSlots];  // nonlinear expert-constant gradients, meaningful on group-0 lanes:
                         // 0 kcir, 1 kci, 2 HillCure, 3 kfb, 4 emax, 5 kdexa, 6 HillPatho, 7 ec50, 8 kel (dose part)
};

// VJP at one stage.  u_out = cotangent of the pre-activations (what the weight gradient contracts with Y).
template <int D, bool ABLATE, bool HILL2, bool NEED_TH>
HODE_DEV void mf_vjp(const RocheTheta& th, const MfConst& c, bool expert_lane, float ln_ec50, DoseVal dose, const float (&Y)[4],
                     const float (&kout)[4], const float (&gk)[4], float (&av)[4], float (&u)[4], MfAcc& acc) {
  constexpr int NG = D / 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    u[r] = expert_lane ? gk[r] : gk[r] * __builtin_fmaf(-kout[r], kout[r], 1.0f);
    acc.db[r] += u[r];
  }
  float ut[4] = {u[0], u[1], u[2], u[3]};
  transpose4(ut);
  mf4 av4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NG; ++s) av4 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.wT[s], ut[s], av4, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) av[r] = av4[r];
  if constexpr (!ABLATE) {
    // group-0 lanes hold the expert state in their own registers; other lanes run the same instructions on learned
    // latents and discard the result through selects (never through a multiply: their values may be NaN for general Hill)
    const float g0 = gk[0], g1 = gk[1], g3 = gk[3];
    const float dis = Y[0], ir = Y[1], imm = Y[2], d2 = Y[3];
    const float immp = pow_hill<HILL2>(imm, th.hc);
    const float irp = pow_hill<HILL2>(ir, th.hp);
    const float ecp = pow_hill<HILL2>(th.ec50, th.hp);
    const float rden = __builtin_amdgcn_rcpf(ecp + irp);
    const float er2 = th.emax * rden * rden;
    const float dirp = dpow_dx<HILL2>(ir, th.hp);
    const float p1 = dis * ir;
    const float i0 = g0 * (-th.kcir * ir - th.kci * immp) + g1 * (th.kfb * ir);
    const float i1 = g0 * (-th.kcir * dis) + g1 * (th.kfb * dis + er2 * ecp * dirp - th.kdexa * d2);
    const float i2 = g0 * (-th.kci * dis * dpow_dx<HILL2>(imm, th.hc));
    const float i3 = g1 * (-th.kdexa * ir);
    av[0] += expert_lane ? i0 : 0.f;
    av[1] += expert_lane ? i1 : 0.f;
    av[2] += expert_lane ? i2 : 0.f;
    av[3] += expert_lane ? i3 : 0.f;
    if constexpr (NEED_TH) {
      const float dP = dpow_dp(ir, th.hp, irp);
      const float dE = (th.ec50 == 0.0f && th.hp >= 0.0f) ? 0.0f : ecp * ln_ec50;
      const float v[kMfSlots] = {-g0 * p1,
                                 -g0 * dis * immp,
                                 -g0 * dis * th.kci * dpow_dp(imm, th.hc, immp),
                                 g1 * p1,
                                 g1 * irp * rden,
                                 -g1 * d2 * ir,
                                 g1 * er2 * (dP * ecp - irp * dE),
                                 -g1 * er2 * irp * dpow_dx<HILL2>(th.ec50, th.hp),
                                 g3 * __builtin_fmaf(th.kel, dose.dk, dose.v)};
#pragma unroll
      for (int i = 0; i < kMfSlots; ++i) acc.slot[i] += expert_lane ? v[i] : 0.f;
    }
  }
}

// dW_ext += U Y^T over the 16 patients of the wave (both operands re-laid patient-minor through LDS)
HODE_DEV void mf_dw(float* __restrict__ ldsU, const float* __restrict__ ldsY, const float (&u)[4], int g, int pc, int lane, MfAcc& acc) {
  const int pos = mf_pos(pc);
#pragma unroll
  for (int r = 0; r < 4; ++r) ldsU[(4 * g + r) * 16 + pos] = u[r];
  __syncthreads();
  const mf4 ua = *reinterpret_cast<const mf4*>(ldsU + (lane & 15) * 16 + (lane >> 4) * 4);
  const mf4 yb = *reinterpret_cast<const mf4*>(ldsY + (lane & 15) * 16 + (lane >> 4) * 4);
#pragma unroll
  for (int s = 0; s < 4; ++s) acc.dW = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[s], yb[s], acc.dW, 0, 0, 0);
  __syncthreads();
}

template <int D, int METHOD, bool ABLATE, bool HILL2, bool NEED_TH, bool K1>
HODE_DEV void mf_bwd_body(const MfArgs& a) {
  __shared__ __attribute__((aligned(16))) float lds[5 * 256];  // Y images of up to 4 stages + one U image
  MfLane ln;
  ln.init<D>(a.B);
  const int lane = threadIdx.x & 63;
  const RocheTheta th = load_theta(a.theta, ABLATE);
  const MfConst c = mf_load_const<D, ABLATE>(th, a.w1, a.b1, lane);
  const DoseSched<K1> ds = mf_load_dose<K1>(a, ln.p);
  const float ln_ec50 = log_f32(th.ec50);
  MfAcc acc;
  acc.dW = mf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) acc.db[r] = 0.f;
#pragma unroll
  for (int i = 0; i < kMfSlots; ++i) acc.slot[i] = 0.f;
  float* ldsU = lds + 4 * 256;
  const int pos = mf_pos(ln.pc);
  auto image_y = [&](int stage, const float (&Y)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[stage * 256 + (4 * ln.g + r) * 16 + pos] = Y[r];
  };

  const size_t row = (size_t)a.B * D;
  const bool has = ln.g < D / 4;
  const size_t off = (size_t)ln.p * D + 4 * (has ? ln.g : 0);
  const float lv = ln.live ? 1.0f : 0.0f;
  const float* hp = a.h + (size_t)(a.T - 1) * row + off;
  const float* gp = a.grad_h + (size_t)(a.T - 1) * row + off;
  float lam[4] = {0.f, 0.f, 0.f, 0.f};
  if (has) mf_load4(gp, lam);
#pragma unroll
  for (int r = 0; r < 4; ++r) lam[r] *= lv;
  float y[4] = {0.f, 0.f, 0.f, 0.f}, gh[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.T > 1 && has) {
    mf_load4(hp - row, y);
    mf_load4(gp - row, gh);
  }
  for (int n = a.T - 2; n >= 0; --n) {
    hp -= row;
    gp -= row;
    float y_nx[4] = {0.f, 0.f, 0.f, 0.f}, gh_nx[4] = {0.f, 0.f, 0.f, 0.f};
    if (n > 0 && has) {
      mf_load4(hp - row, y_nx);
      mf_load4(gp - row, gh_nx);
    }
    const MfTimes st(a.t, n, a.perturb, METHOD);
    const float dt = st.dt;
    float k1[4], av[4], u[4], g[4];
    const DoseVal d1 = ds.at(st.t_first, th.kel);
    mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, d1.v, y, k1);
    image_y(0, y);
    if constexpr (METHOD == HODE_METHOD_EULER) {
#pragma unroll
      for (int r = 0; r < 4; ++r) g[r] = dt * lam[r];
      mf_vjp<D, ABLATE, HILL2, NEED_TH>(th, c, ln.expert, ln_ec50, d1, y, k1, g, av, u, acc);
      mf_dw(ldsU, lds, u, ln.g, ln.pc, lane, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) lam[r] += av[r];
    } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
      float Y2[4], k2[4];
      const float half = 0.5f * dt;
#pragma unroll
      for (int r = 0; r < 4; ++r) Y2[r] = __builtin_fmaf(k1[r], half, y[r]);
      const DoseVal d2 = ds.at(st.ta, th.kel);
      mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, d2.v, Y2, k2);
      image_y(1, Y2);
#pragma unroll
      for (int r = 0; r < 4; ++r) g[r] = dt * lam[r];
      mf_vjp<D, ABLATE, HILL2, NEED_TH>(th, c, ln.expert, ln_ec50, d2, Y2, k2, g, av, u, acc);
      mf_dw(ldsU, lds + 256, u, ln.g, ln.pc, lane, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lam[r] += av[r];
        g[r] = half * av[r];
      }
      mf_vjp<D, ABLATE, HILL2, NEED_TH>(th, c, ln.expert, ln_ec50, d1, y, k1, g, av, u, acc);
      mf_dw(ldsU, lds, u, ln.g, ln.pc, lane, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) lam[r] += av[r];
    } else {
      float Y2[4], Y3[4], Y4[4], k2[4], k3[4], k4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) Y2[r] = __builtin_fmaf(dt * k1[r], kC13, y[r]);
      const DoseVal d2 = ds.at(st.ta, th.kel);
      mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, d2.v, Y2, k2);
      image_y(1, Y2);
#pragma unroll
      for (int r = 0; r < 4; ++r) Y3[r] = __builtin_fmaf(dt, __builtin_fmaf(-k1[r], kC13, k2[r]), y[r]);
      const DoseVal d3 = ds.at(st.tb, th.kel);
      mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, d3.v, Y3, k3);
      image_y(2, Y3);
#pragma unroll
      for (int r = 0; r < 4; ++r) Y4[r] = __builtin_fmaf(dt, (k1[r] - k2[r]) + k3[r], y[r]);
      const DoseVal d4 = ds.at(st.t_last, th.kel);
      mf_rhs<D, ABLATE, HILL2>(th, c, ln.expert, d4.v, Y4, k4);
      image_y(3, Y4);

      const float w1 = dt * 0.125f, w3 = dt * 0.375f;
      float g1[4], g2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) g[r] = w1 * lam[r];
      mf_vjp<D, ABLATE, HILL2, NEED_TH>(th, c, ln.expert, ln_ec50, d4, Y4, k4, g, av, u, acc);
      mf_dw(ldsU, lds + 3 * 256, u, ln.g, ln.pc, lane, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float da = dt * av[r];
        g1[r] = __builtin_fmaf(w1, lam[r], da);
        g2[r] = __builtin_fmaf(w3, lam[r], -da);
        g[r] = __builtin_fmaf(w3, lam[r], da);
        lam[r] += av[r];
      }
      mf_vjp<D, ABLATE, HILL2, NEED_TH>(th, c, ln.expert, ln_ec50, d3, Y3, k3, g, av, u, acc);
      mf_dw(ldsU, lds + 2 * 256, u, ln.g, ln.pc, lane, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float da = dt * av[r];
        g2[r] += da;
        g1[r] = __builtin_fmaf(-kC13, da, g1[r]);
        lam[r] += av[r];
      }
      mf_vjp<D, ABLATE, HILL2, NEED_TH>(th, c, ln.expert, ln_ec50, d2, Y2, k2, g2, av, u, acc);
      mf_dw(ldsU, lds + 256, u, ln.g, ln.pc, lane, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        g1[r] = __builtin_fmaf(kC13, dt * av[r], g1[r]);
        lam[r] += av[r];
      }
      mf_vjp<D, ABLATE, HILL2, NEED_TH>(th, c, ln.expert, ln_ec50, d1, y, k1, g1, av, u, acc);
      mf_dw(ldsU, lds, u, ln.g, ln.pc, lane, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) lam[r] += av[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) lam[r] = __builtin_fmaf(gh[r], lv, lam[r]);
    if (n > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[r] = y_nx[r];
        gh[r] = gh_nx[r];
      }
    }
  }
  mf_store4(a.grad_y0 + off, lam, ln.live);

  // ---- per-wave partial row: dW_ext (256) | db_ext (16) | slots (9)
  float* out = a.partials + (size_t)blockIdx.x * kMfPartials;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc.dW[r];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = acc.db[r];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);  // over the 16 patients of this lane group
    if (ln.pc == 0) out[256 + 4 * ln.g + r] = v;
  }
#pragma unroll
  for (int i = 0; i < kMfSlots; ++i) {
    float v = (NEED_TH && !ABLATE) ? acc.slot[i] : 0.f;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) out[256 + 16 + i] = v;
  }
}

template <int D, int METHOD, bool ABLATE, bool NEED_TH>
__global__ __launch_bounds__(64) void mf_bwd_kernel(MfArgs a) {
  const bool hill2 = ABLATE || (a.theta[0] == 2.0f && a.theta[1] == 2.0f);
  if (hill2 && a.K == 1) mf_bwd_body<D, METHOD, ABLATE, true, NEED_TH, true>(a);
  else if (hill2) mf_bwd_body<D, METHOD, ABLATE, true, NEED_TH, false>(a);
  else mf_bwd_body<D, METHOD, ABLATE, false, NEED_TH, false>(a);
}

// fold the per-wave partial rows in a fixed order and scatter them into (grad_w1, grad_b1, grad_theta) with the signs of
// the extended matrix: one wave per partial column
__global__ __launch_bounds__(64) void mf_fold_kernel(const float* __restrict__ partials, int n_waves, int D, int ablate,
                                                     float* __restrict__ gw, float* __restrict__ gb, float* __restrict__ gth,
                                                     int need_th) {
  const int j = blockIdx.x;
  const int lane = threadIdx.x;
  auto fold = [&](int col) {
    float s = 0.f;
    for (int w = lane; w < n_waves; w += 64) s += partials[(size_t)w * kMfPartials + col];
    return wave_sum(s);
  };
  float s = fold(j);
  // kel enters twice: linearly as W_ext[3][3] = -kel (column 51 of dW_ext) and through the dose slot (last column); one
  // block owns gth[12] so that no two blocks read-modify-write the same word
  const bool kel_col = j == 272 + 8;
  if (kel_col) s -= fold(3 * 16 + 3);
  if (lane != 0) return;
  if (j < 256) {
    const int row = j >> 4, col = j & 15;
    if (row >= 4 && row < D && col < D) {
      if (gw) gw[(row - 4) * D + col] += s;
    } else if (need_th && gth && row < 4) {
      if (!ablate) {
        if (row == 0 && col == 0) gth[7] += s;           // k_disprog
        else if (row == 1 && col == 0) gth[8] += s;      // k_immune_disease
        else if (row == 1 && col == 1) gth[10] -= s;     // k_immune_off (enters as -koff)
        else if (row == 2 && col == 1) gth[11] += s;     // k_immunity
      } else {
        if (row == 1 && col == 0) gth[13] -= s;          // theta_1
        else if (row == 3 && col == 2) gth[14] -= s;     // theta_2
      }
    }
  } else if (j < 272) {
    const int comp = j - 256;
    if (comp >= 4 && comp < D && gb) gb[comp - 4] += s;
  } else if (need_th && gth && !ablate) {
    // slots: 0 kcir, 1 kci, 2 HillCure, 3 kfb, 4 emax, 5 kdexa, 6 HillPatho, 7 ec50, 8 kel (dose part, minus dW_ext[3][3])
    constexpr int map[kMfSlots] = {5, 6, 0, 9, 3, 4, 1, 2, 12};
    gth[map[j - 272]] += s;
  }
}

}  // namespace hode

// ====================================================================================================== host
namespace {

using hode::MfArgs;

template <int D, int METHOD, bool ABLATE>
int mf_launch(const hode_solve_desc* d, const MfArgs& a, bool bwd, hipStream_t s) {
  const dim3 grid((d->batch + 15) / 16), block(64);
  if (!bwd) hipLaunchKernelGGL((hode::mf_fwd_kernel<D, METHOD, ABLATE>), grid, block, 0, s, a);
  else if (d->need_theta_grad) hipLaunchKernelGGL((hode::mf_bwd_kernel<D, METHOD, ABLATE, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((hode::mf_bwd_kernel<D, METHOD, ABLATE, false>), grid, block, 0, s, a);
  return hode::hip_fail(hipGetLastError(), "mf kernel launch");
}

template <int D, bool ABLATE>
int mf_method(const hode_solve_desc* d, const MfArgs& a, bool bwd, hipStream_t s) {
  switch (d->method) {
    case HODE_METHOD_EULER: return mf_launch<D, HODE_METHOD_EULER, ABLATE>(d, a, bwd, s);
    case HODE_METHOD_MIDPOINT: return mf_launch<D, HODE_METHOD_MIDPOINT, ABLATE>(d, a, bwd, s);
    default: return mf_launch<D, HODE_METHOD_RK4_38, ABLATE>(d, a, bwd, s);
  }
}

template <int D>
int mf_dim(const hode_solve_desc* d, const MfArgs& a, bool bwd, hipStream_t s) {
  return d->rhs_kind == HODE_RHS_ROCHE_ABLATE ? mf_method<D, true>(d, a, bwd, s) : mf_method<D, false>(d, a, bwd, s);
}

}  // namespace

namespace hode {

bool mf_supported(const hode_solve_desc* d) { return d->latent_dim == 8 || d->latent_dim == 12 || d->latent_dim == 16; }

size_t mf_workspace_bytes(const hode_solve_desc* d) { return (size_t)((d->batch + 15) / 16) * kMfPartials * sizeof(float); }

int mf_rk(const hode_solve_desc* d, bool bwd, hipStream_t s) {
  MfArgs a{};
  a.t = d->t; a.y0 = d->y0; a.dosage = d->dosage; a.dose_times = d->dose_times; a.theta = d->theta; a.w1 = d->w1; a.b1 = d->b1;
  a.h = d->h; a.grad_h = d->grad_h; a.grad_y0 = d->grad_y0; a.partials = (float*)d->workspace; a.status = d->status;
  a.B = d->batch; a.T = d->n_times; a.K = d->n_dose; a.perturb = d->perturb;
  int e;
  switch (d->latent_dim) {
    case 8: e = mf_dim<8>(d, a, bwd, s); break;
    case 12: e = mf_dim<12>(d, a, bwd, s); break;
    default: e = mf_dim<16>(d, a, bwd, s); break;
  }
  if (e || !bwd) return e;
  hipLaunchKernelGGL(mf_fold_kernel, dim3(kMfPartials), dim3(64), 0, s, (const float*)d->workspace, (d->batch + 15) / 16,
                     d->latent_dim, d->rhs_kind == HODE_RHS_ROCHE_ABLATE ? 1 : 0, d->grad_w1, d->grad_b1, d->grad_theta,
                     d->need_theta_grad);
  return hip_fail(hipGetLastError(), "mf_fold launch");
}

}  // namespace hode
