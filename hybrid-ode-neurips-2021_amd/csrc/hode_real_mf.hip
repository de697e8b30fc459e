// Real-data hybrid rhs (two small MLPs + GRU-ODE block, reference model.py:613-645) on the matrix cores, inside the
// fixed-grid euler / midpoint / rk4(3/8) loop and its discrete adjoint, gfx950.  Same C-ABI contract, tape rows and
// arithmetic (up to summation order) as the one-patient-per-lane kernels of hode_real.hip, which stay as the fallback
// (HODE_REAL_LAYOUT=t, and for shapes outside D = 20, hidden <= 64).
//
// Recipe of hode_neural_mf.hip: a wave owns 16 patients for the whole time loop; with v_mfma_f32_16x16x4_f32 a vector over
// <= 16 rows is one accumulator tile (lane (g, n): rows 4g + r of patient n in register r), every contraction is ordered
// so that k-chunk r is the rows {4g + r}, hence the B fragment of a product IS a register the lane already holds; all
// weight operands are gathered once per launch into that order and stay in registers.  The state is two tiles:
//   X = [x1, x2, x3, Dose2, 0...]  (only lanes g == 0 carry data)          H = the M = 16 GRU states
// and one rhs evaluation is
//   hidden(2 HP)  = tanh([W11; W21] X + [b11; b21])     5 HT MFMAs  (HP = 16 HT >= hidden_dim, both MLPs side by side)
//   (s1, s2)      = [w12 | w22] hidden + (b12, b22)     8 HT MFMAs  -> dx1 = tanh(s1), dx2 = tanh(s2)
//   r, z = sigma(Whr H), sigma(Whz H);  u = tanh(Whh (r*H));  dH = (1 - z)(u - H)      12 MFMAs
//   dx3 = x2 k_immunity,  dx4 = kel Dose(t) - kel2 Dose2          (VALU, lanes g == 0; Dose(t) from the per-patient table)
// i.e. 51 MFMAs instead of ~3 000 fmas with memory-sourced weights per lane; the VJP is 42 MFMAs.
#include <hip/hip_runtime.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"
#include "hode_real_args.hpp"

namespace hode {

typedef float v4 __attribute__((ext_vector_type(4)));

HODE_DEV v4 mfma4(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr float kTanhScale = 2.885390081777927f;   // 2 log2(e)
// tanh(z) = 1 - 2 / (exp2(z') + 1) for z' = kTanhScale z, four values
HODE_DEV v4 tanh_scaled4(const v4& z) {
  const f2 e0 = pair2(__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])) + splat2(1.0f);
  const f2 e1 = pair2(__builtin_amdgcn_exp2f(z[2]), __builtin_amdgcn_exp2f(z[3])) + splat2(1.0f);
  const f2 t0 = __builtin_elementwise_fma(pair2(__builtin_amdgcn_rcpf(e0.x), __builtin_amdgcn_rcpf(e0.y)), splat2(-2.0f), splat2(1.0f));
  const f2 t1 = __builtin_elementwise_fma(pair2(__builtin_amdgcn_rcpf(e1.x), __builtin_amdgcn_rcpf(e1.y)), splat2(-2.0f), splat2(1.0f));
  return v4{t0.x, t0.y, t1.x, t1.y};
}

template <int HT>
struct RealMf {
  static constexpr int M = 16;
  static constexpr int NT2 = 2 * HT;
  float A1a[HT][3], A1b[HT][2];   // hidden layers: W11[16i+m][r], W21[16i+m][r] in lanes g == 0
  float A2[NT2][4];               // output layer: row 0 <- w12, row 1 <- w22
  float A3[NT2];                  // hidden cotangent: w12[16i+m] (chunk 0) / w22[16(i-HT)+m] (chunk 1), lanes g == 0
  float A4[NT2][4];               // input cotangent: W11[16i+4g+r][m] (m < 3) / W21[..][m] (m < 2)
  float Gr[4], Gz[4], Gh[4], GrT[4], GzT[4], GhT[4];
  v4 b1[NT2];
  v4 b2;
  int g, n;
  float kim, kel, kel2;

  HODE_DEV void load(const RealArgs& a, int lane) {
    g = lane >> 4;
    n = lane & 15;
    const int m = lane & 15;
    const int H = a.H;
    const RealW w(a.wflat, H, M);
    kim = a.theta[0]; kel = a.theta[1]; kel2 = a.theta[2];
#pragma unroll
    for (int i = 0; i < HT; ++i) {
      const int j = 16 * i + m;  // hidden unit addressed as an A ROW
      // the hidden layers' tanh takes its argument pre-multiplied by 2 log2(e) (kTanhScale, folded into W11 / W21 / b11 / b21 here):
      // v_exp, add, v_rcp, fma per activation, the fma on pairs -- see NeuralMf::tanh_scaled
#pragma unroll
      for (int r = 0; r < 3; ++r) A1a[i][r] = (g == 0 && j < H) ? kTanhScale * w.W11[3 * j + r] : 0.f;
#pragma unroll
      for (int r = 0; r < 2; ++r) A1b[i][r] = (g == 0 && j < H) ? kTanhScale * w.W21[2 * j + r] : 0.f;
      A3[i] = (g == 0 && j < H) ? w.w12[j] : 0.f;
      A3[HT + i] = (g == 0 && j < H) ? w.w22[j] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jc = 16 * i + 4 * g + r;  // hidden unit addressed as a K index / tile row
        A2[i][r] = (m == 0 && jc < H) ? w.w12[jc] : 0.f;
        A2[HT + i][r] = (m == 1 && jc < H) ? w.w22[jc] : 0.f;
        A4[i][r] = (m < 3 && jc < H) ? w.W11[3 * jc + m] : 0.f;
        A4[HT + i][r] = (m < 2 && jc < H) ? w.W21[2 * jc + m] : 0.f;
        b1[i][r] = jc < H ? kTanhScale * w.b11[jc] : 0.f;
        b1[HT + i][r] = jc < H ? kTanhScale * w.b21[jc] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * g + r;
      Gr[r] = w.Whr[m * M + c]; Gz[r] = w.Whz[m * M + c]; Gh[r] = w.Whh[m * M + c];
      GrT[r] = w.Whr[c * M + m]; GzT[r] = w.Whz[c * M + m]; GhT[r] = w.Whh[c * M + m];
    }
    b2 = v4{0.f, 0.f, 0.f, 0.f};
    if (g == 0) {
      b2[0] = w.b12[0];
      b2[1] = w.b22[0];
    }
  }

  struct Stage {  // what the VJP needs of one evaluation
    v4 X, Hs, KX, ah[NT2], r, z, u, RH;
    float dv, ddk;
  };

  // k = f(X, Hs); dose = (Dose(t), dDose/dkel) in lanes g == 0, zeros elsewhere
  HODE_DEV void rhs(Stage& s, v4& KH) const {
    v4 acc[NT2];
#pragma unroll
    for (int i = 0; i < NT2; ++i) acc[i] = b1[i];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < HT; ++i) {
        acc[i] = mfma4(A1a[i][r], s.X[r], acc[i]);
        if (r < 2) acc[HT + i] = mfma4(A1b[i][r < 2 ? r : 0], s.X[r], acc[HT + i]);
      }
#pragma unroll
    for (int i = 0; i < NT2; ++i) s.ah[i] = tanh_scaled4(acc[i]);
    v4 zz[4];
    zz[0] = b2;
    zz[1] = zz[2] = zz[3] = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NT2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) zz[r] = mfma4(A2[i][r], s.ah[i][r], zz[r]);
    const v4 S = (zz[0] + zz[1]) + (zz[2] + zz[3]);
    s.KX[0] = tanh_f32(S[0]);           // rows 4g + r of lanes g > 0 are not state: everything below is 0 there
    s.KX[1] = tanh_f32(S[1]);
    s.KX[2] = s.X[1] * kim;
    s.KX[3] = __builtin_fmaf(kel, s.dv, -kel2 * s.X[3]);
    // GRU block
    v4 ar = v4{0.f, 0.f, 0.f, 0.f}, az = ar, au = ar;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ar = mfma4(Gr[r], s.Hs[r], ar);
      az = mfma4(Gz[r], s.Hs[r], az);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s.r[r] = sigm(ar[r]);
      s.z[r] = sigm(az[r]);
      s.RH[r] = s.r[r] * s.Hs[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) au = mfma4(Gh[r], s.RH[r], au);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s.u[r] = tanh_f32(au[r]);
      KH[r] = (1.0f - s.z[r]) * (s.u[r] - s.Hs[r]);
    }
  }

  // (aX, aH) = (df/dy)^T (gX, gH) at stage s; cotangent tiles for the tape; scalar gradients into dth
  HODE_DEV void vjp(const Stage& s, const v4& gX, const v4& gH, v4& aX, v4& aH, v4 (&U1)[NT2], float& u12, float& u22,
                    v4& ur, v4& uz, v4& uh, float (&dth)[3]) const {
    u12 = gX[0] * __builtin_fmaf(-s.KX[0], s.KX[0], 1.0f);
    u22 = gX[1] * __builtin_fmaf(-s.KX[1], s.KX[1], 1.0f);
    v4 da[NT2];
#pragma unroll
    for (int i = 0; i < HT; ++i) {
      da[i] = mfma4(A3[i], u12, v4{0.f, 0.f, 0.f, 0.f});
      da[HT + i] = mfma4(A3[HT + i], u22, v4{0.f, 0.f, 0.f, 0.f});
    }
#pragma unroll
    for (int i = 0; i < NT2; ++i)   // the same two operations per value (fma, mul), on pairs
      U1[i] = da[i] * __builtin_elementwise_fma(-s.ah[i], s.ah[i], v4{1.0f, 1.0f, 1.0f, 1.0f});
    v4 zz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) zz[r] = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NT2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) zz[r] = mfma4(A4[i][r], U1[i][r], zz[r]);
    aX = (zz[0] + zz[1]) + (zz[2] + zz[3]);
    aX[1] = __builtin_fmaf(gX[2], kim, aX[1]);
    aX[3] = __builtin_fmaf(-kel2, gX[3], aX[3]);
    dth[0] = __builtin_fmaf(gX[2], s.X[1], dth[0]);
    dth[1] = __builtin_fmaf(gX[3], __builtin_fmaf(kel, s.ddk, s.dv), dth[1]);
    dth[2] = __builtin_fmaf(-gX[3], s.X[3], dth[2]);
    // GRU block
    v4 drh = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uz[r] = -gH[r] * (s.u[r] - s.Hs[r]) * s.z[r] * (1.0f - s.z[r]);
      uh[r] = gH[r] * (1.0f - s.z[r]) * __builtin_fmaf(-s.u[r], s.u[r], 1.0f);
      aH[r] = -gH[r] * (1.0f - s.z[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) drh = mfma4(GhT[r], uh[r], drh);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ur[r] = drh[r] * s.Hs[r] * s.r[r] * (1.0f - s.r[r]);
      aH[r] = __builtin_fmaf(drh[r], s.r[r], aH[r]);
    }
    v4 acc = v4{0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc = mfma4(GrT[r], ur[r], acc);
      acc2 = mfma4(GzT[r], uz[r], acc2);
    }
    aH = aH + (acc + acc2);
  }
};

// ------------------------------------------------------------------------------ weight gradients on the matrix cores
// The eleven weight gradients of the rhs are outer products summed over patients (recipe of NeuralGradAcc in
// hode_neural_mf.hpp: operands transposed through patient-major LDS images so that the patient index becomes the MFMA
// contraction index).  Per stage VJP, over the wave's 16 patients:
//   dWa[i] += U1 tile i (x) [X | 1]      i <  HT: rows of dW11 (cols 0..2) and db11 (col 4, the ones row behind X)
//                                         i >= HT: rows of dW21 (cols 0..1) and db21 (col 4)
//   dWb[i] += [u12; u22] (x) ah tile i    i <  HT: row 0 = dw12;  i >= HT: row 1 = dw22   (the cross terms are not read)
//   dHh += uh (x) r*h,  dHz += uz (x) h,  dHr += ur (x) h;   db12 / db22: per-lane sums of u12 / u22
// 60 MFMAs, 19 ds_write_b128 and 76 ds_read_b32 per stage at HT = 3 instead of ~70 strided tape stores per lane; one block
// of NP floats per wave, folded in a fixed order by real_grad_fold_kernel into the flat weight-gradient buffer.
template <int HT>
struct RealGradAcc {
  static constexpr int NT2 = 2 * HT;
  static constexpr int PH = ((16 * NT2 - 16 + 63) / 64) * 64 + 16;  // pitch == 16 mod 64: conflict-free fragment reads
  static constexpr int kLdsFloats = 2 * 16 * PH + 7 * 256;
  static constexpr int NP = (2 * NT2 + 3) * 256 + 16;
  v4 dWa[NT2], dWb[NT2], dHh, dHz, dHr, db2;
  float *U1i, *AHi, *Xi, *U2i, *UHi, *UZi, *URi, *RHi, *HHi;
  HODE_DEV void init(float* lds) {
    U1i = lds; AHi = lds + 16 * PH;
    float* sm = lds + 32 * PH;
    Xi = sm; U2i = sm + 256; UHi = sm + 512; UZi = sm + 768; URi = sm + 1024; RHi = sm + 1280; HHi = sm + 1536;
    const v4 z = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NT2; ++i) dWa[i] = dWb[i] = z;
    dHh = dHz = dHr = db2 = z;
  }
  HODE_DEV void add(const v4 (&U1)[NT2], const v4 (&ah)[NT2], v4 X, float u12, float u22, const v4& uh, const v4& uz,
                    const v4& ur, const v4& rh, const v4& hh, int g, int n) {
    if (g == 1) X[0] = 1.0f;  // row 4: the bias column
    const v4 U2 = v4{u12, u22, 0.f, 0.f};  // lanes g > 0 carry zeros (gX is zero there)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT2; ++i) {
      *reinterpret_cast<v4*>(U1i + n * PH + 16 * i + 4 * g) = U1[i];
      *reinterpret_cast<v4*>(AHi + n * PH + 16 * i + 4 * g) = ah[i];
    }
    *reinterpret_cast<v4*>(Xi + n * 16 + 4 * g) = X;
    *reinterpret_cast<v4*>(U2i + n * 16 + 4 * g) = U2;
    *reinterpret_cast<v4*>(UHi + n * 16 + 4 * g) = uh;
    *reinterpret_cast<v4*>(UZi + n * 16 + 4 * g) = uz;
    *reinterpret_cast<v4*>(URi + n * 16 + 4 * g) = ur;
    *reinterpret_cast<v4*>(RHi + n * 16 + 4 * g) = rh;
    *reinterpret_cast<v4*>(HHi + n * 16 + 4 * g) = hh;
    __syncthreads();
    const int m = n, kk = g;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int o = (4 * c + kk) * 16 + m;
      const float xB = Xi[o], u2A = U2i[o], rhB = RHi[o], hhB = HHi[o];
      dHh = mfma4(UHi[o], rhB, dHh);
      dHz = mfma4(UZi[o], hhB, dHz);
      dHr = mfma4(URi[o], hhB, dHr);
#pragma unroll
      for (int i = 0; i < NT2; ++i) {
        dWa[i] = mfma4(U1i[(4 * c + kk) * PH + 16 * i + m], xB, dWa[i]);
        dWb[i] = mfma4(u2A, AHi[(4 * c + kk) * PH + 16 * i + m], dWb[i]);
      }
    }
    db2 = db2 + U2;
  }
  HODE_DEV void store(float* __restrict__ out, int lane) {
#pragma unroll
    for (int i = 0; i < NT2; ++i) {
      *reinterpret_cast<v4*>(out + ((size_t)i * 64 + lane) * 4) = dWa[i];
      *reinterpret_cast<v4*>(out + ((size_t)(NT2 + i) * 64 + lane) * 4) = dWb[i];
    }
    *reinterpret_cast<v4*>(out + ((size_t)(2 * NT2) * 64 + lane) * 4) = dHh;
    *reinterpret_cast<v4*>(out + ((size_t)(2 * NT2 + 1) * 64 + lane) * 4) = dHz;
    *reinterpret_cast<v4*>(out + ((size_t)(2 * NT2 + 2) * 64 + lane) * 4) = dHr;
    v4 sv;
#pragma unroll
    for (int r = 0; r < 4; ++r) sv[r] = row_sum(db2[r]);
    if ((lane & 15) == 0) *reinterpret_cast<v4*>(out + (2 * NT2 + 3) * 256 + 4 * (lane >> 4)) = sv;
  }
};

// fixed-order fold of the per-wave blocks into the flat weight gradient (RealW order); one wave per slot of a block
template <int HT>
__global__ __launch_bounds__(64) void real_grad_fold_kernel(const float* __restrict__ partials, int n_waves, int H,
                                                            float* __restrict__ gw) {
  constexpr int NT2 = 2 * HT, NP = RealGradAcc<HT>::NP, M = 16;
  const int j = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int w = lane; w < n_waves; w += 64) s += partials[(size_t)w * NP + j];
  s = wave_sum(s);
  if (lane != 0) return;
  const int oW11 = 0, ob11 = 3 * H, ow12 = 4 * H, ob12 = 5 * H, oW21 = 5 * H + 1, ob21 = 7 * H + 1, ow22 = 8 * H + 1,
            ob22 = 9 * H + 1, oWhh = 9 * H + 2, oWhz = oWhh + M * M, oWhr = oWhz + M * M;
  if (j >= (2 * NT2 + 3) * 256) {
    const int o = j - (2 * NT2 + 3) * 256;
    if (o == 0) gw[ob12] += s;
    else if (o == 1) gw[ob22] += s;
    return;
  }
  const int tile = j / 256, l = (j % 256) / 4, r = j % 4;
  const int rw = 4 * (l >> 4) + r, col = l & 15;
  if (tile < NT2) {
    const bool first = tile < HT;
    const int hid = 16 * (first ? tile : tile - HT) + rw;
    if (hid >= H) return;
    if (first) {
      if (col < 3) gw[oW11 + 3 * hid + col] += s;
      else if (col == 4) gw[ob11 + hid] += s;
    } else {
      if (col < 2) gw[oW21 + 2 * hid + col] += s;
      else if (col == 4) gw[ob21 + hid] += s;
    }
  } else if (tile < 2 * NT2) {
    const int i = tile - NT2;
    const bool first = i < HT;
    const int hid = 16 * (first ? i : i - HT) + col;
    if (hid >= H) return;
    if (first && rw == 0) gw[ow12 + hid] += s;
    else if (!first && rw == 1) gw[ow22 + hid] += s;
  } else {
    const int which = tile - 2 * NT2;
    gw[(which == 0 ? oWhh : (which == 1 ? oWhz : oWhr)) + rw * M + col] += s;
  }
}

// ONCHIP (backward): weight gradients accumulated by the wave (RealGradAcc), one partial block per wave in a.tape;
// otherwise the GEMM operands are taped for the caller (contract of the lane-per-patient kernels).
template <int HT, int METHOD, bool BWD, bool ONCHIP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void real_mf_kernel(RealArgs a) {
  typedef RealMf<HT> Net;
  typedef typename Net::Stage Stage;
  constexpr int D = 20, M = 16, NT2 = 2 * HT;
  constexpr int NS = METHOD == HODE_METHOD_EULER ? 1 : (METHOD == HODE_METHOD_MIDPOINT ? 2 : 4);
  constexpr float c13 = (float)(1.0 / 3.0);
  const int lane = threadIdx.x;
  Net nn;
  nn.load(a, lane);
  const int g = nn.g;
  const bool g0 = g == 0;
  const int pr = blockIdx.x * 16 + nn.n;
  const bool live = pr < a.B;
  const int p = live ? pr : a.B - 1;
  const size_t B = a.B;
  const size_t row = B * D;
  const RealTape tl{a.H, M};
  const size_t tstride = (size_t)tl.rows() * B;
  __shared__ __attribute__((aligned(16))) float lds[(BWD && ONCHIP) ? RealGradAcc<HT>::kLdsFloats : 4];
  RealGradAcc<HT> acc;
  if constexpr (BWD && ONCHIP) acc.init(lds);
  if (g0) real_dose_table(a, p, nn.kel);  // each lane reads back only its own column

  auto load_state = [&](const float* src, v4& X, v4& Hs) {  // src -> this patient's D floats
    X = v4{0.f, 0.f, 0.f, 0.f};
    if (g0) X = *reinterpret_cast<const v4*>(src);
    Hs = *reinterpret_cast<const v4*>(src + 4 + 4 * g);
  };
  auto store_state = [&](float* dst, const v4& X, const v4& Hs) {
    if (!live) return;
    if (g0) *reinterpret_cast<v4*>(dst) = X;
    *reinterpret_cast<v4*>(dst + 4 + 4 * g) = Hs;
  };
  auto set_dose = [&](Stage& s, float t) {
    s.dv = 0.f;
    s.ddk = 0.f;
    if (g0) {
      const DoseK d = real_dose(a, p, t, nn.kel);
      s.dv = d.v;
      s.ddk = d.dk;
    }
  };

  if constexpr (!BWD) {
    v4 X, Hs;
    load_state(a.y0 + (size_t)p * D, X, Hs);
    store_state(a.h + (size_t)p * D, X, Hs);
    Stage s;
    for (int n = 0; n + 1 < a.T; ++n) {
      const RStageTimes st(a.t, n, a.perturb, METHOD);
      const float dt = st.dt;
      v4 k1H;
      s.X = X; s.Hs = Hs; set_dose(s, st.t_first);
      nn.rhs(s, k1H);
      const v4 k1X = s.KX;
      if constexpr (METHOD == HODE_METHOD_EULER) {
        X = X + dt * k1X; Hs = Hs + dt * k1H;
      } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
        v4 k2H;
        s.X = X + (0.5f * dt) * k1X; s.Hs = Hs + (0.5f * dt) * k1H; set_dose(s, st.ta);
        nn.rhs(s, k2H);
        X = X + dt * s.KX; Hs = Hs + dt * k2H;
      } else {
        v4 k2H, k3H, k4H;
        s.X = X + (dt * k1X) * c13; s.Hs = Hs + (dt * k1H) * c13; set_dose(s, st.ta);
        nn.rhs(s, k2H);
        const v4 k2X = s.KX;
        s.X = X + dt * (k2X - k1X * c13); s.Hs = Hs + dt * (k2H - k1H * c13); set_dose(s, st.tb);
        nn.rhs(s, k3H);
        const v4 k3X = s.KX;
        s.X = X + dt * ((k1X - k2X) + k3X); s.Hs = Hs + dt * ((k1H - k2H) + k3H); set_dose(s, st.t_last);
        nn.rhs(s, k4H);
        X = X + ((k1X + 3.0f * (k2X + k3X)) + s.KX) * (dt * 0.125f);
        Hs = Hs + ((k1H + 3.0f * (k2H + k3H)) + k4H) * (dt * 0.125f);
      }
      store_state(a.h + (size_t)(n + 1) * row + (size_t)p * D, X, Hs);
    }
  } else {
    const float lv = live ? 1.0f : 0.0f;
    float dth[3] = {0.f, 0.f, 0.f};
    v4 lamX, lamH;
    load_state(a.grad_h + (size_t)(a.T - 1) * row + (size_t)p * D, lamX, lamH);
    lamX = lv * lamX; lamH = lv * lamH;

    auto tape_tile = [&](float* tp, int first_row, int nrows, const v4& v) {  // rows 4g + r of a 16-row vector
      if (!live) return;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rw = 4 * g + r;
        if (rw < nrows) tp[(size_t)(first_row + rw) * B] = v[r];
      }
    };
    auto tape_hidden = [&](float* tp, int first_row, const v4* v) {  // HT tiles, rows < hidden_dim
      if (!live) return;
#pragma unroll
      for (int i = 0; i < HT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rw = 16 * i + 4 * g + r;
          if (rw < a.H) tp[(size_t)(first_row + rw) * B] = v[i][r];
        }
    };

    for (int n = a.T - 2; n >= 0; --n) {
      const RStageTimes st(a.t, n, a.perturb, METHOD);
      const float dt = st.dt;
      float* tp0 = a.tape + (size_t)n * NS * tstride + p;
      v4 X, Hs;
      load_state(a.h + (size_t)n * row + (size_t)p * D, X, Hs);
      Stage s[NS];
      v4 kH[NS];
      s[0].X = X; s[0].Hs = Hs; set_dose(s[0], st.t_first);
      nn.rhs(s[0], kH[0]);
      if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
        s[1].X = X + (0.5f * dt) * s[0].KX; s[1].Hs = Hs + (0.5f * dt) * kH[0]; set_dose(s[1], st.ta);
        nn.rhs(s[1], kH[1]);
      } else if constexpr (METHOD == HODE_METHOD_RK4_38) {
        s[1].X = X + (dt * s[0].KX) * c13; s[1].Hs = Hs + (dt * kH[0]) * c13; set_dose(s[1], st.ta);
        nn.rhs(s[1], kH[1]);
        s[2].X = X + dt * (s[1].KX - s[0].KX * c13); s[2].Hs = Hs + dt * (kH[1] - kH[0] * c13); set_dose(s[2], st.tb);
        nn.rhs(s[2], kH[2]);
        s[3].X = X + dt * ((s[0].KX - s[1].KX) + s[2].KX); s[3].Hs = Hs + dt * ((kH[0] - kH[1]) + kH[2]);
        set_dose(s[3], st.t_last);
        nn.rhs(s[3], kH[3]);
      }
      auto vjp = [&](int q, const v4& gX, const v4& gH, v4& aX, v4& aH) {
        v4 U1[NT2], ur, uz, uh;
        float u12, u22;
        nn.vjp(s[q], gX, gH, aX, aH, U1, u12, u22, ur, uz, uh, dth);
        if constexpr (ONCHIP) {
          acc.add(U1, s[q].ah, s[q].X, u12, u22, uh, uz, ur, s[q].RH, s[q].Hs, g, nn.n);
          return;
        }
        float* tp = tp0 + (size_t)q * tstride;
        tape_tile(tp, tl.y3(), 3, s[q].X);
        tape_hidden(tp, tl.a11(), s[q].ah);
        tape_hidden(tp, tl.u11(), U1);
        tape_hidden(tp, tl.a21(), s[q].ah + HT);
        tape_hidden(tp, tl.u21(), U1 + HT);
        if (live && g0) {
          tp[(size_t)tl.u12() * B] = u12;
          tp[(size_t)tl.u22() * B] = u22;
        }
        tape_tile(tp, tl.hh(), M, s[q].Hs);
        tape_tile(tp, tl.rh(), M, s[q].RH);
        tape_tile(tp, tl.ur(), M, ur);
        tape_tile(tp, tl.uz(), M, uz);
        tape_tile(tp, tl.uh(), M, uh);
      };
      v4 aX, aH;
      if constexpr (METHOD == HODE_METHOD_EULER) {
        vjp(0, dt * lamX, dt * lamH, aX, aH);
        lamX = lamX + aX; lamH = lamH + aH;
      } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
        vjp(1, dt * lamX, dt * lamH, aX, aH);
        lamX = lamX + aX; lamH = lamH + aH;
        const v4 gX = (0.5f * dt) * aX, gH = (0.5f * dt) * aH;
        vjp(0, gX, gH, aX, aH);
        lamX = lamX + aX; lamH = lamH + aH;
      } else {
        const float w1 = dt * 0.125f, w3 = dt * 0.375f;
        vjp(3, w1 * lamX, w1 * lamH, aX, aH);
        v4 daX = dt * aX, daH = dt * aH;
        v4 g1X = w1 * lamX + daX, g1H = w1 * lamH + daH;
        v4 g2X = w3 * lamX - daX, g2H = w3 * lamH - daH;
        const v4 gX = w3 * lamX + daX, gH = w3 * lamH + daH;
        lamX = lamX + aX; lamH = lamH + aH;
        vjp(2, gX, gH, aX, aH);
        daX = dt * aX; daH = dt * aH;
        g2X = g2X + daX; g2H = g2H + daH;
        g1X = g1X - c13 * daX; g1H = g1H - c13 * daH;
        lamX = lamX + aX; lamH = lamH + aH;
        vjp(1, g2X, g2H, aX, aH);
        g1X = g1X + c13 * (dt * aX); g1H = g1H + c13 * (dt * aH);
        lamX = lamX + aX; lamH = lamH + aH;
        vjp(0, g1X, g1H, aX, aH);
        lamX = lamX + aX; lamH = lamH + aH;
      }
      v4 ghX, ghH;
      load_state(a.grad_h + (size_t)n * row + (size_t)p * D, ghX, ghH);
      lamX = lamX + lv * ghX; lamH = lamH + lv * ghH;
    }
    store_state(a.grad_y0 + (size_t)p * D, lamX, lamH);
    // only lanes g == 0 of live patients carry scalar-gradient terms (X, gX are zero elsewhere; lv zeroes dead patients)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float v = wave_sum(g0 ? dth[j] : 0.f);
      if (lane == 0) a.partials[(size_t)blockIdx.x * 3 + j] = v;
    }
    if constexpr (ONCHIP) acc.store(a.tape + (size_t)blockIdx.x * RealGradAcc<HT>::NP, lane);
  }
}

bool real_mf_supported(const hode_solve_desc* d) {
  return d->latent_dim == 20 && d->hidden_dim >= 1 && d->hidden_dim <= 64;
}

template <int HT, bool BWD>
int launch_real_mf_ht(const hode_solve_desc* d, const RealArgs& a, hipStream_t s) {
  const dim3 grid((d->batch + 15) / 16), block(64);
  const bool onchip = BWD && d->grad_w1 != nullptr;
#define HODE_REAL_MF_LAUNCH(M)                                                                        \
  if (onchip) hipLaunchKernelGGL((real_mf_kernel<HT, M, BWD, BWD>), grid, block, 0, s, a);              \
  else hipLaunchKernelGGL((real_mf_kernel<HT, M, BWD, false>), grid, block, 0, s, a);
  switch (d->method) {
    case HODE_METHOD_EULER: HODE_REAL_MF_LAUNCH(HODE_METHOD_EULER) break;
    case HODE_METHOD_MIDPOINT: HODE_REAL_MF_LAUNCH(HODE_METHOD_MIDPOINT) break;
    default: HODE_REAL_MF_LAUNCH(HODE_METHOD_RK4_38) break;
  }
  if (onchip)
    hipLaunchKernelGGL((real_grad_fold_kernel<HT>), dim3(RealGradAcc<HT>::NP), block, 0, s, a.tape, (int)grid.x, d->hidden_dim,
                       d->grad_w1);
  return hip_fail(hipGetLastError(), "real MFMA kernel launch");
}

// bytes of per-wave gradient partials of the on-chip backward (they take the tape's place in the workspace)
size_t real_mf_partial_bytes(const hode_solve_desc* d) {
  const size_t nw = (d->batch + 15) / 16;
  switch ((d->hidden_dim + 15) / 16) {
    case 1: return nw * RealGradAcc<1>::NP * sizeof(float);
    case 2: return nw * RealGradAcc<2>::NP * sizeof(float);
    case 3: return nw * RealGradAcc<3>::NP * sizeof(float);
    default: return nw * RealGradAcc<4>::NP * sizeof(float);
  }
}

int launch_real_mf(const hode_solve_desc* d, const RealArgs& a, bool bwd, hipStream_t s) {
  const int ht = (d->hidden_dim + 15) / 16;
  switch (ht) {
    case 1: return bwd ? launch_real_mf_ht<1, true>(d, a, s) : launch_real_mf_ht<1, false>(d, a, s);
    case 2: return bwd ? launch_real_mf_ht<2, true>(d, a, s) : launch_real_mf_ht<2, false>(d, a, s);
    case 3: return bwd ? launch_real_mf_ht<3, true>(d, a, s) : launch_real_mf_ht<3, false>(d, a, s);
    default: return bwd ? launch_real_mf_ht<4, true>(d, a, s) : launch_real_mf_ht<4, false>(d, a, s);
  }
}

}  // namespace hode
