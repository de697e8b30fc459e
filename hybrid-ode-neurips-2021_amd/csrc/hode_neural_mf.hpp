// NeuralODE rhs on the matrix cores: the register-resident weight fragments and the rhs / VJP products shared by the
// fixed-grid kernels (hode_neural_mf.hip) and the adaptive ones (hode_neural_dopri5.hip).  Layout and fragment ordering are
// described at the top of hode_neural_mf.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_neural_args.hpp"

namespace hode {

typedef float v4 __attribute__((ext_vector_type(4)));

template <int D>
struct NeuralMf {
  static constexpr int HD = 10 * D;
  static constexpr int HT = (HD + 15) / 16;   // hidden tiles
  static constexpr int GD = D / 4, RD = D % 4;  // tile position of the Dose input (row D)
  float A1[HT][4], A2[HT][4], A3[HT][4], A4[HT][4];
  v4 bias1[HT];
  v4 bias2;
  int g, n;
  // Both tanh layers take their argument pre-multiplied by 2 log2(e) (folded into W1, b1, W2, b2 as they are gathered):
  // tanh(z) = 1 - 2 / (exp2(z') + 1) then is v_exp, add, v_rcp, fma with the add and the fma on packed pairs -- 3 issue slots
  // per value instead of 5.  The transposed operands of the VJP (A3, A4) stay unscaled.
  static constexpr float kTanhScale = 2.885390081777927f;
  static HODE_DEV v4 tanh_scaled(const v4& z) {
    // written on pairs: on the four-vector the compiler kept the additions scalar
    const f2 e0 = pair2(__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])) + splat2(1.0f);
    const f2 e1 = pair2(__builtin_amdgcn_exp2f(z[2]), __builtin_amdgcn_exp2f(z[3])) + splat2(1.0f);
    const f2 t0 = __builtin_elementwise_fma(pair2(__builtin_amdgcn_rcpf(e0.x), __builtin_amdgcn_rcpf(e0.y)), splat2(-2.0f), splat2(1.0f));
    const f2 t1 = __builtin_elementwise_fma(pair2(__builtin_amdgcn_rcpf(e1.x), __builtin_amdgcn_rcpf(e1.y)), splat2(-2.0f), splat2(1.0f));
    return v4{t0.x, t0.y, t1.x, t1.y};
  }

  HODE_DEV void load(const NeuralArgs& a, int lane) {
    g = lane >> 4;
    n = lane & 15;
    const int m = lane & 15;
    const float* W1 = a.w1;   // [HD][D + 1]
    const float* W2 = a.w2;   // [D][HD]
#pragma unroll
    for (int i = 0; i < HT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 4 * g + r;          // k index of this lane within chunk r
        const int hrow = 16 * i + m;        // A-row of hidden-sized products
        const int hcol = 16 * i + 4 * g + r;  // k index over the hidden axis
        A1[i][r] = (hrow < HD && col <= D) ? kTanhScale * W1[(size_t)hrow * (D + 1) + col] : 0.f;   // W1[16i+m][4g+r]
        A2[i][r] = (m < D && hcol < HD) ? kTanhScale * W2[(size_t)m * HD + hcol] : 0.f;             // W2[m][16i+4g+r]
        A3[i][r] = (hrow < HD && col < D) ? W2[(size_t)col * HD + hrow] : 0.f;              // W2^T[16i+m][4g+r]
        A4[i][r] = (hcol < HD && m <= D) ? W1[(size_t)hcol * (D + 1) + m] : 0.f;            // W1^T[m][16i+4g+r]
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + 4 * g + r;
        bias1[i][r] = row < HD ? kTanhScale * a.b1[row] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bias2[r] = (4 * g + r) < D ? kTanhScale * a.b2[4 * g + r] : 0.f;
  }

  // hidden activations a1 = tanh(W1 e + b1) for the input tile e (tile i, register r <-> hidden unit 16i + 4g + r)
  HODE_DEV void hidden(const v4& e, v4 (&a1)[HT]) const {
    v4 acc[HT];
#pragma unroll
    for (int i = 0; i < HT; ++i) acc[i] = bias1[i];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < HT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[i][r], e[r], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < HT; ++i) a1[i] = tanh_scaled(acc[i]);
  }

  // k = f(e) = tanh(W2 a1 + b2); a1 is left for the caller (the fixed-grid adjoint holds it, the adaptive one recomputes it)
  HODE_DEV v4 rhs(const v4& e, v4 (&a1)[HT]) const {
    hidden(e, a1);
    v4 z[4];
    z[0] = bias2;
    z[1] = z[2] = z[3] = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < HT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[i][r], a1[i][r], z[r], 0, 0, 0);
    return tanh_scaled((z[0] + z[1]) + (z[2] + z[3]));
  }

  // VJP at a stage with activations a1 and output k: returns (df/de)^T gk; u2, u1 are the pre-activation cotangents
  HODE_DEV v4 vjp(const v4 (&a1)[HT], const v4& k, const v4& gk, v4& u2, v4 (&u1)[HT]) const {
#pragma unroll
    for (int r = 0; r < 4; ++r) u2[r] = gk[r] * __builtin_fmaf(-k[r], k[r], 1.0f);
    v4 acc[HT];
#pragma unroll
    for (int i = 0; i < HT; ++i) acc[i] = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < HT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A3[i][r], u2[r], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < HT; ++i)   // same two operations per value as before (fma, mul), on packed pairs
      u1[i] = acc[i] * __builtin_elementwise_fma(-a1[i], a1[i], v4{1.0f, 1.0f, 1.0f, 1.0f});
    v4 z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < HT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(A4[i][r], u1[i][r], z[r], 0, 0, 0);
    return (z[0] + z[1]) + (z[2] + z[3]);
  }
};

// the state tile of patient p: rows < D from memory, everything else 0
template <int D>
HODE_DEV v4 mf_load_rows(const float* __restrict__ src, int g) {
  v4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (4 * g + r) < D ? src[4 * g + r] : 0.f;
  return v;
}
template <int D>
HODE_DEV void mf_store_rows(float* __restrict__ dst, int g, const v4& v, bool live) {
  if (!live) return;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if ((4 * g + r) < D) dst[4 * g + r] = v[r];
}
template <int D>
HODE_DEV v4 mf_with_dose(const v4& y, float dose, int g) {  // e = [y, Dose, 0...]: row D lives in tile position (GD, RD)
  v4 e = y;
  if (g == NeuralMf<D>::GD) e[NeuralMf<D>::RD] = dose;
  return e;
}

HODE_DEV v4 splat4(float x) { return v4{x, x, x, x}; }
HODE_DEV float hsum4(const v4& v) { return (v[0] + v[1]) + (v[2] + v[3]); }
HODE_DEV v4 vfma4(float a, const v4& b, const v4& c) {
  v4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(a, b[i], c[i]);
  return r;
}
HODE_DEV v4 vfma4(const v4& a, const v4& b, const v4& c) {
  v4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(a[i], b[i], c[i]);
  return r;
}


// ------------------------------------------------------------------------------ weight gradients on the matrix cores
// dW1[h][i] += sum_n u1[h][n] e[i][n]  (i = D + 1 is a ones row: the column that collects db1),
// dW2[o][h] += sum_n u2[o][n] a1[h][n],  db2[o] += sum_n u2[o][n]  over the wave's 16 patients n, per stage VJP.
// The register tiles hold [row][patient] with the patient in (lane & 15), which is the MFMA's B layout with the ROW as the
// contraction index; the outer products contract over PATIENTS, so the four operands go through LDS patient-major
// (16-byte stores: a lane's four rows are consecutive) and come back with lane (m, kk) reading image[4c + kk][.. + m]:
// A[m][kk] / B[kk][m] fragments of patient chunk c.  72 ds_read_b32 + 18 ds_write_b128 + 64 MFMAs per stage.
template <int D>
struct NeuralGradAcc {
  static constexpr int HT = NeuralMf<D>::HT;
  // image pitches (floats) == 16 mod 64: the fragment read of lane (m, kk) at [4c + kk][16i + m] then hits bank m + 16 kk
  // -- all 64 lanes on different banks -- and the 16-byte writes of lane (g, n) at [n][16i + 4g] spread 4 dwords per bank
  static constexpr int PH = ((16 * HT - 16 + 63) / 64) * 64 + 16;
  static constexpr int PS = 16;
  static constexpr int kLdsFloats = 16 * (2 * PH + 2 * PS);
  static constexpr int NP = 2 * HT * 256 + 16;  // floats per wave in the partial array
  static constexpr int GB = (D + 1) / 4, RB = (D + 1) % 4;  // tile position of the ones row behind [y, Dose]
  v4 dW1[HT], dW2[HT], db2;
  float *U1, *A1, *E, *U2;

  HODE_DEV void init(float* lds) {
    U1 = lds;
    A1 = lds + 16 * PH;
    E = lds + 32 * PH;
    U2 = E + 16 * PS;
#pragma unroll
    for (int i = 0; i < HT; ++i) dW1[i] = dW2[i] = splat4(0.f);
    db2 = splat4(0.f);
  }
  HODE_DEV void add(const v4 (&u1)[HT], v4 e, const v4& u2, const v4 (&a1)[HT], int g, int n) {
    if (g == GB) e[RB] = 1.0f;
    __syncthreads();  // the previous call's reads are done
#pragma unroll
    for (int i = 0; i < HT; ++i) {
      *reinterpret_cast<v4*>(U1 + n * PH + 16 * i + 4 * g) = u1[i];
      *reinterpret_cast<v4*>(A1 + n * PH + 16 * i + 4 * g) = a1[i];
    }
    *reinterpret_cast<v4*>(E + n * PS + 4 * g) = e;
    *reinterpret_cast<v4*>(U2 + n * PS + 4 * g) = u2;
    __syncthreads();
    const int m = n, kk = g;  // fragment coordinates of this lane
    float eB[4], u2A[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      eB[c] = E[(4 * c + kk) * PS + m];
      u2A[c] = U2[(4 * c + kk) * PS + m];
    }
#pragma unroll
    for (int i = 0; i < HT; ++i) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float au = U1[(4 * c + kk) * PH + 16 * i + m];
        const float ba = A1[(4 * c + kk) * PH + 16 * i + m];
        dW1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(au, eB[c], dW1[i], 0, 0, 0);
        dW2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(u2A[c], ba, dW2[i], 0, 0, 0);
      }
    }
    db2 = db2 + u2;
  }
  // one block of NP floats per wave: [dW1 tiles | dW2 tiles] as [tile][lane][4], then db2[16]
  HODE_DEV void store(float* __restrict__ out, int lane) {
#pragma unroll
    for (int i = 0; i < HT; ++i) {
      *reinterpret_cast<v4*>(out + ((size_t)i * 64 + lane) * 4) = dW1[i];
      *reinterpret_cast<v4*>(out + ((size_t)(HT + i) * 64 + lane) * 4) = dW2[i];
    }
    v4 s;
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = row_sum(db2[r]);  // over the 16 patients of this row group
    if ((lane & 15) == 0) *reinterpret_cast<v4*>(out + 2 * HT * 256 + 4 * (lane >> 4)) = s;
  }
  HODE_DEV static void store_zero(float* __restrict__ out, int lane) {
    for (int i = lane; i < NP; i += 64) out[i] = 0.f;
  }
};

// fixed-order fold of the per-wave blocks into the caller's accumulators (one wave per slot of the block)
template <int D>
__global__ __launch_bounds__(64) void neural_grad_fold_kernel(const float* __restrict__ partials, int n_waves, float* __restrict__ gw1,
                                                      float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2) {
  constexpr int HD = 10 * D, HT = NeuralMf<D>::HT, NP = NeuralGradAcc<D>::NP;
  const int j = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int w = lane; w < n_waves; w += 64) s += partials[(size_t)w * NP + j];
  s = wave_sum(s);
  if (lane != 0) return;
  if (j >= 2 * HT * 256) {
    const int o = j - 2 * HT * 256;
    if (o < D && gb2) gb2[o] += s;
    return;
  }
  const int tile = j / 256, l = (j % 256) / 4, r = j % 4;
  const int rw = 4 * (l >> 4) + r, col = l & 15;
  if (tile < HT) {
    const int hid = 16 * tile + rw;
    if (hid >= HD) return;
    if (col <= D) { if (gw1) gw1[(size_t)hid * (D + 1) + col] += s; }
    else if (col == D + 1) { if (gb1) gb1[hid] += s; }
  } else {
    const int hid = 16 * (tile - HT) + col;
    if (rw < D && hid < HD && gw2) gw2[(size_t)rw * HD + hid] += s;
  }
}

}  // namespace hode
