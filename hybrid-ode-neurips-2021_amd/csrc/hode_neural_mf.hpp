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
        A1[i][r] = (hrow < HD && col <= D) ? W1[(size_t)hrow * (D + 1) + col] : 0.f;        // W1[16i+m][4g+r]
        A2[i][r] = (m < D && hcol < HD) ? W2[(size_t)m * HD + hcol] : 0.f;                  // W2[m][16i+4g+r]
        A3[i][r] = (hrow < HD && col < D) ? W2[(size_t)col * HD + hrow] : 0.f;              // W2^T[16i+m][4g+r]
        A4[i][r] = (hcol < HD && m <= D) ? W1[(size_t)hcol * (D + 1) + m] : 0.f;            // W1^T[m][16i+4g+r]
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + 4 * g + r;
        bias1[i][r] = row < HD ? a.b1[row] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bias2[r] = (4 * g + r) < D ? a.b2[4 * g + r] : 0.f;
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
    for (int i = 0; i < HT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) a1[i][r] = tanh_f32(acc[i][r]);
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
    const v4 zs = (z[0] + z[1]) + (z[2] + z[3]);
    v4 k;
#pragma unroll
    for (int r = 0; r < 4; ++r) k[r] = tanh_f32(zs[r]);
    return k;
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
    for (int i = 0; i < HT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) u1[i][r] = acc[i][r] * __builtin_fmaf(-a1[i][r], a1[i][r], 1.0f);
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

}  // namespace hode
