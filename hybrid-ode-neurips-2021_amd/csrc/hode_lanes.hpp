// Per-lane patient mapping and vector load/store helpers shared by the solver kernels.
#pragma once
#include "hode_common.hpp"

namespace hode {

template <int D>
HODE_DEV void load_vec(const float* __restrict__ p, float (&v)[D]) {
  if constexpr (D % 4 == 0) {
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
      const float4 x = reinterpret_cast<const float4*>(p)[c];
      v[4 * c] = x.x; v[4 * c + 1] = x.y; v[4 * c + 2] = x.z; v[4 * c + 3] = x.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i) v[i] = p[i];
  }
}

// store the state of one patient; with LPP = 4 the four lanes share the chunks (one 16-byte store per lane)
template <int D, int LPP>
HODE_DEV void store_vec(float* __restrict__ p, const float (&v)[D], int q, bool live) {
  if (!live) return;
  if constexpr (D % 4 == 0) {
    constexpr int NC = D / 4;
#pragma unroll
    for (int c0 = 0; c0 < NC; c0 += LPP) {
      float4 x = make_float4(v[4 * c0], v[4 * c0 + 1], v[4 * c0 + 2], v[4 * c0 + 3]);
      int c = c0;
      if constexpr (LPP > 1) {
#pragma unroll
        for (int qq = 1; qq < LPP; ++qq) {
          if (c0 + qq < NC) {
            const bool m = (q == qq);
            x.x = m ? v[4 * (c0 + qq)] : x.x;
            x.y = m ? v[4 * (c0 + qq) + 1] : x.y;
            x.z = m ? v[4 * (c0 + qq) + 2] : x.z;
            x.w = m ? v[4 * (c0 + qq) + 3] : x.w;
          }
        }
        c = c0 + q;
      }
      if (c < NC) reinterpret_cast<float4*>(p)[c] = x;
    }
  } else {
    if (q == 0) {
#pragma unroll
      for (int i = 0; i < D; ++i) p[i] = v[i];
    }
  }
}

template <int LPP>
struct LaneMap {
  int p, q;
  bool live;
  // ppw = patients handled by one wave (<= 64 / LPP).  VALU cost is per wave-instruction, not per lane, and extra waves
  // on one SIMD do not overlap (tools/micro/valu_rate.hip), so at small batches the host lowers ppw until the grid has
  // about one wave per SIMD (1024): 10 patients per wave at B = 10 000 instead of 16.
  HODE_DEV LaneMap(int B, int ppw) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int slot = lane / LPP;
    q = lane % LPP;
    const int pp = wave * ppw + slot;
    live = slot < ppw && pp < B;
    p = live ? pp : min(pp, B - 1);  // idle lanes shadow a valid patient so that cross-lane ops stay well defined
    if (slot >= ppw) p = min(wave * ppw, B - 1);
  }
};


// wave-level reduction of one per-lane partial: over patients for the quad layout (same quad position), over all lanes
// for the lane-per-patient layout.  Result valid in every lane.
template <int LPP>
HODE_DEV float wave_sum_patients(float v) {
  if constexpr (LPP == 4) return wave_sum_stride4(v);
  else return wave_sum(v);
}

}  // namespace hode
