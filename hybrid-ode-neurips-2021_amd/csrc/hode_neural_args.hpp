// Shared by hode_neural.hip (one patient per lane) and hode_neural_mf.hip (matrix cores): kernel arguments, stage times and
// the impulse dose of the NeuralODE rhs (reference model.py:969-1026).
#pragma once
#include "hode_common.hpp"

namespace hode {

struct NeuralArgs {
  const float* __restrict__ t;
  const float* __restrict__ y0;
  const float* __restrict__ dosage;
  const float* __restrict__ dose_times;
  const float* __restrict__ w1;   // [HD][D+1]
  const float* __restrict__ b1;   // [HD]
  const float* __restrict__ w2t;  // [HD][D]   (W2 transposed once per call: column n of W2 is contiguous)
  const float* __restrict__ b2;   // [D]
  const float* __restrict__ w2;   // [D][HD] as given by the caller (the matrix-core kernels gather from it)
  float* __restrict__ h;
  const float* __restrict__ grad_h;
  float* __restrict__ grad_y0;
  float* __restrict__ a1t;   // tapes (backward)
  float* __restrict__ u1t;
  float* __restrict__ yet;
  float* __restrict__ u2t;
  int B, T, K, perturb;
};

constexpr float kThird = (float)(1.0 / 3.0);
constexpr float kTwoThird = (float)(2.0 / 3.0);

struct NStageTimes {
  float t0, t1, dt, ta, tb, t_first, t_last;
  HODE_DEV NStageTimes(const float* __restrict__ t, int n, int perturb, int method) {
    t0 = t[n];
    t1 = t[n + 1];
    dt = t1 - t0;
    t_first = perturb ? nextafter_up(t0) : t0;
    t_last = perturb ? nextafter_down(t1) : t1;
    if (method == HODE_METHOD_RK4_38) {
      ta = add_rn(t0, mul_rn(dt, kThird));
      tb = add_rn(t0, mul_rn(dt, kTwoThird));
    } else {
      ta = add_rn(t0, mul_rn(0.5f, dt));
      tb = ta;
    }
  }
};

HODE_DEV float neural_dose(const NeuralArgs& a, int p, float dosage, float t) {
  float cnt = 0.f;
  for (int k = 0; k < a.K; ++k) cnt += (a.dose_times[(size_t)p * a.K + k] == t) ? 1.0f : 0.0f;
  return dosage * cnt;
}


}  // namespace hode
