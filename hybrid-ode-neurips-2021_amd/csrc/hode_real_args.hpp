// Shared by hode_real.hip (one patient per lane) and hode_real_mf.hip (matrix cores): weight views, tape rows, kernel
// arguments, the tabulated dose and the stage times of the real-data hybrid rhs (reference model.py:570-657).
#pragma once
#include "hode_common.hpp"

namespace hode {

// flat weight buffer (reference parameter creation order, model.py:588-607)
struct RealW {
  const float *W11, *b11, *w12, *b12, *W21, *b21, *w22, *b22, *Whh, *Whz, *Whr;
  HODE_DEV RealW(const float* f, int H, int M) {
    W11 = f; f += 3 * H; b11 = f; f += H; w12 = f; f += H; b12 = f; f += 1;
    W21 = f; f += 2 * H; b21 = f; f += H; w22 = f; f += H; b22 = f; f += 1;
    Whh = f; f += M * M; Whz = f; f += M * M; Whr = f;
  }
};

// tape rows per (step, stage) instance, each [rows][B]:  Y3 | A11 | U11 | U12 | A21 | U21 | U22 | HH | RH | UR | UZ | UH
struct RealTape {
  int H, M;
  HODE_DEV int y3() const { return 0; }
  HODE_DEV int a11() const { return 3; }
  HODE_DEV int u11() const { return 3 + H; }
  HODE_DEV int u12() const { return 3 + 2 * H; }
  HODE_DEV int a21() const { return 4 + 2 * H; }
  HODE_DEV int u21() const { return 4 + 3 * H; }
  HODE_DEV int u22() const { return 4 + 4 * H; }
  HODE_DEV int hh() const { return 5 + 4 * H; }
  HODE_DEV int rh() const { return 5 + 4 * H + M; }
  HODE_DEV int ur() const { return 5 + 4 * H + 2 * M; }
  HODE_DEV int uz() const { return 5 + 4 * H + 3 * M; }
  HODE_DEV int uh() const { return 5 + 4 * H + 4 * M; }
  HODE_DEV int rows() const { return 5 + 4 * H + 5 * M; }
};

struct RealArgs {
  const float* __restrict__ t;
  const float* __restrict__ y0;
  const float* __restrict__ act;     // [Ta][B] dose table
  const float* __restrict__ theta;   // k_immunity, kel, kel2
  const float* __restrict__ wflat;
  float* __restrict__ h;
  const float* __restrict__ grad_h;
  float* __restrict__ grad_y0;
  float* __restrict__ tape;          // [inst][rows][B]
  float* __restrict__ partials;      // [n_waves][3]
  float* __restrict__ dose_tab;      // [2][Ta + 1][B]: S_n and dS_n/dkel at the integer times n = 0..Ta (see real_dose)
  int B, T, Ta, H, perturb;
};

struct DoseK {
  float v, dk;
};
// Dose(t) = sum_{k <= t} a[k-1] exp(kel (k - t)) and d Dose / d kel.  The reference re-sums all past doses at every rhs
// call (model.py:653-657: an O(T B) reduction per call); evaluated like that here, the k-loop is a chain of dependent
// global loads (35 us per rhs evaluation at T = 120, 5x everything else in the rhs).  With n = floor(t):
//     Dose(t) = exp(kel (n - t)) S_n,   S_n = sum_{k <= n} a[k-1] exp(kel (k - n)) = a[n-1] + exp(-kel) S_{n-1}
//     dS_n/dkel = exp(-kel) (dS_{n-1}/dkel - S_{n-1})
// so every thread tabulates S_n and dS_n/dkel for its patient once per launch (real_dose_table, Ta steps) and an
// evaluation is two loads and one exp.  Same function of (a, kel, t); the summation order differs from the reference's
// (forward recurrence vs one flat sum), inside the test tolerance of the trajectory (tests/test_hip_real.py).
HODE_DEV void real_dose_table(const RealArgs& a, int p, float kel) {
  const size_t B = a.B;
  float* S = a.dose_tab;
  float* dS = a.dose_tab + (size_t)(a.Ta + 1) * B;
  const float E = exp_f32(-kel);
  float s = 0.f, ds = 0.f;
  S[p] = 0.f;
  dS[p] = 0.f;
#pragma unroll 8
  for (int n = 1; n <= a.Ta; ++n) {
    const float an = a.act[(size_t)(n - 1) * B + p];
    ds = E * (ds - s);
    s = __builtin_fmaf(E, s, an);
    S[(size_t)n * B + p] = s;
    dS[(size_t)n * B + p] = ds;
  }
}
HODE_DEV DoseK real_dose(const RealArgs& a, int p, float t, float kel) {
  const int n = min(a.Ta, (int)__builtin_floorf(t));
  if (n < 1) return DoseK{0.f, 0.f};
  const size_t B = a.B;
  const float s = a.dose_tab[(size_t)n * B + p];
  const float ds = a.dose_tab[(size_t)(a.Ta + 1 + n) * B + p];
  const float dlt = (float)n - t;
  const float e = exp_f32(kel * dlt);
  return DoseK{e * s, e * __builtin_fmaf(dlt, s, ds)};
}

HODE_DEV float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

struct RStageTimes {
  float t0, t1, dt, ta, tb, t_first, t_last;
  HODE_DEV RStageTimes(const float* __restrict__ t, int n, int perturb, int method) {
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

}  // namespace hode
