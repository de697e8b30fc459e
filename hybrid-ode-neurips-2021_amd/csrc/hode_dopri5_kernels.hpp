// Adaptive Dormand-Prince 5(4) solve of the hybrid Roche ODE and its discrete adjoint, gfx950.
//
// Replaces torchdiffeq.odeint(method="dopri5") -- the reference default (sim_config.py:50) -- as called at
// model.py:1116, and the autograd replay of its accepted steps.  CPU restatement: oracle/solvers.py::_odeint_dopri5.
//
// Structure: ONE LAUNCH PER ATTEMPTED STEP.  The controller record (t0, dt, counters, done flag)
// lives in device memory, double-buffered by attempt parity; the batch-global RMS error norm is reduced per wave into
// a partial array that every wave of the NEXT launch folds in the same fixed order, so every lane takes the same
// accept/reject decision without atomics or a grid barrier.  The current state y_n is the tape itself
// (tape_y[n_acc]); a candidate y1 is written to tape_y[n_acc+1] and simply overwritten if the attempt is rejected.
// Stage derivatives of the latest attempt live in kbuf[7][B][D] (kbuf[0] = f0 of the current state, FSAL; the owner-layout
// attempt writes kbuf[1..5] only when an output time lies inside the step, the one case in which they are read back).
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

#include "../../include/hode.h"
#include "hode_host.hpp"
#include "hode_lanes.hpp"
#include "hode_roche.hpp"

namespace hode {

struct DpCtrl {
  double t0;   // start time of the step being attempted
  double dt;   // its size
  float h0, d1;  // Hairer initial-step scratch
  int n_acc, n_rej, j_next, done, status, attempt;
};

// What the backward needs of Hairer's initial step size (include/hode.h: hode_dopri5_init_record has the same layout).
// torchdiffeq computes dt_0 = min(100 h0, h1) from y0, f0, f1 OUTSIDE no_grad (rk_common.py _before_integrate), so the
// reference's loss.backward() differentiates it whenever the first attempt is the one that gets accepted.
struct DpInit {
  float h0, d0, d1, d2, h1;
  int first_accepted;  // 1: attempt 0 (the one that ran with dt_0) was accepted
  float sigma;         // backward: d loss / d dt_0 (diagnostics)
  int pad;
};

struct DpArgs {
  const float* __restrict__ t;
  const float* __restrict__ y0;
  const float* __restrict__ dosage;
  const float* __restrict__ dose_times;
  const float* __restrict__ theta;
  const float* __restrict__ w1;
  const float* __restrict__ b1;
  float* __restrict__ h;
  DpCtrl* ctrl;            // [2]
  DpInit* init;            // [1]
  float* partials;         // [2][2 * n_waves]
  unsigned long long* slots;  // [2][n_waves]: (sequence number << 32 | error-norm partial) of the persistent attempt loop
  int max_iters;           // persistent attempt loop: attempts this launch may run
  float* kbuf;             // [7][B][D]
  double* tape_t;          // [max_steps]
  double* tape_dt;         // [max_steps]
  int* tape_j;             // [2 * max_steps]: first / one-past-last output index interpolated inside the step
  float* tape_y;           // [max_steps + 1][B][D]
  const float* __restrict__ grad_h;
  float* __restrict__ grad_y0;
  float* __restrict__ grad_partials;  // [n_waves][P]
  int B, T, K, n_waves, max_steps, attempt, n_acc, ppw;
  int ring;   // HODE_FLAG_NO_TAPE: tape_y holds two rows (current state / candidate) addressed by step parity
  int hill2;  // -1: decide on the device from theta[0..1]; 0 / 1: decided by the host (attempt launches: one dependent
              // scalar round trip less per launch, hode_dopri5.hip reads the two exponents back once per solve)
  float rtol, atol;
};

// ---- tableau, rounded to fp32 exactly as torchdiffeq casts its float64 tensors to the state dtype
#define F32(x) ((float)(x))
__device__ constexpr float kDpAlpha[6] = {F32(1.0 / 5), F32(3.0 / 10), F32(4.0 / 5), F32(8.0 / 9), 1.0f, 1.0f};
__device__ constexpr float kDpBeta[6][6] = {
    {F32(1.0 / 5), 0, 0, 0, 0, 0},
    {F32(3.0 / 40), F32(9.0 / 40), 0, 0, 0, 0},
    {F32(44.0 / 45), F32(-56.0 / 15), F32(32.0 / 9), 0, 0, 0},
    {F32(19372.0 / 6561), F32(-25360.0 / 2187), F32(64448.0 / 6561), F32(-212.0 / 729), 0, 0},
    {F32(9017.0 / 3168), F32(-355.0 / 33), F32(46732.0 / 5247), F32(49.0 / 176), F32(-5103.0 / 18656), 0},
    {F32(35.0 / 384), 0.0f, F32(500.0 / 1113), F32(125.0 / 192), F32(-2187.0 / 6784), F32(11.0 / 84)},
};
__device__ constexpr float kDpErr[7] = {
    F32(35.0 / 384 - 1951.0 / 21600), 0.0f, F32(500.0 / 1113 - 22642.0 / 50085), F32(125.0 / 192 - 451.0 / 720),
    F32(-2187.0 / 6784 - -12231.0 / 42400), F32(11.0 / 84 - 649.0 / 6300), F32(-1.0 / 60.0)};
__device__ constexpr float kDpMid[7] = {
    F32(6025192743.0 / 30085553152.0 / 2), 0.0f, F32(51252292925.0 / 65400821598.0 / 2),
    F32(-2691868925.0 / 45128329728.0 / 2), F32(187940372067.0 / 1594534317056.0 / 2),
    F32(-1776094331.0 / 19743644256.0 / 2), F32(11237099.0 / 235043384.0 / 2)};
#undef F32

// row of tape_y that holds the state at the start of accepted step n
HODE_DEV size_t dp_tape_row(const DpArgs& a, int n) { return (size_t)(a.ring ? (n & 1) : n); }

template <bool K1>
HODE_DEV DoseSched<K1> dp_load_dose(const DpArgs& a, int p) {
  DoseSched<K1> ds;
  ds.dosage = a.dosage[p];
  ds.K = a.K;
  ds.taus = a.dose_times + (size_t)p * a.K;
  ds.tau0 = K1 ? ds.taus[0] : 0.f;
  return ds;
}

// fold this launch's view of a per-wave partial array (fixed order => identical result in every lane of every wave).
// The loads of a lane are issued together (16 in flight) before they are summed: a dependent load-add chain here costs
// one L2 round trip per 64 partials on EVERY attempt.
HODE_DEV float fold_waves(const float* __restrict__ part, int n_waves, int stride, int off) {
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int base = 0; base < n_waves; base += 64 * 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int w = base + lane + 64 * j;
      v[j] = w < n_waves ? part[(size_t)w * stride + off] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
  }
  return wave_sum(s);
}

// stage time i (2..7) of a step, fp32 like torchdiffeq's _runge_kutta_step; alpha == 1 -> just before t1
HODE_DEV float dp_stage_time(int i, float t0f, float dtf, float t1f) {
  const float al = kDpAlpha[i - 2];
  return al == 1.0f ? nextafter_down(t1f) : add_rn(t0f, mul_rn(al, dtf));
}

// all seven stage derivatives of one attempt from (y0, k[0] = f0).  Y[i] (i = 1..6) = stage state of stage i+1,
// Y[6] = y1.  s[i] = this lane's tanh outputs of stage i+1 (s[0] belongs to k[0] and is filled by the caller).
template <int D, int LPP, bool ABLATE, bool HILL2, bool K1>
HODE_DEV void dp_stages(const RocheTheta& th, const MlSlice<D, LPP>& ml, const DoseSched<K1>& ds, const float (&y0)[D],
                        float t0f, float dtf, float t1f, float (&k)[7][D], float (&Y)[7][D],
                        float (&s)[7][MlSlice<D, LPP>::MR], DoseVal (&dv)[7]) {
#pragma unroll
  for (int i = 2; i <= 7; ++i) {
    const float ti = dp_stage_time(i, t0f, dtf, t1f);
#pragma unroll
    for (int c = 0; c < D; ++c) {
      float acc = y0[c];
#pragma unroll
      for (int m = 0; m < i - 1; ++m) acc = __builtin_fmaf(kDpBeta[i - 2][m] * dtf, k[m][c], acc);
      Y[i - 1][c] = acc;
    }
    dv[i - 1] = ds.at(ti, th.kel);
    roche_rhs<D, LPP, ABLATE, HILL2>(th, ml, dv[i - 1].v, Y[i - 1], k[i - 1], s[i - 1]);
  }
}

// ------------------------------------------------------------------------------------------------ init kernels
// init1: f0 = f(t[0], y0); h[0] = y0; tape_y[0] = y0; kbuf[0] = f0; partial sums of (y0/scale)^2 and (f0/scale)^2
template <int D, int LPP, bool ABLATE, bool HILL2, bool K1>
HODE_DEV void dp_init1_body(const DpArgs& a) {
  using Ml = MlSlice<D, LPP>;
  const LaneMap<LPP> lm(a.B, a.ppw);
  const RocheTheta th = load_theta(a.theta, ABLATE);
  Ml ml;
  ml.load(a.w1, a.b1, lm.q);
  const DoseSched<K1> ds = dp_load_dose<K1>(a, lm.p);
  float y[D], f0[D], own[Ml::MR];
  load_vec<D>(a.y0 + (size_t)lm.p * D, y);
  roche_rhs<D, LPP, ABLATE, HILL2>(th, ml, ds.at(a.t[0], th.kel).v, y, f0, own);
  store_vec<D, LPP>(a.h + (size_t)lm.p * D, y, lm.q, lm.live);
  store_vec<D, LPP>(a.tape_y + (size_t)lm.p * D, y, lm.q, lm.live);
  store_vec<D, LPP>(a.kbuf + (size_t)lm.p * D, f0, lm.q, lm.live);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float scale = a.atol + __builtin_fabsf(y[c]) * a.rtol;
    const float u = div_f32(y[c], scale), v = div_f32(f0[c], scale);
    s0 = __builtin_fmaf(u, u, s0);
    s1 = __builtin_fmaf(v, v, s1);
  }
  const float m = (lm.live && lm.q == 0) ? 1.0f : 0.0f;  // count every patient once
  s0 = wave_sum(s0 * m);
  s1 = wave_sum(s1 * m);
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) {
    a.partials[2 * wave] = s0;
    a.partials[2 * wave + 1] = s1;
  }
}

// init2: h0 from (d0, d1); f1 = f(t0 + h0, y0 + h0 f0); partial of ((f1 - f0)/scale)^2; wave 0 seeds the controller
template <int D, int LPP, bool ABLATE, bool HILL2, bool K1>
HODE_DEV void dp_init2_body(const DpArgs& a) {
  using Ml = MlSlice<D, LPP>;
  const LaneMap<LPP> lm(a.B, a.ppw);
  const RocheTheta th = load_theta(a.theta, ABLATE);
  Ml ml;
  ml.load(a.w1, a.b1, lm.q);
  const DoseSched<K1> ds = dp_load_dose<K1>(a, lm.p);
  const float cnt = (float)a.B * (float)D;
  const float d0 = __builtin_sqrtf(fold_waves(a.partials, a.n_waves, 2, 0) / cnt);
  const float d1 = __builtin_sqrtf(fold_waves(a.partials, a.n_waves, 2, 1) / cnt);
  const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : div_f32(0.01f * d0, d1);
  float y[D], f0[D], y1[D], f1[D], own[Ml::MR];
  load_vec<D>(a.y0 + (size_t)lm.p * D, y);
  load_vec<D>(a.kbuf + (size_t)lm.p * D, f0);
#pragma unroll
  for (int c = 0; c < D; ++c) y1[c] = __builtin_fmaf(h0, f0[c], y[c]);
  const float t0f = a.t[0];
  roche_rhs<D, LPP, ABLATE, HILL2>(th, ml, ds.at(add_rn(t0f, h0), th.kel).v, y1, f1, own);
  float s2 = 0.f;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float scale = a.atol + __builtin_fabsf(y[c]) * a.rtol;
    const float u = div_f32(f1[c] - f0[c], scale);
    s2 = __builtin_fmaf(u, u, s2);
  }
  s2 = wave_sum(s2 * ((lm.live && lm.q == 0) ? 1.0f : 0.0f));
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float* pout = a.partials + (size_t)2 * a.n_waves;  // second half: input of attempt 0
  if ((threadIdx.x & 63) == 0) pout[2 * (gid >> 6)] = s2;
  if (gid == 0) {
    DpCtrl c;
    c.t0 = (double)t0f;
    c.dt = 0.0;
    c.h0 = h0;
    c.d1 = d1;
    c.n_acc = 0; c.n_rej = 0; c.j_next = 1; c.done = (a.T <= 1) ? 1 : 0; c.status = 0; c.attempt = 0;
    a.ctrl[0] = c;
    a.ctrl[1] = c;  // defined contents for the record attempt 0 will fill (its status word is OR-ed into)
    DpInit in{};
    in.h0 = h0; in.d0 = d0; in.d1 = d1;
    *a.init = in;
  }
}

// fold_waves split in two so that the loads can be requested at kernel entry, next to everything else the attempt reads:
// fold_issue requests the first 1 024 partials without a branch (indices clamped, masked when summed), fold_finish adds
// them in fold_waves' order.
struct FoldHead {
  float v[16];
};
HODE_DEV FoldHead fold_issue(const float* __restrict__ part, int n_waves, int stride, int off) {
  const int lane = threadIdx.x & 63;
  FoldHead h;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int w = min(lane + 64 * j, n_waves - 1);
    h.v[j] = part[(size_t)w * stride + off];
  }
  return h;
}
HODE_DEV float fold_finish(const FoldHead& h, const float* __restrict__ part, int n_waves, int stride, int off) {
  const int lane = threadIdx.x & 63;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += (lane + 64 * j < n_waves) ? h.v[j] : 0.f;
  for (int base = 64 * 16; base < n_waves; base += 64 * 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int w = base + lane + 64 * j;
      v[j] = w < n_waves ? part[(size_t)w * stride + off] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
  }
  return wave_sum(s);
}

// Step-size factor of the controller (torchdiffeq _optimal_step_size with dopri5's constants: safety .9, ifactor 10,
// dfactor .2, order 5), in fp64 like the clock: min(10, max(0.9 ratio^(-1/5), ratio < 1 ? 1 : 0.2)); 10 when ratio == 0;
// NaN propagates (torch.max / torch.min do).  Every wave evaluates it on every attempt, so the library pow(double) --
// ~0.7 us with in-kernel clock stamps, DESIGN.md section 5 -- is replaced by the fifth root it is: where the clamps do not decide the result outright
// (5e-6 <= ratio <= 1900), y = ratio^(-1/5) from an fp32 seed and three division-free Newton steps
// y <- y (6 - ratio y^5) / 5 in fp64 (error 3 e^2 per step: 1e-6 -> 3e-12 -> 3e-23; the third is a guard), i.e. the
// value pow() gives to within 4 ulp of fp64 (checked over 2e5 ratios on the host), 5e-16 relative on dt.
HODE_DEV double dp_step_factor(float ratio) {
  if (!(ratio == ratio)) return __builtin_nan("");
  if (ratio == 0.0f) return 10.0;
  const double dfac = ratio < 1.0f ? 1.0 : 0.2;
  if (ratio < 5e-6f) return 10.0;   // 0.9 ratio^(-1/5) > 10.3
  if (ratio > 1900.0f) return 0.2;  // 0.9 ratio^(-1/5) < 0.199
  const double r = (double)ratio;
  double y = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(ratio));
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const double y2 = y * y;
    const double y5 = y2 * y2 * y;
    y = y * (0.2 * __builtin_fma(-r, y5, 6.0));
  }
  return fmin(10.0, fmax(0.9 * y, dfac));
}

// ------------------------------------------------------------------------------------------------ attempt kernel
template <int D, int LPP, bool ABLATE, bool HILL2, bool K1>
HODE_DEV void dp_attempt_body(const DpArgs& a) {
  using Ml = MlSlice<D, LPP>;
  constexpr int MR = Ml::MR;
  const int par = a.attempt & 1;
  const DpCtrl cin = a.ctrl[par];
  DpCtrl* cout = a.ctrl + (par ^ 1);
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (cin.done) {
    if (gid == 0) *cout = cin;
    return;
  }
  // partial layout: init2 wrote attempt 0's input to the second half; attempt k writes half (k & 1), attempt k+1 reads it
  const float* pin = a.partials + (size_t)(par ^ 1) * 2 * a.n_waves;
  float* pout = a.partials + (size_t)par * 2 * a.n_waves;

  const LaneMap<LPP> lm(a.B, a.ppw);
  const size_t row = (size_t)a.B * D;
  const size_t poff = (size_t)lm.p * D;
  const float cnt = (float)a.B * (float)D;

  DpCtrl c = cin;
  float y[D], f0[D];
  if (cin.attempt == 0) {
    // finish the initial step selection (order 4): h1 from d1, d2
    const float d2 = div_f32(__builtin_sqrtf(fold_waves(pin, a.n_waves, 2, 0) / cnt), cin.h0);
    float h1;
    if (cin.d1 <= 1e-15f && d2 <= 1e-15f) h1 = fmaxf(1e-6f, cin.h0 * 1e-3f);
    else h1 = powf(div_f32(0.01f, fmaxf(cin.d1, d2)), 0.2f);
    c.dt = (double)fminf(100.0f * cin.h0, h1);
    if (gid == 0) { a.init->d2 = d2; a.init->h1 = h1; }
    load_vec<D>(a.tape_y + poff, y);
    load_vec<D>(a.kbuf + poff, f0);
  } else {
    const float ratio = __builtin_sqrtf(fold_waves(pin, a.n_waves, 2, 0) / cnt);
    const bool accept = ratio <= 1.0f;
    if (cin.attempt == 1 && gid == 0) a.init->first_accepted = accept ? 1 : 0;
    const double t1 = cin.t0 + cin.dt;
    if (accept) {
      // candidate becomes the state; emit every output time inside (t0, t1] from the quartic dense output
      float ya[D], k7[D];
      load_vec<D>(a.tape_y + dp_tape_row(a, cin.n_acc + 1) * row + poff, y);
      load_vec<D>(a.kbuf + 6 * row + poff, k7);
      int j = cin.j_next;
      if (j < a.T && (double)a.t[j] <= t1) {
        float k1[D], ym[D];
        load_vec<D>(a.tape_y + dp_tape_row(a, cin.n_acc) * row + poff, ya);
        load_vec<D>(a.kbuf + poff, k1);
        const float dtf = (float)cin.dt;
#pragma unroll
        for (int cc = 0; cc < D; ++cc) ym[cc] = ya[cc];
        for (int m = 0; m < 7; ++m) {
          float km[D];
          load_vec<D>(a.kbuf + (size_t)m * row + poff, km);
          const float w = dtf * kDpMid[m];
#pragma unroll
          for (int cc = 0; cc < D; ++cc) ym[cc] = __builtin_fmaf(w, km[cc], ym[cc]);
        }
        float ca[D], cb[D], cc_[D], cd[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const float f0i = k1[i], f1i = k7[i], y0i = ya[i], y1i = y[i], ymi = ym[i];
          ca[i] = 2.0f * dtf * (f1i - f0i) - 8.0f * (y1i + y0i) + 16.0f * ymi;
          cb[i] = dtf * (5.0f * f0i - 3.0f * f1i) + 18.0f * y0i + 14.0f * y1i - 32.0f * ymi;
          cc_[i] = dtf * (f1i - 4.0f * f0i) - 11.0f * y0i - 5.0f * y1i + 16.0f * ymi;
          cd[i] = dtf * f0i;
        }
        for (; j < a.T && (double)a.t[j] <= t1; ++j) {
          const float x = (float)(((double)a.t[j] - cin.t0) / (t1 - cin.t0));
          const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
          float out[D];
#pragma unroll
          for (int i = 0; i < D; ++i) out[i] = (((ya[i] + x * cd[i]) + x2 * cc_[i]) + x3 * cb[i]) + x4 * ca[i];
          store_vec<D, LPP>(a.h + (size_t)j * row + poff, out, lm.q, lm.live);
        }
      }
      if (gid == 0) {
        a.tape_t[cin.n_acc] = cin.t0;
        a.tape_dt[cin.n_acc] = cin.dt;
        a.tape_j[2 * cin.n_acc] = cin.j_next;
        a.tape_j[2 * cin.n_acc + 1] = j;
      }
      c.j_next = j;
      c.n_acc = cin.n_acc + 1;
      c.t0 = t1;
#pragma unroll
      for (int i = 0; i < D; ++i) f0[i] = k7[i];
    } else {
      c.n_rej = cin.n_rej + 1;
      load_vec<D>(a.tape_y + dp_tape_row(a, cin.n_acc) * row + poff, y);
      load_vec<D>(a.kbuf + poff, f0);
    }
    // controller (torchdiffeq _optimal_step_size): fp64 clock, constants of dopri5 (safety .9, ifactor 10, dfactor .2)
    c.dt = cin.dt * dp_step_factor(ratio);
  }
  c.attempt = cin.attempt + 1;

  // ---- termination / failure checks (uniform over the grid: every lane computes the same record)
  bool stop = false;
  if (c.status) { c.done = 1; stop = true; }  // a lane of the previous attempt flagged a non-finite state
  if (!stop && c.j_next >= a.T) { c.done = 1; stop = true; }
  if (!stop && !(c.t0 + c.dt > c.t0)) { c.status |= HODE_STATUS_DT_UNDERFLOW; c.done = 1; stop = true; }
  if (!stop && c.n_acc >= a.max_steps) { c.status |= HODE_STATUS_MAX_STEPS; c.done = 1; stop = true; }
  if (stop) {
    if (gid == 0) *cout = c;
    return;
  }

  // ---- new attempt from (y, f0) at (t0, dt)
  const RocheTheta th = load_theta(a.theta, ABLATE);
  Ml ml;
  ml.load(a.w1, a.b1, lm.q);
  const DoseSched<K1> ds = dp_load_dose<K1>(a, lm.p);
  const float t0f = (float)c.t0, dtf = (float)c.dt, t1f = (float)(c.t0 + c.dt);
  float k[7][D], Y[7][D], s[7][MR];
  DoseVal dv[7];
#pragma unroll
  for (int i = 0; i < D; ++i) k[0][i] = f0[i];
  dp_stages<D, LPP, ABLATE, HILL2, K1>(th, ml, ds, y, t0f, dtf, t1f, k, Y, s, dv);
  float se = 0.f;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float err = 0.f;
#pragma unroll
    for (int m = 0; m < 7; ++m) err = __builtin_fmaf(dtf * kDpErr[m], k[m][i], err);
    const float tol = a.atol + a.rtol * fmaxf(__builtin_fabsf(y[i]), __builtin_fabsf(Y[6][i]));
    const float u = div_f32(err, tol);
    se = __builtin_fmaf(u, u, se);
    bad |= !__builtin_isfinite(y[i]);
  }
  se = wave_sum(se * ((lm.live && lm.q == 0) ? 1.0f : 0.0f));
  if ((threadIdx.x & 63) == 0 && (gid >> 6) < a.n_waves) pout[2 * (gid >> 6)] = se;
  store_vec<D, LPP>(a.tape_y + dp_tape_row(a, c.n_acc + 1) * row + poff, Y[6], lm.q, lm.live);
#pragma unroll
  for (int m = 0; m < 7; ++m) store_vec<D, LPP>(a.kbuf + (size_t)m * row + poff, k[m], lm.q, lm.live);
  if (bad && lm.live) atomicOr(&cout->status, HODE_STATUS_NONFINITE);  // torchdiffeq asserts on the state before a step
  if (gid == 0) {
    // status may be OR-ed concurrently by other lanes: write the other fields, OR our own bits
    cout->t0 = c.t0; cout->dt = c.dt; cout->h0 = c.h0; cout->d1 = c.d1;
    cout->n_acc = c.n_acc; cout->n_rej = c.n_rej; cout->j_next = c.j_next; cout->done = c.done; cout->attempt = c.attempt;
    if (c.status) atomicOr(&cout->status, c.status);
  }
}

// ------------------------------------------------------------------------------------ attempt kernel, owner layout
// Same attempt, same tape, for the quad layout (LPP = 4, D = 8 / 12) -- but nothing is replicated across the quad.  The
// launch is bound by the ~1 750 instructions a wave issues (section 5 of DESIGN.md), and in dp_attempt_body every lane of
// a patient's quad repeats the 252-FMA stage combinations, the error estimate and the dense output of ALL D components,
// and all-gathers every stage derivative.  Here lane q OWNS components {q} u {4 + q MR + r}: the expert component q and
// the learned rows it evaluates anyway (MlSlice).  Stage states, stage derivatives, the error estimate, the dense output
// and every load / store exist only for the NO = 1 + MR owned components; the one thing a stage needs from its
// neighbours -- the full stage state as the rhs operand -- is read through DPP quad broadcasts of the owners' registers.
template <int D>
struct DpOwn {
  static constexpr int MR = (D - 4) / 4;
  static constexpr int NO = 1 + MR;
  // component index of owned slot s for quad position q
  HODE_DEV static int comp(int s, int q) { return s == 0 ? q : 4 + q * MR + (s - 1); }
  HODE_DEV static void load(const float* __restrict__ base, int q, float (&o)[NO]) {
    o[0] = base[q];
#pragma unroll
    for (int r = 0; r < MR; ++r) o[1 + r] = base[4 + q * MR + r];
  }
  HODE_DEV static void store(float* __restrict__ base, int q, const float (&o)[NO], bool live) {
    if (!live) return;
    base[q] = o[0];
#pragma unroll
    for (int r = 0; r < MR; ++r) base[4 + q * MR + r] = o[1 + r];
  }
  // full-state component C (compile time) out of the owners' registers
  template <int C>
  HODE_DEV static float full(const float (&o)[NO]) {
    if constexpr (C < 4) return quad_bcast<C>(o[0]);
    else return quad_bcast<(C - 4) / MR>(o[1 + (C - 4) % MR]);
  }
};

template <int D, bool ABLATE, bool HILL2, int... C>
HODE_DEV void dp_own_rhs_impl(const RocheTheta& th, const MlSlice<D, 4>& ml, float dose, int q,
                              const float (&Yo)[DpOwn<D>::NO], float (&ko)[DpOwn<D>::NO], std::integer_sequence<int, C...>) {
  using Own = DpOwn<D>;
  const float Y[D] = {Own::template full<C>(Yo)...};
  const float dis = Y[0], ir = Y[1], imm = Y[2], d2 = Y[3];
  float k0, k1, k2, k3;
  if constexpr (!ABLATE) {
    const float immp = pow_hill<HILL2>(imm, th.hc);
    const float irp = pow_hill<HILL2>(ir, th.hp);
    const float ecp = pow_hill<HILL2>(th.ec50, th.hp);
    k0 = dis * th.kprog - dis * immp * th.kci - dis * ir * th.kcir;
    k1 = dis * th.kid - ir * th.koff + dis * ir * th.kfb + div_f32(irp * th.emax, ecp + irp) - d2 * ir * th.kdexa;
    k2 = ir * th.kim;
    k3 = th.kel * dose - th.kel * d2;
  } else {
    k0 = ir;
    k1 = -1.0f * dis * th.th1;
    k2 = d2;
    k3 = -1.0f * imm * th.th2;
  }
  ko[0] = q == 0 ? k0 : (q == 1 ? k1 : (q == 2 ? k2 : k3));
#pragma unroll
  for (int r = 0; r < Own::MR; ++r) {
    float z = ml.b[r];
#pragma unroll
    for (int i = 0; i < D; ++i) z = __builtin_fmaf(ml.w[r][i], Y[i], z);
    ko[1 + r] = tanh_f32(z);
  }
}
template <int D, bool ABLATE, bool HILL2>
HODE_DEV void dp_own_rhs(const RocheTheta& th, const MlSlice<D, 4>& ml, float dose, int q, const float (&Yo)[DpOwn<D>::NO],
                         float (&ko)[DpOwn<D>::NO]) {
  dp_own_rhs_impl<D, ABLATE, HILL2>(th, ml, dose, q, Yo, ko, std::make_integer_sequence<int, D>{});
}

template <int D, bool ABLATE, bool HILL2, bool K1>
HODE_DEV void dp_attempt_body_own(const DpArgs& a) {
  using Own = DpOwn<D>;
  constexpr int NO = Own::NO;
  const int par = a.attempt & 1;
  DpCtrl* cout = a.ctrl + (par ^ 1);
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const float* pin = a.partials + (size_t)(par ^ 1) * 2 * a.n_waves;
  float* pout = a.partials + (size_t)par * 2 * a.n_waves;
  const LaneMap<4> lm(a.B, a.ppw);
  const int q = lm.q;
  const size_t row = (size_t)a.B * D;
  const size_t poff = (size_t)lm.p * D;
  const float cnt = (float)a.B * (float)D;
  // Diagnostics (build with HODE_DP_FLAGS=-DHODE_DP_STAMPS, tools/dp_stamp_probe.py): lane 0 of block 0 and of block
  // n_waves / 2 stamp s_memtime behind a full s_waitcnt at the waits of the attempt, into the (forward-idle) gradient
  // partial array.  Measured timeline of an attempt at 10 000 x 12 (2.4 GHz): DESIGN.md section 5.
#ifdef HODE_DP_STAMPS
  const bool stamp = a.attempt < 2300 && (threadIdx.x == 0) && (blockIdx.x == 0 || blockIdx.x == (unsigned)a.n_waves / 2);
  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.grad_partials) + ((size_t)a.attempt * 2 + (blockIdx.x ? 1 : 0)) * 8;
#define HODE_STAMP(i) if (stamp) { __builtin_amdgcn_s_waitcnt(0); dbg[i] = __builtin_amdgcn_s_memtime(); }
#else
#define HODE_STAMP(i)
#endif
  HODE_STAMP(0)
  // level 1: everything addressed by the kernel arguments alone is requested before anything is waited for
  const FoldHead head = fold_issue(pin, a.n_waves, 2, 0);
  const RocheTheta th = load_theta(a.theta, ABLATE);
  MlSlice<D, 4> ml;
  ml.load(a.w1, a.b1, q);
  const DoseSched<K1> ds = dp_load_dose<K1>(a, lm.p);
  const DpCtrl cin = a.ctrl[par];
  __builtin_amdgcn_sched_barrier(0);
  HODE_STAMP(1)   // level-1 loads back (the stamp waits for everything outstanding)
  if (cin.done) {
    if (gid == 0) *cout = cin;
    return;
  }
  // level 2: addressed by the controller record; BOTH candidate states, so that the decision starts no further round trip
  float y_old[NO], k_first[NO], y_new[NO], k_last[NO];
  Own::load(a.tape_y + dp_tape_row(a, cin.n_acc) * row + poff, q, y_old);
  Own::load(a.kbuf + poff, q, k_first);
  Own::load(a.tape_y + dp_tape_row(a, cin.n_acc + 1) * row + poff, q, y_new);  // row n_acc + 1 <= max_steps exists
  Own::load(a.kbuf + 6 * row + poff, q, k_last);
  const float t_next = a.t[min(cin.j_next, a.T - 1)];
  __builtin_amdgcn_sched_barrier(0);
  HODE_STAMP(2)   // level-2 loads back

  DpCtrl c = cin;
  float y[NO], f0[NO];
  if (cin.attempt == 0) {
    const float d2 = div_f32(__builtin_sqrtf(fold_finish(head, pin, a.n_waves, 2, 0) / cnt), cin.h0);
    float h1;
    if (cin.d1 <= 1e-15f && d2 <= 1e-15f) h1 = fmaxf(1e-6f, cin.h0 * 1e-3f);
    else h1 = powf(div_f32(0.01f, fmaxf(cin.d1, d2)), 0.2f);
    c.dt = (double)fminf(100.0f * cin.h0, h1);
    if (gid == 0) { a.init->d2 = d2; a.init->h1 = h1; }
#pragma unroll
    for (int s = 0; s < NO; ++s) {
      y[s] = y_old[s];  // n_acc == 0
      f0[s] = k_first[s];
    }
  } else {
    const float ratio = __builtin_sqrtf(fold_finish(head, pin, a.n_waves, 2, 0) / cnt);
    const double t1 = cin.t0 + cin.dt;
    if (cin.attempt == 1 && gid == 0) a.init->first_accepted = ratio <= 1.0f ? 1 : 0;
    if (ratio <= 1.0f) {
#pragma unroll
      for (int s = 0; s < NO; ++s) {
        y[s] = y_new[s];
        f0[s] = k_last[s];  // FSAL: k7 of the accepted step
      }
      int j = cin.j_next;
      if (j < a.T && (double)t_next <= t1) {
        // quartic dense output of the accepted step on the owned components (the previous launch stored k1..k7 for it)
        float ya[NO], k1[NO], ym[NO];
#pragma unroll
        for (int s = 0; s < NO; ++s) {
          ya[s] = y_old[s];
          k1[s] = k_first[s];
        }
        const float dtf = (float)cin.dt;
#pragma unroll
        for (int s = 0; s < NO; ++s) ym[s] = ya[s];
        for (int m = 0; m < 7; ++m) {
          float km[NO];
          Own::load(a.kbuf + (size_t)m * row + poff, q, km);
          const float w = dtf * kDpMid[m];
#pragma unroll
          for (int s = 0; s < NO; ++s) ym[s] = __builtin_fmaf(w, km[s], ym[s]);
        }
        float ca[NO], cb[NO], cc_[NO], cd[NO];
#pragma unroll
        for (int s = 0; s < NO; ++s) {
          const float f0i = k1[s], f1i = f0[s], y0i = ya[s], y1i = y[s], ymi = ym[s];
          ca[s] = 2.0f * dtf * (f1i - f0i) - 8.0f * (y1i + y0i) + 16.0f * ymi;
          cb[s] = dtf * (5.0f * f0i - 3.0f * f1i) + 18.0f * y0i + 14.0f * y1i - 32.0f * ymi;
          cc_[s] = dtf * (f1i - 4.0f * f0i) - 11.0f * y0i - 5.0f * y1i + 16.0f * ymi;
          cd[s] = dtf * f0i;
        }
        for (; j < a.T && (double)a.t[j] <= t1; ++j) {
          const float x = (float)(((double)a.t[j] - cin.t0) / (t1 - cin.t0));
          const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
          float out[NO];
#pragma unroll
          for (int s = 0; s < NO; ++s) out[s] = (((ya[s] + x * cd[s]) + x2 * cc_[s]) + x3 * cb[s]) + x4 * ca[s];
          Own::store(a.h + (size_t)j * row + poff, q, out, lm.live);
        }
      }
      if (gid == 0) {
        a.tape_t[cin.n_acc] = cin.t0;
        a.tape_dt[cin.n_acc] = cin.dt;
        a.tape_j[2 * cin.n_acc] = cin.j_next;
        a.tape_j[2 * cin.n_acc + 1] = j;
      }
      c.j_next = j;
      c.n_acc = cin.n_acc + 1;
      c.t0 = t1;
    } else {
      c.n_rej = cin.n_rej + 1;
#pragma unroll
      for (int s = 0; s < NO; ++s) {
        y[s] = y_old[s];
        f0[s] = k_first[s];
      }
    }
    c.dt = cin.dt * dp_step_factor(ratio);
  }
  c.attempt = cin.attempt + 1;

  bool stop = false;
  if (c.status) { c.done = 1; stop = true; }
  if (!stop && c.j_next >= a.T) { c.done = 1; stop = true; }
  if (!stop && !(c.t0 + c.dt > c.t0)) { c.status |= HODE_STATUS_DT_UNDERFLOW; c.done = 1; stop = true; }
  if (!stop && c.n_acc >= a.max_steps) { c.status |= HODE_STATUS_MAX_STEPS; c.done = 1; stop = true; }
  if (stop) {
    if (gid == 0) *cout = c;
    return;
  }

  HODE_STAMP(3)   // decision taken, dense output (if any) written
  // ---- new attempt from (y, f0) at (t0, dt), owned components only
  const float t0f = (float)c.t0, dtf = (float)c.dt, t1f = (float)(c.t0 + c.dt);
  float k[7][NO], Yo[NO];
#pragma unroll
  for (int s = 0; s < NO; ++s) k[0][s] = f0[s];
#pragma unroll
  for (int i = 2; i <= 7; ++i) {
    const float ti = dp_stage_time(i, t0f, dtf, t1f);
#pragma unroll
    for (int s = 0; s < NO; ++s) {
      float acc = y[s];
#pragma unroll
      for (int m = 0; m < i - 1; ++m) acc = __builtin_fmaf(kDpBeta[i - 2][m] * dtf, k[m][s], acc);
      Yo[s] = acc;
    }
    dp_own_rhs<D, ABLATE, HILL2>(th, ml, ds.at(ti, th.kel).v, q, Yo, k[i - 1]);
  }
  // Yo is y1 (the last beta row is the solution weights, FSAL)
  float se = 0.f;
  bool bad = false;
#pragma unroll
  for (int s = 0; s < NO; ++s) {
    float err = 0.f;
#pragma unroll
    for (int m = 0; m < 7; ++m) err = __builtin_fmaf(dtf * kDpErr[m], k[m][s], err);
    const float tol = a.atol + a.rtol * fmaxf(__builtin_fabsf(y[s]), __builtin_fabsf(Yo[s]));
    const float u = div_f32(err, tol);
    se = __builtin_fmaf(u, u, se);
    bad |= !__builtin_isfinite(y[s]);
  }
  HODE_STAMP(4)   // stages and error estimate done
  se = wave_sum(lm.live ? se : 0.0f);  // every component of every live patient is owned by exactly one lane
  if ((threadIdx.x & 63) == 0 && (gid >> 6) < a.n_waves) pout[2 * (gid >> 6)] = se;  // the last block may carry idle waves
  Own::store(a.tape_y + dp_tape_row(a, c.n_acc + 1) * row + poff, q, Yo, lm.live);
  Own::store(a.kbuf + poff, q, k[0], lm.live);
  Own::store(a.kbuf + 6 * row + poff, q, k[6], lm.live);
  // k2..k6 are read back only for the dense output, i.e. when an output time lies inside this step and it is accepted
  if ((double)(c.j_next == cin.j_next ? t_next : a.t[c.j_next]) <= c.t0 + c.dt) {
#pragma unroll
    for (int m = 1; m < 6; ++m) Own::store(a.kbuf + (size_t)m * row + poff, q, k[m], lm.live);
  }
  if (bad && lm.live) atomicOr(&cout->status, HODE_STATUS_NONFINITE);
  if (gid == 0) {
    cout->t0 = c.t0; cout->dt = c.dt; cout->h0 = c.h0; cout->d1 = c.d1;
    cout->n_acc = c.n_acc; cout->n_rej = c.n_rej; cout->j_next = c.j_next; cout->done = c.done; cout->attempt = c.attempt;
    if (c.status) atomicOr(&cout->status, c.status);
  }
  HODE_STAMP(5)   // stores drained
#undef HODE_STAMP
}

// ------------------------------------------------------------------------- persistent attempt loop (owner layout)
// The whole attempt loop in ONE launch.  An attempt launched on its own spends ~3.8 of its 5.7 us outside the arithmetic
// (kernel boundary, three dependent cold round trips, store drain -- DESIGN.md 5c); here the state, the stage derivatives,
// the weights and the controller stay in registers across attempts and the kernel boundary is replaced by ONE hop through
// memory: every wave publishes (attempt number, error-norm partial) as a single 8-byte agent-scope store into its own
// slot -- payload and flag are the same word, so there is nothing to order -- and then polls all slots with agent-scope
// loads until they carry the current attempt number.  No read-modify-write atomics (round 1's arrival counter serialised
// 625 of them at the memory side), no cache-wide fences.  The partials are summed in fold_waves' order; every wave takes the
// same decision from the same sum, so the controller needs no exchange at all.
// MEASURED (tools/dp_persist_probe.py, 10 000 x 12): 7.0 us per attempt against 5.7 for one launch per attempt -- the
// all-to-all poll across 8 XCDs costs ~5 us, more than the kernel boundary it replaces.  Opt-in (HODE_DP_PERSIST=1).
// Liveness: all blocks must be resident (625 one-wave blocks on an otherwise idle chip are); the poll is bounded -- a wave
// that does not see its peers within kDpSpinLimit rounds raises kDpStatusBarrierTimeout and leaves, and so does every
// other wave (they wait for each other), so the grid always drains; the host then repeats the solve on the
// launch-per-attempt path.
constexpr int kDpStatusBarrierTimeout = 8;   // internal status bit, never handed to the caller
constexpr int kDpSpinLimit = 200000;
constexpr int kDpMaxPersistWaves = 1024;     // 16 slots per lane

HODE_DEV void dp_slot_publish(unsigned long long* slot, int seq, float v) {
  const unsigned long long w = ((unsigned long long)(unsigned)seq << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
  __hip_atomic_store(slot, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all n_waves partials of attempt `seq`, summed in fold_waves' order; false when the bounded poll gives up
HODE_DEV bool dp_slot_gather(const unsigned long long* slots, int n_waves, int seq, float& sum) {
  const int lane = threadIdx.x & 63;
  unsigned long long v[16];
  for (int spin = 0;; ++spin) {
    // all 16 loads of a round in flight together (indices past the array clamped: a conditional per load would serialise
    // sixteen cross-XCD round trips)
#pragma unroll
    for (int j = 0; j < 16; ++j)
      v[j] = __hip_atomic_load(slots + min(lane + 64 * j, n_waves - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool all_here = true;
#pragma unroll
    for (int j = 0; j < 16; ++j) all_here &= (int)(v[j] >> 32) == seq;
    if (__builtin_amdgcn_ballot_w64(!all_here) == 0) break;
    if (spin >= kDpSpinLimit) return false;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += (lane + 64 * j < n_waves) ? __builtin_bit_cast(float, (unsigned)(v[j] & 0xffffffffull)) : 0.f;
  sum = wave_sum(s);
  return true;
}

template <int D, bool ABLATE, bool HILL2, bool K1>
HODE_DEV void dp_persist_body_own(const DpArgs& a) {
  using Own = DpOwn<D>;
  constexpr int NO = Own::NO;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int wave = gid >> 6;
  const LaneMap<4> lm(a.B, a.ppw);
  const int q = lm.q;
  const size_t row = (size_t)a.B * D;
  const size_t poff = (size_t)lm.p * D;
  const float cnt = (float)a.B * (float)D;
  const RocheTheta th = load_theta(a.theta, ABLATE);
  MlSlice<D, 4> ml;
  ml.load(a.w1, a.b1, q);
  const DoseSched<K1> ds = dp_load_dose<K1>(a, lm.p);
  DpCtrl c = a.ctrl[0];
  if (c.done) return;
  float y[NO], f0[NO];
  Own::load(a.tape_y + dp_tape_row(a, c.n_acc) * row + poff, q, y);
  Own::load(a.kbuf + poff, q, f0);
  if (c.attempt == 0) {
    // finish the initial step selection (order 4): d2 from the partials init2 left, h1, dt_0
    const float d2 = div_f32(__builtin_sqrtf(fold_waves(a.partials + (size_t)2 * a.n_waves, a.n_waves, 2, 0) / cnt), c.h0);
    float h1;
    if (c.d1 <= 1e-15f && d2 <= 1e-15f) h1 = fmaxf(1e-6f, c.h0 * 1e-3f);
    else h1 = powf(div_f32(0.01f, fmaxf(c.d1, d2)), 0.2f);
    c.dt = (double)fminf(100.0f * c.h0, h1);
    if (gid == 0) { a.init->d2 = d2; a.init->h1 = h1; }
    c.attempt = 1;
  }
  // termination checks before the first attempt of this launch (same order as the launch-per-attempt kernel)
  auto stop_now = [&]() {
    if (c.status) { c.done = 1; return true; }
    if (c.j_next >= a.T) { c.done = 1; return true; }
    if (!(c.t0 + c.dt > c.t0)) { c.status |= HODE_STATUS_DT_UNDERFLOW; c.done = 1; return true; }
    if (c.n_acc >= a.max_steps) { c.status |= HODE_STATUS_MAX_STEPS; c.done = 1; return true; }
    return false;
  };
  bool stopped = stop_now();
  for (int it = 0; !stopped && it < a.max_iters; ++it) {
    // ---- attempt number c.attempt from (y, f0) at (t0, dt), owned components only
    const float t0f = (float)c.t0, dtf = (float)c.dt, t1f = (float)(c.t0 + c.dt);
    float k[7][NO], Yo[NO];
#pragma unroll
    for (int s = 0; s < NO; ++s) k[0][s] = f0[s];
#pragma unroll
    for (int i = 2; i <= 7; ++i) {
      const float ti = dp_stage_time(i, t0f, dtf, t1f);
#pragma unroll
      for (int s = 0; s < NO; ++s) {
        float acc = y[s];
#pragma unroll
        for (int m = 0; m < i - 1; ++m) acc = __builtin_fmaf(kDpBeta[i - 2][m] * dtf, k[m][s], acc);
        Yo[s] = acc;
      }
      dp_own_rhs<D, ABLATE, HILL2>(th, ml, ds.at(ti, th.kel).v, q, Yo, k[i - 1]);
    }
    float se = 0.f;
    bool bad = false;
#pragma unroll
    for (int s = 0; s < NO; ++s) {
      float err = 0.f;
#pragma unroll
      for (int m = 0; m < 7; ++m) err = __builtin_fmaf(dtf * kDpErr[m], k[m][s], err);
      const float tol = a.atol + a.rtol * fmaxf(__builtin_fabsf(y[s]), __builtin_fabsf(Yo[s]));
      const float u = div_f32(err, tol);
      se = __builtin_fmaf(u, u, se);
      bad |= !__builtin_isfinite(y[s]);
    }
    se = wave_sum(lm.live ? se : 0.0f);
    if (__builtin_amdgcn_ballot_w64(bad && lm.live) != 0) se = __builtin_nanf("");  // a non-finite state reaches every wave
    // ---- the one hop: publish, gather
    const int seq = c.attempt;
    unsigned long long* sl = a.slots + (size_t)(seq & 1) * a.n_waves;
    if ((threadIdx.x & 63) == 0) dp_slot_publish(sl + wave, seq, se);
    float sum;
    if (!dp_slot_gather(sl, a.n_waves, seq, sum)) {
      c.status |= kDpStatusBarrierTimeout;
      c.done = 1;
      break;
    }
    const float ratio = __builtin_sqrtf(sum / cnt);
    if (!(ratio == ratio)) {  // torchdiffeq asserts on the state before a step; here every wave sees it at once
      c.status |= HODE_STATUS_NONFINITE;
      c.done = 1;
      break;
    }
    // ---- decision (every wave computes the same record)
    const double t1 = c.t0 + c.dt;
    if (seq == 1 && gid == 0) a.init->first_accepted = ratio <= 1.0f ? 1 : 0;
    if (ratio <= 1.0f) {
      int j = c.j_next;
      if (j < a.T && (double)a.t[j] <= t1) {
        // quartic dense output of the accepted step on the owned components: everything it needs is in registers
        float ym[NO], ca[NO], cb[NO], cc_[NO], cd[NO];
#pragma unroll
        for (int s = 0; s < NO; ++s) {
          ym[s] = y[s];
#pragma unroll
          for (int m = 0; m < 7; ++m) ym[s] = __builtin_fmaf(dtf * kDpMid[m], k[m][s], ym[s]);
          const float f0i = k[0][s], f1i = k[6][s], y0i = y[s], y1i = Yo[s], ymi = ym[s];
          ca[s] = 2.0f * dtf * (f1i - f0i) - 8.0f * (y1i + y0i) + 16.0f * ymi;
          cb[s] = dtf * (5.0f * f0i - 3.0f * f1i) + 18.0f * y0i + 14.0f * y1i - 32.0f * ymi;
          cc_[s] = dtf * (f1i - 4.0f * f0i) - 11.0f * y0i - 5.0f * y1i + 16.0f * ymi;
          cd[s] = dtf * f0i;
        }
        for (; j < a.T && (double)a.t[j] <= t1; ++j) {
          const float x = (float)(((double)a.t[j] - c.t0) / (t1 - c.t0));
          const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
          float out[NO];
#pragma unroll
          for (int s = 0; s < NO; ++s) out[s] = (((y[s] + x * cd[s]) + x2 * cc_[s]) + x3 * cb[s]) + x4 * ca[s];
          Own::store(a.h + (size_t)j * row + poff, q, out, lm.live);
        }
      }
      if (gid == 0) {
        a.tape_t[c.n_acc] = c.t0;
        a.tape_dt[c.n_acc] = c.dt;
        a.tape_j[2 * c.n_acc] = c.j_next;
        a.tape_j[2 * c.n_acc + 1] = j;
      }
      Own::store(a.tape_y + dp_tape_row(a, c.n_acc + 1) * row + poff, q, Yo, lm.live);
      c.j_next = j;
      c.n_acc += 1;
      c.t0 = t1;
#pragma unroll
      for (int s = 0; s < NO; ++s) {
        y[s] = Yo[s];
        f0[s] = k[6][s];  // FSAL
      }
    } else {
      c.n_rej += 1;
    }
    c.dt = c.dt * dp_step_factor(ratio);
    c.attempt = seq + 1;
    stopped = stop_now();
  }
  // what a continuation launch (or the launch-per-attempt path) resumes from: f0 of the current state; the state itself is
  // tape_y[n_acc] already
  Own::store(a.kbuf + poff, q, f0, lm.live);
  if (gid == 0) a.ctrl[0] = c;
}

template <int D, bool ABLATE>
__global__ __launch_bounds__(64) void dp_persist_kernel(DpArgs a) {
  const bool hill2 = ABLATE || a.hill2 != 0;
  if (hill2 && a.K == 1) dp_persist_body_own<D, ABLATE, true, true>(a);
  else if (hill2) dp_persist_body_own<D, ABLATE, true, false>(a);
  else dp_persist_body_own<D, ABLATE, false, false>(a);
}

// PHASE 2 (the attempt) runs as workgroups of up to 4 waves: a launch's fixed cost grows with the number of workgroups the
// dispatcher has to place, and an attempt is ~6 us long (wave indices, partials and lane maps are per wave either way)
template <int D, int LPP, bool ABLATE, int PHASE>
__global__ __launch_bounds__(PHASE == 2 ? 256 : 64) void dp_fwd_kernel(DpArgs a) {
  const bool hill2 = ABLATE || (a.hill2 >= 0 ? a.hill2 != 0 : (a.theta[0] == 2.0f && a.theta[1] == 2.0f));
#define HODE_DP_DISPATCH(BODY)                                             \
  if (hill2 && a.K == 1) BODY<D, LPP, ABLATE, true, true>(a);              \
  else if (hill2) BODY<D, LPP, ABLATE, true, false>(a);                    \
  else BODY<D, LPP, ABLATE, false, false>(a);
  if constexpr (PHASE == 0) { HODE_DP_DISPATCH(dp_init1_body) }
  else if constexpr (PHASE == 1) { HODE_DP_DISPATCH(dp_init2_body) }
  else if constexpr (LPP == 4) {
    if (hill2 && a.K == 1) dp_attempt_body_own<D, ABLATE, true, true>(a);
    else if (hill2) dp_attempt_body_own<D, ABLATE, true, false>(a);
    else dp_attempt_body_own<D, ABLATE, false, false>(a);
  } else { HODE_DP_DISPATCH(dp_attempt_body) }
}

// ------------------------------------------------------------------------------------------------ backward
// Reverse sweep over the accepted-step tape.  Per step: recompute the 7 stage derivatives from y_n, then
//   g7 = lam_f (FSAL use in the next step) + dense-output terms;  a7 = J7^T g7;  lam_y1 += a7
//   lam_y0 = lam_y1 (+ dense-output terms);  g_m += dt (b_m lam_y1 + cmid_m lam_ymid)
//   for i = 6..2: a_i = J_i^T g_i; lam_y0 += a_i; g_m += dt beta_{i,m} a_i (m < i)
//   g1 is handed to step n-1 as lam_f (k1 of step n IS k7 of step n-1); at n = 0 it goes through J1 instead.
// Step sizes dt_1, dt_2, ... are constants (torchdiffeq's controller runs under no_grad).  dt_0 is not: Hairer's initial
// step is computed from y0, f0, f1 with autograd on, and when attempt 0 is accepted every later step boundary is
// t_n = t[0] + dt_0 + const.  The sweep therefore also accumulates sigma = d loss / d dt_0 (this lane's share):
//   all steps:  stage times    g_i[3] * (-kel^2 Dose(tau_i)) * d tau_i / d dt_0     (the rhs depends on t through the dose
//                              only; d tau_i / d dt_0 = 1 for n >= 1, alpha_i for the stages of step 0)
//               dense output   G_j . p'(x_j) * d x_j / d dt_0,  x = (t_j - t_n) / dt_n:  -1/dt_n (n >= 1), -x/dt_0 (n = 0)
//   step 0:     dt_0 itself    a_i . (Y_i - y0)/dt + lam_y1 . (y1 - y0)/dt + lam_mid . (y_mid - y0)/dt
//                              + (sum_j Q0_j G_j . f0 + sum_j Q1_j G_j . f1)/dt
// (formulas checked against autograd in fp64; the replay oracle of tests/test_hip_dopri5.py pins the sum).  The kernels
// dp_initbwd_* below push sigma through dt_0 = min(100 h0, h1) into grad_y0 and the parameter gradients.
template <int D, int LPP, bool NEED_TH>
HODE_DEV void dp_store_grad_partials(const GradAcc<D, LPP>& acc, float* __restrict__ out) {
  constexpr int M = D - 4;
  constexpr int MR = MlSlice<D, LPP>::MR;
  const int lane = threadIdx.x & 63;
  if constexpr (M > 0) {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const float v = wave_sum_patients<LPP>(acc.dw[r][i]);
        if (lane < LPP) out[(lane * MR + r) * D + i] = v;
      }
      const float vb = wave_sum_patients<LPP>(acc.db[r]);
      if (lane < LPP) out[M * D + lane * MR + r] = vb;
    }
  }
#pragma unroll
  for (int i = 0; i < kNTheta; ++i) {
    const float v = wave_sum_patients<LPP>(NEED_TH ? acc.dth[i] : 0.f);
    if (lane == 0) out[M * D + M + i] = v;
  }
}

template <int D, int LPP, bool ABLATE, bool HILL2, bool NEED_TH, bool K1>
HODE_DEV void dp_bwd_body(const DpArgs& a) {
  using Ml = MlSlice<D, LPP>;
  constexpr int MR = Ml::MR;
  constexpr int M = D - 4;
  const LaneMap<LPP> lm(a.B, a.ppw);
  const RocheTheta th = load_theta(a.theta, ABLATE);
  Ml ml;
  ml.load(a.w1, a.b1, lm.q);
  MlColSlice<D, LPP> mc;
  mc.load(a.w1, lm.q);
  const float ln_ec50 = log_f32(th.ec50);
  const DoseSched<K1> ds = dp_load_dose<K1>(a, lm.p);
  GradAcc<D, LPP> acc;
  acc.zero();
  const size_t row = (size_t)a.B * D;
  const size_t poff = (size_t)lm.p * D;
  const float live = lm.live ? 1.0f : 0.0f;

  float lam_y[D], lam_f[D];
#pragma unroll
  for (int i = 0; i < D; ++i) lam_y[i] = lam_f[i] = 0.f;
  float sig_t = 0.f, sig_d = 0.f;  // sigma = -kel^2 sig_t + sig_d

  for (int n = a.n_acc - 1; n >= 0; --n) {
    const double t0 = a.tape_t[n], dt = a.tape_dt[n];
    const double t1 = t0 + dt;
    const float t0f = (float)t0, dtf = (float)dt, t1f = (float)t1;
    const bool first = n == 0;
    const float rdt = div_f32(1.0f, dtf);
    float sd = 0.f;  // this step's terms that carry 1/dt
    float y0[D];
    load_vec<D>(a.tape_y + (size_t)n * row + poff, y0);
    float k[7][D], Y[7][D], s[7][MR];
    DoseVal dv[7];
    const float tk1 = (n == 0) ? a.t[0] : nextafter_down(t0f);
    dv[0] = ds.at(tk1, th.kel);
    roche_rhs<D, LPP, ABLATE, HILL2>(th, ml, dv[0].v, y0, k[0], s[0]);
#pragma unroll
    for (int i = 0; i < D; ++i) Y[0][i] = y0[i];
    dp_stages<D, LPP, ABLATE, HILL2, K1>(th, ml, ds, y0, t0f, dtf, t1f, k, Y, s, dv);

    float g[7][D], lam_y0[D], lam_mid[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      g[6][i] = lam_f[i];
      lam_y0[i] = 0.f;
      lam_mid[i] = 0.f;
#pragma unroll
      for (int m = 0; m < 6; ++m) g[m][i] = 0.f;
    }
    // cotangents of the outputs interpolated inside this step
    const int jlo = a.tape_j[2 * n], jhi = a.tape_j[2 * n + 1];
    if (jlo < jhi) {
      // p'(x) / dt of this step's dense-output polynomial.  With y1 = y0 + dt S1, y_mid = y0 + dt Sm the y0 terms of the
      // quartic's coefficients cancel exactly (-8 - 8 + 16 = 18 + 14 - 32 = -11 - 5 + 16 = 0), so the coefficients are dt
      // times combinations of the stage derivatives -- formed from those directly: differencing the O(1) states of a
      // step of size 1e-4 and dividing by dt again would leave percent-level noise in sigma.
      float cd[D], cc_[D], cb[D], ca[D], ymd[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        float s1 = 0.f, sm = 0.f;
#pragma unroll
        for (int m = 0; m < 6; ++m) s1 = __builtin_fmaf(kDpBeta[5][m], k[m][i], s1);
#pragma unroll
        for (int m = 0; m < 7; ++m) sm = __builtin_fmaf(kDpMid[m], k[m][i], sm);
        const float f0i = k[0][i], f1i = k[6][i];
        ca[i] = 4.0f * (2.0f * (f1i - f0i) - 8.0f * s1 + 16.0f * sm);
        cb[i] = 3.0f * ((5.0f * f0i - 3.0f * f1i) + 14.0f * s1 - 32.0f * sm);
        cc_[i] = 2.0f * ((f1i - 4.0f * f0i) - 5.0f * s1 + 16.0f * sm);
        cd[i] = f0i;
        ymd[i] = sm;  // (y_mid - y0) / dt
      }
      for (int j = jlo; j < jhi; ++j) {
        const float x = (float)(((double)a.t[j] - t0) / (t1 - t0));
        const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
        const float P0 = 1.0f - 11.0f * x2 + 18.0f * x3 - 8.0f * x4;
        const float P1 = -5.0f * x2 + 14.0f * x3 - 8.0f * x4;
        const float Pm = 16.0f * x2 - 32.0f * x3 + 16.0f * x4;
        const float Q0 = dtf * (x - 4.0f * x2 + 5.0f * x3 - 2.0f * x4);
        const float Q1 = dtf * (x2 - 3.0f * x3 + 2.0f * x4);
        const float xs = first ? -x : -1.0f;  // d x / d dt_0 times dt
        float G[D];
        load_vec<D>(a.grad_h + (size_t)j * row + poff, G);
        float sx = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const float gi = G[i] * live;
          lam_y0[i] = __builtin_fmaf(P0, gi, lam_y0[i]);
          lam_y[i] = __builtin_fmaf(P1, gi, lam_y[i]);
          lam_mid[i] = __builtin_fmaf(Pm, gi, lam_mid[i]);
          g[0][i] = __builtin_fmaf(Q0, gi, g[0][i]);
          g[6][i] = __builtin_fmaf(Q1, gi, g[6][i]);
          const float dp = cd[i] + x * cc_[i] + x2 * cb[i] + x3 * ca[i];  // p'(x) / dt
          sx = __builtin_fmaf(gi, dp, sx);
        }
        sig_d = __builtin_fmaf(xs, sx, sig_d);
      }
      if (first) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
          sd = __builtin_fmaf(g[0][i], k[0][i], sd);               // g[0] holds sum_j Q0_j G_j so far
          sd = __builtin_fmaf(g[6][i] - lam_f[i], k[6][i], sd);    // sum_j Q1_j G_j
          sig_d = __builtin_fmaf(lam_mid[i], ymd[i], sig_d);
        }
      }
    }
    // y_mid = y0 + dt sum cmid_m k_m
#pragma unroll
    for (int i = 0; i < D; ++i) {
      lam_y0[i] += lam_mid[i];
#pragma unroll
      for (int m = 0; m < 7; ++m) g[m][i] = __builtin_fmaf(dtf * kDpMid[m], lam_mid[i], g[m][i]);
    }
    // stage 7: k7 = f(t1-, y1)
    float a_[D];
    roche_vjp<D, LPP, ABLATE, HILL2, NEED_TH>(th, ml, mc, ln_ec50, dv[6], Y[6], s[6], g[6], lm.q, a_, acc);
    if constexpr (!ABLATE) sig_t = __builtin_fmaf(g[6][3], dv[6].v, sig_t);  // alpha_7 = 1 in every step
#pragma unroll
    for (int i = 0; i < D; ++i) lam_y[i] += a_[i];
    // y1 = y0 + dt sum_{m<=6} beta_6m k_m
#pragma unroll
    for (int i = 0; i < D; ++i) {
      lam_y0[i] += lam_y[i];
      if (first) sd = __builtin_fmaf(lam_y[i], Y[6][i] - y0[i], sd);
#pragma unroll
      for (int m = 0; m < 6; ++m) g[m][i] = __builtin_fmaf(kDpBeta[5][m] * dtf, lam_y[i], g[m][i]);
    }
    // stages 6 .. 2
#pragma unroll
    for (int st = 6; st >= 2; --st) {
      roche_vjp<D, LPP, ABLATE, HILL2, NEED_TH>(th, ml, mc, ln_ec50, dv[st - 1], Y[st - 1], s[st - 1], g[st - 1], lm.q,
                                                a_, acc);
      if constexpr (!ABLATE) sig_t = __builtin_fmaf(g[st - 1][3] * (first ? kDpAlpha[st - 2] : 1.0f), dv[st - 1].v, sig_t);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        lam_y0[i] += a_[i];
        if (first) sd = __builtin_fmaf(a_[i], Y[st - 1][i] - y0[i], sd);
#pragma unroll
        for (int m = 0; m < st - 1; ++m) g[m][i] = __builtin_fmaf(kDpBeta[st - 2][m] * dtf, a_[i], g[m][i]);
      }
    }
    if (n == 0) {
      roche_vjp<D, LPP, ABLATE, HILL2, NEED_TH>(th, ml, mc, ln_ec50, dv[0], Y[0], s[0], g[0], lm.q, a_, acc);
#pragma unroll
      for (int i = 0; i < D; ++i) lam_y0[i] += a_[i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      lam_y[i] = lam_y0[i];
      lam_f[i] = g[0][i];
    }
    sig_d = __builtin_fmaf(sd, rdt, sig_d);
  }
  {
    // every lane of a patient's quad holds the full vectors: count the patient once
    const float mine = (LPP == 1 || lm.q == 0) ? 1.0f : 0.0f;
    const float sig = wave_sum(mine * __builtin_fmaf(-th.kel * th.kel, sig_t, sig_d));
    if ((threadIdx.x & 63) == 0) a.partials[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = sig;
  }
  // output 0 is y0 itself
  {
    float G[D];
    load_vec<D>(a.grad_h + poff, G);
#pragma unroll
    for (int i = 0; i < D; ++i) lam_y[i] = __builtin_fmaf(G[i], live, lam_y[i]);
  }
  store_vec<D, LPP>(a.grad_y0 + poff, lam_y, lm.q, lm.live);

  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int P = M * D + M + kNTheta;
  dp_store_grad_partials<D, LPP, NEED_TH>(acc, a.grad_partials + (size_t)wave * P);
}

// ------------------------------------------------------------------------------------------ backward, owner layout
// dp_bwd_body with the ownership of dp_attempt_body_own: stage states, stage derivatives, the cotangents g_1..g_7,
// lam_y / lam_f and the dense-output terms exist only for the 1 + MR components a lane owns; the rhs and its VJP read the
// full stage state / the full u = g (1 - s^2) of the learned rows through DPP quad broadcasts.  Same tape, same partial
// layout, same arithmetic per component (only the order of the error-free reads differs).
template <int D, bool ABLATE, bool HILL2, bool NEED_TH, int... C>
HODE_DEV void dp_own_vjp_impl(const RocheTheta& th, const float (&wcol)[D - 4][DpOwn<D>::NO], float ln_ec50, DoseVal dose,
                              int q, const float (&Yo)[DpOwn<D>::NO], const float (&ko)[DpOwn<D>::NO],
                              const float (&go)[DpOwn<D>::NO], float (&ao)[DpOwn<D>::NO], GradAcc<D, 4>& acc,
                              std::integer_sequence<int, C...>) {
  using Own = DpOwn<D>;
  constexpr int MR = Own::MR, NO = Own::NO, M = D - 4;
  const float Y[D] = {Own::template full<C>(Yo)...};
  // ---- learned rows of this lane: u_r = g_r (1 - s_r^2); dW, db; a = W^T u on the owned components
  float u[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r) {
    u[r] = go[1 + r] * __builtin_fmaf(-ko[1 + r], ko[1 + r], 1.0f);
    acc.db[r] += u[r];
#pragma unroll
    for (int i = 0; i < D; ++i) acc.dw[r][i] = __builtin_fmaf(u[r], Y[i], acc.dw[r][i]);
  }
  float uf[M];
#pragma unroll
  for (int r = 0; r < MR; ++r) {
    uf[0 * MR + r] = quad_bcast<0>(u[r]);
    uf[1 * MR + r] = quad_bcast<1>(u[r]);
    uf[2 * MR + r] = quad_bcast<2>(u[r]);
    uf[3 * MR + r] = quad_bcast<3>(u[r]);
  }
#pragma unroll
  for (int s = 0; s < NO; ++s) {
    float p = 0.f;
#pragma unroll
    for (int j = 0; j < M; ++j) p = __builtin_fmaf(wcol[j][s], uf[j], p);
    ao[s] = p;
  }
  // ---- expert block (every lane evaluates the four entries, keeps the one it owns; dth is formed redundantly by the quad
  // exactly as in roche_vjp, the epilogue's stride-4 sum counts every patient once)
  const float dis = Y[0], ir = Y[1], imm = Y[2], d2 = Y[3];
  const float g0 = quad_bcast<0>(go[0]), g1 = quad_bcast<1>(go[0]), g2 = quad_bcast<2>(go[0]), g3 = quad_bcast<3>(go[0]);
  float e0, e1, e2, e3;
  if constexpr (!ABLATE) {
    const float immp = pow_hill<HILL2>(imm, th.hc);
    const float irp = pow_hill<HILL2>(ir, th.hp);
    const float ecp = pow_hill<HILL2>(th.ec50, th.hp);
    const float rden = __builtin_amdgcn_rcpf(ecp + irp);
    const float dirp = dpow_dx<HILL2>(ir, th.hp);
    const float g0d = g0 * dis, g1d = g1 * dis, g1i = g1 * ir;
    const float er2 = th.emax * rden * rden;
    e0 = g0 * (th.kprog - immp * th.kci - ir * th.kcir) + g1 * (th.kid + ir * th.kfb);
    e1 = g1 * (dis * th.kfb - th.koff + er2 * ecp * dirp - d2 * th.kdexa) - g0d * th.kcir + g2 * th.kim;
    e2 = -(g0d * th.kci * dpow_dx<HILL2>(imm, th.hc));
    e3 = -(g1i * th.kdexa + g3 * th.kel);
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
    e0 = -(th.th1 * g1);
    e1 = g0;
    e2 = -(th.th2 * g3);
    e3 = g2;
    if constexpr (NEED_TH) {
      acc.dth[13] -= dis * g1;
      acc.dth[14] -= imm * g3;
    }
  }
  ao[0] += q == 0 ? e0 : (q == 1 ? e1 : (q == 2 ? e2 : e3));
}
template <int D, bool ABLATE, bool HILL2, bool NEED_TH>
HODE_DEV void dp_own_vjp(const RocheTheta& th, const float (&wcol)[D - 4][DpOwn<D>::NO], float ln_ec50, DoseVal dose, int q,
                         const float (&Yo)[DpOwn<D>::NO], const float (&ko)[DpOwn<D>::NO], const float (&go)[DpOwn<D>::NO],
                         float (&ao)[DpOwn<D>::NO], GradAcc<D, 4>& acc) {
  dp_own_vjp_impl<D, ABLATE, HILL2, NEED_TH>(th, wcol, ln_ec50, dose, q, Yo, ko, go, ao, acc,
                                             std::make_integer_sequence<int, D>{});
}

template <int D, bool ABLATE, bool HILL2, bool NEED_TH, bool K1>
HODE_DEV void dp_bwd_body_own(const DpArgs& a) {
  using Own = DpOwn<D>;
  constexpr int NO = Own::NO, MR = Own::MR, M = D - 4;
  const LaneMap<4> lm(a.B, a.ppw);
  const int q = lm.q;
  const RocheTheta th = load_theta(a.theta, ABLATE);
  MlSlice<D, 4> ml;
  ml.load(a.w1, a.b1, q);
  float wcol[M][NO];  // W[j][owned component]: the columns of W^T u this lane forms
#pragma unroll
  for (int j = 0; j < M; ++j) {
    wcol[j][0] = a.w1[j * D + q];
#pragma unroll
    for (int r = 0; r < MR; ++r) wcol[j][1 + r] = a.w1[j * D + 4 + q * MR + r];
  }
  const float ln_ec50 = log_f32(th.ec50);
  const DoseSched<K1> ds = dp_load_dose<K1>(a, lm.p);
  GradAcc<D, 4> acc;
  acc.zero();
  const size_t row = (size_t)a.B * D;
  const size_t poff = (size_t)lm.p * D;
  const float live = lm.live ? 1.0f : 0.0f;

  float lam_y[NO], lam_f[NO];
#pragma unroll
  for (int s = 0; s < NO; ++s) lam_y[s] = lam_f[s] = 0.f;
  float sig_t = 0.f, sig_d = 0.f;     // d loss / d dt_0 = -kel^2 sig_t + sig_d, see dp_bwd_body
  const float own3 = q == 3 ? 1.0f : 0.0f;  // the lane that owns Dose2, the one component whose rhs depends on t

  for (int n = a.n_acc - 1; n >= 0; --n) {
    const double t0 = a.tape_t[n], dt = a.tape_dt[n];
    const double t1 = t0 + dt;
    const float t0f = (float)t0, dtf = (float)dt, t1f = (float)t1;
    const bool first = n == 0;
    const float rdt = div_f32(1.0f, dtf);
    float sd = 0.f;
    float k[7][NO], Ys[7][NO];
    DoseVal dv[7];
    Own::load(a.tape_y + (size_t)n * row + poff, q, Ys[0]);
    const float tk1 = (n == 0) ? a.t[0] : nextafter_down(t0f);
    dv[0] = ds.at(tk1, th.kel);
    dp_own_rhs<D, ABLATE, HILL2>(th, ml, dv[0].v, q, Ys[0], k[0]);
#pragma unroll
    for (int i = 2; i <= 7; ++i) {
      const float ti = dp_stage_time(i, t0f, dtf, t1f);
#pragma unroll
      for (int s = 0; s < NO; ++s) {
        float acc_ = Ys[0][s];
#pragma unroll
        for (int m = 0; m < i - 1; ++m) acc_ = __builtin_fmaf(kDpBeta[i - 2][m] * dtf, k[m][s], acc_);
        Ys[i - 1][s] = acc_;
      }
      dv[i - 1] = ds.at(ti, th.kel);
      dp_own_rhs<D, ABLATE, HILL2>(th, ml, dv[i - 1].v, q, Ys[i - 1], k[i - 1]);
    }

    float g[7][NO], lam_y0[NO], lam_mid[NO];
#pragma unroll
    for (int s = 0; s < NO; ++s) {
      g[6][s] = lam_f[s];
      lam_y0[s] = 0.f;
      lam_mid[s] = 0.f;
#pragma unroll
      for (int m = 0; m < 6; ++m) g[m][s] = 0.f;
    }
    const int jlo = a.tape_j[2 * n], jhi = a.tape_j[2 * n + 1];
    if (jlo < jhi) {
      float cd[NO], cc_[NO], cb[NO], ca[NO], ymd[NO];  // p'(x) / dt from the stage derivatives, see dp_bwd_body
#pragma unroll
      for (int s = 0; s < NO; ++s) {
        float s1 = 0.f, sm = 0.f;
#pragma unroll
        for (int m = 0; m < 6; ++m) s1 = __builtin_fmaf(kDpBeta[5][m], k[m][s], s1);
#pragma unroll
        for (int m = 0; m < 7; ++m) sm = __builtin_fmaf(kDpMid[m], k[m][s], sm);
        const float f0i = k[0][s], f1i = k[6][s];
        ca[s] = 4.0f * (2.0f * (f1i - f0i) - 8.0f * s1 + 16.0f * sm);
        cb[s] = 3.0f * ((5.0f * f0i - 3.0f * f1i) + 14.0f * s1 - 32.0f * sm);
        cc_[s] = 2.0f * ((f1i - 4.0f * f0i) - 5.0f * s1 + 16.0f * sm);
        cd[s] = f0i;
        ymd[s] = sm;
      }
      for (int j = jlo; j < jhi; ++j) {
        const float x = (float)(((double)a.t[j] - t0) / (t1 - t0));
        const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
        const float P0 = 1.0f - 11.0f * x2 + 18.0f * x3 - 8.0f * x4;
        const float P1 = -5.0f * x2 + 14.0f * x3 - 8.0f * x4;
        const float Pm = 16.0f * x2 - 32.0f * x3 + 16.0f * x4;
        const float Q0 = dtf * (x - 4.0f * x2 + 5.0f * x3 - 2.0f * x4);
        const float Q1 = dtf * (x2 - 3.0f * x3 + 2.0f * x4);
        const float xs = first ? -x : -1.0f;
        float G[NO];
        Own::load(a.grad_h + (size_t)j * row + poff, q, G);
        float sx = 0.f;
#pragma unroll
        for (int s = 0; s < NO; ++s) {
          const float gi = G[s] * live;
          lam_y0[s] = __builtin_fmaf(P0, gi, lam_y0[s]);
          lam_y[s] = __builtin_fmaf(P1, gi, lam_y[s]);
          lam_mid[s] = __builtin_fmaf(Pm, gi, lam_mid[s]);
          g[0][s] = __builtin_fmaf(Q0, gi, g[0][s]);
          g[6][s] = __builtin_fmaf(Q1, gi, g[6][s]);
          const float dp = cd[s] + x * cc_[s] + x2 * cb[s] + x3 * ca[s];
          sx = __builtin_fmaf(gi, dp, sx);
        }
        sig_d = __builtin_fmaf(xs, sx, sig_d);
      }
      if (first) {
#pragma unroll
        for (int s = 0; s < NO; ++s) {
          sd = __builtin_fmaf(g[0][s], k[0][s], sd);
          sd = __builtin_fmaf(g[6][s] - lam_f[s], k[6][s], sd);
          sig_d = __builtin_fmaf(lam_mid[s], ymd[s], sig_d);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NO; ++s) {
      lam_y0[s] += lam_mid[s];
#pragma unroll
      for (int m = 0; m < 7; ++m) g[m][s] = __builtin_fmaf(dtf * kDpMid[m], lam_mid[s], g[m][s]);
    }
    float a_[NO];
    dp_own_vjp<D, ABLATE, HILL2, NEED_TH>(th, wcol, ln_ec50, dv[6], q, Ys[6], k[6], g[6], a_, acc);
    if constexpr (!ABLATE) sig_t = __builtin_fmaf(g[6][0] * own3, dv[6].v, sig_t);
#pragma unroll
    for (int s = 0; s < NO; ++s) lam_y[s] += a_[s];
#pragma unroll
    for (int s = 0; s < NO; ++s) {
      lam_y0[s] += lam_y[s];
      if (first) sd = __builtin_fmaf(lam_y[s], Ys[6][s] - Ys[0][s], sd);
#pragma unroll
      for (int m = 0; m < 6; ++m) g[m][s] = __builtin_fmaf(kDpBeta[5][m] * dtf, lam_y[s], g[m][s]);
    }
#pragma unroll
    for (int st = 6; st >= 2; --st) {
      dp_own_vjp<D, ABLATE, HILL2, NEED_TH>(th, wcol, ln_ec50, dv[st - 1], q, Ys[st - 1], k[st - 1], g[st - 1], a_, acc);
      if constexpr (!ABLATE)
        sig_t = __builtin_fmaf(g[st - 1][0] * (first ? own3 * kDpAlpha[st - 2] : own3), dv[st - 1].v, sig_t);
#pragma unroll
      for (int s = 0; s < NO; ++s) {
        lam_y0[s] += a_[s];
        if (first) sd = __builtin_fmaf(a_[s], Ys[st - 1][s] - Ys[0][s], sd);
#pragma unroll
        for (int m = 0; m < st - 1; ++m) g[m][s] = __builtin_fmaf(kDpBeta[st - 2][m] * dtf, a_[s], g[m][s]);
      }
    }
    if (n == 0) {
      dp_own_vjp<D, ABLATE, HILL2, NEED_TH>(th, wcol, ln_ec50, dv[0], q, Ys[0], k[0], g[0], a_, acc);
#pragma unroll
      for (int s = 0; s < NO; ++s) lam_y0[s] += a_[s];
    }
#pragma unroll
    for (int s = 0; s < NO; ++s) {
      lam_y[s] = lam_y0[s];
      lam_f[s] = g[0][s];
    }
    sig_d = __builtin_fmaf(sd, rdt, sig_d);
  }
  {
    // every component of every live patient is owned by exactly one lane (dead lanes carry zero cotangents)
    const float sig = wave_sum(__builtin_fmaf(-th.kel * th.kel, sig_t, sig_d));
    if ((threadIdx.x & 63) == 0) a.partials[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = sig;
  }
  {
    float G[NO];
    Own::load(a.grad_h + poff, q, G);
#pragma unroll
    for (int s = 0; s < NO; ++s) lam_y[s] = __builtin_fmaf(G[s], live, lam_y[s]);
  }
  Own::store(a.grad_y0 + poff, q, lam_y, lm.live);

  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int P = M * D + M + kNTheta;
  dp_store_grad_partials<D, 4, NEED_TH>(acc, a.grad_partials + (size_t)wave * P);
}

template <int D, int LPP, bool ABLATE, bool NEED_TH>
__global__ __launch_bounds__(64) void dp_bwd_kernel(DpArgs a) {
  const bool hill2 = ABLATE || (a.theta[0] == 2.0f && a.theta[1] == 2.0f);
  if constexpr (LPP == 4) {
    if (hill2 && a.K == 1) dp_bwd_body_own<D, ABLATE, true, NEED_TH, true>(a);
    else if (hill2) dp_bwd_body_own<D, ABLATE, true, NEED_TH, false>(a);
    else dp_bwd_body_own<D, ABLATE, false, NEED_TH, false>(a);
  } else {
    if (hill2 && a.K == 1) dp_bwd_body<D, LPP, ABLATE, true, NEED_TH, true>(a);
    else if (hill2) dp_bwd_body<D, LPP, ABLATE, true, NEED_TH, false>(a);
    else dp_bwd_body<D, LPP, ABLATE, false, NEED_TH, false>(a);
  }
}

// ------------------------------------------------------------------------------- backward of the initial step size
// dt_0 = min(100 h0, h1) with (torchdiffeq _select_initial_step, order 4; oracle/solvers.py::_initial_step)
//   scale = atol + |y0| rtol;  d0 = rms(y0/scale);  d1 = rms(f0/scale);  h0 = 0.01 d0/d1  (1e-6 if d0 or d1 < 1e-5)
//   f1 = f(t0 + h0, y0 + h0 f0);  d2 = rms((f1 - f0)/scale)/h0;  h1 = (0.01/max(d1, d2))^(1/5)  (degenerate: max(1e-6, 1e-3 h0))
// d0, d1, d2 are batch-global, so sigma = d loss / d dt_0 (folded from the sweep's per-wave partials) reaches EVERY
// patient's y0 and the parameters through f0 and f1.  The cotangent of h0 itself needs a second global sum (h0 enters
// y1 = y0 + h0 f0 and the stage time t0 + h0 of every patient), hence two launches:
//   PASS 1: per-wave partial of  sum_p ( y1bar . f0 + f1bar[3] (-kel^2 Dose(t0 + h0)) )      -> partials[n_waves + wave]
//   PASS 2: everything else; grad_y0 += ..., parameter-gradient partials for a second fold.
// Both recompute f0, y1, f1 (one attempt's worth of work per launch, once per solve).  The branch decisions are re-taken
// from the fp32 values the forward left in the DpInit record, with the forward's own comparisons.
template <int D, int LPP, bool ABLATE, bool HILL2, bool NEED_TH, bool K1, int PASS>
HODE_DEV void dp_initbwd_body(const DpArgs& a) {
  using Ml = MlSlice<D, LPP>;
  constexpr int MR = Ml::MR;
  constexpr int M = D - 4;
  constexpr int P = M * D + M + kNTheta;
  const LaneMap<LPP> lm(a.B, a.ppw);
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const DpInit in = *a.init;
  const float sigma = fold_waves(a.partials, a.n_waves, 1, 0);
  float* gout = a.grad_partials + (size_t)wave * P;
  if (!in.first_accepted || sigma == 0.0f) {
    // dt_0 never reached the outputs (attempt 0 rejected: every later step size is a controller constant)
    if constexpr (PASS == 1) {
      if (lane == 0) a.partials[a.n_waves + wave] = 0.f;
    } else {
      for (int i = lane; i < P; i += 64) gout[i] = 0.f;
      if (wave == 0 && lane == 0) a.init->sigma = in.first_accepted ? sigma : 0.f;
    }
    return;
  }
  const RocheTheta th = load_theta(a.theta, ABLATE);
  Ml ml;
  ml.load(a.w1, a.b1, lm.q);
  MlColSlice<D, LPP> mc;
  mc.load(a.w1, lm.q);
  const float ln_ec50 = log_f32(th.ec50);
  const DoseSched<K1> ds = dp_load_dose<K1>(a, lm.p);
  GradAcc<D, LPP> acc;
  acc.zero();
  const float live = lm.live ? 1.0f : 0.0f;
  const float NN = (float)a.B * (float)D;

  // ---- scalar part of the chain (identical in every lane)
  const float h0 = in.h0, d0 = in.d0, d1 = in.d1, d2 = in.d2, h1 = in.h1;
  const bool deg0 = d0 < 1e-5f || d1 < 1e-5f;
  const bool deg1 = d1 <= 1e-15f && d2 <= 1e-15f;
  const bool use_d2 = d2 > d1;                 // python max(d1, d2) keeps d1 unless d2 > d1
  const bool branch_a = 100.0f * h0 <= h1;     // dt_0 = 100 h0
  float h0b = branch_a ? 100.0f * sigma : 0.0f;
  const float h1b = branch_a ? 0.0f : sigma;
  float d1b = 0.f, d2b = 0.f;
  if (deg1) {
    if (h0 * 1e-3f > 1e-6f) h0b = __builtin_fmaf(1e-3f, h1b, h0b);
  } else {
    const float mb = -0.2f * div_f32(h1, use_d2 ? d2 : d1) * h1b;
    if (use_d2) d2b = mb; else d1b = mb;
  }
  const float r2 = d2 * h0;  // rms((f1 - f0)/scale)
  float r2b = 0.f;
  if (d2b != 0.0f && r2 > 0.0f) {
    r2b = div_f32(d2b, h0);
    h0b -= div_f32(d2b * d2, h0);
  }

  // ---- per patient: recompute f0, y1, f1
  float y[D], f0[D], y1[D], f1[D], own0[MR], own1[MR];
  load_vec<D>(a.y0 + (size_t)lm.p * D, y);
  const float t0f = a.t[0];
  const DoseVal dv0 = ds.at(t0f, th.kel);
  roche_rhs<D, LPP, ABLATE, HILL2>(th, ml, dv0.v, y, f0, own0);
#pragma unroll
  for (int c = 0; c < D; ++c) y1[c] = __builtin_fmaf(h0, f0[c], y[c]);
  const DoseVal dv1 = ds.at(add_rn(t0f, h0), th.kel);
  roche_rhs<D, LPP, ABLATE, HILL2>(th, ml, dv1.v, y1, f1, own1);
  float scale[D], w[D], wb[D], f1b[D], y1b[D];
  const float cw = r2b != 0.0f ? div_f32(r2b, NN * r2) * live : 0.0f;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    scale[c] = a.atol + __builtin_fabsf(y[c]) * a.rtol;
    w[c] = div_f32(f1[c] - f0[c], scale[c]);
    wb[c] = cw * w[c];
    f1b[c] = div_f32(wb[c], scale[c]);
  }
  roche_vjp<D, LPP, ABLATE, HILL2, NEED_TH>(th, ml, mc, ln_ec50, dv1, y1, own1, f1b, lm.q, y1b, acc);
  const float mine = (LPP == 1 || lm.q == 0) ? 1.0f : 0.0f;
  if constexpr (PASS == 1) {
    float sp = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) sp = __builtin_fmaf(y1b[c], f0[c], sp);
    if constexpr (!ABLATE) sp = __builtin_fmaf(f1b[3], -th.kel * th.kel * dv1.v, sp);
    sp = wave_sum(sp * mine);
    if (lane == 0) a.partials[a.n_waves + wave] = sp;
    return;
  } else {
    h0b += fold_waves(a.partials + a.n_waves, a.n_waves, 1, 0);
    float d0b = 0.f;
    if (!deg0) {
      d0b = div_f32(0.01f, d1) * h0b;
      d1b -= div_f32(h0, d1) * h0b;
    }
    const float cv = d1 > 0.0f ? div_f32(d1b, NN * d1) * live : 0.0f;
    const float cu = d0 > 0.0f ? div_f32(d0b, NN * d0) * live : 0.0f;
    float f0b[D], yb[D], a0[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const float rs = div_f32(1.0f, scale[c]);
      const float v = f0[c] * rs, u = y[c] * rs;
      const float vb = cv * v, ub = cu * u;
      f0b[c] = __builtin_fmaf(h0, y1b[c], (vb - wb[c]) * rs);
      const float sb = -(wb[c] * w[c] + vb * v + ub * u) * rs;
      const float sgn = y[c] > 0.0f ? 1.0f : (y[c] < 0.0f ? -1.0f : 0.0f);
      yb[c] = y1b[c] + ub * rs + sb * a.rtol * sgn;
    }
    roche_vjp<D, LPP, ABLATE, HILL2, NEED_TH>(th, ml, mc, ln_ec50, dv0, y, own0, f0b, lm.q, a0, acc);
    float gy[D];
    load_vec<D>(a.grad_y0 + (size_t)lm.p * D, gy);
#pragma unroll
    for (int c = 0; c < D; ++c) gy[c] += yb[c] + a0[c];
    store_vec<D, LPP>(a.grad_y0 + (size_t)lm.p * D, gy, lm.q, lm.live);
    dp_store_grad_partials<D, LPP, NEED_TH>(acc, gout);
    if (wave == 0 && lane == 0) a.init->sigma = sigma;
  }
}

template <int D, int LPP, bool ABLATE, bool NEED_TH, int PASS>
__global__ __launch_bounds__(64) void dp_initbwd_kernel(DpArgs a) {
  const bool hill2 = ABLATE || (a.theta[0] == 2.0f && a.theta[1] == 2.0f);
  if (hill2 && a.K == 1) dp_initbwd_body<D, LPP, ABLATE, true, NEED_TH, true, PASS>(a);
  else if (hill2) dp_initbwd_body<D, LPP, ABLATE, true, NEED_TH, false, PASS>(a);
  else dp_initbwd_body<D, LPP, ABLATE, false, NEED_TH, false, PASS>(a);
}

// ---------------------------------------------------------------------------------------------- launch helpers
struct DpLaunch {
  int lpp;
  bool ablate, need_th;
  int phase;  // 0 init1, 1 init2, 2 attempt, 3 backward sweep, 4 / 5 initial-step backward pass 1 / 2, 6 persistent attempt loop
  int waves_per_block = 1;  // attempt launches only (1..4)
};

template <int D, int LPP, bool ABLATE>
int dp_launch(const DpLaunch& L, const DpArgs& a, hipStream_t s) {
  const dim3 grid(a.n_waves), block(64);
  switch (L.phase) {
    case 0: hipLaunchKernelGGL((dp_fwd_kernel<D, LPP, ABLATE, 0>), grid, block, 0, s, a); break;
    case 1: hipLaunchKernelGGL((dp_fwd_kernel<D, LPP, ABLATE, 1>), grid, block, 0, s, a); break;
    case 2: {
      const int wpb = L.waves_per_block > 0 ? L.waves_per_block : 1;
      hipLaunchKernelGGL((dp_fwd_kernel<D, LPP, ABLATE, 2>), dim3((a.n_waves + wpb - 1) / wpb), dim3(64 * wpb), 0, s, a);
      break;
    }
    case 3:
      if (L.need_th) hipLaunchKernelGGL((dp_bwd_kernel<D, LPP, ABLATE, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((dp_bwd_kernel<D, LPP, ABLATE, false>), grid, block, 0, s, a);
      break;
    case 6:
      if constexpr (LPP == 4) hipLaunchKernelGGL((dp_persist_kernel<D, ABLATE>), grid, block, 0, s, a);
      break;
    case 4:
      // pass 1 only forms a scalar; its parameter accumulators are dead code in either instantiation
      hipLaunchKernelGGL((dp_initbwd_kernel<D, LPP, ABLATE, false, 1>), grid, block, 0, s, a);
      break;
    case 5:
      if (L.need_th) hipLaunchKernelGGL((dp_initbwd_kernel<D, LPP, ABLATE, true, 2>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((dp_initbwd_kernel<D, LPP, ABLATE, false, 2>), grid, block, 0, s, a);
      break;
  }
  return hip_fail(hipGetLastError(), "dopri5 kernel launch");
}

template <int D>
int dp_dispatch(const DpLaunch& L, const DpArgs& a, hipStream_t s) {
  if constexpr (D > 4 && (D - 4) % 4 == 0) {
    if (L.lpp == 4) return L.ablate ? dp_launch<D, 4, true>(L, a, s) : dp_launch<D, 4, false>(L, a, s);
  }
  return L.ablate ? dp_launch<D, 1, true>(L, a, s) : dp_launch<D, 1, false>(L, a, s);
}

}  // namespace hode
