// C-ABI entry points hode_dopri5_fwd / hode_dopri5_bwd (include/hode.h): workspace carving, the attempt loop, checks.
#include <stdlib.h>
#include <string.h>

#include "hode_dopri5_kernels.hpp"

namespace {

using hode::DpArgs;
using hode::DpCtrl;
using hode::DpLaunch;

// Attempts enqueued between two reads of the controller record.  Every read is a host round trip during which the GPU
// idles (~70 us measured: 29-30 ms per solve with a fixed chunk of 32, 26.7 ms with 128 at 4 200 attempts), every attempt
// enqueued past the end costs an early-exit launch (~1.3 us).  The chunk therefore starts small, doubles while nothing is
// known, and then follows an estimate of what is left: attempts so far scaled by the output-grid progress j_next / T.
constexpr int kChunkFirst = 64, kChunkMin = 32, kChunkMax = 2048;
constexpr int kAttemptWavesPerBlock = 1;

int next_chunk(int chunk, long long attempts, int j_next, int n_times) {
  const double done = n_times > 1 ? (double)(j_next - 1) / (double)(n_times - 1) : 1.0;
  if (done <= 0.0) return chunk * 2 > kChunkMax ? kChunkMax : chunk * 2;
  const double left = (double)attempts * (1.0 - done) / done;
  long long c = (long long)(0.75 * left);
  if (c < kChunkMin) c = kChunkMin;
  if (c > kChunkMax) c = kChunkMax;
  return (int)c;
}

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
constexpr size_t kInitOffset = 128;  // DpInit sits behind the two DpCtrl records
static_assert(2 * sizeof(DpCtrl) <= kInitOffset, "controller records overlap the init record");
static_assert(sizeof(hode::DpInit) == sizeof(hode_dopri5_init_record), "DpInit is the ABI's hode_dopri5_init_record");

// Patients per wave of the dopri5 kernels: full waves.  The fixed-grid kernels spread a small batch over ~one wave per
// SIMD (hode::patients_per_wave) because they stream h every step; an attempt launch has no such stream, its cost per
// wave does not depend on the number of live lanes, and every extra wave is one more workgroup to dispatch and one more
// error-norm partial for every wave of the next launch to read: 6.0-6.1 us per attempt with 16 patients per wave against
// 6.3 with 10 at 10 000 patients (A/B, same call); the backward is indifferent.  HODE_PPW still overrides.
int dp_patients_per_wave(const hode_solve_desc* d) {
  const int cap = 64 / hode::choose_lpp(d);
  if (getenv("HODE_PPW")) return hode::patients_per_wave(d->batch, hode::choose_lpp(d));
  return cap;
}
int dp_n_waves(const hode_solve_desc* d) {
  const int ppw = dp_patients_per_wave(d);
  return (d->batch + ppw - 1) / ppw;
}

struct DpLayout {
  size_t ctrl, partials, slots, kbuf, tape_t, tape_dt, tape_j, tape_y, grad_partials, total;
};

DpLayout dp_layout(const hode_solve_desc* d) {
  const int nw = dp_n_waves(d);
  const size_t BD = (size_t)d->batch * d->latent_dim;
  const size_t S = (size_t)(d->max_steps > 0 ? d->max_steps : 1);
  DpLayout L;
  size_t off = 0;
  L.ctrl = off; off = align_up(off + kInitOffset + sizeof(hode::DpInit));  // two controller records + the DpInit record
  L.partials = off; off = align_up(off + (size_t)4 * nw * sizeof(float));
  L.slots = off; off = align_up(off + (size_t)2 * nw * sizeof(unsigned long long));
  L.kbuf = off; off = align_up(off + 7 * BD * sizeof(float));
  L.tape_t = off; off = align_up(off + S * sizeof(double));
  L.tape_dt = off; off = align_up(off + S * sizeof(double));
  L.tape_j = off; off = align_up(off + 2 * S * sizeof(int));
  L.tape_y = off; off = align_up(off + ((d->flags & HODE_FLAG_NO_TAPE) ? 2 : S + 1) * BD * sizeof(float));
  L.grad_partials = off; off = align_up(off + (size_t)nw * hode::n_partials(d) * sizeof(float));
  L.total = off;
  return L;
}

DpArgs dp_args(const hode_solve_desc* d, const DpLayout& L) {
  DpArgs a{};
  char* ws = (char*)d->workspace;
  a.t = d->t; a.y0 = d->y0; a.dosage = d->dosage; a.dose_times = d->dose_times; a.theta = d->theta;
  a.w1 = d->w1; a.b1 = d->b1; a.h = d->h;
  a.ctrl = (DpCtrl*)(ws + L.ctrl);
  a.init = (hode::DpInit*)(ws + L.ctrl + kInitOffset);
  a.partials = (float*)(ws + L.partials);
  a.slots = (unsigned long long*)(ws + L.slots);
  a.kbuf = (float*)(ws + L.kbuf);
  a.tape_t = (double*)(ws + L.tape_t);
  a.tape_dt = (double*)(ws + L.tape_dt);
  a.tape_j = (int*)(ws + L.tape_j);
  a.tape_y = (float*)(ws + L.tape_y);
  a.grad_h = d->grad_h; a.grad_y0 = d->grad_y0;
  a.hill2 = -1;
  a.grad_partials = (float*)(ws + L.grad_partials);
  a.B = d->batch; a.T = d->n_times; a.K = d->n_dose;
  a.n_waves = dp_n_waves(d);
  a.ppw = dp_patients_per_wave(d);
  a.max_steps = d->max_steps;
  a.ring = (d->flags & HODE_FLAG_NO_TAPE) ? 1 : 0;
  a.rtol = (float)d->rtol; a.atol = (float)d->atol;
  return a;
}

int dp_dispatch_dim(const hode_solve_desc* d, const DpLaunch& L, const DpArgs& a, hipStream_t s) {
  switch (d->latent_dim) {
    case 4: return hode::dp_dispatch_d4(L, a, s);
    case 6: return hode::dp_dispatch_d6(L, a, s);
    case 8: return hode::dp_dispatch_d8(L, a, s);
    case 12: return hode::dp_dispatch_d12(L, a, s);
  }
  return hode::fail(HODE_E_UNSUPPORTED, "dopri5: latent_dim %d has no compiled kernel (have 4, 6, 8, 12)", d->latent_dim);
}

bool is_neural(const hode_solve_desc* d) { return d && d->struct_size == sizeof(hode_solve_desc) && d->rhs_kind == HODE_RHS_NEURAL; }

// argument checks shared by both rhs families
int check_common(const hode_solve_desc* d, bool bwd) {
  if (!d) return hode::fail(HODE_E_NULL, "descriptor is NULL");
  if (d->struct_size != sizeof(hode_solve_desc))
    return hode::fail(HODE_E_SIZE, "struct_size %u != %zu (ABI mismatch)", d->struct_size, sizeof(hode_solve_desc));
  if (d->batch <= 0 || d->n_times <= 0 || d->latent_dim < 4 || d->n_dose < 0 || d->max_steps <= 0)
    return hode::fail(HODE_E_SIZE, "bad sizes: batch=%d n_times=%d latent_dim=%d n_dose=%d max_steps=%d", d->batch,
                      d->n_times, d->latent_dim, d->n_dose, d->max_steps);
  if (!(d->rtol > 0) || !(d->atol >= 0)) return hode::fail(HODE_E_SIZE, "rtol must be > 0 and atol >= 0");
  if (!d->t || !d->y0 || !d->dosage || !d->h || (d->n_dose > 0 && !d->dose_times))
    return hode::fail(HODE_E_NULL, "t / y0 / dosage / dose_times / h must be non-NULL");
  if (!d->host_n_accepted) return hode::fail(HODE_E_NULL, "host_n_accepted is required (fwd: out, bwd: in)");
  if (bwd && (!d->grad_h || !d->grad_y0)) return hode::fail(HODE_E_NULL, "grad_h / grad_y0 required by the backward");
  if (bwd && (d->flags & HODE_FLAG_OVERWRITE_GRADS))
    return hode::fail(HODE_E_UNSUPPORTED, "HODE_FLAG_OVERWRITE_GRADS is only implemented by hode_rk_bwd");
  if (bwd && (d->flags & HODE_FLAG_NO_TAPE))
    return hode::fail(HODE_E_UNSUPPORTED, "the forward ran with HODE_FLAG_NO_TAPE: there is no tape to sweep");
  uintptr_t m = (uintptr_t)d->y0 | (uintptr_t)d->h | (uintptr_t)d->workspace;
  if (bwd) m |= (uintptr_t)d->grad_h | (uintptr_t)d->grad_y0;
  if (m & 15) return hode::fail(HODE_E_ALIGN, "y0 / h / grad_h / grad_y0 / workspace must be 16-byte aligned");
  return 0;
}

int check_dp(const hode_solve_desc* d, bool bwd) {
  if (int e = check_common(d, bwd)) return e;
  if (d->rhs_kind != HODE_RHS_ROCHE && d->rhs_kind != HODE_RHS_ROCHE_ABLATE)
    return hode::fail(HODE_E_UNSUPPORTED, "rhs_kind %d has no dopri5 kernels (have ROCHE, ROCHE_ABLATE, NEURAL)", d->rhs_kind);
  if (!d->theta) return hode::fail(HODE_E_NULL, "theta must be non-NULL");
  if (d->latent_dim > 4 && (!d->w1 || !d->b1)) return hode::fail(HODE_E_NULL, "w1 / b1 required when latent_dim > 4");
  const DpLayout L = dp_layout(d);
  if (!d->workspace || d->workspace_bytes < L.total)
    return hode::fail(HODE_E_WORKSPACE, "workspace %zu B < required %zu B", d->workspace_bytes, L.total);
  return 0;
}

}  // namespace

extern "C" size_t hode_dopri5_workspace_bytes(const hode_solve_desc* d) {
  return is_neural(d) ? hode::neural_dopri5_workspace_bytes(d) : dp_layout(d).total;
}

extern "C" int hode_dopri5_fwd(const hode_solve_desc* d, void* stream) {
  if (is_neural(d)) {
    if (int e = check_common(d, false)) return e;
    return hode::neural_dopri5(d, false, (hipStream_t)stream);
  }
  if (int e = check_dp(d, false)) return e;
  hipStream_t s = (hipStream_t)stream;
  const DpLayout lay = dp_layout(d);
  DpArgs a = dp_args(d, lay);
  DpLaunch L;
  L.lpp = hode::choose_lpp(d);
  L.ablate = d->rhs_kind == HODE_RHS_ROCHE_ABLATE;
  L.need_th = false;
  L.phase = 0;
  if (int e = dp_dispatch_dim(d, L, a, s)) return e;
  L.phase = 1;
  if (int e = dp_dispatch_dim(d, L, a, s)) return e;
  L.phase = 2;
  L.waves_per_block = kAttemptWavesPerBlock;
#ifdef HODE_DP_EXPERIMENTS   // diagnostic builds only: a stray environment variable must not change the shipped launch structure
  if (const char* env = getenv("HODE_DP_WPB")) {  // tuning
    const int v = atoi(env);
    if (v >= 1 && v <= 4) L.waves_per_block = v;
  }
#endif
  {
    // which rhs specialisation the attempts run (both Hill exponents == 2 is the shipped configuration) is a property of
    // the parameters: read the two exponents back once, under the init kernels, instead of on the device in every launch
    float hill[2] = {0.f, 0.f};
    if (int e = hode::hip_fail(hipMemcpyAsync(hill, d->theta, sizeof(hill), hipMemcpyDeviceToHost, s), "theta read-back")) return e;
    if (int e = hode::hip_fail(hipStreamSynchronize(s), "theta read-back sync")) return e;
    a.hill2 = (hill[0] == 2.0f && hill[1] == 2.0f) ? 1 : 0;
  }
  DpCtrl host{};
  int attempt = 0;
  // every attempt either accepts (<= max_steps of those) or shrinks dt by >= 5x towards underflow: a generous bound
  const long long max_attempts = 64LL * ((long long)d->max_steps + 64);
  // Experiment builds only (-DHODE_DP_EXPERIMENTS, then HODE_DP_PERSIST=1, quad layout): the whole attempt loop in one persistent launch (dp_persist_body_own: state in
  // registers, one hop through memory per attempt instead of a kernel boundary).  Correct and bounded, but MEASURED SLOWER on
  // this part -- 7.0 us per attempt against 5.7 (tools/dp_persist_probe.py): a fence-free, atomics-free all-to-all exchange
  // across 8 XCDs still costs ~5 us, more than the ~3.8 us the kernel boundary costs.  Kept as the
  // reproducible form of that measurement; a launch that cannot assemble its waves falls back to the loop below.
  bool persist = false;
#ifdef HODE_DP_EXPERIMENTS
  if (const char* env = getenv("HODE_DP_PERSIST")) persist = atoi(env) != 0 && L.lpp == 4 && a.n_waves <= hode::kDpMaxPersistWaves;
#endif
  if (persist) {
    if (int e = hode::hip_fail(hipMemsetAsync(a.slots, 0, (size_t)2 * a.n_waves * sizeof(unsigned long long), s), "slot clear")) return e;
    L.phase = 6;
    a.max_iters = (int)(max_attempts > 0x3fffffff ? 0x3fffffff : max_attempts);
    if (int e = dp_dispatch_dim(d, L, a, s)) return e;
    if (int e = hode::hip_fail(hipMemcpyAsync(&host, a.ctrl, sizeof(DpCtrl), hipMemcpyDeviceToHost, s), "controller read-back")) return e;
    if (int e = hode::hip_fail(hipStreamSynchronize(s), "controller read-back sync")) return e;
    if (!(host.status & hode::kDpStatusBarrierTimeout)) {
      if (!host.done) host.status |= HODE_STATUS_MAX_STEPS;
    } else {
      // the grid could not assemble (another kernel held the CUs): start over, one launch per attempt
      persist = false;
      L.phase = 0;
      if (int e = dp_dispatch_dim(d, L, a, s)) return e;
      L.phase = 1;
      if (int e = dp_dispatch_dim(d, L, a, s)) return e;
      host = DpCtrl{};
    }
  }
  L.phase = 2;
  int chunk = kChunkFirst;
  while (!persist) {
    for (int i = 0; i < chunk; ++i) {
      a.attempt = attempt++;
      if (int e = dp_dispatch_dim(d, L, a, s)) return e;
    }
    // the ONE host synchronisation of the path: the number of adaptive steps is data dependent
    if (int e = hode::hip_fail(hipMemcpyAsync(&host, a.ctrl + (attempt & 1), sizeof(DpCtrl), hipMemcpyDeviceToHost, s),
                               "controller read-back"))
      return e;
    if (int e = hode::hip_fail(hipStreamSynchronize(s), "controller read-back sync")) return e;
    if (host.done) break;
    if (attempt > max_attempts) {
      host.status |= HODE_STATUS_MAX_STEPS;
      break;
    }
    chunk = next_chunk(chunk, attempt, host.j_next, d->n_times);
  }
  *d->host_n_accepted = host.n_acc;
  if (d->host_n_rejected) *d->host_n_rejected = host.n_rej;
  if (d->status && host.status) {
    if (int e = hode::hip_fail(hipMemcpyAsync(d->status, &host.status, sizeof(int), hipMemcpyHostToDevice, s), "status write"))
      return e;
    if (int e = hode::hip_fail(hipStreamSynchronize(s), "status write sync")) return e;
  }
  return 0;
}

extern "C" int hode_dopri5_bwd(const hode_solve_desc* d, void* stream) {
  if (is_neural(d)) {
    if (int e = check_common(d, true)) return e;
    return hode::neural_dopri5(d, true, (hipStream_t)stream);
  }
  if (int e = check_dp(d, true)) return e;
  hipStream_t s = (hipStream_t)stream;
  const DpLayout lay = dp_layout(d);
  DpArgs a = dp_args(d, lay);
  a.n_acc = *d->host_n_accepted;
  if (a.n_acc < 0 || a.n_acc > d->max_steps) return hode::fail(HODE_E_SIZE, "n_accepted %d outside the tape", a.n_acc);
  DpLaunch L;
  L.lpp = hode::choose_lpp(d);
  L.ablate = d->rhs_kind == HODE_RHS_ROCHE_ABLATE;
  L.need_th = d->need_theta_grad != 0;
  L.phase = 3;
  if (int e = dp_dispatch_dim(d, L, a, s)) return e;
  const int M = d->latent_dim - 4;
  if (int e = hode::launch_fold_partials(a.grad_partials, a.n_waves, hode::n_partials(d), M * d->latent_dim, M, d->grad_w1,
                                         d->grad_b1, d->grad_theta, d->need_theta_grad, s))
    return e;
  if (a.n_acc == 0 || (d->flags & HODE_FLAG_DETACH_FIRST_STEP)) return 0;
  // the reference's graph differentiates dt_0 (kernels: "backward of the initial step size"); the partial array is free
  // again once the fold above has read it (same stream)
  L.phase = 4;
  if (int e = dp_dispatch_dim(d, L, a, s)) return e;
  L.phase = 5;
  if (int e = dp_dispatch_dim(d, L, a, s)) return e;
  return hode::launch_fold_partials(a.grad_partials, a.n_waves, hode::n_partials(d), M * d->latent_dim, M, d->grad_w1,
                                    d->grad_b1, d->grad_theta, d->need_theta_grad, s);
}

extern "C" int hode_dopri5_tape_offsets(const hode_solve_desc* d, size_t* out5) {
  if (!d || !out5) return hode::fail(HODE_E_NULL, "descriptor / out5 is NULL");
  if (d->struct_size != sizeof(hode_solve_desc)) return hode::fail(HODE_E_SIZE, "struct_size mismatch");
  if (is_neural(d)) return hode::neural_dopri5_tape_offsets(d, out5);
  const DpLayout L = dp_layout(d);
  out5[0] = L.ctrl + kInitOffset;
  out5[1] = L.tape_t;
  out5[2] = L.tape_dt;
  out5[3] = L.tape_j;
  out5[4] = L.tape_y;
  return 0;
}
