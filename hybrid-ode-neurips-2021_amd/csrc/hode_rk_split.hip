// "Split" layout of the fixed-grid Roche solve: wave-specialised expert / learned pipelines, gfx950 (D in {8, 12}).
//
// Same arithmetic and C-ABI contract as hode_rk_kernels.hpp.  Why it exists: the quad-layout kernels are
// bound by VALU ISSUE on ONE wave -- a SIMD retires one dependent wave-instruction per ~4-5 cycles and does not overlap
// the VALU work of several waves -- while at 10 000 patients 40 % of the SIMDs have no wave at all.  So the work of a
// patient group is split over waves that run CONCURRENTLY ON DIFFERENT SIMDs of one CU:
//   * the expert block (Disease, ImmuneReact, Immunity, Dose2) is AUTONOMOUS: it never reads the learned latents
//     (reference model.py:527-544).  Wave 0 of a workgroup integrates it alone, one patient per lane (48 patients),
//     ONE STEP AHEAD, and publishes the 4 stage states of every step in an LDS ring ([2][4 stages][48][4] floats);
//   * waves 1..3 integrate the learned latents of 16 patients each in the quad layout (a patient per DPP quad, each lane
//     owns (D-4)/4 rows of tanh(W y + b) AND only the matching components of the state -- stage algebra on 1-2
//     components instead of D), reading the expert stage states from LDS (one ds_read_b128, quad-broadcast).
// One __syncthreads per step hands the rings over.  A workgroup = 4 waves = the 4 SIMDs of a CU; 209 workgroups at the
// bench shape.
//
// The kernels are bound by ISSUE SLOTS (every wave-instruction ~2.4 ns), so the instruction streams are
// written for count, not for flops:
//   * packed fp32 (f2 = ext_vector_type(2), v_pk_fma_f32) by hand: the learned block as (row0, row1) accumulators with the
//     state component broadcast through op_sel, two interleaved chains per product (a packed result cannot feed the very
//     next instruction), tanh's scale folded into the weights; the expert wave's stage / cotangent algebra on the pairs
//     (Disease, ImmuneReact), (Immunity, Dose2);
//   * the dose schedule is evaluated by the learned waves, stage q by quad lane q, and handed to the expert wave through a
//     third LDS ring (4 stages in parallel instead of 4 x 9 serial instructions);
//   * HODE_FLAG_TAPE: the forward leaves the expert block's intermediate stage states in the workspace, the backward's
//     expert wave loads them an iteration ahead instead of re-integrating (bit-identical, +16 B x 3 per patient / step);
//     for rk4 also tanh(W Y_s + b) of the last two stages (loaded two iterations ahead by the backward's learned waves);
//   * adjoint with the tape and theta gradients: a FIFTH wave accumulates the 13 (15) expert-parameter gradients one
//     iteration behind the expert wave, off its cotangent chain (stage cotangents and doses through two LDS rings, stage
//     states from the tape); it shares a SIMD with the expert wave (100 -> 80 us together with the tape);
//   * adjoint with the tape: the transposed product c_q = sum_j W[j][q] u_j (the learned block's share of the expert cotangent)
//     runs on a wave of its own between the learned waves and the expert wave, which then lags two iterations; rings three
//     deep (81.0 -> 79.4 us, bit-identical);
//   * the time grid sits in LDS (dynamic shared memory, hence n_times <= 8192 for this layout);
//   * time loops unrolled by two through a generic lambda so that ring parities are immediates; all ring reads of a step
//     are issued up front and pinned with sched_barrier; per-step stores are unpredicated (lanes beyond the batch hold
//     bit-identical copies of patient B-1); the backward epilogue reduces with DPP row rotations + one LDS join.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"
#include "hode_roche.hpp"

namespace hode {

#ifndef HODE_SPLIT_EXPERT_FIRST
#define HODE_SPLIT_EXPERT_FIRST 1   // bias + expert columns of tanh(W y + b) ahead of the stage chain (MlRows::rhs_expert); 0: A/B
#endif
constexpr int kSplitPatients = 48;  // per workgroup: wave 0 holds all 48 (one per lane), waves 1..3 hold 16 each

struct SplitArgs {
  const float* __restrict__ t;
  const float* __restrict__ y0;
  const float* __restrict__ dosage;
  const float* __restrict__ dose_times;
  const float* __restrict__ theta;
  const float* __restrict__ w1;
  const float* __restrict__ b1;
  float* __restrict__ h;
  float* __restrict__ tape;  // TAPE: [T-1][NS-1][B][4] expert stage states 1..NS-1 of every step (stage 0 is h itself)
  unsigned long long* dbg;   // HODE_SPLIT_STAMPS builds only: [4 waves][T] s_memtime at each wave's arrival at the step barrier (block 0)
  float* __restrict__ ltape; // TAPE, rk4: [T-1][B][4 quad lanes][kLearnedTapeStages * MR] learned stage derivatives (see below)
  int* __restrict__ status;
  int B, T, K, perturb;
};

struct SpTimes {
  float t0, t1, dt, ts[4];
  HODE_DEV SpTimes(const float* __restrict__ t, int n, int perturb, int method) {
    t0 = t[n];
    t1 = t[n + 1];
    dt = t1 - t0;
    ts[0] = perturb ? nextafter_up(t0) : t0;
    ts[3] = perturb ? nextafter_down(t1) : t1;
    if (method == HODE_METHOD_RK4_38) {
      ts[1] = add_rn(t0, mul_rn(dt, (float)(1.0 / 3.0)));
      ts[2] = add_rn(t0, mul_rn(dt, (float)(2.0 / 3.0)));
    } else {
      ts[1] = add_rn(t0, mul_rn(0.5f, dt));
      ts[2] = ts[1];
    }
  }
};

// The time of stage q (0..3) of the step [t0, t1] exactly as SpTimes forms it, for a lane that needs only its own stage (the learned
// waves evaluate the dose schedule with one stage per quad lane).  t0 + dt * 0 is t0 exactly, so only q == 3 selects.
template <int METHOD>
HODE_DEV float sp_stage_time(float t0, float t1, int perturb, int q) {
  const float dt = t1 - t0;
  const float c = METHOD == HODE_METHOD_RK4_38 ? (q == 1 ? (float)(1.0 / 3.0) : (q == 2 ? (float)(2.0 / 3.0) : 0.f))
                                               : (q == 0 ? 0.f : 0.5f);
  float tq = add_rn(t0, mul_rn(dt, c));
  tq = q == 3 ? t1 : tq;
  if (perturb) {
    if (q == 0) tq = nextafter_up(t0);
    if (q == 3) tq = nextafter_down(t1);
  }
  return tq;
}

// The time grid staged in LDS (dynamic shared memory, T + 3 floats, the tail padded with t[T-1]): every wave reads its
// 2-4 grid points per step with one or two ds_read2 next to the ring reads, instead of a global load that has to be
// issued iterations ahead and rotated through registers (13 instructions per step on the learned waves).
constexpr int kSplitMaxT = 8192;
HODE_DEV void sp_stage_grid(float* __restrict__ tg, const float* __restrict__ t, int T) {
  for (int i = threadIdx.x; i < T + 3; i += blockDim.x) tg[i] = t[min(i, T - 1)];
}

template <int METHOD>
constexpr int sp_stages() { return METHOD == HODE_METHOD_EULER ? 1 : (METHOD == HODE_METHOD_MIDPOINT ? 2 : 4); }

// stage state i (0-based) of the scheme from y and the earlier stage derivatives; V = float (one component) or f2 (a
// packed pair of components, same operations in the same order: the results are identical)
template <int METHOD, class V>
HODE_DEV V sp_stage_state(int i, V y, float dt, V k1, V k2, V k3) {
  constexpr float c13 = (float)(1.0 / 3.0);
  if constexpr (METHOD == HODE_METHOD_MIDPOINT) return i == 0 ? y : vfma(k1, vsplat<V>(0.5f * dt), y);
  else if constexpr (METHOD == HODE_METHOD_RK4_38) {
    if (i == 0) return y;
    if (i == 1) return vfma(dt * k1, vsplat<V>(c13), y);
    if (i == 2) return vfma(vsplat<V>(dt), vfma(-k1, vsplat<V>(c13), k2), y);
    return vfma(vsplat<V>(dt), (k1 - k2) + k3, y);
  } else return y;
}
template <int METHOD, class V>
HODE_DEV V sp_advance(V y, float dt, V k1, V k2, V k3, V k4) {
  if constexpr (METHOD == HODE_METHOD_EULER) return vfma(vsplat<V>(dt), k1, y);
  else if constexpr (METHOD == HODE_METHOD_MIDPOINT) return vfma(vsplat<V>(dt), k2, y);
  else return vfma((k1 + 3.0f * (k2 + k3)) + k4, vsplat<V>(dt * 0.125f), y);
}

// The time loops are unrolled by two by hand (a lambda per iteration, instantiated for both ring parities): every ring
// offset becomes an immediate instead of s_and / s_mul / v_add per ring and iteration.  (The compiler cannot unroll a loop
// with a barrier and a run-time trip count itself.)
template <int V> struct IC { static constexpr int value = V; };

template <int MR> struct OwnSel { typedef float type; };
template <> struct OwnSel<2> { typedef f2 type; };

template <bool C, class A, class B> struct TypeSel { typedef A type; };
template <class A, class B> struct TypeSel<false, A, B> { typedef B type; };

// Learned-block tape (HODE_FLAG_TAPE, rk4): the forward also leaves tanh(W Y_s + b) of the LAST kLearnedTapeStages stages of
// every step, in the lane order both kernels use ((patient, quad lane) -> kLearnedTapeStages * MR floats, contiguous per
// wave).  The backward's learned waves load them TWO iterations ahead instead of recomputing D packed fmas + a tanh pair
// per stage; the values are bit-identical to the recomputed ones.  Worth it only once the learned waves bound the
// adjoint, i.e. with the theta wave (see split_bwd_body); measured per stage count in DESIGN.md 4.3c.
#ifndef HODE_LTAPE_STAGES
#define HODE_LTAPE_STAGES 2
#endif
constexpr int kLearnedTapeStages = HODE_LTAPE_STAGES;
template <int METHOD> constexpr int sp_ltape_stages() { return METHOD == HODE_METHOD_RK4_38 ? kLearnedTapeStages : 0; }

// The learned block as one lane of a patient's DPP quad sees it: MR = (D-4)/4 rows of tanh(W y + b).  The weights are kept
// pre-multiplied by 2 log2(e), so tanh(z) = 1 - 2 / (exp2(z') + 1) needs no scaling instruction.
//   MR == 2: the two rows are the two halves of packed registers: wp[i] = (W[row0][i], W[row1][i]); a stage is D
//            v_pk_fma_f32 with the state component broadcast through op_sel, the result is the Own pair directly, and the
//            weight gradient dw[i] += u * Y_i has the same shape.  Every broadcast operand is a scalar of its own (an
//            element of a pair costs a v_mov when it is the odd one), so the stage state is kept as D floats;
//   MR == 1: wp[ip] = (W[row][2ip], W[row][2ip+1]); .x sums the even columns from the bias, .y the odd ones, one add
//            joins them; the stage state is kept as D/2 pairs.
template <int D>
struct MlRows {
  static constexpr int M = D - 4, MR = M / 4, DP = D / 2, MP = M / 2;
  static constexpr int NW = MR == 2 ? D : DP;   // packed weight registers per lane
  static constexpr int NU = MR == 2 ? M : MP;   // registers of an all-gathered cotangent
  static constexpr float kTanhScale = 2.885390081777927f;  // 2 log2(e)
  typedef typename OwnSel<MR>::type Own;
  typedef typename TypeSel<MR == 2, float, f2>::type Elem;
  struct Stage { Elem v[NW]; };  // the full stage state Y as this lane holds it
  struct Gath { Elem v[NU]; };   // all-gathered learned-block vector (u of the VJP)
  f2 wp[NW];
  Own bias;
  HODE_DEV void load(const float* __restrict__ W, const float* __restrict__ b, int q) {
    if constexpr (MR == 2) {
#pragma unroll
      for (int i = 0; i < D; ++i) wp[i] = kTanhScale * pair2(W[(2 * q) * D + i], W[(2 * q + 1) * D + i]);
      bias = kTanhScale * pair2(b[2 * q], b[2 * q + 1]);
    } else {
#pragma unroll
      for (int ip = 0; ip < DP; ++ip) wp[ip] = kTanhScale * pair2(W[q * D + 2 * ip], W[q * D + 2 * ip + 1]);
      bias = kTanhScale * b[q];
    }
  }
  // all-gather of a quad's own components (component j lives in lane j / MR)
  static HODE_DEV void gather(Own v, Elem* __restrict__ out) {
    if constexpr (MR == 2) {
      out[0] = quad_bcast<0>(v.x); out[1] = quad_bcast<0>(v.y);
      out[2] = quad_bcast<1>(v.x); out[3] = quad_bcast<1>(v.y);
      out[4] = quad_bcast<2>(v.x); out[5] = quad_bcast<2>(v.y);
      out[6] = quad_bcast<3>(v.x); out[7] = quad_bcast<3>(v.y);
    } else {
      out[0] = pair2(quad_bcast<0>(v), quad_bcast<1>(v));
      out[1] = pair2(quad_bcast<2>(v), quad_bcast<3>(v));
    }
  }
  // stage state from the expert wave's 4 components and this quad's own ones
  static HODE_DEV Stage stage(const float4& e, Own Yo) {
    Stage Y;
    if constexpr (MR == 2) {
      Y.v[0] = e.x; Y.v[1] = e.y; Y.v[2] = e.z; Y.v[3] = e.w;
      gather(Yo, Y.v + 4);
    } else {
      Y.v[0] = pair2(e.x, e.y);
      Y.v[1] = pair2(e.z, e.w);
      gather(Yo, Y.v + 2);
    }
    return Y;
  }
  static HODE_DEV Own tanh_scaled(Own z) {  // z already carries the factor 2 log2(e)
    if constexpr (MR == 2) {
      const f2 e = pair2(__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)) + splat2(1.0f);
      return vfma(pair2(__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)), splat2(-2.0f), splat2(1.0f));
    } else {
      return __builtin_fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z) + 1.0f), -2.0f, 1.0f);
    }
  }
  HODE_DEV Own rhs(const Stage& Y) const {
    if constexpr (MR == 2) {
      // two interleaved accumulators: the result of a packed op cannot be consumed by the very next instruction (the
      // assembler pads a dependent v_pk_fma chain with s_nop, 4.1 ns per link instead of 2.4)
      f2 acc = bias, acc1 = splat2(0.f);
#pragma unroll
      for (int i = 0; i < D; i += 2) {
        acc = vfma(wp[i], Y.v[i], acc);
        acc1 = vfma(wp[i + 1], Y.v[i + 1], acc1);
      }
      return tanh_scaled(acc + acc1);
    } else {
      f2 acc = pair2(bias, 0.f);
#pragma unroll
      for (int ip = 0; ip < DP; ++ip) acc = vfma(wp[ip], Y.v[ip], acc);
      return tanh_scaled(hsum(acc));
    }
  }
  // rhs split at the expert columns: the expert stage states of ALL stages of a step are known when the step starts (the
  // expert wave publishes them one step ahead), the learned components only stage by stage.  rhs_expert() forms the first
  // links of the accumulator chains -- bias and columns 0..3, in the order rhs() takes them -- off the stage-to-stage
  // dependency chain; rhs_rest() continues with the learned columns.  Same operations in the same order as rhs(): the result
  // is bit-identical, the chain through a stage is two packed fmas shorter.
  struct Part { f2 a, b; };
  HODE_DEV Part rhs_expert(const float4& e) const {
    Part p;
    if constexpr (MR == 2) {
      p.a = vfma(wp[2], e.z, vfma(wp[0], e.x, bias));
      p.b = vfma(wp[3], e.w, vfma(wp[1], e.y, splat2(0.f)));
    } else {
      p.a = vfma(wp[1], pair2(e.z, e.w), vfma(wp[0], pair2(e.x, e.y), pair2(bias, 0.f)));
      p.b = splat2(0.f);
    }
    return p;
  }
  HODE_DEV Own rhs_rest(const Part& p, const Stage& Y) const {
    if constexpr (MR == 2) {
      f2 acc = p.a, acc1 = p.b;
#pragma unroll
      for (int i = 4; i < D; i += 2) {
        acc = vfma(wp[i], Y.v[i], acc);
        acc1 = vfma(wp[i + 1], Y.v[i + 1], acc1);
      }
      return tanh_scaled(acc + acc1);
    } else {
      f2 acc = p.a;
#pragma unroll
      for (int ip = 2; ip < DP; ++ip) acc = vfma(wp[ip], Y.v[ip], acc);
      return tanh_scaled(hsum(acc));
    }
  }
  // dw += u (x) Y in the layout of wp
  static HODE_DEV void outer_acc(f2 (&dw)[NW], Own u, const Stage& Y) {
#pragma unroll
    for (int i = 0; i < NW; ++i) dw[i] = vfma(u, Y.v[i], dw[i]);
  }
  static HODE_DEV float dw_at(const f2 (&dw)[NW], int r, int i) {
    if constexpr (MR == 2) return r == 0 ? dw[i].x : dw[i].y;
    else return (i & 1) ? dw[i / 2].y : dw[i / 2].x;
  }
  static HODE_DEV Own load_own(const float* __restrict__ src, int q) {
    if constexpr (MR == 2) {
      const float2 x = *reinterpret_cast<const float2*>(src + 4 + 2 * q);
      return pair2(x.x, x.y);
    } else return src[4 + q];
  }
  static HODE_DEV void store_own(float* __restrict__ dst, int q, Own v) {
    if constexpr (MR == 2) *reinterpret_cast<float2*>(dst + 4 + 2 * q) = make_float2(v.x, v.y);
    else dst[4 + q] = v;
  }
  // NL stage derivatives of this lane as one contiguous record of NL * MR floats (dst / src point at the lane's record)
  template <int NL>
  static HODE_DEV void store_tape(float* __restrict__ dst, const Own* __restrict__ k) {
    if constexpr (MR == 2 && NL % 2 == 0) {
#pragma unroll
      for (int i = 0; i < NL; i += 2)
        *reinterpret_cast<float4*>(dst + 4 * (i / 2)) = make_float4(k[i].x, k[i].y, k[i + 1].x, k[i + 1].y);
    } else if constexpr (MR == 1 && NL == 4) {
      *reinterpret_cast<float4*>(dst) = make_float4(k[0], k[1], k[2], k[3]);
    } else if constexpr (MR == 1 && NL == 2) {
      *reinterpret_cast<float2*>(dst) = make_float2(k[0], k[1]);
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        if constexpr (MR == 2) *reinterpret_cast<float2*>(dst + 2 * i) = make_float2(k[i].x, k[i].y);
        else dst[i] = k[i];
      }
    }
  }
  template <int NL>
  static HODE_DEV void load_tape(const float* __restrict__ src, Own* __restrict__ k) {
    if constexpr (MR == 2 && NL % 2 == 0) {
#pragma unroll
      for (int i = 0; i < NL; i += 2) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * (i / 2));
        k[i] = pair2(v.x, v.y); k[i + 1] = pair2(v.z, v.w);
      }
    } else if constexpr (MR == 1 && NL == 4) {
      const float4 v = *reinterpret_cast<const float4*>(src);
      k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
    } else if constexpr (MR == 1 && NL == 2) {
      const float2 v = *reinterpret_cast<const float2*>(src);
      k[0] = v.x; k[1] = v.y;
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        if constexpr (MR == 2) { const float2 v = *reinterpret_cast<const float2*>(src + 2 * i); k[i] = pair2(v.x, v.y); }
        else k[i] = src[i];
      }
    }
  }
};

template <int D, int METHOD, bool ABLATE, bool HILL2, bool K1, bool TAPE>
HODE_DEV void split_fwd_body(const SplitArgs& a) {
  constexpr int NS = sp_stages<METHOD>();
  typedef MlRows<D> Ml;
  typedef typename Ml::Own Own;
  constexpr int DP = Ml::DP;
  __shared__ __attribute__((aligned(16))) float ring[2][4][kSplitPatients][4];
  // Dose(t_s) of the 4 stages of a step, written by the learned waves (stage q by quad lane q) one step ahead of the
  // expert wave: dring[n & 1] holds step n
  __shared__ __attribute__((aligned(16))) float dring[2][kSplitPatients][4];
  extern __shared__ float tg[];  // time grid, see sp_stage_grid
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int b0 = blockIdx.x * kSplitPatients;
  const RocheTheta th = load_theta(a.theta, ABLATE);
  const size_t row = (size_t)a.B * D;
  sp_stage_grid(tg, a.t, a.T);  // visible after the first __syncthreads of either pipeline
#ifdef HODE_SPLIT_STAMPS
#define HODE_FSTAMP(k) if (a.dbg && blockIdx.x == 0 && lane == 0) a.dbg[(size_t)wave * a.T + (k)] = __builtin_amdgcn_s_memtime();
#else
#define HODE_FSTAMP(k)
#endif

  if (wave == 0) {
    // ------------------------------------------------------------------ expert pipeline: one patient per lane
    const int slot = lane;
    const bool live = slot < kSplitPatients && b0 + slot < a.B;
    const int p = min(b0 + (slot < kSplitPatients ? slot : 0), a.B - 1);
    const int rslot = slot < kSplitPatients ? slot : 0;
    const unsigned lane_h = (unsigned)p * D, lane_t = (unsigned)p * 4;  // 32-bit lane offsets on wave-uniform bases
    MlSlice<4, 1> none;
    float own1[1];
    // the state as two packed pairs (Disease, ImmuneReact), (Immunity, Dose2): the stage algebra is 2 instructions wide
    f2 ya, yb;
    {
      const float4 v = *reinterpret_cast<const float4*>(a.y0 + (size_t)p * D);
      ya = pair2(v.x, v.y);
      yb = pair2(v.z, v.w);
      if (live) *reinterpret_cast<float4*>(a.h + (size_t)p * D) = v;
    }
    __syncthreads();  // the learned waves' prologue fills dring for steps 0 and 1; the grid is in LDS
    auto e_iter = [&](int it, auto PAR) {
      constexpr int par = decltype(PAR)::value;  // == it & 1
      if (it + 1 < a.T) {
        const float dt = tg[it + 1] - tg[it];
        const float4 dz = *reinterpret_cast<const float4*>(&dring[par][rslot][0]);
        const float dose[4] = {dz.x, dz.y, dz.z, dz.w};
        float* __restrict__ tape_it = TAPE ? a.tape + (size_t)it * (NS - 1) * a.B * 4 : nullptr;
        f2 ka[4], kb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ka[s] = kb[s] = splat2(0.f);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const f2 Ya = sp_stage_state<METHOD>(s, ya, dt, ka[0], ka[1], ka[2]);
          const f2 Yb = sp_stage_state<METHOD>(s, yb, dt, kb[0], kb[1], kb[2]);
          if (slot < kSplitPatients) *reinterpret_cast<float4*>(&ring[par][s][slot][0]) = make_float4(Ya.x, Ya.y, Yb.x, Yb.y);
          if constexpr (TAPE) {
            if (s >= 1)
              *reinterpret_cast<float4*>(tape_it + ((unsigned)(s - 1) * (unsigned)a.B * 4u + lane_t)) = make_float4(Ya.x, Ya.y, Yb.x, Yb.y);
          }
          const float Y[4] = {Ya.x, Ya.y, Yb.x, Yb.y};
          float k[4];
          roche_rhs<4, 1, ABLATE, HILL2>(th, none, dose[s], Y, k, own1);
          ka[s] = pair2(k[0], k[1]);
          kb[s] = pair2(k[2], k[3]);
        }
        ya = sp_advance<METHOD>(ya, dt, ka[0], ka[1], ka[2], ka[3]);
        yb = sp_advance<METHOD>(yb, dt, kb[0], kb[1], kb[2], kb[3]);
        // unpredicated (see the learned waves' store): spare lanes hold bit-identical copies of a live patient
        *reinterpret_cast<float4*>(a.h + (size_t)(it + 1) * row + lane_h) = make_float4(ya.x, ya.y, yb.x, yb.y);
      }
      HODE_FSTAMP(it)
      __syncthreads();
    };
    for (int it = 0; it < a.T; it += 2) {
      e_iter(it, IC<0>{});
      if (it + 1 < a.T) e_iter(it + 1, IC<1>{});
    }
    if (a.status) {
      const bool bad = !(vfinite(ya) && vfinite(yb));
      if (bad && live) atomicOr(a.status, HODE_STATUS_NONFINITE);
    }
  } else {
    // ------------------------------------------------------------------ learned pipeline: a patient per DPP quad
    const int q = lane & 3;
    const int slot = (wave - 1) * 16 + (lane >> 2);
    const bool live = b0 + slot < a.B;
    const int p = min(b0 + slot, a.B - 1);
    Ml ml;
    ml.load(a.w1, a.b1, q);
    DoseSched<K1> ds;
    ds.dosage = a.dosage[p];
    ds.K = a.K;
    ds.taus = a.dose_times + (size_t)p * a.K;
    ds.tau0 = K1 ? ds.taus[0] : 0.f;
    // Dose(t) at stage q of step n for the expert wave (it reads all four stages with one ds_read_b128)
    auto dose_step = [&](int parity, float t0, float t1) {  // step n goes to dring[n & 1]; lanes q >= NS: unread
      dring[parity][slot][q] = ds.at(sp_stage_time<METHOD>(t0, t1, a.perturb, q), th.kel).v;
    };
    if (a.T >= 2) dose_step(0, a.t[0], a.t[1]);  // before the first barrier: straight from global memory
    if (a.T >= 3) dose_step(1, a.t[1], a.t[2]);
    Own yo = Ml::load_own(a.y0 + (size_t)p * D, q);
    if (live) Ml::store_own(a.h + (size_t)p * D, q, yo);
    constexpr int NLT = sp_ltape_stages<METHOD>();
    const size_t lt_step = (size_t)a.B * 4 * NLT * Ml::MR;           // floats per step of the learned tape
    const unsigned lt_lane = ((unsigned)p * 4 + q) * NLT * Ml::MR;   // this lane's record
    __syncthreads();  // doses of steps 0 and 1 are in place
    __syncthreads();  // iteration 0: the expert wave fills ring[0]
    auto m_iter = [&](int it, auto PAR) {
      constexpr int par = decltype(PAR)::value;  // == (it - 1) & 1 == (it + 1) & 1
      const int n = it - 1;
      const float t_a = tg[it - 1], t_b = tg[it], t_c = tg[it + 1], t_d = tg[it + 2];  // padded past T-1
      const float dt = t_b - t_a;
      Own k[4];
      float4 e[4];  // all four expert stage states up front: one LDS round trip per step instead of one per stage
#pragma unroll
      for (int s = 0; s < NS; ++s) e[s] = *reinterpret_cast<const float4*>(&ring[par][s][slot][0]);
      __builtin_amdgcn_sched_barrier(0);  // keep the four reads here: the scheduler would sink each next to its use
      // the expert wave is at step `it` now and reads dring[it & 1]; past the last step this writes an unread slot
      dose_step(par, t_c, t_d);
#if HODE_SPLIT_EXPERT_FIRST
      typename Ml::Part pe[4];   // bias + expert columns of every stage, ahead of the stage chain (MlRows::rhs_expert)
#pragma unroll
      for (int s = 0; s < NS; ++s) pe[s] = ml.rhs_expert(e[s]);
#endif
#pragma unroll
      for (int s = 0; s < 4; ++s) k[s] = vsplat<Own>(0.f);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const Own Yo = sp_stage_state<METHOD>(s, yo, dt, k[0], k[1], k[2]);
#if HODE_SPLIT_EXPERT_FIRST
        k[s] = ml.rhs_rest(pe[s], Ml::stage(e[s], Yo));
#else
        k[s] = ml.rhs(Ml::stage(e[s], Yo));
#endif
      }
      yo = sp_advance<METHOD>(yo, dt, k[0], k[1], k[2], k[3]);
      // no `if (live)`: a quad beyond the batch integrates a bit-identical copy of patient B-1 (p is clamped) and
      // stores the same values to the same address -- cheaper than an exec-mask branch every step
      Ml::store_own(a.h + (size_t)(n + 1) * row + (size_t)p * D, q, yo);
      if constexpr (TAPE && NLT > 0) Ml::template store_tape<NLT>(a.ltape + (size_t)n * lt_step + lt_lane, k + (NS - NLT));
      HODE_FSTAMP(it)
      __syncthreads();
    };
    for (int it = 1; it < a.T; it += 2) {
      m_iter(it, IC<0>{});
      if (it + 1 < a.T) m_iter(it + 1, IC<1>{});
    }
    if (a.status) {
      if (!vfinite(yo) && live) atomicOr(a.status, HODE_STATUS_NONFINITE);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// The adjoint system reverses the dependency: the LEARNED cotangents are autonomous (the expert rhs never reads learned
// latents, so no expert term enters d/dy_learned), while the expert cotangent needs c_s = sum_j W[j][0..3] u_j from
// the learned block at every stage.  Per iteration k (ML handles step m_k = T-2-k):
//   wave 0 (expert):  (b) adjoint of step m_{k-1}: reads its own stage states / doses and the learned block's c_s from
//                     the rings of iteration k-1;  (a) publishes the stage states of step m_{k+1} into the rings
//                     (re-integrated from h, or -- TAPE -- loaded from the forward's tape one iteration earlier)
//   waves 1..3 (ML):  recompute + adjoint of step m_k (stage states from the ring), publish c_s; with the tape also the
//                     step's doses (v, d/dkel), stage q by quad lane q
//   one __syncthreads.  Rings are double buffered by iteration parity; (b) runs before (a) because (a) overwrites the
//   buffer (b) reads.
struct SplitBwdArgs {
  const float* __restrict__ t;
  const float* __restrict__ dosage;
  const float* __restrict__ dose_times;
  const float* __restrict__ theta;
  const float* __restrict__ w1;
  const float* __restrict__ b1;
  const float* __restrict__ h;
  const float* __restrict__ grad_h;
  float* __restrict__ grad_y0;
  float* __restrict__ part_ml;   // [3 * nblk][M*D + M]
  float* __restrict__ part_th;   // [nblk][kNTheta]
  const float* __restrict__ tape;  // TAPE: what the forward kernel left (see SplitArgs)
  const float* __restrict__ ltape;
  int B, T, K, perturb;
  unsigned long long* dbg;  // HODE_SPLIT_STAMPS builds only: [5 waves][T] s_memtime at each wave's arrival at the step barrier (block 0)
};

// The cotangent algebra of one step of the scheme, shared by the expert wave (V = f2, two pairs) and the learned waves
// (V = Own).  vjp(s, g) must return (df/dY)^T g at stage s.  lam is updated in place.
template <int METHOD, class V, class F>
HODE_DEV void sp_adjoint_step(V& lam, float dt, F&& vjp) {
  constexpr float c13 = (float)(1.0 / 3.0);
  if constexpr (METHOD == HODE_METHOD_EULER) {
    lam += vjp(0, dt * lam);
  } else if constexpr (METHOD == HODE_METHOD_MIDPOINT) {
    const V a1 = vjp(1, dt * lam);
    lam += a1;
    lam += vjp(0, (0.5f * dt) * a1);
  } else {
    const float w1 = dt * 0.125f, w3 = dt * 0.375f;
    const V a3 = vjp(3, w1 * lam);
    V da = dt * a3;
    V g1 = vfma(w1, lam, da);
    V g2 = vfma(w3, lam, -da);
    const V g = vfma(w3, lam, da);
    lam += a3;
    const V a2 = vjp(2, g);
    da = dt * a2;
    g2 += da;
    g1 = vfma(-c13, da, g1);
    lam += a2;
    const V a1 = vjp(1, g2);
    g1 = vfma(c13, dt * a1, g1);
    lam += a1;
    lam += vjp(0, g1);
  }
}

struct F4 {  // the expert wave's 4 components as two packed pairs
  f2 a, b;
  HODE_DEV F4& operator+=(const F4& o) { a += o.a; b += o.b; return *this; }
  HODE_DEV F4 operator-() const { return F4{-a, -b}; }
};
HODE_DEV F4 operator*(float s, const F4& v) { return F4{s * v.a, s * v.b}; }
HODE_DEV F4 vfma(float s, const F4& x, const F4& y) { return F4{vfma(s, x.a, y.a), vfma(s, x.b, y.b)}; }

// LDS of one backward workgroup, declared ONCE in the kernel and handed to the body instantiation that runs (a `__shared__`
// array inside split_bwd_body exists once per instantiation: three of them -- HILL2 / K1 variants -- made 94 KB of 31).
//   CW (c-wave, with the tape): the transposed product of the expert cotangent, c_q = sum_j W[j][q] u_j, runs on a wave of
//   its own between the learned waves (which publish u through `uring`) and the expert wave, which then lags the learned
//   waves by TWO iterations instead of one: every ring gets a third slot, slot(step) = (learned iteration of the step) % RD.
template <int D, bool CW>
struct SplitBwdShared {
  static constexpr int RD = CW ? 3 : 2;   // ring depth
  static constexpr int M = D - 4;
  float yring[RD][4][kSplitPatients][4];        // expert stage states
  // learned block -> expert cotangent; row kSplitPatients stays zero: the expert wave's 16 spare lanes read it, so that
  // their (masked) cotangent stays exactly zero without a per-stage multiply
  float cring[RD][4][kSplitPatients + 1][4];
  // Dose(t_s) [0..3] and dDose/dkel [4..7] of the 4 stages; written by the expert wave when it re-integrates, by the
  // learned waves (stage q by quad lane q) when the stage states come from the tape
  float dring[RD][kSplitPatients][8];
  float gring[RD][4][kSplitPatients + 1][4];    // THW: stage cotangents for the theta wave; row kSplitPatients: zeros
  float tdring[RD][kSplitPatients][8];          // THW: the doses of the step the expert wave has just adjoined
  float uring[CW ? RD : 1][4][CW ? kSplitPatients : 1][CW ? M : 1];   // CW: u = g (1 - s^2) of the learned rows, per stage
  float red[4][4][M * D + M];                   // epilogue: per-row partial sums of every gradient entry of a wave
};

template <int D, int METHOD, bool ABLATE, bool HILL2, bool NEED_TH, bool K1, bool TAPE>
HODE_DEV void split_bwd_body(const SplitBwdArgs& a, SplitBwdShared<D, TAPE>& sh) {
  constexpr int NS = sp_stages<METHOD>();
  typedef MlRows<D> Ml;
  typedef typename Ml::Own Own;
  constexpr int M = Ml::M, MR = Ml::MR, DP = Ml::DP, MP = Ml::MP;
  // THW: the 13 (15) theta-gradient accumulations of every stage run on a FIFTH wave, one iteration behind the expert wave
  // and off its cotangent chain (they are 40 % of its instructions, two v_log per stage among them): the expert wave hands
  // over the stage cotangents g_s through gring, the theta wave reads the stage states from the forward's tape and gets
  // the doses forwarded (tdring).  Only with the tape (without it the stage states exist nowhere but in the expert wave).
  constexpr bool THW = TAPE && NEED_TH;
  // CW: see SplitBwdShared.  The learned waves' loop loses the M fmas + the LDS write of c_q per stage (-32 of ~350
  // instructions per step; they bound the adjoint) and gains one write of u.
  constexpr bool CW = TAPE;
  constexpr int RD = SplitBwdShared<D, TAPE>::RD;
  constexpr int EL = CW ? 2 : 1;           // iterations the expert wave lags the learned waves
  constexpr int CWAVE = THW ? 5 : 4;       // wave index of the c-wave
  auto& yring = sh.yring; auto& cring = sh.cring; auto& dring = sh.dring; auto& gring = sh.gring;
  auto& tdring = sh.tdring; auto& uring = sh.uring; auto& red = sh.red;
  extern __shared__ float tg[];  // time grid, see sp_stage_grid
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int b0 = blockIdx.x * kSplitPatients;
  const RocheTheta th = load_theta(a.theta, ABLATE);
  const size_t row = (size_t)a.B * D;
  const int T = a.T;
  // Iterations: the learned waves handle step T-2-k in iteration k (k <= T-2), the expert wave adjoins step T-k+EL-2 in
  // iteration k (EL <= k <= T+EL-2).  All waves run the same NIT iterations (one barrier each).
  const int NIT = T + EL - 1;
  sp_stage_grid(tg, a.t, a.T);  // visible after the __syncthreads that precedes both pipelines' loops
#ifdef HODE_SPLIT_STAMPS
#define HODE_SSTAMP(k) if (a.dbg && blockIdx.x == 0 && lane == 0) a.dbg[(size_t)wave * (a.T + 2) + (k)] = __builtin_amdgcn_s_memtime();
// phase stamps inside an iteration of learned wave 3 (alone on its SIMD): [T + 2][8] behind the barrier stamps and the HW_ID block
#define HODE_PSTAMP(k, ph) { __builtin_amdgcn_sched_barrier(0); if (a.dbg && blockIdx.x == 0 && wave == 3 && lane == 0) a.dbg[(size_t)6 * (a.T + 2) + 64 + (size_t)(k) * 8 + (ph)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
  if (a.dbg && blockIdx.x < 8 && lane == 0) a.dbg[(size_t)6 * (a.T + 2) + blockIdx.x * 8 + wave] = __builtin_amdgcn_s_getreg(63492);  // HW_ID: SIMD in bits 5:4
#else
#define HODE_SSTAMP(k)
#define HODE_PSTAMP(k, ph)
#endif
  // the time loops are unrolled by RD through a generic lambda: S = k % RD is a compile-time constant, so every ring offset
  // is an immediate
  auto run = [&](auto&& iter) {
    for (int k = 0; k < NIT; k += RD) {
      iter(k, IC<0>{});
      if (k + 1 < NIT) iter(k + 1, IC<1>{});
      if constexpr (RD == 3) {
        if (k + 2 < NIT) iter(k + 2, IC<2>{});
      }
    }
  };

  if (wave == 0) {
    // ================================================================== expert wave
    const int slot = lane < kSplitPatients ? lane : 0;
    const bool mine = lane < kSplitPatients;
    const bool live = mine && b0 + slot < a.B;
    const int p = min(b0 + slot, a.B - 1);
    const float lv = live ? 1.0f : 0.0f;
    DoseSched<K1> ds;
    ds.dosage = a.dosage[p];
    ds.K = a.K;
    ds.taus = a.dose_times + (size_t)p * a.K;
    ds.tau0 = K1 ? ds.taus[0] : 0.f;
    const float ln_ec50 = log_f32(th.ec50);
    const int cslot = mine ? lane : kSplitPatients;
    if (lane < 16 * RD) (&cring[lane >> 4][(lane >> 2) & 3][kSplitPatients][0])[lane & 3] = 0.f;
    const unsigned lane_h = (unsigned)p * D, lane_t = (unsigned)p * 4;  // 32-bit lane offsets on wave-uniform bases
    MlSlice<4, 1> none;
    MlColSlice<4, 1> nonec;
    GradAcc<4, 1> acc;
    acc.zero();
    float own1[1] = {0.f};

    // (a): stage states + doses of step m into ring slot `par`.  Without a tape the stages are re-integrated; with
    // one, fetch() issues the loads an iteration's worth of work ahead of publish().
    float tp[4][4];  // plain floats: an array of float4 is not promoted to registers
    auto fetch = [&](int m) {
      const float4 v = *reinterpret_cast<const float4*>(a.h + (size_t)m * row + lane_h);
      tp[0][0] = v.x; tp[0][1] = v.y; tp[0][2] = v.z; tp[0][3] = v.w;
      if constexpr (TAPE) {
        const float* __restrict__ tape_m = a.tape + (size_t)m * (NS - 1) * a.B * 4;
#pragma unroll
        for (int s = 1; s < NS; ++s) {
          const float4 u = *reinterpret_cast<const float4*>(tape_m + ((unsigned)(s - 1) * (unsigned)a.B * 4u + lane_t));
          tp[s][0] = u.x; tp[s][1] = u.y; tp[s][2] = u.z; tp[s][3] = u.w;
        }
      }
    };
    auto publish = [&](int m, int par) {
      if constexpr (TAPE) {
        if (mine) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
            *reinterpret_cast<float4*>(&yring[par][s][slot][0]) = make_float4(tp[s][0], tp[s][1], tp[s][2], tp[s][3]);
        }
      } else {
        const SpTimes st(a.t, m, a.perturb, METHOD);
        const f2 ya = pair2(tp[0][0], tp[0][1]), yb = pair2(tp[0][2], tp[0][3]);
        f2 ka[4], kb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ka[s] = kb[s] = splat2(0.f);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const f2 Ya = sp_stage_state<METHOD>(s, ya, st.dt, ka[0], ka[1], ka[2]);
          const f2 Yb = sp_stage_state<METHOD>(s, yb, st.dt, kb[0], kb[1], kb[2]);
          const DoseVal dv = ds.at(st.ts[s], th.kel);
          if (mine) {
            *reinterpret_cast<float4*>(&yring[par][s][slot][0]) = make_float4(Ya.x, Ya.y, Yb.x, Yb.y);
            dring[par][slot][s] = dv.v;
            dring[par][slot][4 + s] = dv.dk;
          }
          if (s + 1 < NS) {
            const float Y[4] = {Ya.x, Ya.y, Yb.x, Yb.y};
            float k[4];
            roche_rhs<4, 1, ABLATE, HILL2>(th, none, dv.v, Y, k, own1);
            ka[s] = pair2(k[0], k[1]);
            kb[s] = pair2(k[2], k[3]);
          }
        }
      }
    };

    F4 lam;
    {
      const float4 g4 = *reinterpret_cast<const float4*>(a.grad_h + (size_t)(T - 1) * row + (size_t)p * D);
      lam.a = lv * pair2(g4.x, g4.y);
      lam.b = lv * pair2(g4.z, g4.w);
    }
    if (T >= 2) {
      fetch(T - 2);
      publish(T - 2, 0);
    }
    __syncthreads();
    auto e_iter = [&](int k, auto SI) {
      constexpr int S = decltype(SI)::value;         // == k % RD
      constexpr int par = (S + 1) % RD;              // slot of the step (b) adjoins == the slot (a) refills for iteration k + 1
      if (T - 3 - k >= 0) fetch(T - 3 - k);
      if (k >= EL && k <= T + EL - 2) {
        // ---- (b) adjoint of step m = T-k+EL-2 (the learned waves handled it in iteration k-EL: slot (k-EL) % RD == par)
        const int m = T - k + EL - 2;
        const float dt = tg[m + 1] - tg[m];
        float Y[4][4];
        F4 cs[4];
        DoseVal dv[4];
        {
          const float4 d0 = *reinterpret_cast<const float4*>(&dring[par][slot][0]);
          const float4 d1 = *reinterpret_cast<const float4*>(&dring[par][slot][4]);
          dv[0].v = d0.x; dv[1].v = d0.y; dv[2].v = d0.z; dv[3].v = d0.w;
          dv[0].dk = d1.x; dv[1].dk = d1.y; dv[2].dk = d1.z; dv[3].dk = d1.w;
          if constexpr (THW) {
            *reinterpret_cast<float4*>(&tdring[S][slot][0]) = d0;
            *reinterpret_cast<float4*>(&tdring[S][slot][4]) = d1;
          }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float4 yv = *reinterpret_cast<const float4*>(&yring[par][s][slot][0]);
          const float4 cv = *reinterpret_cast<const float4*>(&cring[par][s][cslot][0]);
          Y[s][0] = yv.x; Y[s][1] = yv.y; Y[s][2] = yv.z; Y[s][3] = yv.w;
          // no masking: a patient beyond the batch has a zero cotangent in the learned waves too (c_s is linear in it)
          cs[s].a = pair2(cv.x, cv.y);
          cs[s].b = pair2(cv.z, cv.w);
        }
        sp_adjoint_step<METHOD>(lam, dt, [&](int s, const F4& gs) {
          const float g[4] = {gs.a.x, gs.a.y, gs.b.x, gs.b.y};
          float av[4];
          if constexpr (THW) {
            if (mine) *reinterpret_cast<float4*>(&gring[S][s][slot][0]) = make_float4(g[0], g[1], g[2], g[3]);
          }
          roche_vjp<4, 1, ABLATE, HILL2, NEED_TH && !THW>(th, none, nonec, ln_ec50, dv[s], Y[s], own1, g, 0, av, acc);
          F4 r = cs[s];
          r.a += pair2(av[0], av[1]);
          r.b += pair2(av[2], av[3]);
          return r;
        });
        const float4 g4 = *reinterpret_cast<const float4*>(a.grad_h + (size_t)m * row + lane_h);
        lam.a = vfma(lv, pair2(g4.x, g4.y), lam.a);
        lam.b = vfma(lv, pair2(g4.z, g4.w), lam.b);
      }
      // ---- (a) stage states of step T-3-k for the learned waves' next iteration
      if (T - 3 - k >= 0) publish(T - 3 - k, par);
      HODE_SSTAMP(k)
      __syncthreads();
    };
    run(e_iter);
    if (live) *reinterpret_cast<float4*>(a.grad_y0 + (size_t)p * D) = make_float4(lam.a.x, lam.a.y, lam.b.x, lam.b.y);
    // theta partials: 16-lane row sums by DPP rotations, the 4 rows joined through LDS (in-order within the wave)
    if constexpr (!THW) {
#pragma unroll
      for (int i = 0; i < kNTheta; ++i) {
        const float v = row_sum(NEED_TH ? acc.dth[i] : 0.f);
        if ((lane & 15) == 0) red[0][lane >> 4][i] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (lane < kNTheta)
        a.part_th[(size_t)blockIdx.x * kNTheta + lane] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    }
  } else if (THW && wave == 4) {
    // ================================================================== theta wave (one patient per lane, like the expert wave)
    const int slot = lane < kSplitPatients ? lane : 0;
    const bool mine = lane < kSplitPatients;
    const int p = min(b0 + slot, a.B - 1);
    const int gslot = mine ? lane : kSplitPatients;   // spare lanes read the zero row: their contributions are exactly zero
    const float ln_ec50 = log_f32(th.ec50);
    if (lane < 16 * RD) (&gring[lane >> 4][(lane >> 2) & 3][kSplitPatients][0])[lane & 3] = 0.f;
    const unsigned lane_h = (unsigned)p * D, lane_t = (unsigned)p * 4;
    float dth[kNTheta];
#pragma unroll
    for (int i = 0; i < kNTheta; ++i) dth[i] = 0.f;
    float tp[4][4];  // stage states of the step to be processed next (from h and the forward's tape)
    auto fetch = [&](int m) {
      const float4 v = *reinterpret_cast<const float4*>(a.h + (size_t)m * row + lane_h);
      tp[0][0] = v.x; tp[0][1] = v.y; tp[0][2] = v.z; tp[0][3] = v.w;
      const float* __restrict__ tape_m = a.tape + (size_t)m * (NS - 1) * a.B * 4;
#pragma unroll
      for (int s = 1; s < NS; ++s) {
        const float4 u = *reinterpret_cast<const float4*>(tape_m + ((unsigned)(s - 1) * (unsigned)a.B * 4u + lane_t));
        tp[s][0] = u.x; tp[s][1] = u.y; tp[s][2] = u.z; tp[s][3] = u.w;
      }
    };
    // the theta terms of a step; stages in the order the expert wave's adjoint visits them (3, 2, 1, 0), so that every
    // accumulator sees its terms in the same order as when the expert wave accumulates them itself
    auto theta_step = [&](int gpar) {
      const float4 d0 = *reinterpret_cast<const float4*>(&tdring[gpar][slot][0]);
      const float4 d1 = *reinterpret_cast<const float4*>(&tdring[gpar][slot][4]);
      const float dvv[4] = {d0.x, d0.y, d0.z, d0.w}, dvk[4] = {d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int si = 0; si < NS; ++si) {
        const int s = NS - 1 - si;
        const float4 gv = *reinterpret_cast<const float4*>(&gring[gpar][s][gslot][0]);
        DoseVal dv;
        dv.v = dvv[s]; dv.dk = dvk[s];
        const float g[4] = {gv.x, gv.y, gv.z, gv.w};
        roche_theta_grad<ABLATE, HILL2>(th, ln_ec50, dv, tp[s], g, dth);
      }
    };
    __syncthreads();  // the expert wave's prologue
    auto t_iter = [&](int k, auto SI) {
      constexpr int S = decltype(SI)::value;
      constexpr int gpar = (S + RD - 1) % RD;   // what the expert wave wrote in iteration k - 1 (its step T-k+EL-1)
      if (k >= EL + 1) theta_step(gpar);
      // the stage states of the step the expert wave adjoins in THIS iteration (processed here in the next one)
      if (k >= EL && T - k + EL - 2 >= 0) fetch(T - k + EL - 2);
      HODE_SSTAMP(k)
      __syncthreads();
    };
    run(t_iter);
    if (T >= 2) theta_step((NIT - 1) % RD);
#pragma unroll
    for (int i = 0; i < kNTheta; ++i) {
      const float v = row_sum(dth[i]);
      if ((lane & 15) == 0) red[0][lane >> 4][i] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < kNTheta)
      a.part_th[(size_t)blockIdx.x * kNTheta + lane] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
  } else if (CW && wave == CWAVE) {
    // ================================================================== c-wave (one patient per lane)
    // c_q = sum_j W[j][q] u_j for the four expert components q, per stage: the same fma chain in the same order as the
    // learned waves formed it before (MR == 2: one chain over j; MR == 1: even and odd j in two chains joined by one add),
    // so every output stays bit-identical to the tape-less kernel.
    const int slot = lane < kSplitPatients ? lane : 0;
    const bool mine = lane < kSplitPatients;
    float wq[M][4];   // W[j][q]: wave-uniform
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) wq[j][q] = a.w1[j * D + q];
    __syncthreads();  // the expert wave's prologue
    auto c_iter = [&](int k, auto SI) {
      constexpr int S = decltype(SI)::value;
      constexpr int cpar = (S + RD - 1) % RD;   // the step the learned waves handled in iteration k - 1
      if (k >= 1 && k - 1 <= T - 2) {
        float u[NS][M];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int j4 = 0; j4 < M; j4 += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&uring[cpar][s][slot][j4]);
            u[s][j4] = v.x; u[s][j4 + 1] = v.y; u[s][j4 + 2] = v.z; u[s][j4 + 3] = v.w;
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // all ring reads of the step up front
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          float c[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if constexpr (MR == 2) {
              float cq = 0.f;
#pragma unroll
              for (int j = 0; j < M; ++j) cq = __builtin_fmaf(wq[j][q], u[s][j], cq);
              c[q] = cq;
            } else {
              float cx = 0.f, cy = 0.f;
#pragma unroll
              for (int jp = 0; jp < MP; ++jp) {
                cx = __builtin_fmaf(wq[2 * jp][q], u[s][2 * jp], cx);
                cy = __builtin_fmaf(wq[2 * jp + 1][q], u[s][2 * jp + 1], cy);
              }
              c[q] = cx + cy;
            }
          }
          if (mine) *reinterpret_cast<float4*>(&cring[cpar][s][slot][0]) = make_float4(c[0], c[1], c[2], c[3]);
        }
      }
      HODE_SSTAMP(k)
      __syncthreads();
    };
    run(c_iter);
  } else {
    // ================================================================== learned waves (quad layout, own components)
    const int q = lane & 3;
    const int slot = (wave - 1) * 16 + (lane >> 2);
    const bool live = b0 + slot < a.B;
    const int p = min(b0 + slot, a.B - 1);
    const float lv = live ? 1.0f : 0.0f;
    Ml ml;
    ml.load(a.w1, a.b1, q);
    // transposed operands of a = W^T u.  wc: column q of W (the expert component this lane reports to the expert wave;
    // unused with the c-wave).
    // MR == 2: wc[j] scalars, wtp[j] = (W[j][own0], W[j][own1]) (row-paired, the result is the Own pair);
    // MR == 1: wc[jp] = (W[2jp][q], W[2jp+1][q]), wtp[jp] = (W[2jp][own], W[2jp+1][own]) (paired over the rows j).
    typename Ml::Elem wc[Ml::NU];
    f2 wtp[Ml::NU];
    if constexpr (MR == 2) {
#pragma unroll
      for (int j = 0; j < M; ++j) {
        wc[j] = a.w1[j * D + q];
        wtp[j] = pair2(a.w1[j * D + 4 + 2 * q], a.w1[j * D + 4 + 2 * q + 1]);
      }
    } else {
#pragma unroll
      for (int jp = 0; jp < MP; ++jp) {
        wc[jp] = pair2(a.w1[(2 * jp) * D + q], a.w1[(2 * jp + 1) * D + q]);
        wtp[jp] = pair2(a.w1[(2 * jp) * D + 4 + q], a.w1[(2 * jp + 1) * D + 4 + q]);
      }
    }
    f2 dw[Ml::NW];
    Own db = vsplat<Own>(0.f);
#pragma unroll
    for (int i = 0; i < Ml::NW; ++i) dw[i] = splat2(0.f);
    DoseSched<K1> ds;
    ds.dosage = a.dosage[p];
    ds.K = a.K;
    ds.taus = a.dose_times + (size_t)p * a.K;
    ds.tau0 = K1 ? ds.taus[0] : 0.f;
    Own lam = lv * Ml::load_own(a.grad_h + (size_t)(T - 1) * row + (size_t)p * D, q);
    // the operands of iteration k+1 are fetched during iteration k (the state is needed by the very first instruction of
    // an iteration: an un-hidden HBM round trip would cost a quarter of it)
    Own yo_nx, gh_nx;
    constexpr int NLT = TAPE ? sp_ltape_stages<METHOD>() : 0;
    const size_t lt_step = (size_t)a.B * 4 * NLT * MR;
    const unsigned lt_lane = ((unsigned)p * 4 + q) * NLT * MR;
    Own lt_nx[NLT > 0 ? NLT : 1], lt_n2[NLT > 0 ? NLT : 1];   // the tape records of the next two iterations
    {
      const int m0 = T >= 2 ? T - 2 : 0;
      const int m1 = m0 >= 1 ? m0 - 1 : 0;
      yo_nx = Ml::load_own(a.h + (size_t)m0 * row + (size_t)p * D, q);
      gh_nx = Ml::load_own(a.grad_h + (size_t)m0 * row + (size_t)p * D, q);
      if constexpr (NLT > 0) {
        Ml::template load_tape<NLT>(a.ltape + (size_t)m0 * lt_step + lt_lane, lt_nx);
        Ml::template load_tape<NLT>(a.ltape + (size_t)m1 * lt_step + lt_lane, lt_n2);
      }
    }
    __syncthreads();  // the expert wave's prologue fills ring 0
    auto m_iter = [&](int k, auto SI) {
      constexpr int par = decltype(SI)::value;  // == k % RD
      if (k <= T - 2) {
        HODE_PSTAMP(k, 0)
        const int m = T - 2 - k;
        const float t0 = tg[m], t1 = tg[m + 1];
        const float dt = t1 - t0;
        const Own yo = yo_nx, gh = gh_nx;
        Own lt[NLT > 0 ? NLT : 1];
#pragma unroll
        for (int i = 0; i < NLT; ++i) { lt[i] = lt_nx[i]; lt_nx[i] = lt_n2[i]; }
        {
          const int mn = m >= 1 ? m - 1 : 0;  // clamped: the last prefetches are simply unused
          yo_nx = Ml::load_own(a.h + (size_t)mn * row + (size_t)p * D, q);
          gh_nx = Ml::load_own(a.grad_h + (size_t)mn * row + (size_t)p * D, q);
          // the tape record TWO iterations ahead: one iteration (~0.7 us) does not cover its HBM round trip (with one the
          // tape bought nothing: 89.7 us with and without it; with two 80.2).  h / grad_h stay at one: two measured 1 % slower
          if constexpr (NLT > 0) Ml::template load_tape<NLT>(a.ltape + (size_t)(m >= 2 ? m - 2 : 0) * lt_step + lt_lane, lt_n2);
        }
        // ---- recompute the learned stage derivatives
        typename Ml::Stage Y[4];
        Own so[4];
        float4 e[4];  // all four expert stage states up front: one LDS round trip per step instead of one per stage
#pragma unroll
        for (int s = 0; s < NS; ++s) e[s] = *reinterpret_cast<const float4*>(&yring[par][s][slot][0]);
        __builtin_amdgcn_sched_barrier(0);  // keep the four reads here: the scheduler would sink each next to its use
        HODE_PSTAMP(k, 1)
        if constexpr (TAPE) {  // the expert wave adjoins this step EL iterations later: its doses, stage q by quad lane q
          const DoseVal dq = ds.at(sp_stage_time<METHOD>(t0, t1, a.perturb, q), th.kel);  // lanes q >= NS: unread
          dring[par][slot][q] = dq.v;
          if constexpr (NEED_TH) dring[par][slot][4 + q] = dq.dk;
        }
#if HODE_SPLIT_EXPERT_FIRST
        typename Ml::Part pe[4];
#pragma unroll
        for (int s = 0; s < NS - NLT; ++s) pe[s] = ml.rhs_expert(e[s]);
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s) so[s] = vsplat<Own>(0.f);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const Own Yo = sp_stage_state<METHOD>(s, yo, dt, so[0], so[1], so[2]);
          Y[s] = Ml::stage(e[s], Yo);
          if (s >= NS - NLT) so[s] = lt[s - (NS - NLT) < 0 ? 0 : s - (NS - NLT)];  // what the forward computed, bit for bit
#if HODE_SPLIT_EXPERT_FIRST
          else so[s] = ml.rhs_rest(pe[s], Y[s]);
#else
          else so[s] = ml.rhs(Y[s]);
#endif
        }
        // ---- adjoint of the stages
        HODE_PSTAMP(k, 2)
        sp_adjoint_step<METHOD>(lam, dt, [&](int s, Own gs) {
          HODE_PSTAMP(k, 6 - s)
          const Own u = gs * vfma(-so[s], so[s], vsplat<Own>(1.0f));
          db += u;
          Ml::outer_acc(dw, u, Y[s]);
          if constexpr (CW) {   // the c-wave forms c_q from it in the next iteration
            if constexpr (MR == 2) *reinterpret_cast<float2*>(&uring[par][s][slot][2 * q]) = make_float2(u.x, u.y);
            else uring[par][s][slot][q] = u;
          }
          typename Ml::Gath uf;
          Ml::gather(u, uf.v);
          if constexpr (MR == 2) {
            float cq = 0.f;
            f2 av = splat2(0.f), av1 = splat2(0.f);  // two chains, see MlRows::rhs
#pragma unroll
            for (int j = 0; j < M; j += 2) {
              if constexpr (!CW) cq = __builtin_fmaf(wc[j], uf.v[j], cq);
              av = vfma(wtp[j], uf.v[j], av);
              if constexpr (!CW) cq = __builtin_fmaf(wc[j + 1], uf.v[j + 1], cq);
              av1 = vfma(wtp[j + 1], uf.v[j + 1], av1);
            }
            if constexpr (!CW) cring[par][s][slot][q] = cq;
            return av + av1;
          } else {
            f2 c2 = splat2(0.f), a2 = splat2(0.f);
#pragma unroll
            for (int jp = 0; jp < MP; ++jp) {
              if constexpr (!CW) c2 = vfma(wc[jp], uf.v[jp], c2);
              a2 = vfma(wtp[jp], uf.v[jp], a2);
            }
            if constexpr (!CW) cring[par][s][slot][q] = hsum(c2);
            return hsum(a2);
          }
        });
        lam = vfma(lv, gh, lam);
        HODE_PSTAMP(k, 7)
      }
      HODE_SSTAMP(k)
      __syncthreads();
    };
    run(m_iter);
    if (live) Ml::store_own(a.grad_y0 + (size_t)p * D, q, lam);
    // weight-gradient partials of this wave's 16 patients: sum over the 4 quads of a 16-lane row with two DPP
    // rotations per entry, join the 4 rows through LDS, store the M*D + M entries with coalesced lanes
    float* out = a.part_ml + ((size_t)blockIdx.x * 3 + (wave - 1)) * (M * D + M);
    const bool writer = (lane & 15) < 4;  // lane & 15 == q there
    float* mine = &red[wave][lane >> 4][0];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const float v = row_sum_stride4(Ml::dw_at(dw, r, i));
        if (writer) mine[(q * MR + r) * D + i] = v;
      }
      float dbr;
      if constexpr (MR == 2) dbr = r == 0 ? db.x : db.y;
      else dbr = db;
      const float vb = row_sum_stride4(dbr);
      if (writer) mine[M * D + q * MR + r] = vb;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int c = lane; c < M * D + M; c += 64)
      out[c] = (red[wave][0][c] + red[wave][1][c]) + (red[wave][2][c] + red[wave][3][c]);
  }
}

// One launch folds both partial arrays of the split backward in a fixed order (deterministic): block j < P_ml sums column j
// of part_ml over its 3*nblk rows into grad_w1 / grad_b1, the remaining kNTheta blocks sum part_th into grad_theta.
// overwrite != 0 stores instead of accumulating (saves the caller's memset of the accumulators).
__global__ __launch_bounds__(64) void split_fold_kernel(const float* __restrict__ part_ml, const float* __restrict__ part_th,
                                                        int nblk, int P_ml, int n_w, float* __restrict__ gw,
                                                        float* __restrict__ gb, float* __restrict__ gth, int need_th,
                                                        int overwrite) {
  const int j = blockIdx.x;
  const int lane = threadIdx.x;
  const bool ml = j < P_ml;
  const float* src = ml ? part_ml + j : part_th + (j - P_ml);
  const int rows = ml ? 3 * nblk : nblk;
  const int stride = ml ? P_ml : kNTheta;
  float s = 0.f;
  for (int base = 0; base < rows; base += 64 * 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int w = base + lane + 64 * q;
      v[q] = w < rows ? src[(size_t)w * stride] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  s = wave_sum(s);
  if (lane != 0) return;
  float* dst = nullptr;
  if (ml) dst = j < n_w ? (gw ? gw + j : nullptr) : (gb ? gb + (j - n_w) : nullptr);
  else if (gth) dst = gth + (j - P_ml);
  if (!dst) return;
  if (!ml && !need_th) s = 0.f;
  *dst = overwrite ? s : *dst + s;
}

// Occupancy bound the compiler is told about (and which it ENFORCES by padding the register count of the kernel descriptor,
// tools/kernel_descriptor.py).  Forward: two workgroups may share a CU (176 registers): nothing changes up to one
// workgroup per CU (10 000 - 12 288 patients: 51-52 us either way) and past it the second workgroup fills the issue slots
// the first leaves empty -- 97.8 -> 77.3 us at 20 000 patients, 651 -> 507 us at 160 000 (tools/scale_probe.py, same-call
// A/B against the pinned build, profiles/r03_v1_scale_probe.txt; four per CU measure the same as two).  Backward: one
// workgroup per CU regardless -- it holds 94 KB of LDS and 216 registers x 5 waves; 2 waves per SIMD because the theta
// wave shares one.
#ifndef HODE_SPLIT_WPE_FWD
#define HODE_SPLIT_WPE_FWD 2
#endif
#ifndef HODE_SPLIT_WPE_BWD
#define HODE_SPLIT_WPE_BWD 2
#endif
#ifndef HODE_SPLIT_WPE_BWD_MIN
#define HODE_SPLIT_WPE_BWD_MIN 1   // experiment builds: 3 forces <= 168 registers so that two backward workgroups fit a CU
#endif

// 4 waves (expert + 3 learned); with the tape + the c-wave, and with theta gradients + the theta wave: up to 6
template <int D, int METHOD, bool ABLATE, bool NEED_TH, bool TAPE>
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(HODE_SPLIT_WPE_BWD_MIN, HODE_SPLIT_WPE_BWD))) void split_bwd_kernel(SplitBwdArgs a) {
  __shared__ __attribute__((aligned(16))) SplitBwdShared<D, TAPE> sh;
  const bool hill2 = ABLATE || (a.theta[0] == 2.0f && a.theta[1] == 2.0f);
  if (hill2 && a.K == 1) split_bwd_body<D, METHOD, ABLATE, true, NEED_TH, true, TAPE>(a, sh);
  else if (hill2) split_bwd_body<D, METHOD, ABLATE, true, NEED_TH, false, TAPE>(a, sh);
  else split_bwd_body<D, METHOD, ABLATE, false, NEED_TH, false, TAPE>(a, sh);
}

template <int D, int METHOD, bool ABLATE, bool TAPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, HODE_SPLIT_WPE_FWD))) void split_fwd_kernel(SplitArgs a) {
  const bool hill2 = ABLATE || (a.theta[0] == 2.0f && a.theta[1] == 2.0f);
  if (hill2 && a.K == 1) split_fwd_body<D, METHOD, ABLATE, true, true, TAPE>(a);
  else if (hill2) split_fwd_body<D, METHOD, ABLATE, true, false, TAPE>(a);
  else split_fwd_body<D, METHOD, ABLATE, false, false, TAPE>(a);
}

}  // namespace hode

namespace {

template <int D, bool ABLATE>
int split_method(const hode_solve_desc* d, const hode::SplitArgs& a, hipStream_t s) {
  const dim3 grid((d->batch + hode::kSplitPatients - 1) / hode::kSplitPatients), block(256);
  const size_t lds = (size_t)(d->n_times + 3) * sizeof(float);  // the time grid (n_times <= kSplitMaxT)
#define HODE_SPLIT_FWD(M)                                                                                 \
  if (a.tape) hipLaunchKernelGGL((hode::split_fwd_kernel<D, M, ABLATE, true>), grid, block, lds, s, a);   \
  else hipLaunchKernelGGL((hode::split_fwd_kernel<D, M, ABLATE, false>), grid, block, lds, s, a);
  switch (d->method) {
    case HODE_METHOD_EULER: HODE_SPLIT_FWD(HODE_METHOD_EULER) break;
    case HODE_METHOD_MIDPOINT: HODE_SPLIT_FWD(HODE_METHOD_MIDPOINT) break;
    default: HODE_SPLIT_FWD(HODE_METHOD_RK4_38) break;
  }
  return hode::hip_fail(hipGetLastError(), "split kernel launch");
}

}  // namespace

namespace {

template <int D, bool ABLATE>
int split_bwd_method(const hode_solve_desc* d, const hode::SplitBwdArgs& a, hipStream_t s) {
  const dim3 grid((d->batch + hode::kSplitPatients - 1) / hode::kSplitPatients), block(256);
  const size_t lds = (size_t)(d->n_times + 3) * sizeof(float);  // the time grid (n_times <= kSplitMaxT)
#define HODE_SPLIT_BWD(M)                                                                                        \
  if (d->need_theta_grad) {                                                                                      \
    if (a.tape) hipLaunchKernelGGL((hode::split_bwd_kernel<D, M, ABLATE, true, true>), grid, dim3(384), lds, s, a); \
    else hipLaunchKernelGGL((hode::split_bwd_kernel<D, M, ABLATE, true, false>), grid, block, lds, s, a);          \
  } else {                                                                                                       \
    if (a.tape) hipLaunchKernelGGL((hode::split_bwd_kernel<D, M, ABLATE, false, true>), grid, dim3(320), lds, s, a); \
    else hipLaunchKernelGGL((hode::split_bwd_kernel<D, M, ABLATE, false, false>), grid, block, lds, s, a);         \
  }
  switch (d->method) {
    case HODE_METHOD_EULER: HODE_SPLIT_BWD(HODE_METHOD_EULER) break;
    case HODE_METHOD_MIDPOINT: HODE_SPLIT_BWD(HODE_METHOD_MIDPOINT) break;
    default: HODE_SPLIT_BWD(HODE_METHOD_RK4_38) break;
  }
  return hode::hip_fail(hipGetLastError(), "split bwd kernel launch");
}

}  // namespace

namespace hode {

// the grid lives in LDS: longer grids fall back to the quad layout
bool split_supported(const hode_solve_desc* d) { return (d->latent_dim == 8 || d->latent_dim == 12) && d->n_times <= kSplitMaxT; }

static size_t al256s(size_t x) { return (x + 255) / 256 * 256; }

static int n_stages(int method) { return method == HODE_METHOD_EULER ? 1 : (method == HODE_METHOD_MIDPOINT ? 2 : 4); }

// [ learned-block partials | theta partials | tape (HODE_FLAG_TAPE) ]; the same buffer serves hode_rk_fwd and hode_rk_bwd
static size_t split_partials_bytes(const hode_solve_desc* d) {
  const size_t nblk = (d->batch + kSplitPatients - 1) / kSplitPatients;
  const size_t M = d->latent_dim - 4;
  return al256s(nblk * 3 * (M * d->latent_dim + M) * sizeof(float)) + al256s(nblk * kNTheta * sizeof(float));
}
static size_t split_etape_bytes(const hode_solve_desc* d) {   // expert stage states
  if (!(d->flags & HODE_FLAG_TAPE) || d->n_times < 2) return 0;
  return al256s((size_t)(d->n_times - 1) * (n_stages(d->method) - 1) * d->batch * 4 * sizeof(float));
}
static size_t split_ltape_bytes(const hode_solve_desc* d) {   // learned stage derivatives (rk4 only)
  if (!(d->flags & HODE_FLAG_TAPE) || d->n_times < 2 || n_stages(d->method) != 4 || kLearnedTapeStages == 0) return 0;
  return al256s((size_t)(d->n_times - 1) * d->batch * (d->latent_dim - 4) * kLearnedTapeStages * sizeof(float));
}
static size_t split_tape_bytes(const hode_solve_desc* d) { return split_etape_bytes(d) + split_ltape_bytes(d); }
size_t split_workspace_bytes(const hode_solve_desc* d) { return split_partials_bytes(d) + split_tape_bytes(d); }
static float* split_tape(const hode_solve_desc* d) {
  return split_etape_bytes(d) ? (float*)((char*)d->workspace + split_partials_bytes(d)) : nullptr;
}
static float* split_ltape(const hode_solve_desc* d) {
  return split_ltape_bytes(d) ? (float*)((char*)d->workspace + split_partials_bytes(d) + split_etape_bytes(d)) : nullptr;
}

int split_rk_bwd(const hode_solve_desc* d, hipStream_t s) {
  const int nblk = (d->batch + kSplitPatients - 1) / kSplitPatients;
  const int M = d->latent_dim - 4;
  const int Pml = M * d->latent_dim + M;
  char* ws = (char*)d->workspace;
  SplitBwdArgs a{};
  a.t = d->t; a.dosage = d->dosage; a.dose_times = d->dose_times; a.theta = d->theta; a.w1 = d->w1; a.b1 = d->b1;
  a.h = d->h; a.grad_h = d->grad_h; a.grad_y0 = d->grad_y0;
  a.part_ml = (float*)ws;
  a.part_th = (float*)(ws + al256s((size_t)nblk * 3 * Pml * sizeof(float)));
  a.B = d->batch; a.T = d->n_times; a.K = d->n_dose; a.perturb = d->perturb;
  a.tape = split_tape(d);
  a.ltape = split_ltape(d);
#ifdef HODE_SPLIT_STAMPS
  if (const char* env = getenv("HODE_SPLIT_DBG_PTR")) a.dbg = (unsigned long long*)strtoull(env, nullptr, 0);
#endif
  const bool abl = d->rhs_kind == HODE_RHS_ROCHE_ABLATE;
  int e;
  if (d->latent_dim == 8) e = abl ? split_bwd_method<8, true>(d, a, s) : split_bwd_method<8, false>(d, a, s);
  else e = abl ? split_bwd_method<12, true>(d, a, s) : split_bwd_method<12, false>(d, a, s);
  if (e || (d->flags & HODE_FLAG_SKIP_FOLD)) return e;
  const bool th_out = d->grad_theta != nullptr && (d->need_theta_grad || (d->flags & HODE_FLAG_OVERWRITE_GRADS));
  hipLaunchKernelGGL(split_fold_kernel, dim3(Pml + (th_out ? kNTheta : 0)), dim3(64), 0, s, a.part_ml, a.part_th, nblk, Pml,
                     M * d->latent_dim, d->grad_w1, d->grad_b1, d->grad_theta, d->need_theta_grad,
                     (d->flags & HODE_FLAG_OVERWRITE_GRADS) ? 1 : 0);
  return hip_fail(hipGetLastError(), "split_fold launch");
}

int split_rk_fwd(const hode_solve_desc* d, hipStream_t s) {
  SplitArgs a{};
  a.t = d->t; a.y0 = d->y0; a.dosage = d->dosage; a.dose_times = d->dose_times; a.theta = d->theta; a.w1 = d->w1; a.b1 = d->b1;
  a.h = d->h; a.status = d->status;
  a.B = d->batch; a.T = d->n_times; a.K = d->n_dose; a.perturb = d->perturb;
  a.tape = split_tape(d);
  a.ltape = split_ltape(d);
#ifdef HODE_SPLIT_STAMPS
  if (const char* env = getenv("HODE_SPLIT_FWD_DBG_PTR")) a.dbg = (unsigned long long*)strtoull(env, nullptr, 0);
#endif
  const bool abl = d->rhs_kind == HODE_RHS_ROCHE_ABLATE;
  if (d->latent_dim == 8) return abl ? split_method<8, true>(d, a, s) : split_method<8, false>(d, a, s);
  return abl ? split_method<12, true>(d, a, s) : split_method<12, false>(d, a, s);
}

}  // namespace hode
