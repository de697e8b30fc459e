// Masked single-layer LSTM over an observation window on the matrix cores (fp32 MFMA), gfx950.
//
// Replaces the T single-step nn.LSTM calls of EncoderLSTM.forward (reference model.py:420-422; EncoderLSTMReal
// :226-229) and fuses the cat([x, a]) * cat([mask, 1]) that feeds them (model.py:415-421).  CPU restatement:
// oracle/encoder.py::lstm_cell.  Gate order i, f, g, o; gates = x W_ih^T + b_ih + h W_hh^T + b_hh.
//
// Design.  A workgroup (4 waves) owns a tile of BT = 16*NT patients for the WHOLE window:
// h and c never leave the chip.  Per step the gate pre-activations G^T[4H x BT] = Wcat[4H x K] * act^T[K x BT]
// (K = I + H) are accumulated with v_mfma_f32_16x16x4_f32 -- exact fp32, the matrix pipe's rate for this dtype.
//   * Wcat is the A operand.  Its rows are permuted so that one 16-row MFMA tile = 4 hidden units x 4 gates; in the
//     16x16 accumulator layout (row = 4*(lane>>4) + reg, col = lane&15) a lane then holds all four gates of ONE
//     (unit, patient) pair in its 4 registers: the cell update needs no cross-lane traffic at all.
//     Wave w owns hidden units [w*H/4, (w+1)*H/4); its quarter of Wcat is streamed from L2 every step as
//     pre-packed fragments (one global_load_dwordx4 per lane = 4 k-quads of one tile; 1 KiB per wave-instruction).
//   * act^T (B operand) lives in LDS k-major ([k][patient], leading dimension == 16 mod 32 so the 4x16 fragment
//     read is bank-conflict free), double buffered: x_t*mask_t for the next step is fetched from HBM (coalesced,
//     contiguous tile) while the current step's MFMAs run; h_t is written by the cell update.
// Algorithmic bytes per patient: 2*4*T*obs (x, mask) + 4*T*(I-obs) read, 8H written; with save_tape additionally
// 20*T*H written (activated gates + cell state per step, in the lane order the backward kernel reads them).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"

namespace hode {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LstmArgs {
  const float* __restrict__ x;
  const float* __restrict__ a;
  const float* __restrict__ mask;
  const float* __restrict__ wp;     // packed Wcat fragments [4][KQ4][TPW][64][4]; operand row I + Hp is the bias
  float* __restrict__ h_out;
  float* __restrict__ c_out;
  float* __restrict__ tape;         // [T][nblk][Hp/16 (unit tile = wave * TPW + tile of the wave)][NT][5][64] or nullptr
  int T, B, OBS, AD, I, H, Hp, Kq, KQ4, LD, reverse;
  unsigned long long* dbg;  // HODE_LSTM_STAMPS builds only: [T][8] s_memtime stamps of wave 0 of block 0
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release over ALL address
// spaces, i.e. s_waitcnt vmcnt(0): every wave would sit out the HBM write burst of the step (tape stores in the forward,
// dG / h_prev rows in the BPTT -- 32 MB per step over the chip, written by all workgroups at the same moment) before the
// matrix pipe starts again.  Nothing in these kernels reads back what they store, so the stores may drain under the next
// step's MFMAs.
HODE_DEV void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

HODE_DEV float sigmoid_gate(float x) {
  // 1 / (1 + exp(-x)) on v_exp + v_rcp (<= 2 ulp); exp overflow -> rcp(inf) = 0, underflow -> 1
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// NW waves of TPW 16-row tiles each (NW * TPW * 16 = padded H).  Hp = 160 runs as 8 waves x 5 tiles: two waves per
// SIMD, every register in the 256-entry VGPR file -- with 4 x 10 the 120 accumulators went to the AGPR half and came
// back through v_accvgpr moves every step, the compiler spilled, and nothing covered a non-MFMA instruction.
// VEC4: obs_dim % 4 == 0 (x tile fetched in 16-byte groups).  A compile-time switch: as a run-time branch the flat
// path's per-element divisions are hoisted out of the step loop and pinned ~60 registers in every shipped shape.
template <int NT, int TPW, int NW, bool VEC4>
__global__ __launch_bounds__(64 * NW) void lstm_fwd_kernel(LstmArgs p) {
  constexpr int NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BT = 16 * NT;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;  // w in an SGPR: per-wave bases stay scalar
  const int g = l >> 4, pc = l & 15;
  const int b0 = blockIdx.x * BT;
  const int nvalid = min(BT, p.B - b0);
  // compile-time leading dimension (== p.LD, lstm_geom): LDS offsets of the unrolled tile loops become instruction
  // immediates instead of one hoisted address register per (tile, patient column, buffer)
  constexpr int LD = BT + ((BT % 32 == 0) ? 16 : 0);
  const int Krows = 4 * p.Kq;                  // rows of one activation buffer (zero padded past I + Hp)
  float* act0 = lds;
  float* act1 = lds + (size_t)Krows * LD;

  // zero both activation buffers (h_{-1} = 0, padding rows/columns stay 0 forever); row I + Hp is the bias row: ones
  const int one_row = p.I + p.Hp;
  for (int e = tid; e < 2 * Krows * LD; e += NTHR) {
    const int r = (e / LD) % Krows;
    lds[e] = r == one_row ? 1.f : 0.f;
  }

  float cst[TPW][NT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int c = 0; c < NT; ++c) cst[t][c] = 0.f;

  // x tile staging: the tile of one step is BT*OBS contiguous floats (patients are contiguous in [T][B][OBS]).
  // OBS % 4 == 0 (every shipped shape): a thread owns 16-byte groups (patient b = idx % BT, group idx / BT) -- consecutive
  // lanes are consecutive PATIENTS, so the k-major LDS writes are conflict free (the flat element order wrote a wave's
  // 64 values into 2 banks) and no run-time division is needed; the 16-byte loads walk every cache line four times
  // within the step, which L1 / L2 absorb.  Otherwise: flat element order, one division per element.
  // The loads are UNCONDITIONAL (slots past the tile read element 0 and are zeroed when staged) and x * mask is formed
  // when the tile is staged, not when it is fetched: a guarded load followed by the product is one basic block with a
  // vmcnt(0) per 16-byte group -- five serialised HBM round trips (6 us) in front of every step's first MFMA.
  constexpr int XPT = NW == 8 ? 12 : 20;  // staged floats per thread (XPT * NTHR covers BT*OBS <= 5120; multiple of 4)
  const int n_x = nvalid * p.OBS;
  constexpr bool vec4 = VEC4;
  const int Q4 = p.OBS >> 2;
  const bool has_mask = p.mask != nullptr;
  float xs[XPT], ms[XPT];
  auto fetch_x = [&](int t) {
    const size_t base = ((size_t)t * p.B + b0) * p.OBS;
    const float* xb = p.x + base;
    const float* mb = has_mask ? p.mask + base : xb;   // no mask: the second load repeats the first (L1 hit), never used
    if constexpr (vec4) {
#pragma unroll
      for (int j = 0; j < XPT / 4; ++j) {
        const int idx = tid + NTHR * j;
        const int b = idx % BT, i4 = idx / BT;
        const int off = (i4 < Q4 && b < nvalid) ? b * p.OBS + 4 * i4 : 0;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + off);
        const f32x4 m = *reinterpret_cast<const f32x4*>(mb + off);
        xs[4 * j] = v[0]; xs[4 * j + 1] = v[1]; xs[4 * j + 2] = v[2]; xs[4 * j + 3] = v[3];
        ms[4 * j] = m[0]; ms[4 * j + 1] = m[1]; ms[4 * j + 2] = m[2]; ms[4 * j + 3] = m[3];
      }
    } else {
#pragma unroll
      for (int j = 0; j < XPT; ++j) {
        const int e = tid + NTHR * j;
        const int off = e < n_x ? e : 0;
        xs[j] = xb[off];
        ms[j] = mb[off];
      }
    }
  };
  auto stage_x = [&](float* dst, int t) {
    if constexpr (vec4) {
#pragma unroll
      for (int j = 0; j < XPT / 4; ++j) {
        const int idx = tid + NTHR * j;
        const int b = idx % BT, i4 = idx / BT;
        if (i4 < Q4) {
          const bool live = b < nvalid;  // zeros for patients past the batch
#pragma unroll
          for (int c = 0; c < 4; ++c)
            dst[(4 * i4 + c) * LD + b] = live ? (has_mask ? xs[4 * j + c] * ms[4 * j + c] : xs[4 * j + c]) : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < XPT; ++j) {
        const int e = tid + NTHR * j;
        if (e < n_x) {
          const int b = e / p.OBS, i = e - b * p.OBS;
          dst[i * LD + b] = has_mask ? xs[j] * ms[j] : xs[j];
        }
      }
    }
    // action columns (never masked): AD * nvalid values
    for (int e = tid; e < nvalid * p.AD; e += NTHR) {
      const int b = e / p.AD, i = e - b * p.AD;
      dst[(p.OBS + i) * LD + b] = p.a[((size_t)t * p.B + b0 + b) * p.AD + i];
    }
  };

  const int t_first = p.reverse ? p.T - 1 : 0;
  fetch_x(t_first);
  __syncthreads();  // zero fill done
  stage_x(act0, t_first);
  __syncthreads();

  const f32x4* wbase = reinterpret_cast<const f32x4*>(p.wp) + (size_t)w * p.KQ4 * TPW * 64;  // wave-uniform; lane index added per load
  // weight fragment registers live across steps: the first group of step s+1 is requested as soon as step s's MFMA loop
  // ends, so its L2 round trip runs under the cell update, the x staging and the barrier
  f32x4 wa[TPW], wb[TPW];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) wa[tt] = wbase[tt * 64 + l];

#ifdef HODE_LSTM_STAMPS
#define HODE_FSTAMP(i) if (p.dbg && blockIdx.x == 0 && tid == 0) { __builtin_amdgcn_s_waitcnt(0); p.dbg[(size_t)s * 8 + (i)] = __builtin_amdgcn_s_memtime(); }
// no-wait stamps inside the MFMA section: slot [T + s][16]
#define HODE_GSTAMP(i) if (p.dbg && blockIdx.x == 0 && tid == 0) p.dbg[(size_t)(p.T + s) * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define HODE_FSTAMP(i)
#define HODE_GSTAMP(i)
#endif
  for (int s = 0; s < p.T; ++s) {
    const int t = p.reverse ? p.T - 1 - s : s;
    HODE_FSTAMP(0)
    float* cur = (s & 1) ? act1 : act0;
    float* nxt = (s & 1) ? act0 : act1;
    const bool more = s + 1 < p.T;
    const int t_next = p.reverse ? t - 1 : t + 1;
    if (more) fetch_x(t_next);  // HBM loads in flight under the MFMA loop

    f32x4 acc[TPW][NT];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
      for (int c = 0; c < NT; ++c) acc[tt][c] = f32x4{0.f, 0.f, 0.f, 0.f};  // the bias arrives through the ones row

    // software-pipelined weight fragments: group q+1 is loaded while group q feeds the matrix pipe.  The loop over
    // full groups (4 k-quads each) is branch-free; the partial last group is peeled -- with a conditional per k-quad
    // inside the loop the waitcnt pass falls back to vmcnt(0) at every quad, i.e. it waits for the prefetch it has just
    // issued (one L2 round trip per group, 16 per step).
    // B fragments (one LDS row per patient column) are read one k-quad ahead: read-then-use in front of every 30-MFMA
    // block would expose the LDS latency 61 times per step
    auto load_group0 = [&](f32x4 (&wf)[TPW]) {
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) wf[tt] = wbase[tt * 64 + l];
    };
    // Nothing in the loop copies a register: weight fragments alternate between wa / wb over PAIRS of groups and the B
    // fragments between bf / bn over pairs of k-quads -- with one wave per SIMD nothing else covers an instruction
    // that is not an MFMA, and the 40 + 12 moves, the 10 loads and their addresses cost 500 cycles per group (13 %)
    // when they sat between the MFMA blocks.  The prefetch loads are pinned one in front of every 3 MFMAs of the
    // group's first k-quad, where they issue while the matrix pipe is busy.
    float bf[NT], bn[NT];
    const int last_quad = p.Kq - 1;
    auto read_b = [&](float (&dst)[NT], int quad) {
      const float* rowp = cur + (size_t)(4 * min(quad, last_quad) + g) * LD + pc;
#pragma unroll
      for (int c = 0; c < NT; ++c) dst[c] = rowp[16 * c];
    };
    read_b(bf, 0);
    // one full group: 4 k-quads out of wf, B fragments bf -> bn -> bf -> bn -> bf; wn <- group qn during k-quad 0
    auto group4 = [&](const f32x4 (&wf)[TPW], f32x4 (&wn)[TPW], int q, int qn) {
      const f32x4* wq = wbase + (size_t)qn * TPW * 64;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float (&bc)[NT] = (kk & 1) ? bn : bf;
        float (&bx)[NT] = (kk & 1) ? bf : bn;
        read_b(bx, 4 * q + kk + 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the read HERE: the scheduler would sink it next to its use
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
          if (kk == 0) wn[tt] = wq[tt * 64 + l];
#pragma unroll
          for (int c = 0; c < NT; ++c)
            acc[tt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tt][kk], bc[c], acc[tt][c], 0, 0, 0);
        }
        if (kk == 0) {
#pragma unroll
          for (int tt = 0; tt < TPW; ++tt) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one weight load ...
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);  // ... per NT MFMAs
          }
        }
      }
    };
    // the partial last group (its fragments are zero padded): once per step, moves do not matter here
    auto group_tail = [&](const f32x4 (&wf)[TPW], int q, int n) {
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        if (kk < n) {
          read_b(bn, 4 * q + kk + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int c = 0; c < NT; ++c)
              acc[tt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tt][kk], bf[c], acc[tt][c], 0, 0, 0);
#pragma unroll
          for (int c = 0; c < NT; ++c) bf[c] = bn[c];
        }
      }
    };
    const int n_full = p.Kq >> 2;        // groups with all 4 k-quads
    const int tail = p.Kq & 3;           // k-quads of the last, partial group
    const int n_groups = n_full + (tail ? 1 : 0);
    int q = 0;
    HODE_GSTAMP(10)
    for (; q + 2 <= n_full; q += 2) {
      HODE_GSTAMP(q >> 1)
      group4(wa, wb, q, q + 1);
      group4(wb, wa, q + 1, min(q + 2, n_groups - 1));  // clamped: the last prefetch may be a repeat, never out of bounds
    }
    HODE_GSTAMP(8)
    // wa holds group q; the step's last prefetch goes to group 0 of the next step (weights do not change in the launch)
    if (n_full & 1) {
      group4(wa, wb, q, tail ? n_full : 0);
      if (tail) {
        group_tail(wb, n_full, tail);
        load_group0(wa);
      } else {
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) wa[tt] = wb[tt];
      }
    } else {
      if (tail) group_tail(wa, n_full, tail);
      load_group0(wa);
    }
    HODE_GSTAMP(9)
    HODE_FSTAMP(1)

    // cell update: lane (g, pc) holds gates i,f,g,o of unit u = (w*TPW + tt)*4 + g for patient 16c + pc
    float* nxt_lane = nxt + (p.I + w * TPW * 4 + g) * LD + pc;   // lane base; tile / column offsets are immediates
    float* tp = p.tape ? p.tape + (((size_t)t * gridDim.x + blockIdx.x) * NW + w) * TPW * NT * 5 * 64 : nullptr;  // wave-uniform
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
      const int u = (w * TPW + tt) * 4 + g;
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const float gi = sigmoid_gate(acc[tt][c][0]);
        const float gf = sigmoid_gate(acc[tt][c][1]);
        const float gg = tanh_f32(acc[tt][c][2]);
        const float go = sigmoid_gate(acc[tt][c][3]);
        const float cn = __builtin_fmaf(gf, cst[tt][c], gi * gg);
        const float hn = go * tanh_f32(cn);
        cst[tt][c] = cn;
        nxt_lane[4 * tt * LD + 16 * c] = hn;
        if (tp) {
          float* q5 = tp + (tt * NT + c) * 5 * 64;
          q5[l] = gi; q5[64 + l] = gf; q5[128 + l] = gg; q5[192 + l] = go; q5[256 + l] = cn;
        }
      }
      // one tile's NT patient columns at a time: interleaving all TPW * NT exp / rcp chains costs more registers than
      // the file has next to the accumulators (the other wave of the SIMD covers the latency instead)
      __builtin_amdgcn_sched_barrier(0);
    }
    HODE_FSTAMP(2)
    if (more) stage_x(nxt, t_next);
    HODE_FSTAMP(3)
    lds_barrier();
    HODE_FSTAMP(4)
  }
#undef HODE_FSTAMP
#undef HODE_GSTAMP

  // final state: h_T is what the last step left in its output buffer (each lane reads back its own values)
  const float* fin = (p.T & 1) ? act1 : act0;
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int u = (w * TPW + tt) * 4 + g;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const int b = 16 * c + pc;
      if (u < p.H && b < nvalid) {
        p.h_out[(size_t)(b0 + b) * p.H + u] = fin[(size_t)(p.I + u) * LD + b];
        p.c_out[(size_t)(b0 + b) * p.H + u] = cst[tt][c];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------- backward
// Back-propagation through time of the recurrence.  Per step (walked in the reverse of the forward processing order)
//   dh = carry (+ grad_h_out at the last processed step);  tc = tanh(c_t)
//   d_o = dh tc o(1-o);  dc = carry_c + dh o (1 - tc^2);  d_i = dc g i(1-i);  d_f = dc c_prev f(1-f);  d_g = dc i (1-g^2)
//   carry_c = dc f;  carry_h[u'] = sum_rho W_hh[rho][u'] dG[rho]     <- the only contraction: fp32 MFMA
// The gate cotangents dG leave the kernel as grad_gates[T][B][4H] and the entering hidden state as h_prev[T][B][H]
// (both row-major, written coalesced through an LDS transpose); the weight gradients are then three plain GEMMs
// over K = T*B (dG^T x, dG^T h_prev, column sums), which the host wrapper hands to the BLAS library.
//
// MFMA mapping: out[u' (16 per tile) x patient] += W_hh^T[u' x rho] dG^T[rho x patient].  K (rho = gate rows) is
// split over the waves: wave w contracts over the gate rows of ITS units, i.e. exactly the dG values it has just
// produced.  With rho ordered (tile, gate, unit-in-tile) the B-operand fragment of k-quad (tile, gate) -- lane
// (k = lane>>4, j = lane&15) -- IS the register that lane already holds from the element-wise step: no staging.
// The four partial [H x BT] results are exchanged through LDS slabs and summed in a fixed order.
struct LstmBwdArgs {
  const float* __restrict__ tape;      // forward tape [T][nblk][4][TPW][NT][5][64]
  const float* __restrict__ whp;       // packed W_hh^T fragments [4][TPW(tau)][4 (gate row)][ceil(TPW/4)][64][4 (mt)]
  const float* __restrict__ grad_h_out;  // [B][H]
  float* __restrict__ grad_gates;      // [T][B][4H]
  float* __restrict__ h_prev;          // GEMM operand [T][B][W]: (OBS columns left to the caller: x*mask) | action columns
                                       // (AD) | hidden state entering the step (H) | 1.0 | zero padding to W
  const float* __restrict__ a;         // [T][B][AD] or nullptr
  int T, B, H, Hp, LD, reverse, AD, OBS, W;
  unsigned long long* dbg;  // HODE_LSTM_STAMPS builds only: [T][8] s_memtime stamps of wave 0 of block 0
};


#ifndef HODE_BPTT_BODY_HOOK
#define HODE_BPTT_BODY_HOOK 1
#endif
#ifndef HODE_BPTT_BURST
#define HODE_BPTT_BURST 1   // 1: element-wise burst, then the MFMAs with only loads between them; 0: 1 MFMA : 1-2 VALU interleave (3 % slower, tools/micro/mfma_valu_overlap.hip)
#endif
// FLAT: H == 16 TPW (no padded hidden units): the store phase copies whole dG rows (see there); a template parameter so
// that only one of the two store loops -- and its hoisted addresses -- exists in an instantiation
template <int NT, int TPW, bool FLAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_kernel(LstmBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BT = 16 * NT;
  constexpr int Hp = 16 * TPW;
  constexpr int LDG = 4 * Hp + 4;   // row pitch of the dG transpose tile (pad: 2-way instead of 16-way conflicts)
  constexpr int LDH = Hp + 4;
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  const int g = l >> 4, pc = l & 15;
  const int b0 = blockIdx.x * BT;
  const int nvalid = min(BT, p.B - b0);
  constexpr int LD = BT + ((BT % 32 == 0) ? 16 : 0);   // == p.LD (lstm_geom); compile-time: slab offsets become immediates
  float* dgt = lds;                      // [BT][LDG]   (time-shared with the partial slabs [4][Hp][LD])
  float* slab = lds;
  float* hT = lds + (size_t)BT * LDG;    // [BT][LDH]
  const int unit0 = __builtin_amdgcn_readfirstlane(w) * TPW * 4 + g;   // the lane's unit in tile 0 of its wave
  float* dg_lane = dgt + pc * LDG + unit0;
  float* hT_lane = hT + pc * LDH + unit0;
  float* slab_w = slab + (__builtin_amdgcn_readfirstlane(w) * Hp + 4 * g) * LD + pc;   // partial of wave w, rows 4g.., column pc
  const float* slab_r = slab + unit0 * LD + pc;
  const int H = p.H;

  float carry_h[TPW][NT], carry_c[TPW][NT];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int u = (w * TPW + tt) * 4 + g;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const int b = 16 * c + pc;
      carry_h[tt][c] = (u < H && b < nvalid) ? p.grad_h_out[(size_t)(b0 + b) * H + u] : 0.f;
      carry_c[tt][c] = 0.f;
    }
  }
  // wave-uniform bases (SGPRs) + one 32-bit lane offset: with per-lane 64-bit pointers the compiler hoists one address
  // pair per fragment out of the step loop (100 pairs for TPW = 10) and spills them
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const size_t tape_step = (size_t)gridDim.x * 4 * TPW * NT * 5 * 64;
  const size_t tape_blk = ((size_t)blockIdx.x * 4 + wu) * TPW * NT * 5 * 64;
  constexpr int MG = (TPW + 3) / 4;   // 16-byte weight groups per (tile, gate row)
  const f32x4* whb = reinterpret_cast<const f32x4*>(p.whp) + (size_t)wu * TPW * 4 * MG * 64;

  // Operands of one (step, unit tile): the step's activated gates and cell state and the previous step's cell state
  // and output gate for NT patient columns (TapeOps), and the TPW weight fragments the tile's MFMAs read.
  //
  // Software pipeline over the unit tiles of a step (fully unrolled, register sets alternate by renaming):
  //   body(tt) = [tape loads of tile tt+2] [weight loads of tile tt+1] [element-wise step of tile tt+1] [MFMAs of tile tt]
  // The element-wise step of the NEXT tile and the matrix products of THIS one are independent and sit in one basic
  // block, interleaved one VALU group per MFMA: with one wave per SIMD nothing else fills the matrix pipe while the
  // wave does VALU work (26 us per step for 16 us of MFMA time when each tile ran element-wise -> MFMAs in sequence).
  // A load consumed by the next instruction costs an HBM / L2 round trip per tile, hence the two-tile tape distance.
  struct TapeOps { float gi[NT], gf[NT], gg[NT], go[NT], cn[NT], cp[NT], op[NT]; };
  // All tape and weight loads are BUFFER loads: 4-SGPR resource (rebuilt per step with scalar instructions) + one lane
  // offset register + scalar / immediate offsets.  With global loads the compiler kept one 64-bit VGPR address per 4 KiB
  // window (19 pairs) and, behind the pointer laundering that stops it hoisting 100 fragment addresses, fell back to
  // FLAT loads for the weights (they count on lgkmcnt as well as vmcnt: a wait for an LDS write became a wait for L2).
  auto rsrc_of = [](const void* ptr) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, 0x7fffffff, 0x00020000);  // raw, dword format
  };
  auto ldf = [&](__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, l * 4, byte_off, 0));
  };
  auto step_ptrs = [&](int s, const float*& tc, const float*& tpv) {
    const int t = p.reverse ? p.T - 1 - s : s;
    const int t_prev = p.reverse ? t + 1 : t - 1;  // time index processed one step earlier in the forward sweep
    tc = p.tape + (size_t)t * tape_step + tape_blk;
    tpv = (s > 0) ? p.tape + (size_t)t_prev * tape_step + tape_blk : tc;  // s == 0: dummy reads, masked by has_prev
  };
  auto load_tape_col = [&](TapeOps& o, __amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t rp, int tt, int c) {
    const int q5 = (tt * NT + c) * 5 * 64 * 4;   // byte offset of the (tile, column) record: [gi | gf | gg | go | c][64]
    o.gi[c] = ldf(rc, q5); o.gf[c] = ldf(rc, q5 + 256); o.gg[c] = ldf(rc, q5 + 512); o.go[c] = ldf(rc, q5 + 768);
    o.cn[c] = ldf(rc, q5 + 1024);
    o.cp[c] = ldf(rp, q5 + 1024); o.op[c] = ldf(rp, q5 + 768);
  };
  auto load_tape = [&](TapeOps& o, __amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t rp, int tt) {
#pragma unroll
    for (int c = 0; c < NT; ++c) load_tape_col(o, rc, rp, tt, c);
  };
  const __amdgpu_buffer_rsrc_t wrs = rsrc_of(whb);
  // weights of row block R = 4 * tile + gate row
  auto load_w = [&](f32x4 (&wf)[MG], int R) {
#pragma unroll
    for (int j = 0; j < MG; ++j)
      wf[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, l * 16, (R * MG + j) * 64 * 16, 0));
  };
  TapeOps ops[2];          // tile tt lives in ops[tt & 1]: during body(tt) the sets hold tiles tt+1 (consumed) and tt+2 (in flight)
  f32x4 wfr[3][MG];        // row block R = 4 tt + r in wfr[R % 3], loaded two blocks (60 NT/3 MFMAs) ahead
  float dgr[2][NT][4];     // tile tt's gate cotangents (the MFMA B operands) in dgr[tt & 1]
  {
    const float *tc0, *tpv0;
    step_ptrs(p.T - 1, tc0, tpv0);
    load_tape(ops[0], rsrc_of(tc0), rsrc_of(tpv0), 0);
    load_tape(ops[1], rsrc_of(tc0), rsrc_of(tpv0), TPW > 1 ? 1 : 0);
    load_w(wfr[0], 0);
    load_w(wfr[1], 1);
  }

#ifdef HODE_LSTM_STAMPS
#define HODE_LSTAMP(i) if (p.dbg && blockIdx.x == 0 && tid == 0) { __builtin_amdgcn_s_waitcnt(0); p.dbg[(size_t)s * 8 + (i)] = __builtin_amdgcn_s_memtime(); }
#else
#define HODE_LSTAMP(i)
#endif
  const int nA0 = nvalid * p.AD;
  const int act_b = p.AD > 0 ? tid / max(p.AD, 1) : 0, act_u = tid - act_b * p.AD;
  float a_nx = 0.f;
  if (p.a && nA0 <= 256 && tid < nA0) {
    const int t0 = p.reverse ? 0 : p.T - 1;   // time index of the first backward step (s = T - 1)
    a_nx = p.a[((size_t)t0 * p.B + b0) * p.AD + tid];
  }
  for (int s = p.T - 1; s >= 0; --s) {
    const int t = p.reverse ? p.T - 1 - s : s;
    HODE_LSTAMP(0)
    const float *tc, *tpv, *tc_n, *tpv_n;
    step_ptrs(s, tc, tpv);
    step_ptrs(s > 0 ? s - 1 : 0, tc_n, tpv_n);  // next step's first tiles (s == 0: a harmless repeat)
    const float has_prev = s > 0 ? 1.0f : 0.0f;
    const __amdgpu_buffer_rsrc_t rs_c = rsrc_of(tc), rs_p = rsrc_of(tpv), rs_cn = rsrc_of(tc_n), rs_pn = rsrc_of(tpv_n);

    f32x4 acc[TPW][NT];
#pragma unroll
    for (int mt = 0; mt < TPW; ++mt)
#pragma unroll
      for (int c = 0; c < NT; ++c) acc[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The columns of the GEMM operand row that do not depend on the recurrence -- the action columns (a copy of an input),
    // the constant 1 of the bias column, the zero padding -- are written HERE, ahead of the tile loop: inside the store
    // phase each patient's action copy was a dependent global load -> store (12 serial HBM round trips per wave and step,
    // 5.8 us of a 36.6 us step; tools/lstm_stamp_probe.py).
    {
      const int W = p.W, I = p.OBS + p.AD;
      float* hdst = p.h_prev + ((size_t)t * p.B + b0) * W;
      const float* asrc = p.a ? p.a + ((size_t)t * p.B + b0) * p.AD : nullptr;
      const int nA = nvalid * p.AD, nP = nvalid * (W - I - H);
      if (nA <= 256) {
        // one action value per thread, loaded ONE STEP AHEAD (a_nx): a load consumed by the next store stalls its wave for an
        // HBM round trip at the head of the step, and the whole workgroup waits for that wave at the end of the tile loop
        if (tid < nA) hdst[(size_t)act_b * W + p.OBS + act_u] = a_nx;
        if (s > 0 && tid < nA) {
          const int t_n = p.reverse ? p.T - s : s - 1;   // time index of backward step s - 1
          a_nx = p.a[((size_t)t_n * p.B + b0) * p.AD + tid];
        }
      } else {
        for (int e = tid; e < nA; e += 256) {
          const int b = e / p.AD, u = e - b * p.AD;
          hdst[(size_t)b * W + p.OBS + u] = asrc[e];
        }
      }
      for (int e = tid; e < nP; e += 256) {
        const int b = e / (W - I - H), u = e - b * (W - I - H);
        hdst[(size_t)b * W + I + H + u] = u == 0 ? 1.0f : 0.0f;
      }
    }

    // element-wise step of tile tt: gate cotangents -> dgr[tt & 1] (+ the transposed copies for the row-major stores)
    auto elementwise = [&](int tt, const TapeOps& o, float (&dg)[NT][4]) {
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const float gi = o.gi[c], gf = o.gf[c], gg = o.gg[c], go = o.go[c], cn = o.cn[c];
        const float c_prev = has_prev * o.cp[c];
        const float h_prev = has_prev * (o.op[c] * tanh_f32(c_prev));
        const float tcn = tanh_f32(cn);
        const float dh = carry_h[tt][c];
        const float dc = __builtin_fmaf(dh * go, __builtin_fmaf(-tcn, tcn, 1.0f), carry_c[tt][c]);
        dg[c][0] = dc * gg * gi * (1.0f - gi);
        dg[c][1] = dc * c_prev * gf * (1.0f - gf);
        dg[c][2] = dc * gi * __builtin_fmaf(-gg, gg, 1.0f);
        dg[c][3] = dh * tcn * go * (1.0f - go);
        carry_c[tt][c] = dc * gf;
        // lane base + compile-time offset: (patient 16c + pc, unit (w TPW + tt) 4 + g).  Written as one index
        // expression the compiler hoisted one address per (tile, column, gate) out of the step loop and spilled them.
#pragma unroll
        for (int r = 0; r < 4; ++r) dg_lane[16 * c * LDG + r * Hp + 4 * tt] = dg[c][r];
        hT_lane[16 * c * LDH + 4 * tt] = h_prev;
      }
    };
    // body(tt): row block r of the tile = TPW * NT MFMAs.  The tape operands of tile tt + 2 are requested in the first
    // MFMAs of the body (they are needed from the start of the next body: one body = 1.6 us of latency cover), the
    // weights of row block R + 2 at the start of row block R; the group barriers below pin that issue order (left to
    // itself the scheduler sinks every load next to its use, and an in-order vmcnt wait on the newest load waits for all).
    auto body = [&](int tt) {
#if defined(HODE_LSTM_STAMPS) || HODE_BPTT_BODY_HOOK
      // In product builds p.dbg is null and this is a never-taken branch -- KEPT ON PURPOSE: it ends the basic block at every
      // unit tile.  Without it the ten bodies of a step are one block and the kernel is 0.47 ms (15 %) slower at the bench
      // shape (same-call A/B of the two builds, tools/lstm_time_probe.py): the scheduler's / waitcnt pass's choices over a
      // 3 000-instruction block undo part of the issue order pinned below.
      if (p.dbg && blockIdx.x == 0 && tid == 0) p.dbg[(size_t)(p.T + s) * 16 + tt] = __builtin_amdgcn_s_memtime();  // no wait
#endif
      TapeOps& o_nx = ops[tt & 1];
      const bool same_step = tt + 2 < TPW;
      const int tile_nx = same_step ? tt + 2 : min(tt + 2 - TPW, TPW - 1);
      load_w(wfr[(4 * tt + 2) % 3], (4 * tt + 2) % (4 * TPW));
      if (same_step) load_tape(o_nx, rs_c, rs_p, tile_nx);
      else load_tape(o_nx, rs_cn, rs_pn, tile_nx);
      if (tt + 1 < TPW) elementwise(tt + 1, ops[(tt + 1) & 1], dgr[(tt + 1) & 1]);
      if (HODE_BPTT_BURST) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int R = 4 * tt + r;
        if (r > 0) load_w(wfr[(R + 2) % 3], (R + 2) % (4 * TPW));   // past the step's last block: the next step's first two
#pragma unroll
        for (int mt = 0; mt < TPW; ++mt)
#pragma unroll
          for (int c = 0; c < NT; ++c)
            acc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[R % 3][mt >> 2][mt & 3], dgr[tt & 1][c][r], acc[mt][c], 0, 0, 0);
      }
      // issue pipeline of the block: per MFMA one or two VALU of the next tile's element-wise step, the loads in the
      // first MFMAs of each row block, one LDS write every 8 MFMAs
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        constexpr int MF = TPW * NT;
        const int n_loads = HODE_BPTT_BURST ? (r == 0 ? 0 : MG) : MG + (r == 0 ? 7 * NT : 0);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              // 1 MFMA
          if (i < n_loads) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);             // 1 VMEM read
          if (tt + 1 < TPW && !HODE_BPTT_BURST) {
            if (i & 1) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                 // 1 VALU
            else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                       // 2 VALU
            if ((i & 7) == 7) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          // 1 LDS write
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    elementwise(0, ops[0], dgr[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) body(tt);
    // canonical register sets for the next step: its tile 0 -> ops[0], its tile 1 -> ops[1], its row blocks 0, 1 ->
    // wfr[0], wfr[1]
    if constexpr ((TPW & 1) && TPW > 1) {   // TPW == 1: the only body loads the next step's tile 0 straight into ops[0]
      const TapeOps t0 = ops[1], t1 = ops[0];
      ops[0] = t0; ops[1] = t1;
    }
    if constexpr ((4 * TPW) % 3 != 0) {
      f32x4 n0[MG], n1[MG];
#pragma unroll
      for (int j = 0; j < MG; ++j) { n0[j] = wfr[(4 * TPW) % 3][j]; n1[j] = wfr[(4 * TPW + 1) % 3][j]; }
#pragma unroll
      for (int j = 0; j < MG; ++j) { wfr[0][j] = n0[j]; wfr[1][j] = n1[j]; }
    }
    HODE_LSTAMP(1)
    lds_barrier();
    HODE_LSTAMP(2)
    // coalesced row-major stores of this step's dG and h_prev tiles.  Wave w stores patients w, w+4, ...; the loops run
    // over (patient, gate, unit) explicitly -- a flat index would need two integer divisions per element, which made
    // this transposition the longest phase of the step (120 iterations x ~70 instructions per thread).
    {
      float* gdst = p.grad_gates + ((size_t)t * p.B + b0) * 4 * H;
      const int W = p.W, I = p.OBS + p.AD;
      float* hdst = p.h_prev + ((size_t)t * p.B + b0) * W;
      const float* asrc = p.a ? p.a + ((size_t)t * p.B + b0) * p.AD : nullptr;
      // The LDS reads of a patient's rows are all issued before the first store (and two patients are in flight): with one
      // ds_read -> wait -> global_store chain per 16 bytes this phase was LDS-LATENCY bound -- 84 dependent round trips per
      // wave and step, 8.6 us of a 36.6 us step (tools/lstm_stamp_probe.py) -- not bandwidth bound.
      const bool vec = (H & 3) == 0;
      const bool lane_g = 4 * l < H;           // H <= 160: one 16-byte chunk per lane and gate row covers a row
      if constexpr (FLAT) {
        // H a multiple of 16 (every shipped encoder): a patient's dG row is 4H contiguous floats in LDS and in HBM -- a flat
        // copy in 16-byte chunks over ALL 64 lanes (ceil(H/64) instructions instead of 4 with 4H/16 <= 40 lanes busy), software
        // pipelined: the LDS reads of the wave's next patient are in flight while this patient's stores issue
        constexpr int NC = (Hp + 63) / 64;       // 16-byte chunks of the dG row per lane
        constexpr int NHC = (Hp + 63) / 64;      // floats of the h_prev row per lane
        struct Row { f32x4 v[NC]; float h[NHC]; };
        auto ld = [&](int b, Row& r) {
          const float* drow = dgt + (size_t)b * LDG;
          const float* hrow = hT + (size_t)b * LDH;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int f = l + 64 * c;
            r.v[c] = f < Hp ? *reinterpret_cast<const f32x4*>(drow + 4 * f) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int k = 0; k < NHC; ++k) r.h[k] = (l + 64 * k < H) ? hrow[l + 64 * k] : 0.f;
        };
        auto st = [&](int b, const Row& r) {
          float* grow = gdst + (size_t)b * 4 * H;
          float* hd = hdst + (size_t)b * W + I;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int f = l + 64 * c;
            if (f < Hp) *reinterpret_cast<f32x4*>(grow + 4 * f) = r.v[c];
          }
#pragma unroll
          for (int k = 0; k < NHC; ++k)
            if (l + 64 * k < H) hd[l + 64 * k] = r.h[k];
        };
        Row ra, rb;
        int b = w;
        if (b < nvalid) ld(b, ra);
        for (; b < nvalid; b += 8) {
          if (b + 4 < nvalid) ld(b + 4, rb);
          __builtin_amdgcn_sched_barrier(0);   // the next patient's reads stay ahead of this patient's stores
          st(b, ra);
          if (b + 4 < nvalid) {
            if (b + 8 < nvalid) ld(b + 8, ra);
            __builtin_amdgcn_sched_barrier(0);
            st(b + 4, rb);
          }
        }
      } else
      for (int b = w; b < nvalid; b += 8) {
        const int b2 = b + 4;
        const bool two = b2 < nvalid;
        const float* drow = dgt + (size_t)b * LDG;
        const float* drow2 = dgt + (size_t)(two ? b2 : b) * LDG;
        const float* hrow = hT + (size_t)b * LDH;
        const float* hrow2 = hT + (size_t)(two ? b2 : b) * LDH;
        f32x4 v[4], v2[4];
        float hv[3], hv2[3];
        if (vec) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = lane_g ? *reinterpret_cast<const f32x4*>(drow + r * Hp + 4 * l) : f32x4{0.f, 0.f, 0.f, 0.f};
            v2[r] = lane_g ? *reinterpret_cast<const f32x4*>(drow2 + r * Hp + 4 * l) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          hv[k] = (l + 64 * k < H) ? hrow[l + 64 * k] : 0.f;
          hv2[k] = (l + 64 * k < H) ? hrow2[l + 64 * k] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads above the stores
        auto put = [&](int bb, const float* dr, const f32x4 (&vv)[4], const float (&hh)[3]) {
          float* grow = gdst + (size_t)bb * 4 * H;
          if (vec) {
            if (lane_g) {
#pragma unroll
              for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(grow + r * H + 4 * l) = vv[r];
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              for (int u = l; u < H; u += 64) grow[r * H + u] = dr[r * Hp + u];
          }
          float* hd = hdst + (size_t)bb * W;
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (l + 64 * k < H) hd[I + l + 64 * k] = hh[k];
        };
        put(b, drow, v, hv);
        if (two) put(b2, drow2, v2, hv2);
      }
    }
    HODE_LSTAMP(3)
    lds_barrier();
    HODE_LSTAMP(4)
    // exchange the K-split partial products: slab[w][u'][patient]
#pragma unroll
    for (int mt = 0; mt < TPW; ++mt)
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          slab_w[(16 * mt + rr) * LD + 16 * c] = acc[mt][c][rr];
    HODE_LSTAMP(5)
    lds_barrier();
    HODE_LSTAMP(6)
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const float* sp = slab_r + 4 * tt * LD + 16 * c;
        carry_h[tt][c] = ((sp[0] + sp[Hp * LD]) + sp[2 * Hp * LD]) + sp[3 * Hp * LD];
      }
    }
    HODE_LSTAMP(7)
    lds_barrier();
  }
#undef HODE_LSTAMP
}

// ---------------------------------------------------------------------------------------------------- launch geometry
struct LstmGeom {
  int Hp, TPW, NT, BT, nblk, Kq, KQ4, LD;
  int fTPW, fNW;   // the forward kernel's tiles per wave x waves (fTPW * fNW == TPW * 4)
  size_t wp_floats, tape_floats, whp_floats, lds_bytes, lds_bwd_bytes;
};

// Per-size entry points, one translation unit per TPW (csrc/hode_lstm_tpw.hip compiled with -DHODE_LSTM_TPW=<n>, so the
// hidden sizes build in parallel): padded H = 16 TPW.
#define HODE_LSTM_DECL(n)                                                                  \
  int lstm_fwd_tpw##n(const LstmGeom& G, const LstmArgs& a, hipStream_t s);               \
  int lstm_bwd_tpw##n(const LstmGeom& G, const LstmBwdArgs& a, hipStream_t s);
HODE_LSTM_DECL(1) HODE_LSTM_DECL(2) HODE_LSTM_DECL(3) HODE_LSTM_DECL(4) HODE_LSTM_DECL(5) HODE_LSTM_DECL(6) HODE_LSTM_DECL(8)
HODE_LSTM_DECL(10)
#undef HODE_LSTM_DECL

}  // namespace hode
