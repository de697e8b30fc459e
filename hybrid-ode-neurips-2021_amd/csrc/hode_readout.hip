// Fused linear readout + masked sum of squared errors (+ its gradients), gfx950.
//
// Replaces, for the training loss, the pair  x_hat = output_function(h)  (reference model.py:1120, nn.Linear(D -> obs),
// :1097-1100) and  lik = sum((x - x_hat)^2 * mask) / B  (model.py:1179) together with their autograd backward: x_hat
// (T*B*obs floats, 320 MB at the bench shape) is never written to HBM.  SURVEY.md 8f "next" item 1 / row A9.
//
// One pass over the (t, b) rows, HBM-bound by design: per row read h (4D B), x and mask (8 obs B), write grad_h (4D B).
//   e[o]      = (x[o] - (sum_d Wo[o][d] h[d] + bo[o])) * mask[o]         (mask is 0/1 in the reference; general masks work)
//   lik      += e[o] * (x[o] - x_hat[o])            = (x - x_hat)^2 mask
//   grad_xh   = -2 e[o] * scale                      (scale = 1/B; the caller multiplies by the upstream gradient later)
//   grad_h[d] = sum_o grad_xh[o] Wo[o][d];  grad_Wo[o][d] += grad_xh[o] h[d];  grad_bo[o] += grad_xh[o]
// Lane layout: a row is handled by OL = obs/4 adjacent lanes (20 for obs = 80): lane j owns outputs 4j..4j+3 (float4
// loads of x / mask: one contiguous 16*OL-byte run per row), keeps its 4 x D slice of Wo and its 4 x D slice of grad_Wo
// in registers; grad_h is reduced across the row's lanes through LDS.  Per-wave partials are folded in a fixed order.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"

namespace hode {

struct ReadoutArgs {
  const float* __restrict__ h;      // [R][D]
  const float* __restrict__ x;      // [R][OBS]
  const float* __restrict__ mask;   // [R][OBS]
  const float* __restrict__ wo;     // [OBS][D]
  const float* __restrict__ bo;     // [OBS]
  float* __restrict__ grad_h;       // [R][D] or nullptr (loss only)
  float* __restrict__ partials;     // [n_waves][1 + OBS*D + OBS]
  long long R;
  int OBS;
  float scale;
};

// D = latent dim (compile time), rows per wave-iteration RPI = 64 / OL computed at run time (OL = OBS / 4 <= 32)
template <int D, bool GRAD>
__global__ __launch_bounds__(64) void readout_sse_kernel(ReadoutArgs a) {
  __shared__ float red[64 * D];
  const int lane = threadIdx.x;
  const int OL = a.OBS >> 2;
  const int RPI = 64 / OL;
  const int slot = lane / OL;          // which of the RPI rows of this iteration
  const int j = lane - slot * OL;      // which group of 4 outputs
  const bool active = slot < RPI;
  const int o0 = 4 * (active ? j : 0);

  float w[4][D], bias[4], dw[4][D], db[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      w[q][d] = a.wo[(size_t)(o0 + q) * D + d];
      dw[q][d] = 0.f;
    }
    bias[q] = a.bo[o0 + q];
    db[q] = 0.f;
  }
  float lik = 0.f;
  const long long n_iter = (a.R + RPI - 1) / RPI;
  for (long long it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const long long r = it * RPI + slot;
    const bool live = active && r < a.R;
    const long long rr = live ? r : 0;
    float hv[D];
    if constexpr (D % 4 == 0) {
#pragma unroll
      for (int c = 0; c < D / 4; ++c) {
        const float4 v = reinterpret_cast<const float4*>(a.h + rr * D)[c];
        hv[4 * c] = v.x; hv[4 * c + 1] = v.y; hv[4 * c + 2] = v.z; hv[4 * c + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int d = 0; d < D; ++d) hv[d] = a.h[rr * D + d];
    }
    const float4 xv = *reinterpret_cast<const float4*>(a.x + rr * a.OBS + o0);
    const float4 mv = *reinterpret_cast<const float4*>(a.mask + rr * a.OBS + o0);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    const float ms[4] = {mv.x, mv.y, mv.z, mv.w};
    float gh[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gh[d] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float xh = bias[q];
#pragma unroll
      for (int d = 0; d < D; ++d) xh = __builtin_fmaf(w[q][d], hv[d], xh);
      const float diff = live ? xs[q] - xh : 0.f;
      const float e = diff * ms[q];
      lik = __builtin_fmaf(e, diff, lik);
      if constexpr (GRAD) {
        const float gx = -2.0f * a.scale * e;
        db[q] += gx;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          dw[q][d] = __builtin_fmaf(gx, hv[d], dw[q][d]);
          gh[d] = __builtin_fmaf(gx, w[q][d], gh[d]);
        }
      }
    }
    if constexpr (GRAD) {
      // sum gh over the OL lanes of the row: stage through LDS, the first D lanes of each row do the adds
#pragma unroll
      for (int d = 0; d < D; ++d) red[d * 64 + lane] = gh[d];
      __syncthreads();
      if (live) {
        for (int d = j; d < D; d += OL) {   // the row's OL lanes share its D columns
          float s = 0.f;
          const float* col = red + d * 64 + slot * OL;
          for (int k = 0; k < OL; ++k) s += col[k];
          a.grad_h[rr * D + d] = s;
        }
      }
      __syncthreads();
    }
  }
  // per-wave partial row: [lik | dWo (OBS*D) | dbo (OBS)]
  const size_t P = 1 + (size_t)a.OBS * D + a.OBS;
  float* out = a.partials + (size_t)blockIdx.x * P;
  const float liksum = wave_sum(lik);
  if (lane == 0) out[0] = liksum;
  if constexpr (GRAD) {
    // lanes with the same j (different row slots) hold partial sums for the same outputs: fold them through LDS
    __shared__ float red2[64];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int d = 0; d <= D; ++d) {
        const float v = d < D ? dw[q][d < D ? d : 0] : db[q];
        red2[lane] = active ? v : 0.f;
        __syncthreads();
        if (slot == 0) {
          float s = 0.f;
          for (int k = 0; k < RPI; ++k) s += red2[k * OL + j];
          if (d < D) out[1 + (size_t)(o0 + q) * D + d] = s;
          else out[1 + (size_t)a.OBS * D + o0 + q] = s;
        }
        __syncthreads();
      }
    }
  }
}

// ---- matrix-core variant for the two shipped shapes (D = 12 / obs <= 80, D = 8 / obs <= 48).
// The lane-per-4-outputs kernel above handles 3 rows (2.2 KB) per wave-iteration with every load consumed at once: at two
// waves per SIMD that is ONE HBM round trip per 2.2 KB and SIMD -- 357 us for 736 MB (2 TB/s).  Here a wave-iteration is 16
// rows (10.8 KB), the next iteration's x / mask / h are in flight while this one computes, and the three small products run
// on v_mfma_f32_16x16x4_f32 (exact fp32) in the layouts the loads already have:
//   x_hat^T[o x row] = Wo[o x d] h^T[d x row]          C layout (o = 16 mt + 4 g + r, row = pc) == the float4 the lane
//                                                       loads from x / mask at [row pc][16 mt + 4 g ..]
//   grad_h^T[d x row] = Wo^T[d x o] gx[o x row]         K ordered (mt, r | g): the B fragment IS the lane's own gx[mt][r];
//                                                       C layout (d = 4 g + r, row = pc) == one float4 store per lane
//   [dWo | dbo][o x (d | 1)] += gx[o x row] [h | 1][row x (d | 1)]   contraction over the 16 rows: gx transposed through a
//                                                       5 KB LDS tile; accumulators live across the whole launch
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D, int MT, bool GRAD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void readout_mf_kernel(ReadoutArgs a) {
  constexpr int KQ = D / 4;
  constexpr int LDP = 17;
  __shared__ float gxt[GRAD ? MT * 16 * LDP : 1];  // gx^T tile [o][row]
  const int l = threadIdx.x, g = l >> 4, pc = l & 15;
  const int OBS = a.OBS;
  const long long R = a.R;

  float wa[MT][KQ];   // A of x_hat: Wo[16 mt + pc][4 kq + g]
  f32x4 bias[MT];     // C layout: o = 16 mt + 4 g + r
  float wt[MT][4];    // A of grad_h, k-quad (mt, r): Wo[16 mt + 4 g + r][d = pc]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int oa = 16 * mt + pc;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) wa[mt][kq] = oa < OBS ? a.wo[(size_t)oa * D + 4 * kq + g] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * mt + 4 * g + r;
      bias[mt][r] = o < OBS ? a.bo[o] : 0.f;
      wt[mt][r] = (o < OBS && pc < D) ? a.wo[(size_t)o * D + pc] : 0.f;
    }
  }
  f32x4 dW[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) dW[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float lik = 0.f;

  const long long n_iter = (R + 15) / 16;
  struct Tile { f32x4 x[MT], m[MT]; float h[KQ]; float hb[4]; };
  auto fetch = [&](long long it, Tile& t) {
    const long long row = min(it * 16 + pc, R - 1);   // clamped: rows past the end are zeroed when used
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int o0 = 16 * mt + 4 * g;                 // OBS % 4 == 0: the group of 4 is inside or outside as a whole
      const size_t off = (size_t)row * OBS + (o0 < OBS ? o0 : 0);
      t.x[mt] = *reinterpret_cast<const f32x4*>(a.x + off);
      t.m[mt] = *reinterpret_cast<const f32x4*>(a.mask + off);
    }
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) t.h[kq] = a.h[(size_t)row * D + 4 * kq + g];
    if constexpr (GRAD) {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {                // B of the weight-gradient product: [h | 1][row = 4 kq + g][col = pc]
        const long long r2 = min(it * 16 + 4 * kq + g, R - 1);
        t.hb[kq] = pc < D ? a.h[(size_t)r2 * D + pc] : (pc == D ? 1.0f : 0.f);
      }
    }
  };
  Tile cur, nxt;
  if ((long long)blockIdx.x < n_iter) fetch(blockIdx.x, cur);
  for (long long it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const long long it_n = it + gridDim.x;
    fetch(it_n < n_iter ? it_n : it, nxt);            // the last prefetch repeats this tile (unused)
    const bool live = it * 16 + pc < R;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][kq], cur.h[kq], acc[mt], 0, 0, 0);
    float gx[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool ok = live && 16 * mt + 4 * g < OBS;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float diff = ok ? cur.x[mt][r] - acc[mt][r] : 0.f;
        const float e = ok ? diff * cur.m[mt][r] : 0.f;
        lik = __builtin_fmaf(e, diff, lik);
        gx[mt][r] = -2.0f * a.scale * e;
      }
    }
    if constexpr (GRAD) {
      f32x4 gh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gh = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[mt][r], gx[mt][r], gh, 0, 0, 0);
          gxt[(16 * mt + 4 * g + r) * LDP + pc] = gx[mt][r];
        }
      if (live && 4 * g < D) *reinterpret_cast<f32x4*>(a.grad_h + (size_t)(it * 16 + pc) * D + 4 * g) = gh;
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
          dW[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gxt[(16 * mt + pc) * LDP + 4 * kq + g], cur.hb[kq], dW[mt], 0, 0, 0);
      __syncthreads();
    }
    cur = nxt;
  }
  // per-wave partial row: [lik | dWo (OBS*D) | dbo (OBS)]; lane (g, pc) holds dWo[16 mt + 4 g + r][d = pc], column D = dbo
  const size_t P = 1 + (size_t)OBS * D + OBS;
  float* out = a.partials + (size_t)blockIdx.x * P;
  const float liksum = wave_sum(lik);
  if (l == 0) out[0] = liksum;
  if constexpr (GRAD) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * mt + 4 * g + r;
        if (o < OBS) {
          if (pc < D) out[1 + (size_t)o * D + pc] = dW[mt][r];
          else if (pc == D) out[1 + (size_t)OBS * D + o] = dW[mt][r];
        }
      }
  }
}

// out[j] (+)= fixed-order sum over waves; j = 0: lik, then dWo, then dbo
__global__ __launch_bounds__(64) void readout_fold_kernel(const float* __restrict__ partials, int n_waves, int P, int n_w,
                                                          float* __restrict__ lik, float* __restrict__ gw, float* __restrict__ gb) {
  const int j = blockIdx.x;
  const int lane = threadIdx.x;
  float s = 0.f;
  for (int w = lane; w < n_waves; w += 64) s += partials[(size_t)w * P + j];
  s = wave_sum(s);
  if (lane != 0) return;
  if (j == 0) lik[0] = s;
  else if (j <= n_w) { if (gw) gw[j - 1] += s; }
  else if (gb) gb[j - 1 - n_w] += s;
}

}  // namespace hode

namespace {
constexpr int kReadoutWaves = 2048;  // grid-stride: two waves per SIMD keep enough loads in flight for an HBM-bound pass

// matrix-core kernel: the two shipped shapes
bool readout_mf(int latent, int obs) {
  if (getenv("HODE_READOUT_VALU")) return false;   // A/B switch for the lane-per-4-outputs kernel
  return (latent == 12 && obs > 48 && obs <= 80) || (latent == 8 && obs > 32 && obs <= 48);
}

int readout_waves(long long rows, int obs, int latent) {
  const int rpi = readout_mf(latent, obs) ? 16 : 64 / (obs / 4);
  const long long iters = (rows + rpi - 1) / rpi;
  return (int)(iters < kReadoutWaves ? (iters > 0 ? iters : 1) : kReadoutWaves);
}
}  // namespace

extern "C" size_t hode_readout_workspace_bytes(const hode_readout_desc* d) {
  if (!d || d->struct_size != sizeof(hode_readout_desc) || d->obs_dim <= 0 || d->obs_dim % 4 || d->obs_dim > 128) return 0;
  const size_t P = 1 + (size_t)d->obs_dim * d->latent_dim + d->obs_dim;
  return (size_t)readout_waves(d->rows, d->obs_dim, d->latent_dim) * P * sizeof(float);
}

extern "C" int hode_readout_sse(const hode_readout_desc* d, void* stream) {
  if (!d) return hode::fail(HODE_E_NULL, "descriptor is NULL");
  if (d->struct_size != sizeof(hode_readout_desc)) return hode::fail(HODE_E_SIZE, "struct_size mismatch (ABI)");
  if (d->rows <= 0 || d->obs_dim <= 0) return hode::fail(HODE_E_SIZE, "bad sizes rows=%lld obs=%d", (long long)d->rows, d->obs_dim);
  if (d->obs_dim % 4 != 0 || d->obs_dim > 128)
    return hode::fail(HODE_E_UNSUPPORTED, "readout: obs_dim %d must be a multiple of 4 and <= 128", d->obs_dim);
  if (d->latent_dim != 4 && d->latent_dim != 6 && d->latent_dim != 8 && d->latent_dim != 12)
    return hode::fail(HODE_E_UNSUPPORTED, "readout: latent_dim %d has no compiled kernel (have 4, 6, 8, 12)", d->latent_dim);
  if (!d->h || !d->x || !d->mask || !d->w || !d->b || !d->lik) return hode::fail(HODE_E_NULL, "h / x / mask / w / b / lik must be non-NULL");
  if (((uintptr_t)d->x | (uintptr_t)d->mask | (uintptr_t)d->h) & 15) return hode::fail(HODE_E_ALIGN, "h / x / mask must be 16-byte aligned");
  const size_t need = hode_readout_workspace_bytes(d);
  if (!d->workspace || d->workspace_bytes < need) return hode::fail(HODE_E_WORKSPACE, "workspace %zu B < required %zu B", d->workspace_bytes, need);
  const bool grad = d->grad_h != nullptr;
  hode::ReadoutArgs a{};
  a.h = d->h; a.x = d->x; a.mask = d->mask; a.wo = d->w; a.bo = d->b; a.grad_h = d->grad_h; a.partials = (float*)d->workspace;
  a.R = d->rows; a.OBS = d->obs_dim; a.scale = d->scale;
  const int nw = readout_waves(d->rows, d->obs_dim, d->latent_dim);
  hipStream_t s = (hipStream_t)stream;
  if (readout_mf(d->latent_dim, d->obs_dim)) {
    if (d->latent_dim == 12) {
      if (grad) hipLaunchKernelGGL((hode::readout_mf_kernel<12, 5, true>), dim3(nw), dim3(64), 0, s, a);
      else hipLaunchKernelGGL((hode::readout_mf_kernel<12, 5, false>), dim3(nw), dim3(64), 0, s, a);
    } else {
      if (grad) hipLaunchKernelGGL((hode::readout_mf_kernel<8, 3, true>), dim3(nw), dim3(64), 0, s, a);
      else hipLaunchKernelGGL((hode::readout_mf_kernel<8, 3, false>), dim3(nw), dim3(64), 0, s, a);
    }
  } else {
#define HODE_RO(DD)                                                                                          \
  if (grad) hipLaunchKernelGGL((hode::readout_sse_kernel<DD, true>), dim3(nw), dim3(64), 0, s, a);            \
  else hipLaunchKernelGGL((hode::readout_sse_kernel<DD, false>), dim3(nw), dim3(64), 0, s, a);
  switch (d->latent_dim) {
    case 4: HODE_RO(4) break;
    case 6: HODE_RO(6) break;
    case 8: HODE_RO(8) break;
    default: HODE_RO(12) break;
  }
  }
  if (int e = hode::hip_fail(hipGetLastError(), "readout_sse launch")) return e;
  const int n_w = d->obs_dim * d->latent_dim;
  const int P = 1 + n_w + d->obs_dim;
  hipLaunchKernelGGL(hode::readout_fold_kernel, dim3(grad ? P : 1), dim3(64), 0, s, (const float*)d->workspace, nw, P, n_w, d->lik,
                     d->grad_w, d->grad_b);
  return hode::hip_fail(hipGetLastError(), "readout_fold launch");
}
