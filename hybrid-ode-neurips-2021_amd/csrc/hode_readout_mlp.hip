// Fused two-layer readout + masked (time-weighted) sum of squared errors and ALL its gradients on the matrix cores, gfx950.
//
// Replaces, for the real-data training loss, x_hat = output_function(h)[1:] with output_function = Linear(D -> D+1), ELU,
// Linear(D+1 -> obs) (reference model.py:809-813, :859) and lik = sum((x[t0:] - x_hat)^2 * mask[t0:] * weight) / B
// (VariationalInferenceReal.loss, model.py:1243-1247) together with their autograd backward.  As eager torch ops this is
// two tall GEMMs forward, six backward (four of them reductions over ~0.8 M rows into 20 x 21 outputs, which the BLAS
// library runs at a few per cent of the HBM rate) and a dozen element-wise passes: 4.3 of the 6.9 ms of a config-5
// training step.  Here: ONE pass over the (t, b) rows -- read h, x, mask, write grad_h -- and one small fold.
//
// Layout (the recipe of hode_neural_mf.hpp).  A wave handles 16 rows at a time; a vector of up to 24 components is two
// accumulator-shaped tiles held by lane (g, n) (n = row, g = lane >> 4):
//     t0[r] = component 4g + r            (components 0..15)
//     t1[r] = component 16 + 4r + g       (components 16..23: register r is ONE k-chunk of four components)
// so that for every product W v the B fragment of a k-chunk IS a register of v, and the C tile an MFMA leaves IS a tile
// of the result (the rows of the second output tile are ordered to land in the t1 pattern).  Weights live in registers
// as A fragments (46 floats per lane).  Per 16 rows: 10 + 12 MFMAs forward, 12 + 12 for the two transposed products, and
// 32 for the weight gradients, which contract over the 16 ROWS: the four operand vectors go through patient-major LDS
// images (pitch == 16 mod 64: conflict-free fragment reads) exactly as in NeuralGradAcc; a ones entry in a free slot of the
// layer inputs collects the bias gradients.  78 v_mfma_f32_16x16x4_f32 per 16 rows, exact fp32.
#include <hip/hip_runtime.h>

#include "../../include/hode.h"
#include "hode_common.hpp"
#include "hode_host.hpp"

namespace hode {

typedef float rv4 __attribute__((ext_vector_type(4)));

struct ReadoutMlpArgs {
  const float* __restrict__ h;     // [R][DL]
  const float* __restrict__ x;     // [R][DO]
  const float* __restrict__ mask;  // [R][DO]
  const float* __restrict__ tw;    // [R / B] per-time weight or nullptr
  const float* __restrict__ w1;    // [DH][DL]
  const float* __restrict__ b1;    // [DH]
  const float* __restrict__ w2;    // [DO][DH]
  const float* __restrict__ b2;    // [DO]
  float* __restrict__ grad_h;      // [R][DL] or nullptr
  float* __restrict__ partials;    // [n_waves][NP]
  long long R;
  int B;
  float scale;                     // 1 / B
};

HODE_DEV rv4 rmfma(float a, float b, rv4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// component held by storage slot (tile, 4g + r)
HODE_DEV constexpr int rm_comp(int tile, int g, int r) { return tile == 0 ? 4 * g + r : 16 + 4 * r + g; }
// component of row / column position i (0..15) of an MFMA tile that holds storage tile `tile`
HODE_DEV constexpr int rm_pos_comp(int tile, int i) { return rm_comp(tile, i >> 2, i & 3); }

struct RmVec {
  rv4 t0, t1;
};

// A fragments of y = W v, W [NO][NI] row-major: for output tile mt and k-chunk (s, r), lane (m, kk) holds
// W[comp(mt, m)][comp(s, 4 kk + r)]; chunks: r = 0..3 of tile 0, r = 0..NC1-1 of tile 1
template <int NO, int NI>
struct RmWeights {
  static constexpr int NC1 = NI > 16 ? (NI - 16 + 3) / 4 : 0;
  float a0[2][4], a1[2][NC1 > 0 ? NC1 : 1];
  HODE_DEV void load(const float* __restrict__ W, int lane, bool transposed, int ld) {
    const int m = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int o = rm_pos_comp(mt, m);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = rm_comp(0, kk, r);
        a0[mt][r] = (o < NO && i < NI) ? (transposed ? W[(size_t)i * ld + o] : W[(size_t)o * ld + i]) : 0.f;
      }
#pragma unroll
      for (int r = 0; r < NC1; ++r) {
        const int i = rm_comp(1, kk, r);
        a1[mt][r] = (o < NO && i < NI) ? (transposed ? W[(size_t)i * ld + o] : W[(size_t)o * ld + i]) : 0.f;
      }
    }
  }
  HODE_DEV RmVec mul(const RmVec& v, const RmVec& init) const {
    RmVec y = init;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      y.t0 = rmfma(a0[0][r], v.t0[r], y.t0);
      y.t1 = rmfma(a0[1][r], v.t0[r], y.t1);
    }
#pragma unroll
    for (int r = 0; r < NC1; ++r) {
      y.t0 = rmfma(a1[0][r], v.t1[r], y.t0);
      y.t1 = rmfma(a1[1][r], v.t1[r], y.t1);
    }
    return y;
  }
};

template <int N>
HODE_DEV RmVec rm_load_param(const float* __restrict__ b, int g) {  // a parameter vector in the tile layout (zeros past N)
  RmVec v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c0 = rm_comp(0, g, r), c1 = rm_comp(1, g, r);
    v.t0[r] = c0 < N ? b[c0] : 0.f;
    v.t1[r] = (r < 2 && c1 < N) ? b[c1] : 0.f;
  }
  return v;
}
template <int N>
HODE_DEV RmVec rm_load_row(const float* __restrict__ src, int g, bool live) {  // one row of an [R][N] array, N % 4 == 0
  RmVec v;
  v.t0 = rv4{0.f, 0.f, 0.f, 0.f};
  v.t1 = v.t0;
  if (live) {
    if (4 * g < N && 4 * g < 16) v.t0 = *reinterpret_cast<const rv4*>(src + 4 * g);
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (rm_comp(1, g, r) < N) v.t1[r] = src[rm_comp(1, g, r)];
  }
  return v;
}
template <int N>
HODE_DEV void rm_store_row(float* __restrict__ dst, int g, const RmVec& v, bool live) {
  if (!live) return;
  if (4 * g < N && 4 * g < 16) *reinterpret_cast<rv4*>(dst + 4 * g) = v.t0;
#pragma unroll
  for (int r = 0; r < 2; ++r)
    if (rm_comp(1, g, r) < N) dst[rm_comp(1, g, r)] = v.t1[r];
}

template <int DL, int DO>
struct ReadoutMlp {
  static constexpr int DH = DL + 1;
  static constexpr int P = 80;                     // LDS image pitch (floats), == 16 mod 64
  static constexpr int kLdsFloats = 4 * 16 * P;    // images of gx, a, gz, h: [16 rows][32 slots (+ pad)]
  static constexpr int NP = 8 * 256 + 16;          // per-wave partial block: 8 gradient tiles + lik
  static_assert(DL % 4 == 0 && DO % 4 == 0 && DL <= 24 && DH <= 24 && DO <= 24, "readout MLP: dims up to 24, multiples of 4");
};

template <int DL, int DO, bool GRAD>
__global__ __launch_bounds__(64) void readout_mlp_kernel(ReadoutMlpArgs a) {
  typedef ReadoutMlp<DL, DO> Cfg;
  constexpr int DH = Cfg::DH, P = Cfg::P;
  __shared__ __attribute__((aligned(16))) float lds[GRAD ? Cfg::kLdsFloats : 4];
  const int lane = threadIdx.x, g = lane >> 4, n = lane & 15;
  RmWeights<DH, DL> W1;
  RmWeights<DO, DH> W2;
  RmWeights<DH, DO> W2T;
  RmWeights<DL, DH> W1T;
  W1.load(a.w1, lane, false, DL);
  W2.load(a.w2, lane, false, DH);
  if constexpr (GRAD) {
    W2T.load(a.w2, lane, true, DH);
    W1T.load(a.w1, lane, true, DL);
  }
  const RmVec bias1 = rm_load_param<DH>(a.b1, g), bias2 = rm_load_param<DO>(a.b2, g);
  const rv4 z4 = rv4{0.f, 0.f, 0.f, 0.f};
  const RmVec zero{z4, z4};
  rv4 dW2[4], dW1[4];  // gradient tiles (out tile mt, in tile s) at index 2 mt + s
#pragma unroll
  for (int i = 0; i < 4; ++i) dW2[i] = dW1[i] = z4;
  float lik = 0.f;
  float* GX = lds;
  float* AI = lds + 16 * P;
  float* GZ = lds + 32 * P;
  float* HI = lds + 48 * P;

  const long long n_tiles = (a.R + 15) / 16;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long row = tile * 16 + n;
    const bool live = row < a.R;
    const long long rr = live ? row : 0;
    const RmVec hv = rm_load_row<DL>(a.h + rr * DL, g, live);
    const RmVec xv = rm_load_row<DO>(a.x + rr * DO, g, live);
    const RmVec mv = rm_load_row<DO>(a.mask + rr * DO, g, live);
    const float wt = (a.tw && live) ? a.tw[rr / a.B] : 1.0f;
    // forward
    const RmVec z1 = W1.mul(hv, bias1);
    RmVec av;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      av.t0[r] = z1.t0[r] > 0.f ? z1.t0[r] : exp_f32(z1.t0[r]) - 1.0f;  // ELU, alpha = 1 (slots past DH: ELU(0) = 0)
      av.t1[r] = z1.t1[r] > 0.f ? z1.t1[r] : exp_f32(z1.t1[r]) - 1.0f;
    }
    const RmVec xh = W2.mul(av, bias2);
    RmVec gx;
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d0 = xv.t0[r] - xh.t0[r], d1 = xv.t1[r] - xh.t1[r];  // slots past DO: x = 0 (not loaded), x_hat = 0
      const float e0 = d0 * mv.t0[r] * wt, e1 = d1 * mv.t1[r] * wt;    // mask = 0 there and in rows past R
      lsum = __builtin_fmaf(e0, d0, lsum);
      lsum = __builtin_fmaf(e1, d1, lsum);
      gx.t0[r] = -2.0f * a.scale * e0;
      gx.t1[r] = -2.0f * a.scale * e1;
    }
    lik += lsum;
    if constexpr (GRAD) {
      const RmVec ga = W2T.mul(gx, zero);
      RmVec gz;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gz.t0[r] = ga.t0[r] * (z1.t0[r] > 0.f ? 1.0f : av.t0[r] + 1.0f);  // ELU'(z) = exp(z) = a + 1 for z <= 0
        gz.t1[r] = ga.t1[r] * (z1.t1[r] > 0.f ? 1.0f : av.t1[r] + 1.0f);
      }
      const RmVec gh = W1T.mul(gz, zero);
      rm_store_row<DL>(a.grad_h + rr * DL, g, gh, live);
      // ---- weight gradients: contract over the 16 rows (patient-major images, lane (m, kk) reads [4c + kk][16 t + m])
      RmVec ab = av, hb = hv;
      if (g == 3) {  // storage slot 31 (tile 1, g = 3, r = 3) holds no component of a (21) or h (20): the bias column
        ab.t1[3] = 1.0f;
        hb.t1[3] = 1.0f;
      }
      __syncthreads();
      *reinterpret_cast<rv4*>(GX + n * P + 4 * g) = gx.t0;
      *reinterpret_cast<rv4*>(GX + n * P + 16 + 4 * g) = gx.t1;
      *reinterpret_cast<rv4*>(AI + n * P + 4 * g) = ab.t0;
      *reinterpret_cast<rv4*>(AI + n * P + 16 + 4 * g) = ab.t1;
      *reinterpret_cast<rv4*>(GZ + n * P + 4 * g) = gz.t0;
      *reinterpret_cast<rv4*>(GZ + n * P + 16 + 4 * g) = gz.t1;
      *reinterpret_cast<rv4*>(HI + n * P + 4 * g) = hb.t0;
      *reinterpret_cast<rv4*>(HI + n * P + 16 + 4 * g) = hb.t1;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int o = (4 * c + g) * P + n;
        const float gx0 = GX[o], gx1 = GX[o + 16], a0 = AI[o], a1 = AI[o + 16];
        const float gz0 = GZ[o], gz1 = GZ[o + 16], h0 = HI[o], h1 = HI[o + 16];
        dW2[0] = rmfma(gx0, a0, dW2[0]);
        dW2[1] = rmfma(gx0, a1, dW2[1]);
        dW2[2] = rmfma(gx1, a0, dW2[2]);
        dW2[3] = rmfma(gx1, a1, dW2[3]);
        dW1[0] = rmfma(gz0, h0, dW1[0]);
        dW1[1] = rmfma(gz0, h1, dW1[1]);
        dW1[2] = rmfma(gz1, h0, dW1[2]);
        dW1[3] = rmfma(gz1, h1, dW1[3]);
      }
    }
  }
  float* out = a.partials + (size_t)blockIdx.x * Cfg::NP;
  if constexpr (GRAD) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<rv4*>(out + ((size_t)i * 64 + lane) * 4) = dW2[i];
      *reinterpret_cast<rv4*>(out + ((size_t)(4 + i) * 64 + lane) * 4) = dW1[i];
    }
  }
  const float ls = wave_sum(lik);
  if (lane == 0) out[8 * 256] = ls;
}

// the data layout stores slot (tile 1, 4g + r) in image column 16 + 4g + r; component of image column j of tile `tile`
HODE_DEV int rm_col_comp(int tile, int j) { return rm_comp(tile, j >> 2, j & 3); }

template <int DL, int DO>
__global__ __launch_bounds__(64) void readout_mlp_fold_kernel(const float* __restrict__ partials, int n_waves, int j0, float* __restrict__ lik,
                                                              float* __restrict__ gw1, float* __restrict__ gb1,
                                                              float* __restrict__ gw2, float* __restrict__ gb2) {
  typedef ReadoutMlp<DL, DO> Cfg;
  constexpr int DH = Cfg::DH, NP = Cfg::NP;
  const int j = blockIdx.x + j0, lane = threadIdx.x;  // loss only: one block at j0 = the lik slot
  float s = 0.f;
  for (int w = lane; w < n_waves; w += 64) s += partials[(size_t)w * NP + j];
  s = wave_sum(s);
  if (lane != 0) return;
  if (j == 8 * 256) {
    lik[0] = s;
    return;
  }
  if (j > 8 * 256) return;
  const int tile = j / 256, l = (j % 256) / 4, r = j % 4;
  const int which = tile >> 2, mt = (tile >> 1) & 1, st = tile & 1;
  const int oc = rm_pos_comp(mt, 4 * (l >> 4) + r);   // C row 4 g' + r' of output tile mt
  const int jc = l & 15;                              // C column = image column within input tile st
  const bool is_bias = st == 1 && jc == 15;           // slot 31: the ones entry
  const int ic = rm_col_comp(st, jc);
  if (which == 0) {  // dW2 [DO][DH], db2
    if (oc >= DO) return;
    if (is_bias) { if (gb2) gb2[oc] += s; }
    else if (ic < DH && gw2) gw2[(size_t)oc * DH + ic] += s;
  } else {           // dW1 [DH][DL], db1
    if (oc >= DH) return;
    if (is_bias) { if (gb1) gb1[oc] += s; }
    else if (ic < DL && gw1) gw1[(size_t)oc * DL + ic] += s;
  }
}

}  // namespace hode

namespace {
constexpr int kRmWaves = 2048;  // grid-stride: two waves per SIMD keep the row loads of the next tile in flight

int rm_waves(long long rows) {
  const long long tiles = (rows + 15) / 16;
  return (int)(tiles < kRmWaves ? (tiles > 0 ? tiles : 1) : kRmWaves);
}
bool rm_supported(const hode_readout_mlp_desc* d) {
  return d->hidden_dim == d->latent_dim + 1 && d->obs_dim == 24 && (d->latent_dim == 20 || d->latent_dim == 4);
}
}  // namespace

extern "C" size_t hode_readout_mlp_workspace_bytes(const hode_readout_mlp_desc* d) {
  if (!d || d->struct_size != sizeof(hode_readout_mlp_desc) || !rm_supported(d)) return 0;
  return (size_t)rm_waves(d->rows) * hode::ReadoutMlp<20, 24>::NP * sizeof(float);
}

extern "C" int hode_readout_mlp_sse(const hode_readout_mlp_desc* d, void* stream) {
  if (!d) return hode::fail(HODE_E_NULL, "descriptor is NULL");
  if (d->struct_size != sizeof(hode_readout_mlp_desc)) return hode::fail(HODE_E_SIZE, "struct_size mismatch (ABI)");
  if (d->rows <= 0 || d->batch <= 0) return hode::fail(HODE_E_SIZE, "bad sizes rows=%lld batch=%d", (long long)d->rows, d->batch);
  if (!rm_supported(d))
    return hode::fail(HODE_E_UNSUPPORTED, "readout MLP: (latent %d, hidden %d, obs %d) has no compiled kernel (have latent 20 / 4, "
                      "hidden = latent + 1, obs 24: DecoderReal, model.py:809-813)", d->latent_dim, d->hidden_dim, d->obs_dim);
  if (!d->h || !d->x || !d->mask || !d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->lik)
    return hode::fail(HODE_E_NULL, "h / x / mask / w1 / b1 / w2 / b2 / lik must be non-NULL");
  if (((uintptr_t)d->x | (uintptr_t)d->mask | (uintptr_t)d->h | (uintptr_t)d->grad_h) & 15)
    return hode::fail(HODE_E_ALIGN, "h / x / mask / grad_h must be 16-byte aligned");
  const size_t need = hode_readout_mlp_workspace_bytes(d);
  if (!d->workspace || d->workspace_bytes < need) return hode::fail(HODE_E_WORKSPACE, "workspace %zu B < required %zu B", d->workspace_bytes, need);
  const bool grad = d->grad_h != nullptr;
  if (grad && (!d->grad_w1 || !d->grad_b1 || !d->grad_w2 || !d->grad_b2))
    return hode::fail(HODE_E_NULL, "grad_h given: grad_w1 / grad_b1 / grad_w2 / grad_b2 (accumulators) are required too");
  hode::ReadoutMlpArgs a{};
  a.h = d->h; a.x = d->x; a.mask = d->mask; a.tw = d->time_weight; a.w1 = d->w1; a.b1 = d->b1; a.w2 = d->w2; a.b2 = d->b2;
  a.grad_h = d->grad_h; a.partials = (float*)d->workspace; a.R = d->rows; a.B = d->batch; a.scale = d->scale;
  const int nw = rm_waves(d->rows);
  hipStream_t s = (hipStream_t)stream;
#define HODE_RM(DL)                                                                                                   \
  {                                                                                                                     \
    if (grad) hipLaunchKernelGGL((hode::readout_mlp_kernel<DL, 24, true>), dim3(nw), dim3(64), 0, s, a);               \
    else hipLaunchKernelGGL((hode::readout_mlp_kernel<DL, 24, false>), dim3(nw), dim3(64), 0, s, a);                   \
    if (int e = hode::hip_fail(hipGetLastError(), "readout_mlp launch")) return e;                                      \
    hipLaunchKernelGGL((hode::readout_mlp_fold_kernel<DL, 24>), dim3(grad ? 8 * 256 + 1 : 1), dim3(64), 0, s,           \
                       (const float*)d->workspace, nw, grad ? 0 : 8 * 256, d->lik, d->grad_w1, d->grad_b1, d->grad_w2, d->grad_b2); \
  }
  if (d->latent_dim == 20) HODE_RM(20) else HODE_RM(4)
  return hode::hip_fail(hipGetLastError(), "readout_mlp fold launch");
}
