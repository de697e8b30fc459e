// Host side of the masked LSTM encoder kernels (csrc/hode_lstm_kernels.hpp): weight packing, launch geometry, the
// hode_lstm_* entry points of include/hode.h.  The kernels themselves are instantiated per padded hidden size in
// csrc/hode_lstm_tpw.hip.
#include "hode_lstm_kernels.hpp"

namespace hode {

// ---- weight / bias packing (once per forward: the optimiser changes the weights every step).  The bias b_ih + b_hh is
// operand row k = I + Hp of Wcat: the activation buffers hold a row of ones there, so the accumulators start from zero
// and no register holds a bias.
__global__ void lstm_pack_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                 const float* __restrict__ b_ih, const float* __restrict__ b_hh, float* __restrict__ wp,
                                 int I, int H, int TPW, int NW, int KQ4) {
  const long long n_w = (long long)NW * KQ4 * TPW * 64 * 4;   // NW waves x TPW tiles x (4 units x 4 gates)
  const int Hp = 4 * TPW * NW;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n_w; idx += (long long)gridDim.x * blockDim.x) {
    const int kk = idx & 3;
    const int l = (idx >> 2) & 63;
    long long r = idx >> 8;
    const int tau = (int)(r % TPW); r /= TPW;
    const int kq4 = (int)(r % KQ4);
    const int w = (int)(r / KQ4);
    const int i = l & 15;
    const int u = (w * TPW + tau) * 4 + (i >> 2);
    const int gate = i & 3;
    const int k = 4 * (4 * kq4 + kk) + (l >> 4);
    float v = 0.f;
    if (u < H) {
      const int row = gate * H + u;
      if (k < I) v = w_ih[(size_t)row * I + k];
      else if (k - I < H) v = w_hh[(size_t)row * H + (k - I)];
      else if (k == I + Hp) v = b_ih[row] + b_hh[row];
    }
    wp[idx] = v;
  }
}

// ---- forward
// W_hh^T fragments, one 16-byte group per (wave, contracted tile tau, gate row r, group j of 4 output tiles, lane):
// element i of the group is the A-operand value of output tile mt = 4j + i.  Gate-row-major so that the kernel streams
// 4 * ceil(TPW/4) registers per 30-MFMA block instead of holding all 4 * TPW of a tile.
__global__ void lstm_pack_hh_kernel(const float* __restrict__ w_hh, float* __restrict__ whp, int H, int TPW) {
  const int MG = (TPW + 3) / 4;
  const long long n = 4LL * TPW * 4 * MG * 64 * 4;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    const int i = idx & 3;
    const int l = (idx >> 2) & 63;
    long long q = idx >> 8;
    const int j = (int)(q % MG); q /= MG;
    const int r = (int)(q & 3); q >>= 2;
    const int tau = (int)(q % TPW);
    const int w = (int)(q / TPW);
    const int mt = 4 * j + i;
    const int u = (w * TPW + tau) * 4 + (l >> 4);   // contracted unit (its gate r)
    const int uo = 16 * mt + (l & 15);              // output unit
    whp[idx] = (mt < TPW && u < H && uo < H) ? w_hh[((size_t)r * H + u) * H + uo] : 0.f;
  }
}

}  // namespace hode

// ====================================================================================================== host
namespace {

using hode::LstmArgs;
using hode::LstmGeom;

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// patients per workgroup: the smallest cost rounds(blocks over 256 CUs) * NT, ties to the larger tile
int choose_nt(int B, bool bwd) {
  int best = 1;
  long long best_cost = -1;
  for (int nt = 1; nt <= (bwd ? 3 : 4); ++nt) {
    const long long blocks = (B + 16 * nt - 1) / (16 * nt);
    const long long cost = ((blocks + 255) / 256) * nt;
    if (best_cost < 0 || cost <= best_cost) { best = nt; best_cost = cost; }
  }
  return best;
}

int lstm_geom(const hode_lstm_desc* d, LstmGeom* G, bool bwd_compatible) {
  // padded hidden size: the next compiled one (hidden units past H are zero rows / columns of the packed weights and
  // stay exactly 0 through the recurrence)
  static const int kSizes[] = {16, 32, 48, 64, 80, 96, 128, 160};
  int Hp = 0;
  for (int v : kSizes)
    if (d->hidden_dim <= v) { Hp = v; break; }
  if (!Hp)
    return hode::fail(HODE_E_UNSUPPORTED, "lstm: hidden_dim %d exceeds the largest compiled kernel (160)", d->hidden_dim);
  G->Hp = Hp;
  G->TPW = Hp / 16;
  G->fNW = G->TPW <= 5 ? 4 : 8;
  G->fTPW = G->TPW * 4 / G->fNW;
  G->NT = choose_nt(d->batch, bwd_compatible);
  if (const char* env = getenv("HODE_LSTM_NT")) {  // tuning / test override of the patient tile (16 * NT)
    const int v = atoi(env);
    if (v >= 1 && v <= (bwd_compatible ? 3 : 4)) G->NT = v;
  }
  G->BT = 16 * G->NT;
  G->nblk = (d->batch + G->BT - 1) / G->BT;
  G->Kq = (d->input_dim + Hp + 1 + 3) / 4;   // + the ones row that carries the bias
  G->KQ4 = (G->Kq + 3) / 4;
  G->LD = G->BT + ((G->BT % 32 == 0) ? 16 : 0);
  G->wp_floats = (size_t)4 * G->KQ4 * G->TPW * 64 * 4;
  G->tape_floats = (size_t)d->seq_len * G->nblk * 4 * G->TPW * G->NT * 5 * 64;
  G->lds_bytes = (size_t)2 * 4 * G->Kq * G->LD * sizeof(float);
  if (G->lds_bytes > 160 * 1024)
    return hode::fail(HODE_E_UNSUPPORTED, "lstm: activation tile needs %zu B of LDS (> 160 KiB)", G->lds_bytes);
  while (G->NT > 1 && (size_t)16 * G->NT * d->obs_dim > 5120) --G->NT;
  G->BT = 16 * G->NT;
  G->nblk = (d->batch + G->BT - 1) / G->BT;
  G->LD = G->BT + ((G->BT % 32 == 0) ? 16 : 0);
  G->tape_floats = (size_t)d->seq_len * G->nblk * 4 * G->TPW * G->NT * 5 * 64;
  G->lds_bytes = (size_t)2 * 4 * G->Kq * G->LD * sizeof(float);
  G->whp_floats = (size_t)4 * G->TPW * 4 * ((G->TPW + 3) / 4) * 64 * 4;
  {
    const size_t tile = (size_t)G->BT * (4 * Hp + 4), slabs = (size_t)4 * Hp * G->LD;
    G->lds_bwd_bytes = ((tile > slabs ? tile : slabs) + (size_t)G->BT * (Hp + 4)) * sizeof(float);
  }
  if ((size_t)G->BT * d->obs_dim > 5120)
    return hode::fail(HODE_E_UNSUPPORTED, "lstm: obs_dim %d too wide for the staging registers", d->obs_dim);
  return 0;
}

int check_lstm(const hode_lstm_desc* d) {
  if (!d) return hode::fail(HODE_E_NULL, "descriptor is NULL");
  if (d->struct_size != sizeof(hode_lstm_desc))
    return hode::fail(HODE_E_SIZE, "struct_size %u != %zu (ABI mismatch)", d->struct_size, sizeof(hode_lstm_desc));
  if (d->seq_len <= 0 || d->batch <= 0 || d->input_dim <= 0 || d->hidden_dim <= 0 || d->obs_dim <= 0 ||
      d->obs_dim > d->input_dim)
    return hode::fail(HODE_E_SIZE, "bad sizes: T=%d B=%d I=%d H=%d obs=%d", d->seq_len, d->batch, d->input_dim,
                      d->hidden_dim, d->obs_dim);
  if (!d->x || !d->w_ih || !d->w_hh || !d->b_ih || !d->b_hh || !d->h_out || !d->c_out)
    return hode::fail(HODE_E_NULL, "x / w_ih / w_hh / b_ih / b_hh / h_out / c_out must be non-NULL");
  if (d->input_dim > d->obs_dim && !d->a) return hode::fail(HODE_E_NULL, "a is required when input_dim > obs_dim");
  return 0;
}

int launch_fwd(const LstmGeom& G, const LstmArgs& a, hipStream_t s) {
  switch (G.TPW) {
#define HODE_CASE(n) case n: return hode::lstm_fwd_tpw##n(G, a, s);
    HODE_CASE(1) HODE_CASE(2) HODE_CASE(3) HODE_CASE(4) HODE_CASE(5) HODE_CASE(6) HODE_CASE(8) HODE_CASE(10)
#undef HODE_CASE
  }
  return hode::fail(HODE_E_UNSUPPORTED, "lstm_fwd: no kernel for %d tiles", G.TPW);
}

}  // namespace

// workspace: [packed Wcat (bias row included) | packed W_hh^T (tape runs only) | tape (tape runs only)]
extern "C" size_t hode_lstm_workspace_bytes(const hode_lstm_desc* d) {
  LstmGeom G;
  if (!d || d->struct_size != sizeof(hode_lstm_desc) || lstm_geom(d, &G, d->save_tape != 0)) return 0;
  size_t n = align256(G.wp_floats * 4);
  if (d->save_tape) n += align256(G.whp_floats * 4) + align256(G.tape_floats * 4);
  return n;
}

extern "C" int hode_lstm_fwd(const hode_lstm_desc* d, void* stream) {
  if (int e = check_lstm(d)) return e;
  LstmGeom G;
  if (int e = lstm_geom(d, &G, d->save_tape != 0)) return e;
  const size_t need = hode_lstm_workspace_bytes(d);
  if (!d->workspace || d->workspace_bytes < need)
    return hode::fail(HODE_E_WORKSPACE, "workspace %zu B < required %zu B", d->workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)d->workspace;
  float* wp = (float*)ws;
  float* tape = d->save_tape ? (float*)(ws + align256(G.wp_floats * 4) + align256(G.whp_floats * 4)) : nullptr;
  hipLaunchKernelGGL(hode::lstm_pack_kernel, dim3(256), dim3(256), 0, s, d->w_ih, d->w_hh, d->b_ih, d->b_hh, wp,
                     d->input_dim, d->hidden_dim, G.fTPW, G.fNW, G.KQ4);  // the forward kernel's waves x tiles
  if (int e = hode::hip_fail(hipGetLastError(), "lstm_pack launch")) return e;
  LstmArgs a{};
  a.x = d->x; a.a = d->a; a.mask = d->mask; a.wp = wp; a.h_out = d->h_out; a.c_out = d->c_out; a.tape = tape;
  a.T = d->seq_len; a.B = d->batch; a.OBS = d->obs_dim; a.AD = d->input_dim - d->obs_dim; a.I = d->input_dim;
  a.H = d->hidden_dim; a.Hp = G.Hp; a.Kq = G.Kq; a.KQ4 = G.KQ4; a.LD = G.LD; a.reverse = d->reverse;
#ifdef HODE_LSTM_STAMPS
  if (const char* env = getenv("HODE_LSTM_FWD_DBG_PTR")) a.dbg = (unsigned long long*)strtoull(env, nullptr, 0);
#endif
  return launch_fwd(G, a, s);
}

namespace {

int launch_bwd(const LstmGeom& G, const hode::LstmBwdArgs& a, hipStream_t s) {
  switch (G.TPW) {
#define HODE_CASE(n) case n: return hode::lstm_bwd_tpw##n(G, a, s);
    HODE_CASE(1) HODE_CASE(2) HODE_CASE(3) HODE_CASE(4) HODE_CASE(5) HODE_CASE(6) HODE_CASE(8) HODE_CASE(10)
#undef HODE_CASE
  }
  return hode::fail(HODE_E_UNSUPPORTED, "lstm_bwd: no kernel for TPW %d", G.TPW);
}

}  // namespace

// Backward of hode_lstm_fwd(save_tape = 1) with the SAME descriptor sizes and workspace: fills grad_gates[T][B][4H]
// and the GEMM operand h_prev[T][B][W] = [obs_dim columns the caller fills with x*mask | action columns | hidden state
// entering the step | 1 | 0-pad], W = roundup4(I + H + 1); ONE product grad_gates^T h_prev over K = T*B then is
// [grad_w_ih | grad_w_hh | grad_b | 0]: dG is read once, the N dimension fills the BLAS tile (244 of 256 instead of 80 of
// 128 and 162 of 256: 3.4 -> 2.4 ms) and the bias sum needs no pass of its own.  (Writing x*mask from this kernel was
// tried: +0.55 ms here against 0.15 ms for the caller's element-wise kernel.)
extern "C" int hode_lstm_bwd(const hode_lstm_desc* d, void* stream) {
  if (int e = check_lstm(d)) return e;
  if (!d->save_tape) return hode::fail(HODE_E_UNSUPPORTED, "hode_lstm_bwd needs the tape of a forward run with save_tape = 1");
  if (!d->grad_h_out || !d->grad_gates || !d->h_prev)
    return hode::fail(HODE_E_NULL, "grad_h_out / grad_gates / h_prev must be non-NULL");
  LstmGeom G;
  if (int e = lstm_geom(d, &G, true)) return e;
  const size_t need = hode_lstm_workspace_bytes(d);
  if (!d->workspace || d->workspace_bytes < need)
    return hode::fail(HODE_E_WORKSPACE, "workspace %zu B < required %zu B", d->workspace_bytes, need);
  if (G.lds_bwd_bytes > 160 * 1024)
    return hode::fail(HODE_E_UNSUPPORTED, "lstm_bwd: tile needs %zu B of LDS (> 160 KiB)", G.lds_bwd_bytes);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)d->workspace;
  float* whp = (float*)(ws + align256(G.wp_floats * 4));
  const float* tape = (const float*)((char*)whp + align256(G.whp_floats * 4));
  hipLaunchKernelGGL(hode::lstm_pack_hh_kernel, dim3(128), dim3(256), 0, s, d->w_hh, whp, d->hidden_dim, G.TPW);
  if (int e = hode::hip_fail(hipGetLastError(), "lstm_pack_hh launch")) return e;
  hode::LstmBwdArgs a{};
  a.tape = tape; a.whp = whp; a.grad_h_out = d->grad_h_out; a.grad_gates = d->grad_gates; a.h_prev = d->h_prev;
  a.T = d->seq_len; a.B = d->batch; a.H = d->hidden_dim; a.Hp = G.Hp; a.LD = G.LD; a.reverse = d->reverse;
  a.a = d->a; a.AD = d->input_dim - d->obs_dim;
  a.OBS = d->obs_dim; a.W = (d->input_dim + d->hidden_dim + 1 + 3) / 4 * 4;
#ifdef HODE_LSTM_STAMPS
  if (const char* env = getenv("HODE_LSTM_DBG_PTR")) a.dbg = (unsigned long long*)strtoull(env, nullptr, 0);
#endif
  return launch_bwd(G, a, s);
}

// The first obs_dim columns of the weight-gradient GEMM operand rows, h_prev[t][b][0 .. obs) = x[t][b][:] * mask[t][b][:] (x
// itself without a mask): the one part of the rows that does not depend on the recurrence.  A pure streaming pass -- 16-byte
// loads and stores, a workgroup per run of rows -- that the caller launches on a second stream beside hode_lstm_bwd; as a
// framework expression (a product into a strided 80-of-244-column view) it ran as six non-vectorised launches, 1.07 ms of device time
// at the bench shape against 0.2 ms of traffic.
namespace hode {
template <bool VEC4>
__global__ __launch_bounds__(256) void lstm_fill_operand_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                                float* __restrict__ hp, long long rows, int obs, int W) {
  if constexpr (VEC4) {
    const int q = obs >> 2;   // 16-byte groups per row
    const long long n = rows * q;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
      const long long r = e / q;
      const int c = (int)(e - r * q);
      f32x4 v = *reinterpret_cast<const f32x4*>(x + r * obs + 4 * c);
      if (mask) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mask + r * obs + 4 * c);
        v[0] *= m[0]; v[1] *= m[1]; v[2] *= m[2]; v[3] *= m[3];
      }
      *reinterpret_cast<f32x4*>(hp + r * W + 4 * c) = v;
    }
  } else {
    const long long n = rows * obs;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
      const long long r = e / obs;
      const int c = (int)(e - r * obs);
      hp[r * W + c] = mask ? x[e] * mask[e] : x[e];
    }
  }
}
}  // namespace hode

extern "C" int hode_lstm_fill_operand(const hode_lstm_desc* d, void* stream) {
  if (int e = check_lstm(d)) return e;
  if (!d->h_prev) return hode::fail(HODE_E_NULL, "h_prev must be non-NULL");
  const int obs = d->obs_dim, W = (d->input_dim + d->hidden_dim + 1 + 3) / 4 * 4;
  const long long rows = (long long)d->seq_len * d->batch;
  const bool vec4 = (obs & 3) == 0 && !(((uintptr_t)d->x | (uintptr_t)d->h_prev | (uintptr_t)(d->mask ? d->mask : d->x)) & 15);
  const long long work = vec4 ? rows * (obs >> 2) : rows * obs;
  const int grid = (int)std::min<long long>((work + 255) / 256, 256 * 16);
  if (grid <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (vec4) hipLaunchKernelGGL((hode::lstm_fill_operand_kernel<true>), dim3(grid), dim3(256), 0, s, d->x, d->mask, d->h_prev, rows, obs, W);
  else hipLaunchKernelGGL((hode::lstm_fill_operand_kernel<false>), dim3(grid), dim3(256), 0, s, d->x, d->mask, d->h_prev, rows, obs, W);
  return hode::hip_fail(hipGetLastError(), "lstm_fill_operand launch");
}
