// Instantiates the LSTM window / BPTT kernels for ONE padded hidden size Hp = 16 * HODE_LSTM_TPW (the backward runs 4
// waves x TPW unit tiles; the forward TPW <= 5: 4 waves x TPW tiles, above: 8 waves x TPW/2 tiles -- two waves per SIMD, all
// accumulators in the VGPR file).  Compiled once per -DHODE_LSTM_TPW=<n> so the sizes build in parallel (build_hip.py).
#include "hode_lstm_kernels.hpp"

#ifndef HODE_LSTM_TPW
#error "compile with -DHODE_LSTM_TPW=<unit tiles per wave of the backward>"
#endif

#define HODE_CAT_(a, b) a##b
#define HODE_CAT(a, b) HODE_CAT_(a, b)

namespace {

using hode::LstmArgs;
using hode::LstmGeom;

constexpr int kTPW = HODE_LSTM_TPW;
constexpr int kFwdNW = kTPW <= 5 ? 4 : 8;
constexpr int kFwdTPW = kTPW <= 5 ? kTPW : kTPW / 2;
static_assert(kFwdTPW * kFwdNW == kTPW * 4, "forward geometry must cover the padded hidden size");

template <int NT, bool VEC4>
int launch_fwd_vec(const LstmGeom& G, const LstmArgs& a, hipStream_t s) {
  if (int e = hode::hip_fail(hipFuncSetAttribute((const void*)hode::lstm_fwd_kernel<NT, kFwdTPW, kFwdNW, VEC4>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)G.lds_bytes),
                             "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"))
    return e;
  hipLaunchKernelGGL((hode::lstm_fwd_kernel<NT, kFwdTPW, kFwdNW, VEC4>), dim3(G.nblk), dim3(64 * kFwdNW), G.lds_bytes, s, a);
  return hode::hip_fail(hipGetLastError(), "lstm_fwd launch");
}

template <int NT>
int launch_fwd_one(const LstmGeom& G, const LstmArgs& a, hipStream_t s) {
  return (a.OBS & 3) == 0 ? launch_fwd_vec<NT, true>(G, a, s) : launch_fwd_vec<NT, false>(G, a, s);
}

template <int NT, bool FLAT>
int launch_bwd_flat(const LstmGeom& G, const hode::LstmBwdArgs& a, hipStream_t s) {
  if (int e = hode::hip_fail(hipFuncSetAttribute((const void*)hode::lstm_bwd_kernel<NT, kTPW, FLAT>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)G.lds_bwd_bytes),
                             "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"))
    return e;
  hipLaunchKernelGGL((hode::lstm_bwd_kernel<NT, kTPW, FLAT>), dim3(G.nblk), dim3(256), G.lds_bwd_bytes, s, a);
  return hode::hip_fail(hipGetLastError(), "lstm_bwd launch");
}

template <int NT>
int launch_bwd_one(const LstmGeom& G, const hode::LstmBwdArgs& a, hipStream_t s) {
  return a.H == 16 * kTPW ? launch_bwd_flat<NT, true>(G, a, s) : launch_bwd_flat<NT, false>(G, a, s);
}

}  // namespace

namespace hode {

int HODE_CAT(lstm_fwd_tpw, HODE_LSTM_TPW)(const LstmGeom& G, const LstmArgs& a, hipStream_t s) {
  if (G.fTPW != kFwdTPW || G.fNW != kFwdNW) return fail(HODE_E_UNSUPPORTED, "lstm_fwd: geometry %d x %d is not this unit's", G.fNW, G.fTPW);
  switch (G.NT) {
    case 1: return launch_fwd_one<1>(G, a, s);
    case 2: return launch_fwd_one<2>(G, a, s);
    case 3: return launch_fwd_one<3>(G, a, s);
    case 4: return launch_fwd_one<4>(G, a, s);
  }
  return fail(HODE_E_UNSUPPORTED, "lstm_fwd: NT %d", G.NT);
}

int HODE_CAT(lstm_bwd_tpw, HODE_LSTM_TPW)(const LstmGeom& G, const LstmBwdArgs& a, hipStream_t s) {
  switch (G.NT) {
    case 1: return launch_bwd_one<1>(G, a, s);
    case 2: return launch_bwd_one<2>(G, a, s);
    case 3: return launch_bwd_one<3>(G, a, s);
  }
  return fail(HODE_E_UNSUPPORTED, "lstm_bwd: NT %d", G.NT);
}

}  // namespace hode
