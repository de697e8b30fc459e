// Device-side helpers shared by the hode kernels (gfx950 only: wave64, DPP, v_exp/v_rcp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HODE_DEV __device__ __forceinline__

namespace hode {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------------------------
// transcendental helpers.  Accuracy targets are stated per function and checked in tests/test_hip_math.py.
// ---------------------------------------------------------------------------------------------------------

// exp(x), x <= 0 in practice (dose decay kel*(tau - t)): hardware exp2 of the rounded product x*log2(e), corrected
// by the product's rounding residual (exact via fma).  <= 2 ulp for |x| < 80; 5 instructions.
HODE_DEV float exp_f32(float x) {
  const float l2e = 1.4426950408889634f;
  const float t = x * l2e;
  const float lo = __builtin_fmaf(x, l2e, -t);
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, lo * 0.6931471805599453f, e);
}

// natural log through the hardware log2 (only used by the Hill-exponent gradients)
HODE_DEV float log_f32(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// tanh(x) = 1 - 2 / (exp(2x) + 1): v_mul, v_exp, v_add, v_rcp, v_fma.  ABSOLUTE error <= 1.5e-7 over the whole
// line (checked against fp64 in tests/test_hip_math.py); the relative error grows for |x| << 1, which does not
// matter here because tanh feeds an additive rate dy/dt.  NaN propagates, +-inf -> +-1.
HODE_DEV float tanh_f32(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);  // exp(2x)
  return __builtin_fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
}

// tanh with ~1.3 ulp RELATIVE accuracy (odd polynomial below 0.625, exp form above); kept for the encoder gates
HODE_DEV float tanh_precise_f32(float x) {
  float ax = __builtin_fabsf(x);
  float u = x * x;
  float p = -0.005508354399353266f;
  p = __builtin_fmaf(p, u, 0.020461998879909515f);
  p = __builtin_fmaf(p, u, -0.05368518456816673f);
  p = __builtin_fmaf(p, u, 0.13330785930156708f);
  p = __builtin_fmaf(p, u, -0.3333325684070587f);
  float small = __builtin_fmaf(ax * u, p, ax);
  float e = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);
  float big = __builtin_fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
  float r = ax < 0.625f ? small : big;
  return __builtin_copysignf(r, x);
}

// logistic sigmoid via tanh: sigma(x) = 0.5 + 0.5 tanh(x/2)
HODE_DEV float sigmoid_f32(float x) { return __builtin_fmaf(tanh_f32(0.5f * x), 0.5f, 0.5f); }

// ---------------------------------------------------------------------------------------------------------
// packed fp32: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 retire TWO fp32 operations in the issue slot of one
// (tools/micro/issue_rates.hip: 2.28 ns per wave-instruction for v_fma_f32 and for v_pk_fma_f32 alike, dependent or
// not) -- the solver kernels are bound by issue slots, so the paired algebra is written with this type explicitly
// (the SLP vectoriser's automatic pairing costs more v_mov than it saves and is switched off for these files).
// ---------------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
HODE_DEV f2 splat2(float x) { f2 r = {x, x}; return r; }
HODE_DEV f2 pair2(float a, float b) { f2 r = {a, b}; return r; }
HODE_DEV float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
HODE_DEV f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
HODE_DEV f2 vfma(float a, f2 b, f2 c) { return __builtin_elementwise_fma(splat2(a), b, c); }
HODE_DEV f2 vfma(f2 a, float b, f2 c) { return __builtin_elementwise_fma(a, splat2(b), c); }
template <class V> HODE_DEV V vsplat(float x);
template <> HODE_DEV float vsplat<float>(float x) { return x; }
template <> HODE_DEV f2 vsplat<f2>(float x) { return splat2(x); }
HODE_DEV float hsum(f2 v) { return v.x + v.y; }
HODE_DEV f2 tanh_f32(f2 x) {  // the two transcendentals stay scalar, the three arithmetic steps are packed
  const f2 t = x * splat2(2.885390081777927f);
  const f2 e = pair2(__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)) + splat2(1.0f);
  return vfma(pair2(__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)), splat2(-2.0f), splat2(1.0f));
}
HODE_DEV bool vfinite(float v) { return __builtin_isfinite(v); }
HODE_DEV bool vfinite(f2 v) { return __builtin_isfinite(v.x) && __builtin_isfinite(v.y); }

// IEEE-correct-ish division (v_rcp + one Newton step; result within 1 ulp for normal operands)
HODE_DEV float div_f32(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  float q = a * r;
  float e = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(e, r, q);
}

// ---------------------------------------------------------------------------------------------------------
// cross-lane helpers for the "4 lanes per patient" layout: a patient occupies one DPP quad.
// ---------------------------------------------------------------------------------------------------------

// broadcast the value held by lane SRC (0..3) of each quad to all four lanes of the quad (one v_mov_dpp)
template <int SRC>
HODE_DEV float quad_bcast(float v) {
  constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);  // quad_perm:[SRC,SRC,SRC,SRC]
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}

// sum over the four lanes of each quad, result in all four lanes (two DPP adds)
HODE_DEV float quad_sum(float v) {
  // quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E
  float a = v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  return a + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0x4E, 0xf, 0xf, true));
}

// DPP row rotations (a "row" = 16 lanes): plain VALU adds with a DPP operand, no LDS crossbar round trip like __shfl_xor
template <int CTRL>
HODE_DEV float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 4 lanes of a row that share (lane & 3), result in every lane of the row   (row_ror:4, row_ror:8)
HODE_DEV float row_sum_stride4(float v) {
  v += dpp_f32<0x124>(v);
  v += dpp_f32<0x128>(v);
  return v;
}
// sum over the 16 lanes of a row, result in every lane of the row   (row_ror:1, 2, 4, 8)
HODE_DEV float row_sum(float v) {
  v += dpp_f32<0x121>(v);
  v += dpp_f32<0x122>(v);
  v += dpp_f32<0x124>(v);
  v += dpp_f32<0x128>(v);
  return v;
}

// sum across the lanes of a wave that hold the same quad position (xor over lane bits 2..5), result everywhere
HODE_DEV float wave_sum_stride4(float v) {
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// full wave sum, result in every lane: four DPP row rotations, then the four row sums through SGPRs.  (Six __shfl_xor
// steps are six dependent LDS-crossbar round trips; a dopri5 attempt does two such sums on its critical path,
// DESIGN.md section 5.)
HODE_DEV float wave_sum(float v) {
  v = row_sum(v);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

HODE_DEV float nextafter_up(float x) {
  // nextafter(x, +inf) for finite x (torchdiffeq Perturb.NEXT, oracle/solvers.py::_nextafter)
  if (x == 0.0f) return __builtin_bit_cast(float, 1u);
  uint32_t u = __builtin_bit_cast(uint32_t, x);
  return __builtin_bit_cast(float, x > 0.0f ? u + 1u : u - 1u);
}
HODE_DEV float nextafter_down(float x) {
  if (x == 0.0f) return __builtin_bit_cast(float, 0x80000001u);
  uint32_t u = __builtin_bit_cast(uint32_t, x);
  return __builtin_bit_cast(float, x > 0.0f ? u - 1u : u + 1u);
}

// fp32 ops that must NOT be contracted into an fma (stage times are compared against dose times with >= / ==,
// so they have to round exactly like the reference's separate mul and add)
HODE_DEV float mul_rn(float a, float b) { return __fmul_rn(a, b); }
HODE_DEV float add_rn(float a, float b) { return __fadd_rn(a, b); }

}  // namespace hode
