// Instantiates the fixed-grid Roche kernels for ONE latent dimension; compiled once per -DHODE_DIM=<D> so the
// dimensions build in parallel (see __graft_entry__.build).
#include "hode_rk_kernels.hpp"

#ifndef HODE_DIM
#error "compile with -DHODE_DIM=<latent dim>"
#endif

#define HODE_CAT_(a, b) a##b
#define HODE_CAT(a, b) HODE_CAT_(a, b)

namespace hode {
int HODE_CAT(rk_dispatch_d, HODE_DIM)(const RkLaunch& L, const RkArgs& a, hipStream_t s) {
  return dispatch_lpp<HODE_DIM>(L, a, s);
}
}  // namespace hode
