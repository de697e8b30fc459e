// Instantiates the dopri5 kernels for ONE latent dimension (-DHODE_DIM=<D>), see hode_rk_dim.hip.
#include "hode_dopri5_kernels.hpp"

#ifndef HODE_DIM
#error "compile with -DHODE_DIM=<latent dim>"
#endif
#define HODE_CAT_(a, b) a##b
#define HODE_CAT(a, b) HODE_CAT_(a, b)

namespace hode {
int HODE_CAT(dp_dispatch_d, HODE_DIM)(const DpLaunch& L, const DpArgs& a, hipStream_t s) {
  return dp_dispatch<HODE_DIM>(L, a, s);
}
}  // namespace hode
