// Test helper: the PRODUCT's host-side EPnP refit (vdo_slam_amd/csrc/epnp_refit.hpp, plain C++) behind a C entry point, so that
// it can be compared with the independent oracle restatement (oracle/epnp_oracle.hpp) on the CPU, without a GPU.
#include "../../vdo_slam_amd/csrc/epnp_refit.hpp"

extern "C" double product_host_epnp(int n, const double* X, const double* uv, const double* K4, double* T_out) {
  vdo::epnp::Scratch scr;
  const vdo::epnp::Result r = vdo::epnp::solve(n, X, uv, K4, scr);
  for (int i = 0; i < 16; ++i) T_out[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T_out[4 * i + j] = r.R[3 * i + j]; T_out[4 * i + 3] = r.t[i]; }
  return r.err;
}
