// chz_launch.h -- runtime (R1,R2) -> compiled kernel instantiation dispatch.
// Shared by chz_engine.hip (hipcc, real launches on a HIP stream) and the CPU
// test harness (tests/hipemu), so both exercise the same template instances.
#pragma once
#include "chz_kernels.h"
#include "chz_plan.h"

namespace chz {

inline int launch_first_real(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const FirstRealParams& p) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { hipLaunchKernelGGL((fwd_first_real<a, b>), dim3(grid), dim3(block), lds, s, p); return 0; }
  CHZ_FWD_MENU(X)
#undef X
  return -1;
}
inline int launch_cols(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const ColsParams& p) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { hipLaunchKernelGGL((fwd_cols<a, b>), dim3(grid), dim3(block), lds, s, p); return 0; }
  CHZ_FWD_MENU(X)
#undef X
  return -1;
}
inline int launch_rows(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const RowsParams& p) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { hipLaunchKernelGGL((fwd_rows<a, b>), dim3(grid), dim3(block), lds, s, p); return 0; }
  CHZ_FWD_MENU(X)
#undef X
  return -1;
}
inline int launch_chan(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const ChanParams& p) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { hipLaunchKernelGGL((chan_ifft<a, b>), dim3(grid), dim3(block), lds, s, p); return 0; }
  CHZ_CHAN_MENU(X)
#undef X
  return -1;
}
}  // namespace chz
