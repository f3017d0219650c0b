// chz_launch.h -- runtime (R1,R2) -> compiled kernel instantiation dispatch.
// Shared by chz_engine.hip (hipcc, real launches on a HIP stream) and the CPU
// test harness (tests/hipemu), so both exercise the same template instances.
#pragma once
#include "chz_kernels.h"
#include "chz_plan.h"

namespace chz {

// Plain launch, or -- when an event pair is supplied -- hipExtLaunchKernelGGL, whose events carry
// the dispatch packet's own begin/end timestamps (the same clock rocprofv3 --kernel-trace reads), so
// per-kernel times measured in-process agree with the profiler.
#define CHZ_LAUNCH(kern, grid, block, lds, s, ev0, ev1, p)                                              \
  do {                                                                                                  \
    if (ev0) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (unsigned)(lds), s, ev0, ev1, 0, p); \
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, p);                                \
  } while (0)

inline int launch_first_real(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const FirstRealParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { CHZ_LAUNCH((fwd_first_real<a, b>), grid, block, lds, s, e0, e1, p); return 0; }
  CHZ_FWD_MENU(X)
#undef X
  return -1;
}
inline int launch_cols(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const ColsParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { CHZ_LAUNCH((fwd_cols<a, b>), grid, block, lds, s, e0, e1, p); return 0; }
  CHZ_FWD_MENU(X)
#undef X
  return -1;
}
inline int launch_rows(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const RowsParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { CHZ_LAUNCH((fwd_rows<a, b>), grid, block, lds, s, e0, e1, p); return 0; }
  CHZ_FWD_MENU(X)
#undef X
  return -1;
}
inline int launch_chan(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const ChanParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { CHZ_LAUNCH((chan_ifft<a, b>), grid, block, lds, s, e0, e1, p); return 0; }
  CHZ_CHAN_MENU(X)
#undef X
  return -1;
}
}  // namespace chz
