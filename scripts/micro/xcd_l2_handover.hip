// scripts/micro/xcd_l2_handover.hip -- does a line written by one launch survive in the writing XCD's L2 for the NEXT launch?
// (round 4, decision record profiles/r04_xcd_affine.txt).  Kernel W writes a 2 MiB slab with PLAIN stores from workgroups of ONE XCD
// (blocks b with b % 8 == xw; the XCC id every block really ran on is recorded), kernel R -- a separate launch on the same stream --
// reads the slab from workgroups of XCD xr and is timed with HIP events (dispatch timestamps).  If the L2 kept the lines, xr == xw
// reads at the L2's rate and xr != xw at the fabric's; if every launch starts with this memory dropped from its L2, both read at the
// fabric's rate.  A third case reads the slab inside the SAME launch that wrote it (same workgroups): the L2-hit reference.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_l2_handover.bin xcd_l2_handover.hip && ./xcd_l2_handover.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
constexpr int PER = 32;                       // workgroups per XCD that do the work
__global__ void __launch_bounds__(256) wr(float4* p, size_t n16, int xcd, int* where, float seed) {
  if (threadIdx.x == 0) where[blockIdx.x] = xcc_id();
  if (((int)blockIdx.x & 7) != xcd) return;
  const size_t w = blockIdx.x >> 3;
  for (size_t i = w * 256 + threadIdx.x; i < n16; i += (size_t)PER * 256) p[i] = make_float4(seed, (float)i, 1.f, 2.f);
}
__global__ void __launch_bounds__(256) rd(const float4* p, size_t n16, int xcd, float* sink) {
  if (((int)blockIdx.x & 7) != xcd) return;
  const size_t w = blockIdx.x >> 3;
  float a = 0.f;
  for (size_t i = w * 256 + threadIdx.x; i < n16; i += (size_t)PER * 256) { const float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
  if (a == 12345.678f) sink[0] = a;
}
__global__ void __launch_bounds__(256) wr_rd(float4* p, size_t n16, int xcd, float* sink, float seed) {    // same launch, same workgroups
  if (((int)blockIdx.x & 7) != xcd) return;
  const size_t w = blockIdx.x >> 3;
  for (size_t i = w * 256 + threadIdx.x; i < n16; i += (size_t)PER * 256) p[i] = make_float4(seed, (float)i, 1.f, 2.f);
  __syncthreads();
  float a = 0.f;
  for (size_t i = w * 256 + threadIdx.x; i < n16; i += (size_t)PER * 256) { const float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
  if (a == 12345.678f) sink[0] = a;
}
int main() {
  const size_t bytes = 2u << 20, n16 = bytes / 16;
  float4* buf; float* sink; int* where;
  OK(hipMalloc((void**)&buf, bytes)); OK(hipMalloc((void**)&sink, 4)); OK(hipMalloc((void**)&where, 4 * 8 * PER));
  hipStream_t st; OK(hipStreamCreate(&st));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  std::vector<int> wh(8 * PER);
  auto med = [&](int xw, int xr, bool plain_wr) -> float {
    std::vector<float> t;
    for (int rep = 0; rep < 41; rep++) {
      if (plain_wr) hipLaunchKernelGGL(wr, dim3(8 * PER), dim3(256), 0, st, buf, n16, xw, where, (float)rep);
      hipExtLaunchKernelGGL(rd, dim3(8 * PER), dim3(256), 0, st, e0, e1, 0, (const float4*)buf, n16, xr, sink);
      hipStreamSynchronize(st);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end()); return t[t.size() / 2];
  };
  hipLaunchKernelGGL(wr, dim3(8 * PER), dim3(256), 0, st, buf, n16, 0, where, 0.f);
  OK(hipStreamSynchronize(st));
  OK(hipMemcpy(wh.data(), where, 4 * 8 * PER, hipMemcpyDeviceToHost));
  int agree = 0; for (int b = 0; b < 8 * PER; b++) agree += wh[b] == (b & 7);
  printf("placement: %d of %d blocks ran on XCC id == blockIdx %% 8\n", agree, 8 * PER);
  printf("read of a 2 MiB slab by 32 workgroups of one XCD, median of 41, us (dispatch timestamps)\n");
  for (int xw : {0, 3}) {
    printf("  written by the PREVIOUS launch on XCD %d, plain stores:  read on XCD %d: %.2f   read on XCD %d: %.2f   read on XCD %d: %.2f\n",
           xw, xw, med(xw, xw, true), (xw + 1) & 7, med(xw, (xw + 1) & 7, true), (xw + 5) & 7, med(xw, (xw + 5) & 7, true));
  }
  {
    std::vector<float> t;
    for (int rep = 0; rep < 41; rep++) {            // read twice in a row by the same XCD, nothing written in between: does a READ line survive a launch boundary?
      hipLaunchKernelGGL(rd, dim3(8 * PER), dim3(256), 0, st, (const float4*)buf, n16, 2, sink);
      hipExtLaunchKernelGGL(rd, dim3(8 * PER), dim3(256), 0, st, e0, e1, 0, (const float4*)buf, n16, 2, sink);
      hipStreamSynchronize(st);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    printf("  read by the PREVIOUS launch on the same XCD (clean lines): %.2f\n", t[t.size() / 2]);
  }
  {
    std::vector<float> t, t2;
    for (int rep = 0; rep < 41; rep++) {
      hipExtLaunchKernelGGL(wr_rd, dim3(8 * PER), dim3(256), 0, st, e0, e1, 0, buf, n16, 1, sink, (float)rep);
      hipStreamSynchronize(st);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
      hipExtLaunchKernelGGL(wr, dim3(8 * PER), dim3(256), 0, st, e0, e1, 0, buf, n16, 1, where, (float)rep);
      hipStreamSynchronize(st);
      hipEventElapsedTime(&ms, e0, e1); t2.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end()); std::sort(t2.begin(), t2.end());
    printf("  written and read back inside ONE launch (same workgroups): %.2f, of which the write alone is %.2f -> the read costs %.2f\n",
           t[t.size() / 2], t2[t2.size() / 2], t[t.size() / 2] - t2[t2.size() / 2]);
  }
  return 0;
}
