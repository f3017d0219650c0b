// scripts/micro/hbm_phased.hip -- does HBM stream faster when the WHOLE CHIP reads for a while and then writes for a while, instead of every
// wavefront mixing its loads and stores?  (round 6; not product code.)  scripts/micro/hbm_stream.hip: read-only 6.9 TB/s, fill 5.4-5.6, but any
// mixed stream (float4 copy, the channel kernel's 2400 B in / 1920 B out rows) tops out at 4.9-5.0 TB/s -- what C_rt sits on.  If the loss is the
// read/write turnaround of the memory, phases long enough to amortise it should approach 1 / (r/6.9 + w/5.5) ~ 6.1 TB/s for a copy.
// Persistent kernel: G resident workgroups; each loads U float4 per thread (all workgroups in the read phase together), grid barrier, stores
// them, grid barrier, next chunk.  Chip-wide chunk = G x 256 x U x 16 B.    build: hipcc --offload-arch=gfx950 -O3 hbm_phased.hip -o hbm_phased.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    __threadfence();
  }
  __syncthreads();
}
template <int U, bool PHASED> __global__ void __launch_bounds__(256) k_phased(const f4* __restrict__ in, f4* __restrict__ out, size_t n, unsigned* counter) {
  const size_t chunk = (size_t)gridDim.x * 256 * U;            // float4s per chip-wide phase
  unsigned gen = 0;
  for (size_t base = 0; base + chunk <= n; base += chunk) {
    const size_t i0 = base + (size_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(in + i0 + (size_t)u * 256);
    if (PHASED) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); gen++; grid_barrier(counter, gen * gridDim.x); }
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u], out + i0 + (size_t)u * 256);
    if (PHASED) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); gen++; grid_barrier(counter, gen * gridDim.x); }
  }
}
template <class F> static double timed(F launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e-3;
}
int main(int argc, char** argv) {
  const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 8;
  const size_t bytes = gb << 30, n = bytes / sizeof(f4);
  f4 *a = nullptr, *b = nullptr; unsigned* ctr = nullptr;
  CK(hipMalloc((void**)&a, bytes)); CK(hipMalloc((void**)&b, bytes)); CK(hipMalloc((void**)&ctr, 64));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  printf("%zu GB per buffer, %d CUs; TB/s = (bytes read + bytes written) / s; chunk = bytes the chip reads (then writes) per phase\n", gb, cus);
#define CASE(U, WPC) { const int grid = cus * WPC; const double chunk_mb = (double)grid * 256 * U * 16 / 1e6; \
    const double t0 = timed([&] { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_phased<U, false>), dim3(grid), dim3(256), 0, 0, a, b, n, ctr); }, 3); \
    const double t1 = timed([&] { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_phased<U, true>), dim3(grid), dim3(256), 0, 0, a, b, n, ctr); }, 3); \
    printf("  U=%2d  %d workgroups per CU  chunk %6.1f MB   mixed %6.3f TB/s   phased %6.3f TB/s\n", U, WPC, chunk_mb, 2.0 * bytes / t0 / 1e12, 2.0 * bytes / t1 / 1e12); }
  CASE(4, 2) CASE(8, 2) CASE(16, 2) CASE(24, 2)
  CASE(4, 4) CASE(8, 4) CASE(16, 4)
  CASE(8, 1) CASE(16, 1) CASE(32, 1)
  return 0;
}
