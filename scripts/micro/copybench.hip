// scripts/micro/copybench.hip -- reference points for the forward-pass kernels (not product code):
// how long do an empty launch, a plain 13 MB float2 copy, and a tiled strided-read / contiguous-write
// copy with the same grid shapes take on this GPU?  build: hipcc --offload-arch=gfx950 -O3 copybench.hip -o copybench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_copy(const float2* __restrict__ in, float2* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
template <int U> __global__ void k_copy_u(const float2* __restrict__ in, float2* __restrict__ out, long n) {
  long base = (long)blockIdx.x * blockDim.x * U + threadIdx.x;
  float2 v[U];
#pragma unroll
  for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; v[u] = i < n ? in[i] : make_float2(0, 0); }
#pragma unroll
  for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n) out[i] = v[u]; }
}
// column-tile pattern of fwd_first_real: tile of T contiguous float2 x R rows at stride `inner`
template <int R> __global__ void k_coltile(const float2* __restrict__ in, float2* __restrict__ out, int inner, int T, int rows2) {
  int tid = threadIdx.x; int j = tid / T, t = tid - j * T; int c0 = blockIdx.x * T;
  if (j >= rows2) return;
  float2 v[R];
#pragma unroll
  for (int q = 0; q < R; q++) v[q] = in[(long)(j + q * rows2) * inner + c0 + t];
#pragma unroll
  for (int q = 0; q < R; q++) out[(long)(j + q * rows2) * inner + c0 + t] = v[q];
}
template <int R> __global__ void k_coltile_lds(const float2* __restrict__ in, float2* __restrict__ out, int inner, int T, int rows2) {
  extern __shared__ float2 lds[];
  int tid = threadIdx.x; int j = tid / T, t = tid - j * T; int c0 = blockIdx.x * T;
  float2 v[R];
  if (j < rows2) {
#pragma unroll
    for (int q = 0; q < R; q++) v[q] = in[(long)(j + q * rows2) * inner + c0 + t];
#pragma unroll
    for (int q = 0; q < R; q++) lds[(q * rows2 + j) * T + t] = v[q];
  }
  __syncthreads();
  if (j < rows2) {
#pragma unroll
    for (int q = 0; q < R; q++) v[q] = lds[(j * R + q) * T + t];
#pragma unroll
    for (int q = 0; q < R; q++) out[(long)(j + q * rows2) * inner + c0 + t] = v[q];
  }
}

int main() {
  const long n = 1620000;             // float2 elements = 12.96 MB
  float2 *a, *b;
  CK(hipMalloc(&a, n * sizeof(float2) * 2)); CK(hipMalloc(&b, n * sizeof(float2) * 2));
  CK(hipMemset(a, 1, n * sizeof(float2))); CK(hipMemset(b, 0, n * sizeof(float2)));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(e0, s);
    const int reps = 400;
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.2f us/launch  (%.0f GB/s read+write)\n", name, ms / reps * 1e3, 2.0 * n * 8 / (ms / reps * 1e-3) / 1e9);
  };
  timeit("empty 256x64", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(64), 0, s); });
  timeit("empty 900x192", [&] { hipLaunchKernelGGL(k_empty, dim3(900), dim3(192), 0, s); });
  timeit("copy grid-stride 2048x256", [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s, a, b, n); });
  timeit("copy grid-stride 1024x256", [&] { hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n); });
  timeit("copy 1 elem/thread 6329x256", [&] { hipLaunchKernelGGL(k_copy_u<1>, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n); });
  timeit("copy 10/thread 633x256", [&] { hipLaunchKernelGGL(k_copy_u<10>, dim3((n + 2559) / 2560), dim3(256), 0, s, a, b, n); });
  timeit("copy 10/thread in place", [&] { hipLaunchKernelGGL(k_copy_u<10>, dim3((n + 2559) / 2560), dim3(256), 0, s, a, a, n); });
  // pass-1-like: 120 rows x 13500 cols (float2), T=15, R=10, rows2=12 -> 900 WGs x 192 thr (180 active)
  timeit("coltile R=10 T=15 900x192", [&] { hipLaunchKernelGGL(k_coltile<10>, dim3(900), dim3(192), 0, s, a, b, 13500, 15, 12); });
  timeit("coltile+lds R=10 T=15 900x192", [&] { hipLaunchKernelGGL(k_coltile_lds<10>, dim3(900), dim3(192), 16384, s, a, b, 13500, 15, 12); });
  // T=30: 450 WGs x 384
  timeit("coltile R=10 T=30 450x384", [&] { hipLaunchKernelGGL(k_coltile<10>, dim3(450), dim3(384), 0, s, a, b, 13500, 30, 12); });
  return 0;
}
