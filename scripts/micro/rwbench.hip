// scripts/micro/rwbench.hip -- read-only vs write-only vs copy bandwidth for a 13 MB buffer (not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int U, class V> __global__ void k_write(V* __restrict__ out, long n) {
  long base = (long)blockIdx.x * blockDim.x * U + threadIdx.x;
  V v; v.x = (float)threadIdx.x; v.y = 1.f;
#pragma unroll
  for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n) out[i] = v; }
}
template <int U, class V> __global__ void k_read(const V* __restrict__ in, V* __restrict__ out, long n) {
  long base = (long)blockIdx.x * blockDim.x * U + threadIdx.x;
  float acc = 0;
#pragma unroll
  for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n) { V v = in[i]; acc += v.x + v.y; } }
  if (acc == 1.2345e30f) out[0].x = acc;
}
template <int U, class V> __global__ void k_copy(const V* __restrict__ in, V* __restrict__ out, long n) {
  long base = (long)blockIdx.x * blockDim.x * U + threadIdx.x;
  V v[U];
#pragma unroll
  for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n) v[u] = in[i]; }
#pragma unroll
  for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n) out[i] = v[u]; }
}
int main() {
  const long n = 1620000;
  float2 *a, *b;
  CK(hipMalloc(&a, n * 8 * 2)); CK(hipMalloc(&b, n * 8 * 2));
  CK(hipMemset(a, 1, n * 8 * 2)); CK(hipMemset(b, 0, n * 8 * 2));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 20; i++) launch();
    (void)hipEventRecord(e0, s);
    const int reps = 400;
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f us  %6.0f GB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
  };
  const int g10 = (n + 2559) / 2560, g1 = (n + 255) / 256;
  timeit("write-only 13MB float2 x10/thr", n * 8.0, [&] { hipLaunchKernelGGL((k_write<10, float2>), dim3(g10), dim3(256), 0, s, b, n); });
  timeit("write-only 13MB float2 x1/thr", n * 8.0, [&] { hipLaunchKernelGGL((k_write<1, float2>), dim3(g1), dim3(256), 0, s, b, n); });
  timeit("write-only 13MB float4 x5/thr", n * 8.0, [&] { hipLaunchKernelGGL((k_write<5, float4>), dim3(g10), dim3(256), 0, s, (float4*)b, n / 2); });
  timeit("read-only 13MB float2 x10/thr", n * 8.0, [&] { hipLaunchKernelGGL((k_read<10, float2>), dim3(g10), dim3(256), 0, s, a, b, n); });
  timeit("read-only 13MB float4 x5/thr", n * 8.0, [&] { hipLaunchKernelGGL((k_read<5, float4>), dim3(g10), dim3(256), 0, s, (const float4*)a, (float4*)b, n / 2); });
  timeit("copy 13MB float2 x10/thr", n * 16.0, [&] { hipLaunchKernelGGL((k_copy<10, float2>), dim3(g10), dim3(256), 0, s, a, b, n); });
  timeit("copy 13MB float4 x5/thr", n * 16.0, [&] { hipLaunchKernelGGL((k_copy<5, float4>), dim3(g10), dim3(256), 0, s, (const float4*)a, (float4*)b, n / 2); });
  // alternate two output buffers (like the 4 spectrum slots): are repeated writes to the same 13 MB faster?
  float2* c; CK(hipMalloc(&c, n * 8 * 8));
  int tog = 0;
  timeit("write-only rotating over 8 x 13MB", n * 8.0, [&] { hipLaunchKernelGGL((k_write<10, float2>), dim3(g10), dim3(256), 0, s, c + (long)(tog++ & 7) * n, n); });
  return 0;
}
