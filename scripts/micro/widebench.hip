// scripts/micro/widebench.hip -- would 16-byte accesses (two adjacent columns per lane) shorten a forward pass?
// Tile copy with the passes' pattern (16 float2 columns x 120 rows per workgroup, 13 MB in + 13 MB out, write-through
// stores) with 8-byte or 16-byte accesses per lane, plus an optional dependent-arithmetic block.  Not product code.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int R, int WORK>
__global__ void k8(const float2* __restrict__ in, float2* __restrict__ out, int inner, int rows2) {
  const int tid = threadIdx.x, j = tid / 16, t = tid - j * 16;
  if (j >= rows2) return;
  const int base = blockIdx.x * 16 + j * inner + t;
  const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7ffffffc, 0x00020000);
  float2 v[R];
#pragma unroll
  for (int q = 0; q < R; q++) v[q] = in[base + q * rows2 * inner];
#pragma unroll 1
  for (int it = 0; it < WORK; it++)
#pragma unroll
    for (int q = 0; q < R; q++) { v[q].x = fmaf(v[q].x, 1.0000001f, v[(q + 1) % R].y * 1e-9f); v[q].y = fmaf(v[q].y, 0.9999999f, v[q].x * 1e-9f); }
#pragma unroll
  for (int q = 0; q < R; q++)
    __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(v[q].x), __float_as_uint(v[q].y)}, d, (base + q * rows2 * inner) * 8, 0, 16);
}
template <int R, int WORK>
__global__ void k16(const float4* __restrict__ in, float4* __restrict__ out, int inner2, int rows2) {
  const int tid = threadIdx.x, j = tid / 8, t = tid - j * 8;          // 8 lanes x 16 B = one 128-byte line per row
  if (j >= rows2) return;
  const int base = blockIdx.x * 8 + j * inner2 + t;
  const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7ffffffc, 0x00020000);
  float4 v[R];
#pragma unroll
  for (int q = 0; q < R; q++) v[q] = in[base + q * rows2 * inner2];
#pragma unroll 1
  for (int it = 0; it < WORK; it++)
#pragma unroll
    for (int q = 0; q < R; q++) { v[q].x = fmaf(v[q].x, 1.0000001f, v[(q + 1) % R].y * 1e-9f); v[q].y = fmaf(v[q].y, 0.9999999f, v[q].x * 1e-9f);
                                  v[q].z = fmaf(v[q].z, 1.0000001f, v[(q + 1) % R].w * 1e-9f); v[q].w = fmaf(v[q].w, 0.9999999f, v[q].z * 1e-9f); }
#pragma unroll
  for (int q = 0; q < R; q++)
    __builtin_amdgcn_raw_buffer_store_b128(u4{__float_as_uint(v[q].x), __float_as_uint(v[q].y), __float_as_uint(v[q].z), __float_as_uint(v[q].w)}, d, (base + q * rows2 * inner2) * 16, 0, 16);
}
int main() {
  const int inner = 13504, NP = 120, rows2 = 12;
  const long n = (long)inner * NP;
  float2 *a, *b[4];
  CK(hipMalloc(&a, n * 8)); CK(hipMemset(a, 1, n * 8));
  for (int i = 0; i < 4; i++) { CK(hipMalloc(&b[i], n * 8)); CK(hipMemset(b[i], 0, n * 8)); }
  CK(hipDeviceSynchronize());
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int tiles = inner / 16, reps = 200;
  hipEvent_t ev[2 * reps];
  for (auto& x : ev) CK(hipEventCreate(&x));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 20; i++) launch(i, nullptr, nullptr);
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < reps; i++) launch(i, ev[2 * i], ev[2 * i + 1]);
    CK(hipStreamSynchronize(s));
    double tot = 0; float ms;
    for (int i = 0; i < reps; i++) { (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); tot += ms; }
    printf("%-40s kernel %6.2f us\n", name, tot / reps * 1e3);
    return 0;
  };
#define L8(W, threads) [&](int i, hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL((k8<10, W>), dim3(tiles), dim3(threads), 0, s, e0, e1, 0, a, b[i % 4], inner, rows2); }
#define L16(W, threads) [&](int i, hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL((k16<10, W>), dim3(tiles), dim3(threads), 0, s, e0, e1, 0, (const float4*)a, (float4*)b[i % 4], inner / 2, rows2); }
  run("8-byte lanes (192 thr), copy", L8(0, 192));
  run("16-byte lanes (96 -> 128 thr), copy", L16(0, 128));
  run("8-byte lanes, work 12", L8(12, 192));
  run("16-byte lanes, work 12", L16(12, 128));
  run("8-byte lanes, work 24", L8(24, 192));
  run("16-byte lanes, work 24", L16(24, 128));
  return 0;
}
