// scripts/micro/ntbench.hip -- do non-temporal loads/stores shorten a pass?  (not product code)
// Tile copy with the forward passes' access pattern (16 float2 columns x 120 rows, 13 MB in + 13 MB out), with an
// optional block of dependent arithmetic between the loads and the stores standing in for the butterflies.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R, int NTL, int NTS, int WORK>
__global__ void k_tile(const float2* __restrict__ in, float2* __restrict__ out, long inner, int T, int rows2) {
  int tid = threadIdx.x; int j = tid / T, t = tid - j * T;
  long base = (long)blockIdx.x * T;
  if (j >= rows2) return;
  float2 v[R];
#pragma unroll
  for (int q = 0; q < R; q++) {
    const float2* p = in + base + (long)(j + q * rows2) * inner + t;
    if (NTL) { v[q].x = __builtin_nontemporal_load(&p->x); v[q].y = __builtin_nontemporal_load(&p->y); } else v[q] = *p;
  }
  if (WORK) {
#pragma unroll 1
    for (int it = 0; it < WORK; it++) {
#pragma unroll
      for (int q = 0; q < R; q++) { v[q].x = fmaf(v[q].x, 1.0000001f, v[(q + 1) % R].y * 1e-9f); v[q].y = fmaf(v[q].y, 0.9999999f, v[q].x * 1e-9f); }
    }
  }
#pragma unroll
  for (int q = 0; q < R; q++) {
    float2* p = out + base + (long)(j + q * rows2) * inner + t;
    if (NTS) { __builtin_nontemporal_store(v[q].x, &p->x); __builtin_nontemporal_store(v[q].y, &p->y); } else *p = v[q];
  }
}

int main() {
  const long inner = 13504; const int NP = 120, R = 10, rows2 = 12, T = 16;
  const long n = inner * NP;
  float2 *a, *b[4];
  CK(hipMalloc(&a, n * 8)); CK(hipMemset(a, 1, n * 8));
  for (int i = 0; i < 4; i++) { CK(hipMalloc(&b[i], n * 8)); CK(hipMemset(b[i], 0, n * 8)); }
  CK(hipDeviceSynchronize());
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int tiles = (int)(inner / T), threads = 192;
  const int reps = 200;
  hipEvent_t ev[2 * reps];
  for (auto& evt : ev) CK(hipEventCreate(&evt));
  auto run = [&](const char* name, auto kern) {
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, dim3(tiles), dim3(threads), 0, s, a, b[i % 4], inner, T, rows2);
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < reps; i++)
      hipExtLaunchKernelGGL(kern, dim3(tiles), dim3(threads), 0, s, ev[2 * i], ev[2 * i + 1], 0, a, b[i % 4], inner, T, rows2);
    CK(hipStreamSynchronize(s));
    double tot = 0; float ms;
    for (int i = 0; i < reps; i++) { (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); tot += ms; }
    (void)hipEventElapsedTime(&ms, ev[0], ev[2 * reps - 1]);
    printf("%-44s kernel %6.2f us   back-to-back %6.2f us/launch\n", name, tot / reps * 1e3, ms / reps * 1e3);
    return 0;
  };
  run("copy            ", k_tile<R, 0, 0, 0>);
  run("copy  nt-store  ", k_tile<R, 0, 1, 0>);
  run("copy  nt-load   ", k_tile<R, 1, 0, 0>);
  run("copy  nt-both   ", k_tile<R, 1, 1, 0>);
  run("work200         ", k_tile<R, 0, 0, 200>);
  run("work200 nt-store", k_tile<R, 0, 1, 200>);
  run("work200 nt-both ", k_tile<R, 1, 1, 200>);
  run("work400         ", k_tile<R, 0, 0, 400>);
  run("work400 nt-store", k_tile<R, 0, 1, 400>);
  return 0;
}
