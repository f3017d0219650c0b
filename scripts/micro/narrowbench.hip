// scripts/micro/narrowbench.hip -- would a TWO-axis plan (1800 x 1800, 4-column tiles) move data fast enough?
// Copy kernels with the access patterns of such a plan, run alone and pipelined over 4 streams (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// pattern A: tile = T float2 columns x NP rows (row stride inner); XCD-aware: block b -> tile (b%8)*(tiles/8)+b/8
template <int R> __global__ void k_colsA(const float2* __restrict__ in, float2* __restrict__ out, int inner, int T, int rows2, int tiles, int xcd) {
  int b = blockIdx.x; int tile = xcd ? (b % 8) * (tiles / 8) + b / 8 : b; if (tile >= tiles) return;
  int tid = threadIdx.x; int j = tid / T, t = tid - j * T; if (j >= rows2) return;
  const float2* g = in + (long)j * inner + tile * T + t; float2* o = out + (long)j * inner + tile * T + t;
  float2 v[R];
#pragma unroll
  for (int q = 0; q < R; q++) v[q] = g[(long)q * rows2 * inner];
#pragma unroll
  for (int q = 0; q < R; q++) o[(long)q * rows2 * inner] = v[q];
}
// pattern B: tile = T rows (consecutive ka) of NC contiguous elements; write transposed: out[ka + NA*kk] (T-wide segments)
template <int R> __global__ void k_rowsB(const float2* __restrict__ in, float2* __restrict__ out, int NC, int NA, int T, int tiles, int xcd) {
  int b = blockIdx.x; int tile = xcd ? (b % 8) * (tiles / 8) + b / 8 : b; if (tile >= tiles) return;
  int tid = threadIdx.x; int kk0 = tid / T, r = tid - kk0 * T; int per = NC / R; if (kk0 >= per) return;
  int ka = tile * T + r;
  float2 v[R];
#pragma unroll
  for (int q = 0; q < R; q++) v[q] = in[(long)ka * NC + kk0 + q * per];          // strided read stands in for the LDS-transposed load
#pragma unroll
  for (int q = 0; q < R; q++) out[ka + (long)NA * (kk0 + q * per)] = v[q];
}
int main() {
  const long n = 1800L * 904;
  float2 *a, *b[4];
  CK(hipMalloc(&a, n * 8)); CK(hipMemset(a, 1, n * 8));
  for (int i = 0; i < 4; i++) { CK(hipMalloc(&b[i], n * 8)); CK(hipMemset(b[i], 0, n * 8)); }
  hipStream_t s[4]; for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipEvent_t done[4]; for (int i = 0; i < 4; i++) CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
  auto timeit = [&](const char* name, int nstreams, auto launch) {
    for (int i = 0; i < 16; i++) launch(s[i % nstreams], i % 4);
    CK(hipDeviceSynchronize());
    (void)hipEventRecord(e0, s[0]);
    for (int i = 1; i < nstreams; i++) (void)hipStreamWaitEvent(s[i], e0, 0);
    const int reps = 400;
    for (int i = 0; i < reps; i++) launch(s[i % nstreams], i % 4);
    for (int i = 1; i < nstreams; i++) { (void)hipEventRecord(done[i], s[i]); (void)hipStreamWaitEvent(s[0], done[i], 0); }
    (void)hipEventRecord(e1, s[0]); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %d streams %7.2f us/launch %6.0f GB/s\n", name, nstreams, ms / reps * 1e3, 2.0 * 1800 * 900 * 8 / (ms / reps * 1e-3) / 1e9);
    return 0;
  };
  for (int ns : {1, 4}) {
    for (int xcd : {0, 1}) {
      char nm[96];
      snprintf(nm, sizeof nm, "A: 1800 rows x 4-col tiles (R=12, 600 thr) xcd=%d", xcd);
      timeit(nm, ns, [&](hipStream_t st, int i) { hipLaunchKernelGGL((k_colsA<12>), dim3(232), dim3(640), 0, st, a, b[i], 900, 4, 150, 225, xcd); });
      snprintf(nm, sizeof nm, "A8: 1800 rows x 8-col tiles (R=12, 1200->1024? no: 600x2) xcd=%d", xcd);
      timeit(nm, ns, [&](hipStream_t st, int i) { hipLaunchKernelGGL((k_colsA<24>), dim3(120), dim3(640), 0, st, a, b[i], 904, 8, 75, 113, xcd); });
      snprintf(nm, sizeof nm, "B: 4 rows x 1800, write 4-wide segs stride 904 (R=15) xcd=%d", xcd);
      timeit(nm, ns, [&](hipStream_t st, int i) { hipLaunchKernelGGL((k_rowsB<15>), dim3(232), dim3(512), 0, st, a, b[i], 1800, 904, 4, 226, xcd); });
    }
    timeit("ref: aligned 16-col tiles NP=120 (today's pattern)", ns, [&](hipStream_t st, int i) { hipLaunchKernelGGL((k_colsA<10>), dim3(844), dim3(192), 0, st, a, b[i], 13504, 16, 12, 844, 0); });
  }
  return 0;
}
