// scripts/micro/patbench.hip -- access-pattern microbenchmark behind the tile-shape decisions in DESIGN.md
// (not product code).  A workgroup copies a tile of T contiguous float2 columns x NP rows (row stride `inner`),
// each thread moving R elements, optionally as float4 (two adjacent columns per thread).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R, class V> __global__ void k_tile(const V* __restrict__ in, V* __restrict__ out, long inner, int T, int rows2, long off, int tiles_per_row, long rowblock) {
  int tid = threadIdx.x; int j = tid / T, t = tid - j * T;
  int rb = blockIdx.x / tiles_per_row, ct = blockIdx.x - rb * tiles_per_row;
  long base = rb * rowblock + (long)ct * T + off;
  if (j >= rows2) return;
  V v[R];
#pragma unroll
  for (int q = 0; q < R; q++) v[q] = in[base + (long)(j + q * rows2) * inner + t];
#pragma unroll
  for (int q = 0; q < R; q++) out[base + (long)(j + q * rows2) * inner + t] = v[q];
}

int main(int argc, char** argv) {
  const long n = 1700000;
  float2 *a, *b;
  CK(hipMalloc(&a, n * sizeof(float2) * 2)); CK(hipMalloc(&b, n * sizeof(float2) * 2));
  CK(hipMemset(a, 1, n * sizeof(float2) * 2)); CK(hipMemset(b, 0, n * sizeof(float2) * 2));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, long elems, auto launch) {
    for (int i = 0; i < 20; i++) launch();
    (void)hipEventRecord(e0, s);
    const int reps = 300;
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.2f us  (%.0f GB/s r+w)\n", name, ms / reps * 1e3, 2.0 * elems * 8 / (ms / reps * 1e-3) / 1e9);
  };
  char name[128];
  // axis-a-like: NP rows x inner columns (packed), one row block.  NP=120 (R=10, rows2=12) total ~1.62M elements
  struct { int NP, R, rows2; long inner; int T; long off; int vec; } cases[] = {
    {120, 10, 12, 13500, 15, 0, 1}, {120, 10, 12, 13500, 30, 0, 1}, {120, 10, 12, 13500, 45, 0, 1},
    {120, 10, 12, 13504, 16, 0, 1}, {120, 10, 12, 13504, 32, 0, 1}, {120, 10, 12, 13504, 64, 0, 1},
    {120, 10, 12, 13504, 16, 1, 1}, {120, 10, 12, 13504, 32, 1, 1}, {120, 10, 12, 13504, 16, 8, 1}, {120, 10, 12, 13504, 8, 0, 1},
    {120, 10, 12, 13504, 16, 0, 2}, {120, 10, 12, 13504, 32, 0, 2}, {120, 10, 12, 13500, 30, 0, 2},
    {150, 10, 15, 10800, 16, 0, 1}, {150, 10, 15, 10800, 32, 0, 1}, {150, 15, 10, 10800, 16, 0, 1}, {150, 15, 10, 10800, 32, 0, 1},
    {250, 10, 25, 6480, 16, 0, 1}, {250, 10, 25, 6480, 24, 0, 1},
  };
  for (auto& c : cases) {
    long cols = c.inner / c.vec;
    int T = c.T / c.vec;                       // threads per row of the tile
    int tiles = (int)(c.inner / c.T);
    int threads = ((c.rows2 * T + 63) / 64) * 64;
    long elems = (long)c.NP * tiles * c.T;
    snprintf(name, sizeof name, "cols NP=%d R=%d inner=%ld T=%d off=%ld vec=%d  grid %dx%d", c.NP, c.R, c.inner, c.T, c.off, c.vec, tiles, threads);
    if (c.vec == 1) {
      if (c.R == 10) timeit(name, elems, [&] { hipLaunchKernelGGL((k_tile<10, float2>), dim3(tiles), dim3(threads), 0, s, a, b, cols, T, c.rows2, c.off, tiles, 0L); });
      else timeit(name, elems, [&] { hipLaunchKernelGGL((k_tile<15, float2>), dim3(tiles), dim3(threads), 0, s, a, b, cols, T, c.rows2, c.off, tiles, 0L); });
    } else {
      timeit(name, elems, [&] { hipLaunchKernelGGL((k_tile<10, float4>), dim3(tiles), dim3(threads), 0, s, (float4*)a, (float4*)b, cols, T, c.rows2, c.off / 2, tiles, 0L); });
    }
  }
  // axis-b-like: 61 row blocks of [NP][Nc]; tile T columns of Nc
  struct { int NP, R, rows2, Nc, T, rbs; } cb[] = { {150, 10, 15, 180, 20, 61}, {150, 10, 15, 180, 30, 61}, {150, 10, 15, 180, 60, 61},
    {100, 10, 10, 224, 16, 73}, {100, 10, 10, 224, 32, 73}, {90, 10, 9, 240, 16, 76}, {90, 10, 9, 240, 48, 76}, {81, 9, 9, 160, 16, 126}, {81, 9, 9, 160, 32, 126} };
  for (auto& c : cb) {
    int tpr = c.Nc / c.T; int tiles = tpr * c.rbs; int threads = ((c.rows2 * c.T + 63) / 64) * 64;
    long elems = (long)c.NP * c.Nc * c.rbs;
    snprintf(name, sizeof name, "axis-b NP=%d Nc=%d T=%d rowblocks=%d  grid %dx%d", c.NP, c.Nc, c.T, c.rbs, tiles, threads);
    if (c.R == 10) timeit(name, elems, [&] { hipLaunchKernelGGL((k_tile<10, float2>), dim3(tiles), dim3(threads), 0, s, a, b, (long)c.Nc, c.T, c.rows2, 0L, tpr, (long)c.NP * c.Nc); });
    else timeit(name, elems, [&] { hipLaunchKernelGGL((k_tile<9, float2>), dim3(tiles), dim3(threads), 0, s, a, b, (long)c.Nc, c.T, c.rows2, 0L, tpr, (long)c.NP * c.Nc); });
  }
  return 0;
}
