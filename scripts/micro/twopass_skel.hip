// scripts/micro/twopass_skel.hip -- upper bound for a TWO-pass forward transform of N = 3,240,000 = 2025 x 1600
// (not product code).  Skeleton workgroups with the data movement, register footprint, LDS exchanges, barriers,
// table loads and arithmetic volume of a register-resident three-layer axis transform, without the butterflies'
// wiring:
//   pass 1  tile = T packed columns x 2025 rows of the sample ring (row stride 800 float2), two LDS exchanges,
//           result written as contiguous chunks of a blocked intermediate ([row block][tile][rows x columns])
//   pass 2  tile = T adjacent rows x 1600 points, read as ONE contiguous slab of the blocked intermediate,
//           two LDS exchanges, result written as T-bin segments at a pitch of 2032 bins (the spectrum layout)
// Variants: T = 16 (128-byte segments, one workgroup per CU) / T = 8 (64-byte segments, two per CU, optional
// pairing of line-sharing tiles on one XCD); exchanges of re and im separately (half the LDS) or together.
// Each pass is timed alone and chained, on 1 stream and pipelined over 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u2 __attribute__((ext_vector_type(2)));

struct SkelParams {
  const float2* in; float2* out; const float2* tw;
  int ax;        // axis length (rows of the tile in pass 1, points per row in pass 2)
  int inner;     // pass 1: row stride of the input in float2
  int ntiles;    // pass 1: tiles per row; pass 2: row blocks
  int ntiles1;   // tiles of pass 1 (layout of the blocked intermediate)
  int pitch;     // pass 2: output pitch
  int pairmap;   // pass 1/2: WGs b and b+8 take neighbouring tiles
  int work;      // arithmetic iterations per point and layer
};

template <int T, int NT, int PT, int HALF, int KIND, int WPE, int CHK>
__global__ void __launch_bounds__(NT, WPE) skel(SkelParams p) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  int tile = blockIdx.x;
  if (p.pairmap) { const int g = tile >> 4, r = tile & 15; tile = g * 16 + (r & 7) * 2 + (r >> 3); }
  if (tile >= p.ntiles) return;
  const int total = p.ax * T;
  float2 v[PT];
  const __amdgpu_buffer_rsrc_t din = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0x7ffffffc, 0x00020000);
  const __amdgpu_buffer_rsrc_t dtw = __builtin_amdgcn_make_buffer_rsrc((void*)p.tw, 0, 2048 * 8, 0x00020000);
  // ---- load: one VGPR offset per thread, the per-point stride rides in the scalar offset
  {
    const int base = KIND == 1 ? (tid / T) * p.inner + tile * T + (tid & (T - 1)) : tile * total + tid;
    const int step = KIND == 1 ? (NT / T) * p.inner : NT;
#pragma unroll
    for (int q = 0; q < PT; q++) {
      const u2 r = __builtin_amdgcn_raw_buffer_load_b64(din, base * 8, q * step * 8, 0);   // a few rows past the tile on the last q: harmless
      v[q] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
    }
  }
  // ---- three layers with two exchanges
#pragma unroll 1
  for (int layer = 0; layer < 3; layer++) {
#pragma unroll 1
    for (int it = 0; it < p.work; it++)
#pragma unroll
      for (int q = 0; q < PT; q++) {
        v[q].x = fmaf(v[q].x, 1.0000001f, v[(q + 1) % PT].y * 1e-9f);
        v[q].y = fmaf(v[q].y, 0.9999999f, v[(q + 3) % PT].x * 1e-9f);
      }
    if (layer == 2) break;
#pragma unroll
    for (int q = 0; q < PT; q++) { const float2 a = v[q]; const u2 wr = __builtin_amdgcn_raw_buffer_load_b64(dtw, ((tid * 5 + q * 131 + layer * 977) & 2047) * 8, 0, 0); const float2 w = make_float2(__uint_as_float(wr.x), __uint_as_float(wr.y)); v[q] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
    if (HALF) {
#pragma unroll
      for (int q = 0; q < PT; q++) { const int e = q * NT + tid; lds[e] = v[q].x; }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PT; q++) { const int e = ((q * 7 + 3) % PT) * NT + tid; v[q].x = lds[e]; }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PT; q++) { const int e = q * NT + tid; lds[e] = v[q].y; }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PT; q++) { const int e = ((q * 7 + 3) % PT) * NT + tid; v[q].y = lds[e]; }
    } else {
      float2* l2 = reinterpret_cast<float2*>(lds);
      if (layer == 1) __syncthreads();
#pragma unroll
      for (int q = 0; q < PT; q++) { const int e = q * NT + tid; l2[e] = v[q]; }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PT; q++) { const int e = ((q * 7 + 3) % PT) * NT + tid; v[q] = l2[e]; }
    }
  }
  // ---- store (write-through, as the product's passes do)
  const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7ffffffc, 0x00020000);
  {
    const int base = KIND == 1 ? ((tid / CHK) * p.ntiles1 + tile) * CHK + (tid % CHK) : (tid / T) * p.pitch + tile * T + (tid & (T - 1));
    const int step = KIND == 1 ? (NT / CHK) * p.ntiles1 * CHK : (NT / T) * p.pitch;
#pragma unroll
    for (int q = 0; q < PT; q++) {
      int at = base * 8;
      if (q == PT - 1 && q * NT + tid >= total) at = (int)0x80000000;         // past the tile: dropped by the buffer hardware
      __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(v[q].x), __float_as_uint(v[q].y)}, d, at, q * step * 8, 16);
    }
  }
}

__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}

static hipStream_t s[4];
static hipEvent_t e0, e1, done[4];
template <class F> static double timeit(int nstreams, int reps, F launch) {
  for (int i = 0; i < 16; i++) launch(s[i % nstreams], i % 4);
  CK(hipDeviceSynchronize());
  (void)hipEventRecord(e0, s[0]);
  for (int i = 1; i < nstreams; i++) (void)hipStreamWaitEvent(s[i], e0, 0);
  for (int i = 0; i < reps; i++) launch(s[i % nstreams], i % 4);
  for (int i = 1; i < nstreams; i++) { (void)hipEventRecord(done[i], s[i]); (void)hipStreamWaitEvent(s[0], done[i], 0); }
  (void)hipEventRecord(e1, s[0]); CK(hipEventSynchronize(e1));
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3;
}

template <int T, int NT1, int PT1, int NT2, int PT2, int HALF, int WPE, int CHK>
static void variant(const char* name, const float2* ring, float2** mid, float2** spec, const float2* tw, int pairmap, int work) {
  constexpr int NA = 2025, NBC = 1600, RA = 1013;
  SkelParams p1{}; p1.in = ring; p1.tw = tw; p1.ax = NA; p1.inner = NBC / 2; p1.ntiles = NBC / 2 / T; p1.ntiles1 = p1.ntiles; p1.pairmap = pairmap; p1.work = work;
  SkelParams p2{}; p2.tw = tw; p2.ax = NBC; p2.ntiles = (RA + T - 1) / T; p2.ntiles1 = p1.ntiles; p2.pitch = 2032; p2.pairmap = pairmap; p2.work = work;
  const int lds1 = NT1 * PT1 * (HALF ? 4 : 8), lds2 = NT2 * PT2 * (HALF ? 4 : 8);   // padded to whole threads
  auto k1 = skel<T, NT1, PT1, HALF, 1, WPE, CHK>; auto k2 = skel<T, NT2, PT2, HALF, 2, WPE, CHK>;
  CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
  CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
  const int g1 = pairmap ? ((p1.ntiles + 15) / 16) * 16 : p1.ntiles, g2 = pairmap ? ((p2.ntiles + 15) / 16) * 16 : p2.ntiles;
  auto l1 = [&](hipStream_t st, int i) { SkelParams q = p1; q.out = mid[i]; hipLaunchKernelGGL(k1, dim3(g1), dim3(NT1), lds1, st, q); };
  auto l2 = [&](hipStream_t st, int i) { SkelParams q = p2; q.in = mid[i]; q.out = spec[i]; hipLaunchKernelGGL(k2, dim3(g2), dim3(NT2), lds2, st, q); };
  auto l12 = [&](hipStream_t st, int i) { l1(st, i); l2(st, i); };
  const double a1 = timeit(1, 200, l1), a4 = timeit(4, 800, l1);
  const double b1 = timeit(1, 200, l2), b4 = timeit(4, 800, l2);
  const double c1 = timeit(1, 200, l12), c4 = timeit(4, 800, l12);
  printf("%-44s work=%2d  pass1 %6.2f / %6.2f   pass2 %6.2f / %6.2f   both %6.2f / %6.2f us (1 stream / 4 streams)  grids %d+%d x %d/%d thr, lds %d/%d KB\n",
         name, work, a1, a4, b1, b4, c1, c4, g1, g2, NT1, NT2, lds1 >> 10, lds2 >> 10);
  fflush(stdout);
}

int main() {
  const long nring = 8L * 2592000 / 2;                 // the product's sample ring in float2
  const long nmid = 1016L * 1600 + 65536, nspec = 2032L * 1600 + 65536;
  float2 *ring, *tw, *mid[4], *spec[4];
  CK(hipMalloc(&ring, nring * 8)); CK(hipMemset(ring, 1, nring * 8));
  CK(hipMalloc(&tw, 2048 * 8)); CK(hipMemset(tw, 0, 2048 * 8));
  for (int i = 0; i < 4; i++) { CK(hipMalloc(&mid[i], nmid * 8)); CK(hipMemset(mid[i], 0, nmid * 8)); CK(hipMalloc(&spec[i], nspec * 8)); CK(hipMemset(spec[i], 0, nspec * 8)); }
  for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 4; i++) CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
  CK(hipDeviceSynchronize());
  {
    const long n4 = 1620000L / 2;                       // 12.96 MB in, 12.96 MB out
    auto lc = [&](hipStream_t st, int i) { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, st, (const float4*)ring, (float4*)spec[i], n4); };
    printf("copy 12.96 MB -> 12.96 MB: %.2f us (1 stream), %.2f us (4 streams)\n", timeit(1, 400, lc), timeit(4, 1600, lc));
  }
  for (int work : {0, 6, 12}) {
    variant<16, 768, 43, 640, 40, 1, 3, 256>("A  T=16 half-exchange (1 WG/CU)", ring, mid, spec, tw, 0, work);
    variant<16, 1024, 32, 1024, 25, 1, 4, 512>("A' T=16 half-exchange, 1024 thr", ring, mid, spec, tw, 0, work);
    variant<8, 384, 43, 320, 40, 1, 3, 128>("B  T=8 half-exchange (2 WG/CU)", ring, mid, spec, tw, 0, work);
    variant<8, 384, 43, 320, 40, 1, 3, 128>("B  T=8 half-exchange, XCD-paired tiles", ring, mid, spec, tw, 1, work);
    variant<8, 512, 32, 512, 25, 1, 4, 128>("B' T=8 half-exchange, 512 thr, paired", ring, mid, spec, tw, 1, work);
    variant<8, 384, 43, 320, 40, 0, 2, 128>("C  T=8 full exchange (1 WG/CU), paired", ring, mid, spec, tw, 1, work);
  }
  return 0;
}
