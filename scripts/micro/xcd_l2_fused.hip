// scripts/micro/xcd_l2_fused.hip -- the follow-up to xcd_l2_handover.hip (round 4, profiles/r04_xcd_affine.txt): a launch boundary drops
// what a kernel left in its XCD's L2, so does a hand-over INSIDE one launch keep it?  One kernel, 8 slabs of 2 MiB (one per XCD, 16 MiB in
// all -- the size of the forward transform's intermediate), 32 producer workgroups per XCD write their slab, then 32 consumer workgroups
// (higher block indices: dispatched after every producer, so a consumer that spins on the slab's counter cannot starve a producer) read
// the slab of XCD (own + d) % 8.  Store flavours: plain, sc1 (write-through, what fwd_cols ships), nt.  The consumers' phase is timed
// inside the kernel with the constant-rate counter (first consumer through its wait .. last consumer done) and checked against the
// data (a stale line shows up as a wrong sum).
//   hipcc --offload-arch=gfx950 -O3 -o xcd_l2_fused.bin xcd_l2_fused.hip && ./xcd_l2_fused.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int PER = 32;
constexpr size_t SLAB16 = (2u << 20) / 16;       // float4 per slab
__device__ __forceinline__ int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
template <int FLAVOUR>
__device__ __forceinline__ void put(float4* base, size_t i, float4 v) {      // aux bits of the gfx950 buffer store: 0 plain, 16 sc1, 2 nt, 17 sc0 sc1
  constexpr int AUX = FLAVOUR == 0 ? 0 : FLAVOUR == 1 ? 16 : FLAVOUR == 2 ? 2 : 17;
  const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffffc, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(u4_t{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, d, (int)(i * 16), 0, AUX);
}
struct Stamp { long long t0, t1; int xcc; float sum; };
// blocks [0, 8*PER): producers, block b -> slab b % 8.  blocks [8*PER, 16*PER): consumers, block b -> reads slab ((b % 8) + d) % 8.
template <int FLAVOUR>
__global__ void __launch_bounds__(256) fused(float4* buf, unsigned* done, unsigned epoch, int d, float seed, Stamp* st, int wait_on) {
  const int b = blockIdx.x, x = b & 7;
  if (b < 8 * PER) {
    const size_t w = b >> 3;
    float4* p = buf + (size_t)x * SLAB16;
    const long long t0 = wall_clock64();
    if constexpr (FLAVOUR <= 3) {
      for (size_t i = w * 256 + threadIdx.x; i < SLAB16; i += (size_t)PER * 256) put<FLAVOUR>(p, i, make_float4(seed, (float)(i & 1023), 1.f, 2.f));
    } else if constexpr (FLAVOUR == 4) {               // producers only READ their slab: the consumers then find clean lines
      float a = 0.f;
      for (size_t i = w * 256 + threadIdx.x; i < SLAB16; i += (size_t)PER * 256) { const float4 v = p[i]; a += v.x + v.y; }
      if (a == 1.2345f) st[0].sum = a;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(done + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      st[b] = Stamp{t0, wall_clock64(), xcc_id(), 0.f};
    }
    return;
  }
  const int slab = (x + d) & 7;
  const size_t w = (b - 8 * PER) >> 3;
  if (wait_on) {
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(done + slab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * PER) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
  }
  const long long t0 = wall_clock64();
  const float4* p = buf + (size_t)slab * SLAB16;
  float a = 0.f;
  for (size_t i = w * 256 + threadIdx.x; i < SLAB16; i += (size_t)PER * 256) { const float4 v = p[i]; a += (v.x - seed) + (v.y - (float)(i & 1023)) + (v.z - 1.f) + (v.w - 2.f); }
  // a != 0 for any stale element
  __shared__ float red[256];
  red[threadIdx.x] = a * a; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) st[b] = Stamp{t0, wall_clock64(), xcc_id(), red[0]};
}
// the two-launch reference: the same consumer, launched on its own behind the producers
template <int FLAVOUR>
int run(const char* name, float4* buf, unsigned* done, Stamp* st, hipStream_t s, double tick_us) {
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  std::vector<Stamp> h(16 * PER);
  unsigned epoch = 0;
  OK(hipMemsetAsync(done, 0, 32, s));
  for (int d : {0, 1, 5}) {
    std::vector<float> tk, tc, tp, tw; double stale = 0; int misplaced = 0;
    for (int rep = 0; rep < 41; rep++) {
      epoch++;
      hipExtLaunchKernelGGL(fused<FLAVOUR>, dim3(16 * PER), dim3(256), 0, s, e0, e1, 0, buf, done, epoch, d, (float)(rep + 1), st, 1);
      OK(hipStreamSynchronize(s));
      float ms = 0; hipEventElapsedTime(&ms, e0, e1); tk.push_back(ms * 1e3f);
      OK(hipMemcpy(h.data(), st, sizeof(Stamp) * 16 * PER, hipMemcpyDeviceToHost));
      long long p0 = h[0].t0, p1 = h[0].t1, c0 = h[8 * PER].t0, c1 = h[8 * PER].t1;
      for (int b = 0; b < 8 * PER; b++) { p0 = std::min(p0, h[b].t0); p1 = std::max(p1, h[b].t1); misplaced += h[b].xcc != (b & 7); }
      for (int b = 8 * PER; b < 16 * PER; b++) { c0 = std::min(c0, h[b].t0); c1 = std::max(c1, h[b].t1); stale += h[b].sum; misplaced += h[b].xcc != (b & 7); }
      tp.push_back((float)((p1 - p0) * tick_us)); tc.push_back((float)((c1 - c0) * tick_us));
      { std::vector<long long> dur; for (int b = 8 * PER; b < 16 * PER; b++) dur.push_back(h[b].t1 - h[b].t0); std::sort(dur.begin(), dur.end()); tw.push_back((float)(dur[dur.size() / 2] * tick_us)); }
    }
    std::sort(tk.begin(), tk.end()); std::sort(tc.begin(), tc.end()); std::sort(tp.begin(), tp.end()); std::sort(tw.begin(), tw.end());
    printf("  %-22s consumers on XCD own+%d: kernel %6.2f us  producers' phase %6.2f  consumers' phase %6.2f (median workgroup %5.2f)   (16 MiB written, 16 MiB read; stale sum %g, misplaced %d)\n",
           name, d, tk[20], tp[20], tc[20], tw[20], stale, misplaced);
  }
  // two launches: producers (consumers leave at once: d = -1 trick not needed, launch only the first half), then consumers alone
  {
    std::vector<float> t1, t2;
    for (int rep = 0; rep < 41; rep++) {
      epoch++;
      hipExtLaunchKernelGGL(fused<FLAVOUR>, dim3(8 * PER), dim3(256), 0, s, e0, e1, 0, buf, done, epoch, 0, (float)(rep + 1), st, 1);
      OK(hipStreamSynchronize(s));
      float ms = 0; hipEventElapsedTime(&ms, e0, e1); t1.push_back(ms * 1e3f);
    }
    std::sort(t1.begin(), t1.end());
    printf("  %-22s producers alone as a launch: %6.2f us\n", name, t1[20]);
  }
  return 0;
}
// consumers alone (second launch of a pair): grid offset by a kernel argument is not there, so a thin wrapper kernel
__global__ void __launch_bounds__(256) consume(const float4* buf, int d, float seed, Stamp* st) {
  const int b = blockIdx.x, x = b & 7, slab = (x + d) & 7;
  const size_t w = b >> 3;
  const long long t0 = wall_clock64();
  const float4* p = buf + (size_t)slab * SLAB16;
  float a = 0.f;
  for (size_t i = w * 256 + threadIdx.x; i < SLAB16; i += (size_t)PER * 256) { const float4 v = p[i]; a += (v.x - seed) + (v.y - (float)(i & 1023)) + (v.z - 1.f) + (v.w - 2.f); }
  __shared__ float red[256];
  red[threadIdx.x] = a * a; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) st[b] = Stamp{t0, wall_clock64(), 0, red[0]};
}
int main() {
  float4* buf; unsigned* done; Stamp* st;
  OK(hipMalloc((void**)&buf, 8 * SLAB16 * 16)); OK(hipMalloc((void**)&done, 32)); OK(hipMalloc((void**)&st, sizeof(Stamp) * 16 * PER));
  hipStream_t s; OK(hipStreamCreate(&s));
  int rate_khz = 0; OK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const double tick_us = 1e3 / (double)rate_khz;
  printf("producers then consumers inside ONE launch (8 x 32 + 8 x 32 workgroups of 256, 2 MiB per XCD), median of 41\n");
  if (run<0>("plain stores", buf, done, st, s, tick_us)) return 1;
  if (run<1>("sc1 (write-through)", buf, done, st, s, tick_us)) return 1;
  if (run<2>("nt", buf, done, st, s, tick_us)) return 1;
  if (run<3>("sc0 sc1", buf, done, st, s, tick_us)) return 1;
  if (run<4>("producers READ (clean)", buf, done, st, s, tick_us)) return 1;
  if (run<5>("producers idle (cold)", buf, done, st, s, tick_us)) return 1;
  {
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int rep = 0; rep < 41; rep++) {
      hipLaunchKernelGGL(fused<0>, dim3(8 * PER), dim3(256), 0, s, buf, done, 0u, 0, 7.f, st, 1);
      hipExtLaunchKernelGGL(consume, dim3(8 * PER), dim3(256), 0, s, e0, e1, 0, (const float4*)buf, 0, 7.f, st);
      OK(hipStreamSynchronize(s));
      float ms = 0; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    printf("  consumers as a launch of their own behind the producers' launch (plain stores, same XCD): %6.2f us\n", t[20]);
  }
  return 0;
}
