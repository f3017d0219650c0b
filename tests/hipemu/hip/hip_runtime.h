// tests/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny wavefront/workgroup emulator so the *unmodified* HIP kernel sources in
// ka9q-radio_amd/csrc/ can be executed by g++ in the GPU-less build container
// (`pytest -m "not gpu"`): every HIP thread is a ucontext fiber, __syncthreads()
// and the wave shuffles are yield points.  It exists to check index math and
// butterfly wiring before spending GPU-box minutes; it is never linked into
// the product library, which is built by hipcc from the same sources and fails
// loudly without a GPU.  Nothing here is a "CPU fallback".
#pragma once
#define HIPEMU 1   /* lets kernel sources tell this emulator from hipcc's host pass */
#include <ucontext.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <vector>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __restrict__ __restrict
#define __launch_bounds__(...)

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint3 { unsigned x, y, z; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

// ThreadSanitizer has to be told about every stack switch (scripts/engine_emulated.sh TSAN=1 builds the engine's host code with it)
#if defined(__SANITIZE_THREAD__)
extern "C" { void* __tsan_get_current_fiber(void); void* __tsan_create_fiber(unsigned flags); void __tsan_destroy_fiber(void* fiber);
             void __tsan_switch_to_fiber(void* fiber, unsigned flags); }
#define HIPEMU_TSAN 1
#else
#define HIPEMU_TSAN 0
#endif
namespace hipemu {
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  uint3 tid{0, 0, 0};
  void* tsan = nullptr;
};
struct State {
  ucontext_t sched;
  void* sched_tsan = nullptr;
  std::vector<Fiber> fibers;
  int cur = -1;
  uint3 bid{0, 0, 0};
  dim3 bdim, gdim;
  std::function<void()> body;
  char* dyn_smem = nullptr;
  // wave-level exchange slots (64 lanes x up to 64 waves)
  std::vector<uint64_t> slots;
  // counting barriers: whole workgroup, and one per wave
  size_t nthreads = 0;
  size_t bar_count = 0; unsigned bar_gen = 0;
  std::vector<size_t> wave_count; std::vector<unsigned> wave_gen;
  ~State() { for (Fiber& f : fibers) free(f.stack); }     // the stack pool lives as long as the process; keeps LeakSanitizer quiet
};
inline State& st() { static State s; return s; }
inline void to_sched(State& s) {
#if HIPEMU_TSAN
  __tsan_switch_to_fiber(s.sched_tsan, 0);
#endif
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}
inline void yield() { to_sched(st()); }
static void trampoline() {
  State& s = st();
  s.body();
  s.fibers[s.cur].done = true;
  to_sched(s);
}
// Run one workgroup: fibers are resumed round-robin; each runs until it blocks in a counting
// barrier (workgroup or wave) or finishes.  A kernel whose threads do not all reach a barrier
// would spin here forever -- exactly the kernels that hang on hardware.
inline void run_block() {
  State& s = st();
  size_t n = (size_t)s.bdim.x * s.bdim.y * s.bdim.z;
  const size_t STK = 256 * 1024;
  if (s.fibers.size() < n) s.fibers.resize(n);
  for (size_t i = 0; i < n; i++) {
    Fiber& f = s.fibers[i];
    if (!f.stack) f.stack = (char*)malloc(STK);
    f.done = false;
    f.tid = uint3{(unsigned)(i % s.bdim.x), (unsigned)((i / s.bdim.x) % s.bdim.y), (unsigned)(i / ((size_t)s.bdim.x * s.bdim.y))};
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = STK;
    f.ctx.uc_link = &s.sched;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
#if HIPEMU_TSAN
    if (f.tsan) __tsan_destroy_fiber(f.tsan);
    f.tsan = __tsan_create_fiber(0);
#endif
  }
#if HIPEMU_TSAN
  s.sched_tsan = __tsan_get_current_fiber();
#endif
  s.slots.assign(n, 0);
  s.nthreads = n; s.bar_count = 0; s.bar_gen = 0;
  s.wave_count.assign((n + 63) / 64, 0); s.wave_gen.assign((n + 63) / 64, 0);
  bool alive = true;
  while (alive) {
    alive = false;
    for (size_t i = 0; i < n; i++) {
      if (s.fibers[i].done) continue;
      s.cur = (int)i;
#if HIPEMU_TSAN
      __tsan_switch_to_fiber(s.fibers[i].tsan, 0);
#endif
      swapcontext(&s.sched, &s.fibers[i].ctx);
      if (!s.fibers[i].done) alive = true;
    }
  }
}
template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t shmem, A... args) {
  State& s = st();
  s.bdim = block; s.gdim = grid;
  std::vector<char> smem(shmem + 64);
  s.dyn_smem = smem.data() + ((64 - ((uintptr_t)smem.data() & 63)) & 63);
  s.body = [=]() { kernel(args...); };
  for (unsigned z = 0; z < grid.z; z++)
    for (unsigned y = 0; y < grid.y; y++)
      for (unsigned x = 0; x < grid.x; x++) { s.bid = uint3{x, y, z}; run_block(); }
  s.dyn_smem = nullptr;
}
}  // namespace hipemu

// threadIdx.x etc. as expressions evaluated at use time
struct hipemu_tid_t { struct X { operator unsigned() const { return hipemu::st().fibers[hipemu::st().cur].tid.x; } } x;
                      struct Y { operator unsigned() const { return hipemu::st().fibers[hipemu::st().cur].tid.y; } } y;
                      struct Z { operator unsigned() const { return hipemu::st().fibers[hipemu::st().cur].tid.z; } } z; };
struct hipemu_bid_t { struct X { operator unsigned() const { return hipemu::st().bid.x; } } x;
                      struct Y { operator unsigned() const { return hipemu::st().bid.y; } } y;
                      struct Z { operator unsigned() const { return hipemu::st().bid.z; } } z; };
struct hipemu_bdim_t { struct X { operator unsigned() const { return hipemu::st().bdim.x; } } x;
                       struct Y { operator unsigned() const { return hipemu::st().bdim.y; } } y;
                       struct Z { operator unsigned() const { return hipemu::st().bdim.z; } } z; };
struct hipemu_gdim_t { struct X { operator unsigned() const { return hipemu::st().gdim.x; } } x;
                       struct Y { operator unsigned() const { return hipemu::st().gdim.y; } } y;
                       struct Z { operator unsigned() const { return hipemu::st().gdim.z; } } z; };
static hipemu_tid_t threadIdx;
static hipemu_bid_t blockIdx;
static hipemu_bdim_t blockDim;
static hipemu_gdim_t gridDim;

#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::st().dyn_smem);

// Counting barrier over the whole workgroup: the last arriver opens the next
// generation, everybody else yields until it changes.
static inline void __syncthreads() {
  hipemu::State& s = hipemu::st();
  unsigned gen = s.bar_gen;
  if (++s.bar_count == s.nthreads) { s.bar_count = 0; s.bar_gen++; return; }
  while (s.bar_gen == gen) hipemu::yield();
}
static inline void hipemu_wave_barrier(unsigned me) {
  hipemu::State& s = hipemu::st();
  unsigned w = me >> 6;
  size_t lanes = s.nthreads - (size_t)w * 64; if (lanes > 64) lanes = 64;
  unsigned gen = s.wave_gen[w];
  if (++s.wave_count[w] == lanes) { s.wave_count[w] = 0; s.wave_gen[w]++; return; }
  while (s.wave_gen[w] == gen) hipemu::yield();
}

// 64-lane wave shuffles through per-thread slots.  Lanes of one wave are the 64
// consecutive linear thread ids, as on hardware.
static inline unsigned hipemu_linear_tid() {
  const uint3& t = hipemu::st().fibers[hipemu::st().cur].tid; const dim3& b = hipemu::st().bdim;
  return t.x + b.x * (t.y + b.y * t.z);
}
template <class T> static inline T hipemu_shfl_idx(T v, int srcLaneInWave) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  hipemu::State& s = hipemu::st();
  unsigned me = hipemu_linear_tid();
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  s.slots[me] = raw;
  hipemu_wave_barrier(me);               // every lane of the wave has published
  unsigned base = me & ~63u;
  unsigned src = base + ((unsigned)srcLaneInWave & 63u);
  size_t n = (size_t)s.bdim.x * s.bdim.y * s.bdim.z;
  uint64_t got = src < n ? s.slots[src] : raw;
  hipemu_wave_barrier(me);               // every lane has read
  T out; memcpy(&out, &got, sizeof(T));
  return out;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
  int lane = hipemu_linear_tid() & 63;
  int s = (lane & ~(width - 1)) | (src & (width - 1));
  return hipemu_shfl_idx(v, s);
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  int lane = hipemu_linear_tid() & 63; (void)width;
  return hipemu_shfl_idx(v, lane ^ mask);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = hipemu_linear_tid() & 63; (void)width;
  int s = lane + (int)d; if (s > 63) s = lane;
  return hipemu_shfl_idx(v, s);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int lane = hipemu_linear_tid() & 63; (void)width;
  int s = lane - (int)d; if (s < 0) s = lane;
  return hipemu_shfl_idx(v, s);
}
static inline unsigned long long __ballot(int pred) {
  hipemu::State& s = hipemu::st();
  unsigned me = hipemu_linear_tid();
  s.slots[me] = pred ? 1u : 0u;
  hipemu_wave_barrier(me);
  unsigned base = me & ~63u;
  size_t n = (size_t)s.bdim.x * s.bdim.y * s.bdim.z;
  unsigned long long m = 0;
  for (unsigned l = 0; l < 64; l++) if (base + l < n && s.slots[base + l]) m |= 1ull << l;
  hipemu_wave_barrier(me);
  return m;
}
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __lane_id() { return hipemu_linear_tid() & 63; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)

static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
// sin(pi x), cos(pi x) in double with exact argument reduction (device: ocml sincospi)
static inline void sincospi(double x, double* s, double* c) {
  double y = fmod(x, 2.0); if (y < 0) y += 2.0;
  int q = (int)floor(2.0 * y); if (q > 3) q = 3;
  const double r = y - 0.5 * q, pi = 3.14159265358979323846;
  double sr, cr;
  if (r > 0.25) { sr = cos(pi * (0.5 - r)); cr = sin(pi * (0.5 - r)); } else { sr = sin(pi * r); cr = cos(pi * r); }
  switch (q) { case 0: *s = sr; *c = cr; break; case 1: *s = cr; *c = -sr; break; case 2: *s = -sr; *c = -cr; break; default: *s = -cr; *c = sr; }
}

typedef void* hipStream_t;
typedef void* hipEvent_t;
#ifdef HIPEMU_HOST
// the engine's host code on the CPU (tests/test_engine_emulated.py): the rest of the runtime API, launches one at a time
#include "hip_host_stub.h"
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  do { std::lock_guard<std::recursive_mutex> hipemu_lk(hipemu::launch_mutex()); \
       hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__); } while (0)
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, e0, e1, flags, ...) \
  do { std::lock_guard<std::recursive_mutex> hipemu_lk(hipemu::launch_mutex()); \
       hipEventRecord(e0); hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__); hipEventRecord(e1); } while (0)
#else
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, e0, e1, flags, ...) \
  hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)
#endif
// single-threaded fibers: agent-scope atomics degrade to plain accesses
#define __HIP_MEMORY_SCOPE_AGENT 4
template <class T> static inline T __hip_atomic_load(T* p, int, int) { return *p; }
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T, class V> static inline void __hip_atomic_store(T* p, V v, int, int) { *p = (T)v; }
static inline void __builtin_amdgcn_s_sleep(int) {}
