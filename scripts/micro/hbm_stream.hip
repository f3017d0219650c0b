// scripts/micro/hbm_stream.hip -- what does HBM stream on this card with HAND-WRITTEN kernels, far beyond the 256 MiB Infinity Cache?
// (round 6; not product code).  The yardstick DESIGN.md holds the channel kernel against at C_rt -- 4.85-4.88 TB/s "measured stream copy" --
// came from torch's elementwise copy (scripts/hbm_stream_probe.py: copy 4.88, read-only 4.0, fill 6.9 TB/s).  If a plain f4 copy with
// enough loads in flight streams faster than that, the channel kernel has head-room the 93 % figure hides.
// Kernels: copy (16 B per lane, U loads in flight per lane, grid-stride), read-only (the same loads, one add each, one store per workgroup),
// fill; plain and non-temporal variants; 2048 x CUs workgroups.   build: hipcc --offload-arch=gfx950 -O3 hbm_stream.hip -o hbm_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, bool NT> __global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) out[i] = in[i];
}
template <int U, bool NT> __global__ void __launch_bounds__(256) k_read(const f4* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n; i += stride) acc += in[i].x;
  if (acc == 12345.678f) out[blockIdx.x] = acc;                 // (never true: keeps the loads alive without a store stream)
}
template <bool NT> __global__ void __launch_bounds__(256) k_fill(f4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
}
// the channel kernel's shape: per "channel" a contiguous row of RB bytes read once and a contiguous row of WB bytes written once (2400 / 1920 at P = 300),
// one wavefront per CPW rows, nothing else -- the DRAM side of chan_ifft without its arithmetic or its cached gathers
template <bool NT> __global__ void __launch_bounds__(64) k_rows(const f2* __restrict__ in, f2* __restrict__ out, int nrows, int rin, int rout) {
  const int lane = threadIdx.x;
  for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
    const f2* src = in + (size_t)r * rin; f2* dst = out + (size_t)r * rout;
    f2 v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { const int i = lane + 64 * k; v[k] = i < rin ? (NT ? __builtin_nontemporal_load(src + i) : src[i]) : f2{0.f, 0.f}; }
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 5; k++) { s.x += v[k].x; s.y += v[k].y; }
#pragma unroll
    for (int k = 0; k < 4; k++) { const int i = lane + 64 * k; if (i < rout) { const f2 o = {v[k].x + s.x * 0.f, v[k].y}; if (NT) __builtin_nontemporal_store(o, dst + i); else dst[i] = o; } }
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
  const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 16;
  const size_t bytes = gb << 30, n = bytes / sizeof(f4);
  f4 *a = nullptr, *b = nullptr; float* o = nullptr;
  CK(hipMalloc((void**)&a, bytes)); CK(hipMalloc((void**)&b, bytes)); CK(hipMalloc((void**)&o, 1 << 20));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  printf("%zu GB per buffer, %d CUs; TB/s = (bytes read + bytes written) / s\n", gb, cus);
  for (int wgs_per_cu : {4, 8, 16, 32}) {
    const int grid = cus * wgs_per_cu;
    auto run = [&](const char* name, auto fn, double moved) { const double t = timed(fn, 3); printf("  %-34s grid %6d  %7.3f TB/s\n", name, grid, moved / t / 1e12); };
    run("copy  U=1", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy<1, false>), dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    run("copy  U=4", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy<4, false>), dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    run("copy  U=8", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy<8, false>), dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    run("copy  U=4 non-temporal", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy<4, true>), dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    run("read  U=4", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_read<4, false>), dim3(grid), dim3(256), 0, 0, a, o, n); }, 1.0 * bytes);
    run("read  U=8", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_read<8, false>), dim3(grid), dim3(256), 0, 0, a, o, n); }, 1.0 * bytes);
    run("read  U=8 non-temporal", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_read<8, true>), dim3(grid), dim3(256), 0, 0, a, o, n); }, 1.0 * bytes);
    run("fill", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<false>), dim3(grid), dim3(256), 0, 0, b, n); }, 1.0 * bytes);
    run("fill non-temporal", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<true>), dim3(grid), dim3(256), 0, 0, b, n); }, 1.0 * bytes);
  }
  // rows: 2400 B in, 1920 B out per row, as many rows as fit
  const int rin = 300, rout = 240;
  const int nrows = (int)(bytes / (rin * sizeof(f2)) < bytes / (rout * sizeof(f2)) ? bytes / (rin * sizeof(f2)) : bytes / (rout * sizeof(f2)));
  for (int wpc : {8, 16, 32, 64}) {
    const int grid = cus * wpc;
    auto run = [&](const char* name, auto fn, double moved) { const double t = timed(fn, 3); printf("  %-34s grid %6d  %7.3f TB/s\n", name, grid, moved / t / 1e12); };
    run("rows 2400 B in / 1920 B out", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rows<false>), dim3(grid), dim3(64), 0, 0, (const f2*)a, (f2*)b, nrows, rin, rout); }, (double)nrows * (rin + rout) * 8.0);
    run("rows non-temporal", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rows<true>), dim3(grid), dim3(64), 0, 0, (const f2*)a, (f2*)b, nrows, rin, rout); }, (double)nrows * (rin + rout) * 8.0);
  }
  return 0;
}
