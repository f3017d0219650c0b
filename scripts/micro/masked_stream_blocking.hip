// scripts/micro/masked_stream_blocking.hip -- reproducer for the round-5 hang (not product code).
// Question: is a stream made by hipExtStreamCreateWithCUMask a BLOCKING stream (hipStreamDefault semantics: implicitly ordered with the
// legacy null stream), although the engine treats its streams as hipStreamNonBlocking?  The call takes no flags argument.
// Test per kind of stream: a kernel on the stream spins until the HOST sets a flag; meanwhile a one-thread kernel goes to the null stream.
//   non-blocking stream: the null-stream kernel runs while the other still spins
//   blocking stream:     it does not run until the flag is set (null-stream work waits for every blocking stream, and the
//                        next work on any blocking stream waits for the null stream) -- with a device-side ticket wait between two such
//                        streams and ANY null-stream operation in between (the engine's synchronous hipMemcpy / hipMemset, torch's
//                        default stream in bench.py) that is a deadlock: kernel A spins for kernel B, B queues behind the null-stream
//                        operation, the null-stream operation waits for A.
// Part 2 shows that deadlock shape with a time-out instead of a hang.
// build: hipcc --offload-arch=gfx950 -O2 masked_stream_blocking.hip -o masked_stream_blocking.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin_until(volatile int* flag, int want, long long budget_ticks) {
  const long long t0 = wall_clock64();
  while (*flag != want && wall_clock64() - t0 < budget_ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void set_flag(volatile int* flag, int v) { *flag = v; __threadfence_system(); }

static int full_mask_stream(hipStream_t* s) {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  uint32_t mask[32] = {};
  for (int i = 0; i < cus; i++) mask[i / 32] |= 1u << (i % 32);
  CK(hipExtStreamCreateWithCUMask(s, (uint32_t)((cus + 31) / 32), mask));
  return 0;
}

static int probe(const char* name, hipStream_t s, volatile int* hflag, int* dflag, int* scratch) {
  (void)scratch;
  hflag[0] = 0; hflag[8] = 0;
  hipLaunchKernelGGL(spin_until, dim3(1), dim3(1), 0, s, (volatile int*)dflag, 1, 100000000LL * 5);     // <= 5 s at 100 MHz
  hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, 0, (volatile int*)(dflag + 8), 1);                    // legacy null stream: a kernel the host can SEE finishing
  std::this_thread::sleep_for(std::chrono::milliseconds(200));
  const bool ran = hflag[8] == 1;                     // (hipStreamQuery(0) is no witness: on the null stream it reports on every stream of the device)
  unsigned flags = 99;
  (void)hipStreamGetFlags(s, &flags);
  printf("%-44s null-stream kernel after 200 ms: %-8s -> the stream is %s   (hipStreamGetFlags = %u, hipStreamNonBlocking = %u)\n", name,
         ran ? "ran" : "WAITING", ran ? "non-blocking" : "BLOCKING", flags, (unsigned)hipStreamNonBlocking);
  *hflag = 1;
  CK(hipStreamSynchronize(s));
  CK(hipDeviceSynchronize());
  return 0;
}

int main() {
  int *hflag = nullptr, *dflag = nullptr, *scratch = nullptr;
  CK(hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&dflag, hflag, 0));
  CK(hipMalloc((void**)&scratch, 64));
  hipStream_t plain, masked, prio, masked2;
  CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  if (full_mask_stream(&masked)) return 1;
  if (full_mask_stream(&masked2)) return 1;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&prio, hipStreamNonBlocking, hi));
  printf("stream priority range: least %d .. greatest %d\n", lo, hi);
  if (probe("hipStreamCreateWithFlags(NonBlocking)", plain, hflag, dflag, scratch)) return 1;
  if (probe("hipExtStreamCreateWithCUMask(all CUs)", masked, hflag, dflag, scratch)) return 1;
  if (probe("hipStreamCreateWithPriority(NonBlocking,hi)", prio, hflag, dflag, scratch)) return 1;

  // Part 2: the engine's shape.  Kernel A on lane 1 waits (device side, bounded 2 s) for a ticket that kernel B on lane 2 publishes;
  // between the two launches the host issues ONE null-stream operation (what a synchronous hipMemcpy / hipMemset, or torch's default
  // stream, amounts to).  With CU-masked lanes and with plain non-blocking lanes.
  hipStream_t plain2;
  CK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
  for (int kind = 0; kind < 2; kind++) {
    hipStream_t l1 = kind ? plain : masked, l2 = kind ? plain2 : masked2;
    for (int with_null_op = 0; with_null_op < 2; with_null_op++) {
      *hflag = 0;
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(spin_until, dim3(1), dim3(1), 0, l1, (volatile int*)dflag, 7, 100000000LL * 2);
      if (with_null_op) CK(hipMemsetAsync(scratch, 0, 4, 0));
      hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, l2, (volatile int*)dflag, 7);
      CK(hipDeviceSynchronize());
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("ticket wait across two %s streams, %s null-stream operation in between: %.1f ms %s\n", kind ? "plain non-blocking" : "CU-masked", with_null_op ? "ONE" : "no", ms,
             ms > 1000 ? "= the waiter sat out its whole budget: DEADLOCK SHAPE (an unbounded wait hangs here)" : "= handed over at once");
    }
  }
  return 0;
}
