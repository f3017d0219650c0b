// tests/hipemu/hip/hip_host_stub.h -- TEST INFRASTRUCTURE ONLY.  The HOST side of the HIP runtime, as much of it as
// ka9q-radio_amd/csrc/chz_engine.hip uses, on the CPU: "device" memory is host memory, every stream operation completes before the call
// returns, kernels run on the fiber emulator (one launch at a time, whichever host thread issues it).  With it the UNMODIFIED engine
// source builds into a CPU library behind the same C ABI, so that `-m "not gpu"` tests can drive the engine's orchestration -- lanes,
// issuing threads, per-slot descriptors, response-row recycling, the demodulator stream, inline-master pools -- against the oracle.
// Not a product path and not a fallback: the product library is hipcc-built and refuses to run without a GPU.
#pragma once
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorNotSupported = 801 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventReleaseToDevice = 0x40000000, hipHostMallocDefault = 0, hipHostMallocMapped = 2,
       hipHostRegisterDefault = 0, hipHostRegisterPortable = 1, hipHostMallocPortable = 1, hipStreamCaptureModeThreadLocal = 1 };
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipemu_event { double t_ms = 0.0; };
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void (*hipHostFn_t)(void*);

// marks a library built with this stand-in: ka9q-radio_amd/engine.py refuses to load one unless a test asked for it
extern "C" __attribute__((weak, visibility("default"))) int chz_emulated_build(void) { return 1; }

namespace hipemu {
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline std::recursive_mutex& launch_mutex() { static std::recursive_mutex m; return m; }
}

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorNotSupported ? "not supported by the CPU stand-in" : "error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 256, n ? n : 256) == 0 ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, void* = nullptr) { memset(p, v, n); return hipSuccess; }
typedef void* hipDeviceptr_t;
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t p, int v, size_t count, void* = nullptr) { for (size_t i = 0; i < count; i++) static_cast<int*>(p)[i] = v; return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, void* = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, void* = nullptr) {
  for (size_t r = 0; r < height; r++) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(void** s, unsigned) { *s = malloc(8); return *s ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipStreamDestroy(void* s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(void*) { return hipSuccess; }
static inline hipError_t hipLaunchHostFunc(void*, hipHostFn_t fn, void* arg) { fn(arg); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(void** e, unsigned) { *e = new hipemu_event; return hipSuccess; }
static inline hipError_t hipEventCreate(void** e) { return hipEventCreateWithFlags(e, 0); }
static inline hipError_t hipEventDestroy(void* e) { delete static_cast<hipemu_event*>(e); return hipSuccess; }
static inline hipError_t hipEventRecord(void* e, void* = nullptr) { if (e) static_cast<hipemu_event*>(e)->t_ms = hipemu::now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(void*) { return hipSuccess; }
static inline hipError_t hipEventQuery(void*) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(void*, void*, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, void* a, void* b) {
  *ms = (float)(static_cast<hipemu_event*>(b)->t_ms - static_cast<hipemu_event*>(a)->t_ms); return hipSuccess;
}
// no stream capture on the CPU: the engine's graph mode reports the refusal
static inline hipError_t hipStreamBeginCapture(void*, int) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(void*, hipGraph_t*) { return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, void*) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
