// tests/hipemu/rccl/rccl.h -- TEST INFRASTRUCTURE: declarations only, so that chz_comm.inc (which binds RCCL at run time with
// dlopen) compiles into the CPU build of the engine.  No collective can run there: chz_comm_create fails loudly without librccl.
#pragma once
#include <stddef.h>
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
