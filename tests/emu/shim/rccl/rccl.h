// TEST INFRASTRUCTURE (tests/emu): the RCCL types hq_shard.hip names (the library itself is dlopen()ed at run time and is
// never loaded in the emulation build).
#pragma once
#include <hip/hip_runtime.h>
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
