// TEST INFRASTRUCTURE (tests/emu): a lane-exact host emulation of the small part of the HIP programming model that
// hybridq_amd/csrc uses, so that the REAL host planners and the REAL kernel bodies of libhq_hip.so can be executed on a box
// without a GPU (tests/emu/build.py compiles the five .hip translation units as plain C++ against this header into
// tests/emu/_build/libhq_emu.so; only tests load it -- hybridq_amd.core never does, the product has no CPU path).
//
// Model: a kernel launch runs its workgroups one after the other; the threads of a workgroup are cooperative fibers on one
// OS thread that switch only at synchronisation points -- __syncthreads / s_barrier (workgroup) and the wave-level
// operations (MFMA, readfirstlane, shuffles), which rendezvous the 64 lanes of a wave and exchange operands with the
// lane layouts of the gfx950 ISA (v_mfma_f32_16x16x4_f32: A[i][k] in lane i+16k, B[k][j] in lane j+16k, D[i][j] in lane
// j+16(i/4) register i%4; v_mfma_f64_16x16x4_f64: D[i][j] in lane j+16(i%4) register i/4 -- the layouts the kernels were
// validated with on hardware in rounds 1-3).  LDS is one mapping in the low 4 GiB, so that the kernels' 32-bit absolute LDS
// addresses (address_space(3) accesses, ds_write2 operands) are host addresses as they stand.  What this does NOT
// model: timing, the memory hierarchy, bank conflicts, scheduling hazards (s_nop / s_waitcnt are ignored).  Races between
// waves show up only as a dependence on the order in which the scheduler runs the waves (HQ_EMU_ORDER=reverse / random).
#pragma once
#define HQ_EMU 1
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

// ---- language -----------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...)
#define amdgpu_flat_work_group_size(...)
#define __shared__ HQ_EMU_ERROR_use_HQ_LDS_or_HQ_DYN_LDS
#define HIP_SYMBOL(x) x
// inline assembly in the kernels is scheduling / register-allocation advice (empty bodies with "+v" / "+s" constraints,
// s_waitcnt, s_nop) and has no effect on values; the two ds_write2 forms are emulated at their definition (HQ_EMU branch
// of lds_write2).  `asm volatile(...)` disappears: `asm` expands to nothing and `volatile(` swallows its argument list
// (`volatile` as a qualifier is never followed by a parenthesis in these sources).
#pragma clang diagnostic ignored "-Wkeyword-macro"
#define asm
#define volatile(...)

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };

// ---- runtime API (the subset hybridq_amd/csrc calls) -----------------------------------------------------------------
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801, hipErrorNotReady = 600 };
typedef struct hq_emu_stream* hipStream_t;
typedef struct hq_emu_event* hipEvent_t;
typedef void* hipDeviceptr_t;
typedef struct hq_emu_graph* hipGraph_t;
typedef struct hq_emu_graph_exec* hipGraphExec_t;
typedef uint64_t hipMemGenericAllocationHandle_t;
struct hipIpcMemHandle_t { char reserved[64]; };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnified = 3, hipMemoryTypeManaged = 4 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive, hipStreamCaptureStatusInvalidated };
enum hipMemAllocationType { hipMemAllocationTypePinned = 1 };
enum hipMemLocationType { hipMemLocationTypeDevice = 1 };
enum hipMemAllocationGranularity_flags { hipMemAllocationGranularityMinimum = 0 };
enum hipMemAccessFlags { hipMemAccessFlagsProtReadWrite = 3 };
struct hipMemLocation { hipMemLocationType type; int id; };
struct hipMemAllocationProp { hipMemAllocationType type; int requestedHandleType; hipMemLocation location; void* win32; unsigned char pad[16]; };
struct hipMemAccessDesc { hipMemLocation location; hipMemAccessFlags flags; };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipDeviceMallocContiguous = 0x4,
                   hipIpcMemLazyEnablePeerAccess = 1;

const char* hipGetErrorString(hipError_t);
hipError_t hipGetLastError();
hipError_t hipGetDevice(int*);
hipError_t hipGetDeviceCount(int*);
hipError_t hipDeviceSynchronize();
hipError_t hipMalloc(void**, size_t);
template <typename T> hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
hipError_t hipExtMallocWithFlags(void**, size_t, unsigned);
hipError_t hipFree(void*);
hipError_t hipHostMalloc(void**, size_t, unsigned);
template <typename T> hipError_t hipHostMalloc(T** p, size_t n, unsigned f) { return hipHostMalloc(reinterpret_cast<void**>(p), n, f); }
hipError_t hipHostFree(void*);
hipError_t hipMemGetInfo(size_t*, size_t*);
hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind);
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t);
hipError_t hipMemcpyFromSymbol(void*, const void*, size_t);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t*, const void*);
hipError_t hipMemGetAddressRange(hipDeviceptr_t*, size_t*, hipDeviceptr_t);
hipError_t hipStreamCreate(hipStream_t*);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamQuery(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode);
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*);
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus*);
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t);
hipError_t hipGraphDestroy(hipGraph_t);
hipError_t hipGraphExecDestroy(hipGraphExec_t);
hipError_t hipEventCreate(hipEvent_t*);
hipError_t hipEventCreateWithFlags(hipEvent_t*, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t);
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int);
hipError_t hipMemGetAllocationGranularity(size_t*, const hipMemAllocationProp*, hipMemAllocationGranularity_flags);
hipError_t hipMemAddressReserve(void**, size_t, size_t, void*, unsigned long long);
hipError_t hipMemAddressFree(void*, size_t);
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t*, size_t, const hipMemAllocationProp*, unsigned long long);
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t);
hipError_t hipMemMap(void*, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long);
hipError_t hipMemUnmap(void*, size_t);
hipError_t hipMemSetAccess(void*, size_t, const hipMemAccessDesc*, size_t);
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*);
hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned);
hipError_t hipIpcCloseMemHandle(void*);

// ---- execution model --------------------------------------------------------------------------------------------------
namespace hq_emu {
struct Lane {
  dim3 tid, bid;
  unsigned nops;  // wave-level operations this lane has taken part in (selects the exchange buffer)
};
extern Lane* cur;
extern dim3 cur_grid, cur_block;
unsigned char* dyn_lds();
void block_barrier();
int readfirstlane(int);
void mfma_f32(float a, float b, const float* c, float* d);
void mfma_f64(double a, double b, const double* c, double* d);
uint64_t shfl_bits(uint64_t bits, int src_lane_delta, int mode);  // mode 0: down, 1: xor, 2: idx
// runs the workgroups now -- or, while `stream` is being captured into a graph, records the launch
void launch(dim3 grid, dim3 block, size_t lds, hipStream_t stream, std::function<void()> body);
// arguments are evaluated and copied at the call, as a kernel launch does
template <typename K, typename... A>
inline void launch_kernel(dim3 grid, dim3 block, size_t lds, hipStream_t stream, K kern, A... args) {
  launch(grid, block, lds, stream, [=]() { kern(args...); });
}
inline unsigned char* lds_at(unsigned a) { return reinterpret_cast<unsigned char*>((uintptr_t)a); }
}  // namespace hq_emu

#define threadIdx (hq_emu::cur->tid)
#define blockIdx (hq_emu::cur->bid)
#define blockDim (hq_emu::cur_block)
#define gridDim (hq_emu::cur_grid)
#define __syncthreads() hq_emu::block_barrier()
#define __builtin_amdgcn_s_barrier() hq_emu::block_barrier()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) hq_emu::readfirstlane(x)
#define __popcll(x) __builtin_popcountll(x)

typedef float hq_emu_f32x4 __attribute__((ext_vector_type(4)));
typedef double hq_emu_f64x4 __attribute__((ext_vector_type(4)));
inline hq_emu_f32x4 hq_emu_mfma(float a, float b, hq_emu_f32x4 c) {
  float ci[4] = {c[0], c[1], c[2], c[3]}, d[4];
  hq_emu::mfma_f32(a, b, ci, d);
  return hq_emu_f32x4{d[0], d[1], d[2], d[3]};
}
inline hq_emu_f64x4 hq_emu_mfma(double a, double b, hq_emu_f64x4 c) {
  double ci[4] = {c[0], c[1], c[2], c[3]}, d[4];
  hq_emu::mfma_f64(a, b, ci, d);
  return hq_emu_f64x4{d[0], d[1], d[2], d[3]};
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hq_emu_mfma((float)(a), (float)(b), c)
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) hq_emu_mfma((double)(a), (double)(b), c)

template <typename T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
  static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
  (void)width;
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  bits = hq_emu::shfl_bits(bits, (int)delta, 0);
  T out;
  memcpy(&out, &bits, sizeof(T));
  return out;
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  bits = hq_emu::shfl_bits(bits, mask, 1);
  T out;
  memcpy(&out, &bits, sizeof(T));
  return out;
}
// one OS thread runs every lane: plain read-modify-write is atomic
template <typename T, typename U> inline T atomicAdd(T* p, U v) { T old = *p; *p = old + (T)v; return old; }

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
  hq_emu::launch_kernel(grid, block, lds, stream, kern, __VA_ARGS__)
