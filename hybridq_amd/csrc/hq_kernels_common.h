// hq_kernels_common.h / hq_kernels_{apply,swap,aux}.h -- device kernels of the MI355X (gfx950) state-vector evolution core.
//
// Semantics implemented (reference: /root/reference/include/U.h:28-102,123-202,
// include/swap.h:28-95, include/python_U.cpp:114-123), written from the index
// formula, CDNA4-first:
//
//   * apply_direct   k <= 3: pure HBM streaming.  Each lane owns 16-byte vectors
//                    (index bits 0..1 for f32, bit 0 for f64) and ALL 2^k partner
//                    vectors of its tile, so the butterfly is register-local: no
//                    LDS, no cross-lane traffic.  The lane -> address map skips the
//                    target bits, i.e. a wave's loads are contiguous 1 KiB runs
//                    whenever the targets sit at positions >= 8, and degrade to
//                    interleaved 16/32/64-byte pieces of the same cache lines (both
//                    halves issued back to back by the same wave) for lower targets.
//                    U lives in SGPRs / the scalar cache (kernel argument).
//   * apply_mfma     f32, k <= 4: the gate as a real-embedded GEMM on the matrix cores with
//                    role-assigned index digits (see the kernel's header comment).
//   * apply_generic  any k <= 10: workgroup tile of 2^(k+c) amplitudes staged
//                    through LDS (c lowest non-target bits = contiguous columns),
//                    dense complex mat-mat on the tile, results streamed back.
//   * apply_naive    out-of-place one-thread-per-amplitude fallback for tiny states.
//   * swap_lds / swap_gather, interleave (to_complex), init_state, norm2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// LDS declarations go through two macros so that tests/emu (a host emulation of the HIP model used to execute these very
// kernels on a box without a GPU -- test infrastructure, never part of the product) can place them; for the device build
// they are the plain HIP spellings.
#ifdef HQ_EMU
#define HQ_DYN_LDS(name) unsigned char* const name = hq_emu::dyn_lds()
#define HQ_LDS static
#else
#define HQ_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define HQ_LDS __shared__
#endif

namespace hq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Vec;
template <> struct Vec<float> {
  using type = f32x4;   // 16-byte lane vector
  using quad = f32x4;   // 4 consecutive elements
  static constexpr int VB = 2;
};
template <> struct Vec<double> {
  using type = f64x2;
  using quad = f64x4;
  static constexpr int VB = 1;
};

__host__ __device__ constexpr int popc_c(int x) { return x == 0 ? 0 : (x & 1) + popc_c(x >> 1); }
// gather the bits of v selected by mask into a compact integer
__host__ __device__ constexpr int pext_c(int v, int mask) {
  int out = 0, o = 0;
  for (int b = 0; b < 8; ++b)
    if ((mask >> b) & 1) { out |= ((v >> b) & 1) << o; ++o; }
  return out;
}
// scatter the low bits of v to the positions selected by mask
__host__ __device__ constexpr int pdep_c(int v, int mask) {
  int out = 0, o = 0;
  for (int b = 0; b < 8; ++b)
    if ((mask >> b) & 1) { out |= ((v >> o) & 1) << b; ++o; }
  return out;
}

constexpr int kBlock = 256;

// fma in the TYPE of its operands: `__builtin_fma` is the double builtin, so a float call site
// silently converts to f64 and back (found in round 1: the float butterfly kernels ran v_fma_f64)
__device__ __forceinline__ float hq_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double hq_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

constexpr int kMaxK = 10;
constexpr int kTileBits = 12;
// Copy a host-pinned (device-mapped) buffer into device memory in-stream (operand tables, self chunks of the exchange).
// static: this header is included by several translation units of the library.
static __global__ void __launch_bounds__(kBlock)
upload_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, const size_t n16) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBlock) dst[i] = src[i];
}

}  // namespace hq
