// TEST INFRASTRUCTURE: a deliberately racy kernel, compiled into libhq_emu.so only, that shows what the emulator's wave
// schedules can and cannot see: wave w publishes a value in LDS and reads its neighbour's.  With the barrier the result is
// the same under every schedule; without it, it depends on which wave ran first -- the forward (fair) schedule happens to
// give the "right" answer, the reverse (greedy) one does not.  tests/test_emu_kernels.py::test_wave_order relies on that.
#include <hip/hip_runtime.h>

static __global__ void race_kernel(unsigned* out, int with_barrier) {
  static unsigned slot[8];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // a wave-level operation first, so that the lanes of a wave move together and the scheduler has a switch point
  const int w = __builtin_amdgcn_readfirstlane((int)wave);
  if (lane == 0) slot[w] = 0;
  __syncthreads();
  if (lane == 0) slot[w] = 100 + (unsigned)w;
  (void)__builtin_amdgcn_readfirstlane((int)lane);
  if (with_barrier) __syncthreads();
  const unsigned seen = slot[(w + 1) & 7];
  (void)__builtin_amdgcn_readfirstlane((int)seen);
  if (lane == 0) out[w] = seen;
}

extern "C" int hq_emu_selftest_race(int with_barrier, unsigned* out8) {
  hipLaunchKernelGGL(race_kernel, dim3(1), dim3(512), 0, (hipStream_t) nullptr, out8, with_barrier);
  return 0;
}
