// Developer microbenchmark: sustained rate of back-to-back f32 MFMAs on gfx950, 16x16x4 vs 32x32x2,
// NACC independent accumulators per wave, W waves per SIMD.   hipcc --offload-arch=gfx950 tools/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flops_per_mfma, int nacc, int iters, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)blocks * 4 * iters * 8 * nacc;
  std::printf("%-22s blocks=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, n_mfma * flops_per_mfma / ms / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000;
  for (int blocks : {256, 512, 1024, 2048}) {
    run("16x16x4 nacc=1", [&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); }, 2048, 1, iters, blocks);
    run("16x16x4 nacc=2", [&] { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); }, 2048, 2, iters, blocks);
    run("16x16x4 nacc=4", [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); }, 2048, 4, iters, blocks);
    run("32x32x2 nacc=1", [&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); }, 4096, 1, iters, blocks);
    run("32x32x2 nacc=2", [&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); }, 4096, 2, iters, blocks);
  }
  return 0;
}
