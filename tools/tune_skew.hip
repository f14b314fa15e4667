// tools/tune_skew.hip -- developer experiment: how much would a padded (bank-skewed) state
// layout buy for targets at high positions?  The partner stream of a k=1 gate is displaced
// by an extra `skew` bytes, which reproduces the traffic pattern a padded layout would give.
#include "../hybridq_amd/csrc/hq_kernels.h"
#include <cstdio>
#include <cstdlib>
using namespace hq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ILP, bool NT>
__global__ void __launch_bounds__(256) k1_skew(float* __restrict__ re, float* __restrict__ im, float ur, float ui, unsigned p /*vec pos*/, uint64_t skew_vec) {
  f32x4* vr = (f32x4*)re; f32x4* vi = (f32x4*)im;
  const uint64_t g0 = (uint64_t)blockIdx.x * (ILP * 256) + threadIdx.x;
  f32x4 a[ILP][2], b[ILP][2]; uint64_t v0[ILP], v1[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
    uint64_t v = g0 + (uint64_t)i * 256; const uint64_t lo = (1ull << p) - 1; v = ((v & ~lo) << 1) | (v & lo);
    v0[i] = v; v1[i] = (v | (1ull << p)) + skew_vec;
    if (NT) { a[i][0] = __builtin_nontemporal_load(&vr[v0[i]]); b[i][0] = __builtin_nontemporal_load(&vi[v0[i]]); a[i][1] = __builtin_nontemporal_load(&vr[v1[i]]); b[i][1] = __builtin_nontemporal_load(&vi[v1[i]]); }
    else { a[i][0] = vr[v0[i]]; b[i][0] = vi[v0[i]]; a[i][1] = vr[v1[i]]; b[i][1] = vi[v1[i]]; }
  }
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
    f32x4 o0 = a[i][0] * ur - b[i][1] * ui, o1 = a[i][1] * ur + b[i][0] * ui, q0 = b[i][0] * ur + a[i][1] * ui, q1 = b[i][1] * ur - a[i][0] * ui;
    if (NT) { __builtin_nontemporal_store(o0, &vr[v0[i]]); __builtin_nontemporal_store(q0, &vi[v0[i]]); __builtin_nontemporal_store(o1, &vr[v1[i]]); __builtin_nontemporal_store(q1, &vi[v1[i]]); }
    else { vr[v0[i]] = o0; vi[v0[i]] = q0; vr[v1[i]] = o1; vi[v1[i]] = q1; }
  }
}

int main(int argc, char** argv) {
  const unsigned n = 30; float *buf; const size_t N = 1ull << n;
  CK(hipMalloc(&buf, (2 * N + (64u << 20)) * sizeof(float)));
  CK(hipMemset(buf, 0, (2 * N + (64u << 20)) * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t plane_pad = 12288 / 4;
  float* re = buf; float* im = buf + N + (16u << 20) + plane_pad;  // room for the skewed partner
  for (unsigned pos : {13u, 20u, 22u, 25u, 27u, 29u}) {
    for (uint64_t skew_bytes : {0ull, 4096ull, 8192ull, 12288ull, 65536ull + 4096, 1048576ull + 4096, 2097152ull + 8192}) {
      const unsigned p = pos - 2; const uint64_t ngroups = (N / 4 / 2) / (2 * 256);
      auto run = [&] { hipLaunchKernelGGL((k1_skew<2, true>), dim3((unsigned)ngroups), dim3(256), 0, 0, re, im, 0.6f, 0.8f, p, skew_bytes / 16); };
      run(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int i = 0; i < 8; ++i) run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
      printf("pos=%2u skew=%8llu B  %7.3f ms  %7.1f GB/s\n", pos, (unsigned long long)skew_bytes, ms, 16.0 * N / ms / 1e6); fflush(stdout);
    }
  }
  return 0;
}
