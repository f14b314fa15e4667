// tools/tune_direct.hip -- developer micro-benchmark (not part of the product): times
// variants of the streaming gate kernel and an in-place scale kernel (the RMW ceiling)
// on one GPU.  Build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_direct.hip -o /tmp/tune && /tmp/tune [n]
#include "../hybridq_amd/csrc/hq_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
using namespace hq;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// in-place scale, float4, ILP vectors per thread, optional grid-stride
template <int ILP, bool NT>
__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ re, float* __restrict__ im, float a, uint64_t nvec) {
  f32x4* vr = (f32x4*)re; f32x4* vi = (f32x4*)im;
  for (uint64_t g = (uint64_t)blockIdx.x * (ILP * 256) + threadIdx.x; g < nvec; g += (uint64_t)gridDim.x * (ILP * 256)) {
    f32x4 x[ILP], y[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (NT) { x[i] = __builtin_nontemporal_load(&vr[g + i * 256]); y[i] = __builtin_nontemporal_load(&vi[g + i * 256]); }
      else { x[i] = vr[g + i * 256]; y[i] = vi[g + i * 256]; }
    }
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      f32x4 o = x[i] * a - y[i] * 0.5f, p = y[i] * a + x[i] * 0.5f;
      if (NT) { __builtin_nontemporal_store(o, &vr[g + i * 256]); __builtin_nontemporal_store(p, &vi[g + i * 256]); }
      else { vr[g + i * 256] = o; vi[g + i * 256] = p; }
    }
  }
}

// grid-stride variant of the direct kernel (persistent blocks)
template <typename T, int K, int VMASK, int ILP, bool NT>
__global__ void __launch_bounds__(256)
apply_direct_gs(T* __restrict__ re, T* __restrict__ im, const GateArg<T, K> U, const RegPos rp, uint64_t ngroups) {
  using V = typename Vec<T>::type;
  constexpr int VB = Vec<T>::VB, VE = 1 << VB;
  constexpr int KV = popc_c(VMASK), KR = K - KV, R = 1 << KR, D = 1 << K;
  V* __restrict__ vre = reinterpret_cast<V*>(re);
  V* __restrict__ vim = reinterpret_cast<V*>(im);
  uint64_t off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { uint64_t o = 0;
#pragma unroll
    for (int j = 0; j < KR; ++j) o |= (uint64_t)((r >> j) & 1) << rp.p[j];
    off[r] = o; }
  for (uint64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const uint64_t g0 = grp * (ILP * 256) + threadIdx.x;
    uint64_t vb[ILP]; V xr[ILP][R], xi[ILP][R];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      uint64_t v = g0 + (uint64_t)i * 256;
#pragma unroll
      for (int j = 0; j < KR; ++j) { const uint64_t lo = (1ull << rp.p[j]) - 1; v = ((v & ~lo) << 1) | (v & lo); }
      vb[i] = v;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (NT) { xr[i][r] = __builtin_nontemporal_load(&vre[v | off[r]]); xi[i][r] = __builtin_nontemporal_load(&vim[v | off[r]]); }
        else { xr[i][r] = vre[v | off[r]]; xi[i][r] = vim[v | off[r]]; }
      }
    }
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
#pragma unroll
      for (int ro = 0; ro < R; ++ro) {
        V yr, yi;
#pragma unroll
        for (int co = 0; co < VE; ++co) {
          const int to = pext_c(co, VMASK) | (ro << KV); const int cfree = co & ~VMASK;
          T ar = 0, ai = 0;
#pragma unroll
          for (int ti = 0; ti < D; ++ti) {
            const int ci = pdep_c(ti & ((1 << KV) - 1), VMASK) | cfree; const int ri = ti >> KV;
            const T ur = U.re[to * D + ti], ui = U.im[to * D + ti];
            const T pr = xr[i][ri][ci], pi = xi[i][ri][ci];
            ar = __builtin_fma(ur, pr, ar); ar = __builtin_fma(-ui, pi, ar);
            ai = __builtin_fma(ur, pi, ai); ai = __builtin_fma(ui, pr, ai);
          }
          yr[co] = ar; yi[co] = ai;
        }
        if (NT) { __builtin_nontemporal_store(yr, &vre[vb[i] | off[ro]]); __builtin_nontemporal_store(yi, &vim[vb[i] | off[ro]]); }
        else { vre[vb[i] | off[ro]] = yr; vim[vb[i] | off[ro]] = yi; }
      }
    }
  }
}

static float *re, *im; static unsigned n; static hipEvent_t e0, e1;

template <typename F> static double time_ms(F f, int reps = 10) {
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}
static void report(const char* name, double ms) {
  const double bytes = 16.0 * (double)(1ull << n);
  printf("%-44s %8.3f ms  %7.1f GB/s  %5.1f%% of 8TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
  fflush(stdout);
}

template <int K> static GateArg<float, K> rand_gate() {
  GateArg<float, K> g; for (int i = 0; i < (1 << (2 * K)); ++i) { g.re[i] = 0.3f * (float)(rand() % 7 - 3) / 3.0f; g.im[i] = 0.3f * (float)(rand() % 5 - 2) / 2.0f; } return g;
}

template <int K, int VMASK, int ILP, bool NT> static void run_direct(std::vector<unsigned> pos, bool gs, unsigned gridcap = 0) {
  constexpr int KV = popc_c(VMASK), KR = K - KV;
  RegPos rp = {{0,0,0,0}}; for (int j = 0; j < KR; ++j) rp.p[j] = pos[KV + j] - 2;
  auto g = rand_gate<K>();
  const uint64_t nslots = 1ull << (n - 2 - KR); const uint64_t ngroups = nslots / (ILP * 256);
  char name[128]; std::string ps; for (auto p : pos) ps += std::to_string(p) + ",";
  if (!gs) {
    snprintf(name, sizeof name, "direct K=%d pos=%s ILP=%d NT=%d", K, ps.c_str(), ILP, (int)NT);
    report(name, time_ms([&] { hipLaunchKernelGGL((apply_direct_kernel<float, K, VMASK, ILP, NT>), dim3((unsigned)ngroups), dim3(256), 0, 0, re, im, g, rp); }));
  } else {
    snprintf(name, sizeof name, "direct-gs K=%d pos=%s ILP=%d NT=%d grid=%u", K, ps.c_str(), ILP, (int)NT, gridcap);
    report(name, time_ms([&] { hipLaunchKernelGGL((apply_direct_gs<float, K, VMASK, ILP, NT>), dim3(gridcap), dim3(256), 0, 0, re, im, g, rp, ngroups); }));
  }
}

template <int ILP, bool NT> static void run_scale(unsigned grid) {
  const uint64_t nvec = (1ull << n) / 4; char name[128];
  unsigned g = grid ? grid : (unsigned)(nvec / (ILP * 256));
  snprintf(name, sizeof name, "scale in-place ILP=%d NT=%d grid=%u", ILP, (int)NT, g);
  report(name, time_ms([&] { hipLaunchKernelGGL((scale_kernel<ILP, NT>), dim3(g), dim3(256), 0, 0, re, im, 0.7f, nvec); }));
}

int main(int argc, char** argv) {
  n = argc > 1 ? atoi(argv[1]) : 30;
  CK(hipMalloc(&re, sizeof(float) << n)); CK(hipMalloc(&im, sizeof(float) << n));
  CK(hipMemset(re, 0, sizeof(float) << n)); CK(hipMemset(im, 0, sizeof(float) << n));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((init_state_kernel<float>), dim3(8192), dim3(256), 0, 0, re, im, 1ull << n, 1, 0ull, 1e-3f);
  printf("# n=%u, %.1f GiB per plane; algorithmic bytes per pass = 16*2^n\n", n, (double)(sizeof(float) << n) / (1 << 30));
  // D2D memcpy as a reference point (read 1 plane + write 1 plane = 8*2^n bytes)
  { double ms = time_ms([&] { CK(hipMemcpyAsync(im, re, sizeof(float) << n, hipMemcpyDeviceToDevice, 0)); }); printf("%-44s %8.3f ms  %7.1f GB/s (8*2^n bytes)\n", "hipMemcpy D2D plane->plane", ms, 8.0 * (1ull << n) / ms / 1e6); }
  run_scale<1, false>(0); run_scale<2, false>(0); run_scale<4, false>(0); run_scale<8, false>(0);
  run_scale<1, true>(0); run_scale<2, true>(0); run_scale<4, true>(0);
  run_scale<2, false>(256 * 8); run_scale<4, false>(256 * 8); run_scale<4, false>(256 * 4); run_scale<4, true>(256 * 8); run_scale<8, false>(256*4);
  // K=1
  run_direct<1, 0, 1, false>({12}, false); run_direct<1, 0, 2, false>({12}, false); run_direct<1, 0, 4, false>({12}, false);
  run_direct<1, 0, 2, true>({12}, false); run_direct<1, 0, 4, true>({12}, false);
  run_direct<1, 0, 2, false>({12}, true, 256 * 8); run_direct<1, 0, 2, false>({12}, true, 256 * 4); run_direct<1, 0, 4, false>({12}, true, 256 * 4);
  run_direct<1, 0, 2, true>({12}, true, 256 * 8);
  for (unsigned p : {2u, 3u, 4u, 5u, 6u, 7u, 8u, 9u, 10u, 16u, 20u, 25u, 29u}) run_direct<1, 0, 2, false>({p}, false);
  for (unsigned p : {2u, 4u, 7u, 20u, 29u}) run_direct<1, 0, 2, true>({p}, false);
  run_direct<1, 1, 4, false>({0}, false); run_direct<1, 2, 4, false>({1}, false); run_direct<1, 1, 8, false>({0}, false);
  // K=2
  run_direct<2, 0, 1, false>({12, 20}, false); run_direct<2, 0, 2, false>({12, 20}, false); run_direct<2, 0, 1, true>({12, 20}, false);
  run_direct<2, 0, 1, false>({12, 20}, true, 256 * 4); run_direct<2, 0, 1, false>({12, 20}, true, 256 * 8);
  run_direct<2, 0, 1, false>({2, 3}, false); run_direct<2, 0, 1, false>({4, 5}, false); run_direct<2, 0, 1, false>({6, 20}, false); run_direct<2, 0, 1, false>({28, 29}, false);
  run_direct<2, 1, 2, false>({0, 15}, false); run_direct<2, 3, 4, false>({0, 1}, false);
  // K=3
  run_direct<3, 0, 1, false>({10, 15, 20}, false); run_direct<3, 0, 1, true>({10, 15, 20}, false); run_direct<3, 0, 1, false>({2, 3, 4}, false);
  run_direct<3, 0, 1, false>({10, 15, 20}, true, 256 * 2);
  return 0;
}
