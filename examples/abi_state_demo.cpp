// The state allocator of the C ABI (include/hq_hip.h: hq_alloc_state / hq_free_state / hq_state_info), no Python:
//   hipcc --offload-arch=gfx950 -I include examples/abi_state_demo.cpp -o /tmp/abi_state_demo -ldl
//   /tmp/abi_state_demo hybridq_amd/csrc/libhq_hip.so [n_qubits = 30]
// Allocates an n-qubit complex64 state twice -- HQ_STATE_PLAIN (hipMalloc, what a caller's own buffers are) and the
// default tuned placement (draw-probe-keep inside the library) --, streams the same 1-/2-qubit gates through
// apply_U_float32 (the reference's entry point, python_U.cpp:131-136) on both and prints the HBM rate of each
// (16 * 2^n bytes per gate application); then frees the tuned state and allocates it again: the second
// allocation must come from the pool (no new search).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "hq_hip.h"

#define LOAD(name) auto p_##name = reinterpret_cast<decltype(&name)>(dlsym(lib, #name)); \
  if (!p_##name) { std::fprintf(stderr, "missing symbol %s\n", #name); return 2; }

int main(int argc, char** argv) {
  void* lib = dlopen(argc > 1 ? argv[1] : "libhq_hip.so", RTLD_NOW);
  if (!lib) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  LOAD(apply_U_float32) LOAD(hq_alloc_state) LOAD(hq_free_state) LOAD(hq_state_info) LOAD(hq_init_state_float32)
  LOAD(hq_norm2_float32) LOAD(hq_sync) LOAD(hq_last_error)
  const unsigned n = argc > 2 ? (unsigned)std::atoi(argv[2]) : 30;
  const float s = 1.0f / std::sqrt(2.0f);
  const float H[8] = {s, 0, s, 0, s, 0, -s, 0};
  float HH[32] = {0};
  for (int r = 0; r < 4; ++r)
    for (int q = 0; q < 4; ++q) HH[2 * (4 * r + q)] = 0.5f * (__builtin_parity(r & q) ? -1.0f : 1.0f);
  auto rate = [&](float* re, float* im, double* tbps) -> int {
    if (p_hq_init_state_float32(re, im, n, 1 /* |+...+> */, 0)) return 1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) || hipEventCreate(&e1)) return 1;
    int gates = 0;
    for (int rep = 0; rep < 3; ++rep) {  // rep 0 warms up
      if (rep == 1 && (p_hq_sync() || hipEventRecord(e0, nullptr))) return 1;
      for (unsigned q = 0; q < n; q += 2) {
        const unsigned p1[1] = {q}, p2[2] = {q, (q + n / 2) % n};
        if (p_apply_U_float32(re, im, H, p1, n, 1) || p_apply_U_float32(re, im, HH, p2, n, 2)) return 1;
        if (rep) gates += 2;
      }
    }
    if (hipEventRecord(e1, nullptr) || hipEventSynchronize(e1)) return 1;
    float ms = 0;
    if (hipEventElapsedTime(&ms, e0, e1)) return 1;
    *tbps = gates * 16.0 * std::pow(2.0, (double)n) / (ms * 1e-3) / 1e12;
    double norm2 = 0;
    if (p_hq_norm2_float32(re, im, (uint64_t)1 << n, &norm2) || std::fabs(norm2 - 1.0) > 1e-3) return 1;
    return 0;
  };
  char info[8192];
  void *re = nullptr, *im = nullptr;
  double plain = 0, tuned = 0;
  if (p_hq_alloc_state(n, 32, HQ_STATE_PLAIN, &re, &im)) { std::puts(p_hq_last_error()); return 3; }
  if (rate((float*)re, (float*)im, &plain)) { std::puts(p_hq_last_error()); return 4; }
  if (p_hq_free_state(re)) return 5;
  auto t0 = std::chrono::steady_clock::now();
  if (p_hq_alloc_state(n, 32, 0, &re, &im)) { std::puts(p_hq_last_error()); return 6; }
  const double search_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (p_hq_state_info(re, info, sizeof(info))) return 7;
  std::printf("placement: %s\n", info);
  if (rate((float*)re, (float*)im, &tuned)) { std::puts(p_hq_last_error()); return 8; }
  if (p_hq_free_state(re)) return 9;
  t0 = std::chrono::steady_clock::now();
  void *re2 = nullptr, *im2 = nullptr;
  if (p_hq_alloc_state(n, 32, 0, &re2, &im2)) { std::puts(p_hq_last_error()); return 10; }
  const double again_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (p_hq_state_info(re2, info, sizeof(info))) return 11;
  const bool pooled = std::strstr(info, "from_pool") != nullptr && re2 == re;
  if (p_hq_free_state(re2)) return 12;
  std::printf("n=%u plain_TBps=%.3f tuned_TBps=%.3f search_s=%.2f realloc_s=%.4f from_pool=%d\n", n, plain, tuned, search_s,
              again_s, pooled ? 1 : 0);
  return 0;
}
