// AddressSanitizer smoke of the kernels that index LDS / global memory with computed addresses (VERDICT r02 next #6b).
// The library is built with `-fsanitize=address --offload-arch=gfx950:xnack+ -DHQ_ASAN` (tools/asan_smoke.sh); this driver
// (plain host code) dlopen()s it and runs every kernel family once on small states (n = 16..18).  An out-of-bounds or
// misaligned device access aborts the process with an ASAN report; the driver itself only checks return codes.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "hq_hip.h"

#define LOAD(name) auto p_##name = reinterpret_cast<decltype(&name)>(dlsym(lib, #name)); \
  if (!p_##name) { std::fprintf(stderr, "missing symbol %s\n", #name); return 2; }
#define OK(expr) do { if ((expr) != 0) { std::fprintf(stderr, "FAILED %s: %s\n", #expr, p_hq_last_error()); return 1; } ++calls; } while (0)

int main(int argc, char** argv) {
  void* lib = dlopen(argc > 1 ? argv[1] : "libhq_hip_asan.so", RTLD_NOW);
  if (!lib) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  LOAD(apply_U_float32) LOAD(apply_U_float64) LOAD(swap_float32) LOAD(swap_float64) LOAD(hq_permute_bits_32)
  LOAD(hq_permute_bits_64) LOAD(hq_apply_blocked_float32) LOAD(hq_apply_blocked_float64) LOAD(hq_exchange_float32)
  LOAD(hq_init_state_float32) LOAD(hq_init_state_float64) LOAD(hq_init_product_state_float32) LOAD(hq_to_complex64)
  LOAD(hq_norm2_float32) LOAD(hq_norm2_float64) LOAD(hq_probabilities_float32) LOAD(hq_project_float32) LOAD(hq_sync)
  LOAD(hq_last_error) LOAD(hq_set_apply_mode)
  const unsigned n = argc > 2 ? (unsigned)std::atoi(argv[2]) : 17;
  const size_t size = (size_t)1 << n;
  int calls = 0;
  std::mt19937 rng(3);
  std::normal_distribution<double> nd;
  float *re = nullptr, *im = nullptr, *tmp = nullptr, *tmp2 = nullptr;
  double *dre = nullptr, *dim_ = nullptr;
  if (hipMalloc(&re, size * 4) || hipMalloc(&im, size * 4) || hipMalloc(&tmp, size * 8) || hipMalloc(&tmp2, size * 8) ||
      hipMalloc(&dre, size * 8) || hipMalloc(&dim_, size * 8)) return 3;
  OK(p_hq_init_state_float32(re, im, n, 1, 0));
  OK(p_hq_init_state_float64(dre, dim_, n, 1, 0));
  for (const char* mode : {"auto", "direct", "generic", "tile", "gemm"}) {
    OK(p_hq_set_apply_mode(mode));
    for (unsigned k = 1; k <= 8; ++k)
      for (int pat = 0; pat < 3; ++pat) {
        std::vector<unsigned> pos(n);
        std::iota(pos.begin(), pos.end(), 0u);
        if (pat == 0) std::shuffle(pos.begin(), pos.end(), rng);
        if (pat == 2) std::reverse(pos.begin(), pos.end());
        const size_t D = (size_t)1 << k;
        std::vector<float> U(2 * D * D);
        std::vector<double> Ud(2 * D * D);
        for (size_t i = 0; i < U.size(); ++i) { Ud[i] = nd(rng) / std::sqrt((double)D); U[i] = (float)Ud[i]; }
        OK(p_apply_U_float32(re, im, U.data(), pos.data(), n, k));
        OK(p_apply_U_float64(dre, dim_, Ud.data(), pos.data(), n, k));
      }
  }
  OK(p_hq_set_apply_mode("auto"));
  // cache-blocked passes: several gates inside one LDS tile (table-driven and computed-address forms by size)
  for (unsigned ng : {3u, 40u}) {
    std::vector<unsigned> tile32(13), tile64(12), pos, pos64, ks;
    std::iota(tile32.begin(), tile32.end(), 0u);
    std::iota(tile64.begin(), tile64.end(), 0u);
    tile32[12] = n - 1;
    tile64[11] = n - 2;
    std::vector<float> U;
    std::vector<double> Ud;
    for (unsigned g = 0; g < ng; ++g) {
      const unsigned k = 1 + g % 4;
      std::vector<unsigned> t32 = tile32, t64 = tile64;
      std::shuffle(t32.begin(), t32.end(), rng);
      std::shuffle(t64.begin(), t64.end(), rng);
      for (unsigned j = 0; j < k; ++j) { pos.push_back(t32[j]); pos64.push_back(t64[j]); }
      ks.push_back(k);
      const size_t D = (size_t)1 << k;
      for (size_t i = 0; i < 2 * D * D; ++i) { const double v = nd(rng) / std::sqrt((double)D); U.push_back((float)v); Ud.push_back(v); }
    }
    OK(p_hq_apply_blocked_float32(re, im, n, tile32.data(), 13, ng, U.data(), pos.data(), ks.data()));
    OK(p_hq_apply_blocked_float64(dre, dim_, n, tile64.data(), 12, ng, Ud.data(), pos64.data(), ks.data()));
  }
  for (unsigned s : {3u, 8u, 12u, 13u, 14u, 15u, 16u}) {  // low-bit swaps: table kernel, one-pass tiles, two-pass
    if (s > n) continue;
    std::vector<unsigned> pos(s);
    std::iota(pos.begin(), pos.end(), 0u);
    if (s == 16) std::rotate(pos.begin(), pos.begin() + 3, pos.end()); else std::shuffle(pos.begin(), pos.end(), rng);
    OK(p_swap_float32(re, pos.data(), n, s));
    OK(p_swap_float64(dre, pos.data(), n, s));
  }
  for (int pat = 0; pat < 3; ++pat) {  // arbitrary bit permutations and the exchange pack (one rank)
    std::vector<unsigned> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    if (pat == 0) std::shuffle(perm.begin(), perm.end(), rng);
    if (pat == 1) std::reverse(perm.begin(), perm.end());
    if (pat == 2) std::rotate(perm.begin(), perm.begin() + 5, perm.end());
    OK(p_hq_permute_bits_32(re, tmp, perm.data(), n));
    OK(p_hq_permute_bits_64(dre, tmp2, perm.data(), n));
    int where = 0;
    OK(p_hq_exchange_float32(re, im, tmp, tmp + size, n, perm.data(), &where));
  }
  OK(p_hq_init_product_state_float32(re, im, n, 0, 5, 1, 8, n - 2));
  OK(p_hq_to_complex64(re, im, tmp, size));
  double out[1024], nrm = 0;
  const unsigned mp[3] = {2, n / 2, n - 1};
  OK(p_hq_probabilities_float32(re, im, n, mp, 3, out));
  OK(p_hq_project_float32(re, im, n, mp, 3, 1, 1.0));
  OK(p_hq_norm2_float32(re, im, size, &nrm));
  OK(p_hq_norm2_float64(dre, dim_, size, &nrm));
  OK(p_hq_sync());
  if (hipDeviceSynchronize() != hipSuccess) { std::fprintf(stderr, "device error\n"); return 1; }
  std::printf("ASAN smoke: %d library calls on n=%u completed without a sanitizer report\n", calls, n);
  return 0;
}
