// Minimal C/C++ user of the drop-in boundary (include/hq_hip.h), no Python involved:
//   hipcc --offload-arch=gfx950 -I include examples/abi_demo.cpp -o /tmp/abi_demo -ldl
//   /tmp/abi_demo hybridq_amd/csrc/libhq_hip.so
// Allocates a 2^n split re/im state vector in HBM, prepares |0...0>, applies H on every qubit and
// a CZ chain through the reference's own entry point (apply_U_float32, python_U.cpp:133-141),
// checks norm and amplitudes, interleaves with hq_to_complex64.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#include "hq_hip.h"

#define LOAD(name) auto p_##name = reinterpret_cast<decltype(&name)>(dlsym(lib, #name)); \
  if (!p_##name) { std::fprintf(stderr, "missing symbol %s\n", #name); return 2; }

int main(int argc, char** argv) {
  void* lib = dlopen(argc > 1 ? argv[1] : "libhq_hip.so", RTLD_NOW);
  if (!lib) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  LOAD(apply_U_float32) LOAD(get_log2_pack_size) LOAD(hq_init_state_float32) LOAD(hq_norm2_float32)
  LOAD(hq_to_complex64) LOAD(hq_sync) LOAD(hq_last_error)
  const unsigned n = 20;
  const size_t size = (size_t)1 << n;
  float *re = nullptr, *im = nullptr, *out = nullptr;
  if (hipMalloc(&re, size * 4) || hipMalloc(&im, size * 4) || hipMalloc(&out, size * 8)) return 3;
  if (p_hq_init_state_float32(re, im, n, 0 /* basis state */, 0)) { std::puts(p_hq_last_error()); return 4; }
  const float s = 1.0f / std::sqrt(2.0f);
  const float H[8] = {s, 0, s, 0, s, 0, -s, 0};  // row-major, interleaved (re, im)
  float CZ[32] = {0};
  for (int d = 0; d < 4; ++d) CZ[2 * (d * 4 + d)] = d == 3 ? -1.0f : 1.0f;
  for (unsigned q = 0; q < n; ++q)
    if (p_apply_U_float32(re, im, H, &q, n, 1)) { std::puts(p_hq_last_error()); return 5; }
  for (unsigned q = 0; q + 1 < n; ++q) {
    const unsigned pos[2] = {q, q + 1};
    if (p_apply_U_float32(re, im, CZ, pos, n, 2)) { std::puts(p_hq_last_error()); return 6; }
  }
  double norm2 = 0;
  if (p_hq_norm2_float32(re, im, size, &norm2)) return 7;
  if (p_hq_to_complex64(re, im, out, size) || p_hq_sync()) return 8;
  std::vector<float> host(2 * size);
  if (hipMemcpy(host.data(), out, size * 8, hipMemcpyDeviceToHost)) return 9;
  // graph state: every amplitude is +-2^(-n/2), sign = parity of the number of adjacent 11 pairs
  const float a = std::pow(2.0f, -0.5f * n);
  size_t bad = 0;
  for (size_t x = 0; x < size; ++x) {
    const int sign = __builtin_parityll(x & (x >> 1)) ? -1 : 1;
    if (std::fabs(host[2 * x] - sign * a) > 1e-6f * a * 10 || host[2 * x + 1] != 0.0f) ++bad;
  }
  std::printf("log2_pack_size=%u norm2=%.9f wrong_amplitudes=%zu\n", p_get_log2_pack_size(), norm2, bad);
  return (bad == 0 && std::fabs(norm2 - 1.0) < 1e-5) ? 0 : 1;
}
