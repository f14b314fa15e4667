// hq_state.hip -- state memory (hq_alloc*, hq_free), initial states (the device side of prepare_state,
// /root/reference/hybridq/circuit/simulation/utils.py:41-156), reductions and the device side of Measure / Projection.
#include "hq_common.h"
#include "hq_kernels_aux.h"

namespace hq {

template <typename T>
static int init_state_entry(T* re, T* im, unsigned n, int kind, uint64_t basis) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  if (!re || !im || n > 62) return fail("init_state: bad arguments");
  if (!is_device_pointer(re) || !is_device_pointer(im)) return fail("init_state: device pointers only");
  const uint64_t size = 1ull << n;
  if (kind == 0 && basis >= size) return fail("init_state: basis out of range");
  if (kind != 0 && kind != 1) return fail("init_state: unknown kind");
  const T amp = (T)std::pow(2.0, -0.5 * (double)n);
  const unsigned grid = (unsigned)std::min<uint64_t>((size + kBlock - 1) / kBlock, 256 * 32);
  HQ_LAUNCH(c, (init_state_kernel<T>), dim3(grid), dim3(kBlock), 0, re, im, size, kind, basis, amp);
  HQ_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename T>
static int init_product_entry(T* re, T* im, unsigned n_local, uint64_t hi_bits, uint64_t mask01, uint64_t val01,
                              uint64_t mask_minus, unsigned n_pm) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  if (!re || !im || n_local > 62) return fail("init_product_state: bad arguments");
  if (!is_device_pointer(re) || !is_device_pointer(im)) return fail("init_product_state: device pointers only");
  if ((val01 & ~mask01) || (mask01 & mask_minus)) return fail("init_product_state: inconsistent masks");
  if (n_local < 2 || (reinterpret_cast<uintptr_t>(re) % 32) || (reinterpret_cast<uintptr_t>(im) % 32))
    return fail("init_product_state: needs >= 2 local qubits and 32-byte aligned planes");
  if (n_local < 62 && (hi_bits & ((1ull << n_local) - 1))) return fail("init_product_state: hi_bits overlaps the local index");
  const uint64_t nquads = (1ull << n_local) / 4;
  const T amp = (T)std::pow(2.0, -0.5 * (double)n_pm);
  const unsigned grid = (unsigned)std::min<uint64_t>((nquads + kBlock - 1) / kBlock, 256 * 32);
  HQ_LAUNCH(c, (init_product_kernel<T>), dim3(grid), dim3(kBlock), 0, re, im, nquads, hi_bits, mask01, val01, mask_minus, amp);
  HQ_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename T>
static int norm2_entry(const T* re, const T* im, uint64_t size, double* out) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  HQ_NOT_RECORDABLE(c, "a reduction that returns a value to the host");
  if (!re || !im || !out) return fail("norm2: null pointer");
  if (!is_device_pointer(re) || !is_device_pointer(im)) return fail("norm2: device pointers only");
  void* s1 = nullptr;
  if (get_scratch(c, 1, 256, &s1)) return 1;
  HQ_HIP_CHECK(hipMemsetAsync(s1, 0, sizeof(double), c.stream));
  const unsigned grid = (unsigned)std::min<uint64_t>((size + kBlock - 1) / kBlock, 256 * 16);
  hipLaunchKernelGGL((norm2_kernel<T>), dim3(grid), dim3(kBlock), 0, c.stream, re, im, size,
                     (double*)s1);
  HQ_HIP_CHECK(hipGetLastError());
  HQ_HIP_CHECK(hipMemcpyAsync(out, s1, sizeof(double), hipMemcpyDeviceToHost, c.stream));
  HQ_HIP_CHECK(hipStreamSynchronize(c.stream));
  return 0;
}

template <typename T>
static int probabilities_entry(const T* re, const T* im, unsigned n, const unsigned* pos, unsigned k,
                               double* out) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  HQ_NOT_RECORDABLE(c, "a reduction that returns a value to the host");
  if (!re || !im || !pos || !out) return fail("probabilities: null pointer");
  if (k > kMaxK || check_positions(pos, n, k)) return fail("probabilities: invalid positions");
  if (!is_device_pointer(re) || !is_device_pointer(im)) return fail("probabilities: device pointers only");
  BitsArg ba;
  memset(&ba, 0, sizeof(ba));
  ba.k = k;
  for (unsigned j = 0; j < k; ++j) ba.pos[j] = pos[j];
  const size_t nb = (size_t)1 << k;
  void* s1 = nullptr;
  if (get_scratch(c, 1, std::max<size_t>(256, nb * sizeof(double)), &s1)) return 1;
  HQ_HIP_CHECK(hipMemsetAsync(s1, 0, nb * sizeof(double), c.stream));
  const uint64_t size = 1ull << n;
  constexpr unsigned kChunkBits = Vec<T>::VB + 8 + 6;  // probabilities_stream_kernel
  if (n >= kChunkBits && kBlock == 256) {
    const unsigned grid = (unsigned)std::min<uint64_t>(1ull << (n - kChunkBits), 256 * 8);
    hipLaunchKernelGGL((probabilities_stream_kernel<T>), dim3(grid), dim3(kBlock), nb * sizeof(double), c.stream,
                       re, im, n, ba, (double*)s1);
  } else {
    const unsigned grid = (unsigned)std::min<uint64_t>((size + kBlock - 1) / kBlock, 256 * 8);
    hipLaunchKernelGGL((probabilities_kernel<T>), dim3(grid), dim3(kBlock), nb * sizeof(double), c.stream, re,
                       im, size, ba, (double*)s1);
  }
  HQ_HIP_CHECK(hipGetLastError());
  HQ_HIP_CHECK(hipMemcpyAsync(out, s1, nb * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  HQ_HIP_CHECK(hipStreamSynchronize(c.stream));
  return 0;
}

template <typename T>
static int project_entry(T* re, T* im, unsigned n, const unsigned* pos, unsigned k, uint64_t state,
                         double scale) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  if (!re || !im || !pos) return fail("project: null pointer");
  if (k > 62 || check_positions(pos, n, k)) return fail("project: invalid positions");
  if (!is_device_pointer(re) || !is_device_pointer(im)) return fail("project: device pointers only");
  uint64_t mask = 0, want = 0;
  for (unsigned j = 0; j < k; ++j) {
    mask |= 1ull << pos[j];
    want |= ((state >> j) & 1ull) << pos[j];
  }
  const uint64_t size = 1ull << n;
  const unsigned grid = (unsigned)std::min<uint64_t>((size + kBlock - 1) / kBlock, 256 * 32);
  HQ_LAUNCH(c, (project_kernel<T>), dim3(grid), dim3(kBlock), 0, re, im, size, mask, want, (T)scale);
  HQ_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename T>
static int vdot_entry(const T* are, const T* aim, const T* bre, const T* bim, uint64_t size, double* out) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  HQ_NOT_RECORDABLE(c, "a reduction that returns a value to the host");
  if (!are || !aim || !bre || !bim || !out) return fail("vdot: null pointer");
  if (!is_device_pointer(are) || !is_device_pointer(aim) || !is_device_pointer(bre) || !is_device_pointer(bim))
    return fail("vdot: device pointers only");
  void* s1 = nullptr;
  if (get_scratch(c, 1, 256, &s1)) return 1;
  HQ_HIP_CHECK(hipMemsetAsync(s1, 0, 2 * sizeof(double), c.stream));
  const unsigned grid = (unsigned)std::min<uint64_t>((size + kBlock - 1) / kBlock, 256 * 16);
  hipLaunchKernelGGL((vdot_kernel<T>), dim3(grid), dim3(kBlock), 0, c.stream, are, aim, bre, bim, size,
                     (double*)s1);
  HQ_HIP_CHECK(hipGetLastError());
  HQ_HIP_CHECK(hipMemcpyAsync(out, s1, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  HQ_HIP_CHECK(hipStreamSynchronize(c.stream));
  return 0;
}

}  // namespace hq

extern "C" {

int hq_probabilities_float32(const float* re, const float* im, unsigned int n, const unsigned int* pos,
                             unsigned int k, double* out) {
  return hq::probabilities_entry<float>(re, im, n, pos, k, out);
}

int hq_probabilities_float64(const double* re, const double* im, unsigned int n, const unsigned int* pos,
                             unsigned int k, double* out) {
  return hq::probabilities_entry<double>(re, im, n, pos, k, out);
}

int hq_project_float32(float* re, float* im, unsigned int n, const unsigned int* pos, unsigned int k,
                       uint64_t state, double scale) {
  return hq::project_entry<float>(re, im, n, pos, k, state, scale);
}

int hq_project_float64(double* re, double* im, unsigned int n, const unsigned int* pos, unsigned int k,
                       uint64_t state, double scale) {
  return hq::project_entry<double>(re, im, n, pos, k, state, scale);
}

int hq_vdot_float32(const float* are, const float* aim, const float* bre, const float* bim, uint64_t size,
                    double* out) {
  return hq::vdot_entry<float>(are, aim, bre, bim, size, out);
}

int hq_vdot_float64(const double* are, const double* aim, const double* bre, const double* bim,
                    uint64_t size, double* out) {
  return hq::vdot_entry<double>(are, aim, bre, bim, size, out);
}

// State memory.  flags: bit 0 = physically contiguous VRAM (hipDeviceMallocContiguous): one PTE fragment
// covers a large range, which is worth ~14 % of streaming bandwidth on this part.
int hq_alloc(void** dev_ptr, uint64_t bytes, int flags) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  if (!dev_ptr || !bytes) return hq::fail("hq_alloc: bad arguments");
  if (hq::check_device(c)) return 1;
  hipError_t e = (flags & 1) ? hipExtMallocWithFlags(dev_ptr, (size_t)bytes, hipDeviceMallocContiguous)
                             : hipMalloc(dev_ptr, (size_t)bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *dev_ptr = nullptr;
    return hq::fail(std::string("hq_alloc: ") + hipGetErrorString(e));
  }
  return 0;
}

// Scattered placement: a VA-contiguous buffer whose physical granules (hipMemCreate, `granule` bytes
// each) are mapped in a seeded pseudo-random order (hipMemMap).
struct HqVmm { void* va; size_t size, granule; std::vector<hipMemGenericAllocationHandle_t> handles; };
static std::vector<HqVmm>& hq_vmm_registry() { static std::vector<HqVmm> r; return r; }

// Explicit placement: n_granules physical granules of `granule` bytes, created in sequence, granule i mapped at
// virtual slot va_slot[i] (a permutation of 0..n_granules-1).  *granule_min receives the driver's minimum.
int hq_alloc_mapped(void** dev_ptr, uint64_t granule, uint64_t n_granules, const uint32_t* va_slot, uint64_t* granule_min) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  if (hq::check_device(c)) return 1;
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = c.device;
  size_t gmin = 0;
  HQ_HIP_CHECK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  if (granule_min) *granule_min = gmin;
  if (!dev_ptr || !va_slot || !n_granules) return hq::fail("hq_alloc_mapped: bad arguments");
  if (granule % gmin) return hq::fail("hq_alloc_mapped: granule is not a multiple of the driver minimum " + std::to_string(gmin));
  HqVmm v;
  v.granule = granule;
  v.size = (size_t)n_granules * granule;
  v.va = nullptr;
  HQ_HIP_CHECK(hipMemAddressReserve(&v.va, v.size, (size_t)1 << 21, nullptr, 0));
  v.handles.resize(n_granules);
  for (size_t i = 0; i < n_granules; ++i) {
    hipError_t e = hipMemCreate(&v.handles[i], granule, &prop, 0);
    if (e != hipSuccess) return hq::fail(std::string("hipMemCreate: ") + hipGetErrorString(e));
  }
  for (size_t i = 0; i < n_granules; ++i) {
    if (va_slot[i] >= n_granules) return hq::fail("hq_alloc_mapped: slot out of range");
    hipError_t e = hipMemMap(reinterpret_cast<unsigned char*>(v.va) + (size_t)va_slot[i] * granule, granule, 0, v.handles[i], 0);
    if (e != hipSuccess) return hq::fail(std::string("hipMemMap: ") + hipGetErrorString(e));
  }
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = c.device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  HQ_HIP_CHECK(hipMemSetAccess(v.va, v.size, &acc, 1));
  hq_vmm_registry().push_back(v);
  *dev_ptr = v.va;
  return 0;
}

int hq_alloc_scattered(void** dev_ptr, uint64_t bytes, uint64_t granule, uint64_t seed) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  if (!dev_ptr || !bytes) return hq::fail("hq_alloc_scattered: bad arguments");
  if (hq::check_device(c)) return 1;
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = c.device;
  size_t gmin = 0;
  HQ_HIP_CHECK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  if (granule < gmin) granule = gmin;
  granule = (granule + gmin - 1) / gmin * gmin;
  const size_t ng = ((size_t)bytes + granule - 1) / granule;
  HqVmm v;
  v.granule = granule;
  v.size = ng * granule;
  v.va = nullptr;
  HQ_HIP_CHECK(hipMemAddressReserve(&v.va, v.size, (size_t)1 << 21, nullptr, 0));
  std::vector<size_t> order(ng);
  for (size_t i = 0; i < ng; ++i) order[i] = i;
  uint64_t st = seed * 6364136223846793005ull + 1442695040888963407ull;
  if (seed)
    for (size_t i = ng - 1; i > 0; --i) {  // Fisher-Yates with a 64-bit LCG
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      std::swap(order[i], order[(size_t)((st >> 33) % (i + 1))]);
    }
  v.handles.resize(ng);
  for (size_t i = 0; i < ng; ++i) {  // physical granules are created in sequence ...
    hipError_t e = hipMemCreate(&v.handles[i], granule, &prop, 0);
    if (e != hipSuccess) return hq::fail(std::string("hipMemCreate: ") + hipGetErrorString(e));
  }
  for (size_t i = 0; i < ng; ++i) {  // ... and mapped at shuffled virtual slots
    hipError_t e = hipMemMap(reinterpret_cast<unsigned char*>(v.va) + order[i] * granule, granule, 0, v.handles[i], 0);
    if (e != hipSuccess) return hq::fail(std::string("hipMemMap: ") + hipGetErrorString(e));
  }
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = c.device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  HQ_HIP_CHECK(hipMemSetAccess(v.va, v.size, &acc, 1));
  hq_vmm_registry().push_back(v);
  *dev_ptr = v.va;
  return 0;
}

int hq_free(void* dev_ptr) {
  if (!dev_ptr) return 0;
  auto& reg = hq_vmm_registry();
  for (size_t i = 0; i < reg.size(); ++i)
    if (reg[i].va == dev_ptr) {
      // The virtual range is NOT given back (hipMemAddressFree): on this stack (ROCm 7.0 runtime under torch,
      // measured with tools/vmm_integrity.py) a range that is unmapped and immediately reserved + mapped
      // again keeps stale translations -- reads and writes land in the old granules.  Address space is
      // 47 bits wide; the physical granules are what matters and they are released.
      (void)hipDeviceSynchronize();
      (void)hipMemUnmap(reg[i].va, reg[i].size);
      for (auto h : reg[i].handles) (void)hipMemRelease(h);
      static const bool free_va = getenv("HQ_VMM_FREE_VA") && atoi(getenv("HQ_VMM_FREE_VA")) != 0;
      if (free_va) (void)hipMemAddressFree(reg[i].va, reg[i].size);
      reg.erase(reg.begin() + (long)i);
      return 0;
    }
  hipError_t e = hipFree(dev_ptr);
  if (e != hipSuccess) return hq::fail(std::string("hq_free: ") + hipGetErrorString(e));
  return 0;
}

int hq_init_state_float32(float* re, float* im, unsigned int n, int kind, uint64_t basis) {
  return hq::init_state_entry<float>(re, im, n, kind, basis);
}

int hq_init_state_float64(double* re, double* im, unsigned int n, int kind, uint64_t basis) {
  return hq::init_state_entry<double>(re, im, n, kind, basis);
}

int hq_init_product_state_float32(float* re, float* im, unsigned int n_local, uint64_t hi_bits, uint64_t mask01,
                                  uint64_t val01, uint64_t mask_minus, unsigned int n_pm) {
  return hq::init_product_entry<float>(re, im, n_local, hi_bits, mask01, val01, mask_minus, n_pm);
}

int hq_init_product_state_float64(double* re, double* im, unsigned int n_local, uint64_t hi_bits, uint64_t mask01,
                                  uint64_t val01, uint64_t mask_minus, unsigned int n_pm) {
  return hq::init_product_entry<double>(re, im, n_local, hi_bits, mask01, val01, mask_minus, n_pm);
}

int hq_norm2_float32(const float* re, const float* im, uint64_t size, double* out) {
  return hq::norm2_entry<float>(re, im, size, out);
}

int hq_norm2_float64(const double* re, const double* im, uint64_t size, double* out) {
  return hq::norm2_entry<double>(re, im, size, out);
}

}  // extern "C"
