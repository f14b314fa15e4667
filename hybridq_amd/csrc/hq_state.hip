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


// ---------------------------------------------------------------------------------
// state memory: VMM-mapped placements, the draw-probe-keep search and the pool of winning placements
// ---------------------------------------------------------------------------------
// A VA-contiguous buffer whose physical granules (hipMemCreate, `granule` bytes each) are mapped in a chosen order.
struct Vmm {
  void* va = nullptr;
  size_t size = 0, granule = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
  size_t mapped = 0;     // granules currently mapped
  bool touched = false;  // the range has held a mapping at some point (never handed back: see vmm_destroy)
};
static std::vector<Vmm>& vmm_registry() { static std::vector<Vmm> r; return r; }

static hipMemAllocationProp vmm_prop(const Context& c) {
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = c.device;
  return prop;
}

static int vmm_granule_min(const Context& c, size_t* gmin) {
  const hipMemAllocationProp prop = vmm_prop(c);
  HQ_HIP_CHECK(hipMemGetAllocationGranularity(gmin, &prop, hipMemAllocationGranularityMinimum));
  return 0;
}

// Releases the physical granules.  The virtual range is NOT given back (hipMemAddressFree) unless HQ_VMM_FREE_VA=1:
// on this stack (ROCm 7.0 runtime under torch, tools/vmm_integrity.py) a range that is unmapped and immediately
// reserved + mapped again keeps stale translations -- reads and writes land in the old granules.  Address space is 47
// bits wide; the pool below makes retirements rare (one search per state size and process).
static void vmm_destroy(Vmm& v) {
  if (v.va && v.mapped) (void)hipMemUnmap(v.va, v.size);
  for (auto h : v.handles) (void)hipMemRelease(h);
  static const bool free_va = env_int("HQ_VMM_FREE_VA", 0) != 0;
  if (v.va && (free_va || !v.touched)) (void)hipMemAddressFree(v.va, v.size);  // a range that never held a mapping is safe to return
  v = Vmm();
}

// granule i (creation order) -> virtual slot order[i].  Every failure releases what was created.
static int vmm_create(const Context& c, size_t granule, const std::vector<size_t>& order, Vmm& out) {
  const size_t ng = order.size();
  Vmm v;
  v.granule = granule;
  v.size = ng * granule;
  hipError_t e = hipMemAddressReserve(&v.va, v.size, (size_t)1 << 21, nullptr, 0);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(std::string("hipMemAddressReserve: ") + hipGetErrorString(e)); }
  const hipMemAllocationProp prop = vmm_prop(c);
  v.handles.reserve(ng);
  for (size_t i = 0; i < ng && e == hipSuccess; ++i) {  // physical granules are created in sequence ...
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, granule, &prop, 0);
    if (e == hipSuccess) v.handles.push_back(h);
  }
  const char* what = "hipMemCreate";
  if (e == hipSuccess) {
    what = "hipMemMap";
    for (size_t i = 0; i < ng && e == hipSuccess; ++i) {  // ... and mapped at the chosen virtual slots
      e = hipMemMap(reinterpret_cast<unsigned char*>(v.va) + order[i] * granule, granule, 0, v.handles[i], 0);
      if (e == hipSuccess) { v.mapped = i + 1; v.touched = true; }
    }
  }
  if (e == hipSuccess) {
    what = "hipMemSetAccess";
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = c.device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(v.va, v.size, &acc, 1);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (v.mapped) {  // unmap granule by granule what was mapped (a partial range cannot be unmapped in one call)
      for (size_t i = 0; i < v.mapped; ++i)
        (void)hipMemUnmap(reinterpret_cast<unsigned char*>(v.va) + order[i] * granule, granule);
      v.mapped = 0;
    }
    vmm_destroy(v);
    return fail(std::string(what) + ": " + hipGetErrorString(e));
  }
  v.mapped = ng;
  out = v;
  return 0;
}

static std::vector<size_t> vmm_order(size_t ng, uint64_t seed) {
  std::vector<size_t> order(ng);
  for (size_t i = 0; i < ng; ++i) order[i] = i;
  uint64_t st = seed * 6364136223846793005ull + 1442695040888963407ull;
  if (seed)
    for (size_t i = ng - 1; i > 0; --i) {  // Fisher-Yates with a 64-bit LCG
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      std::swap(order[i], order[(size_t)((st >> 33) % (i + 1))]);
    }
  return order;
}

// The SAME physical granules mapped in another order into a fresh virtual range; `v` is unmapped (its range retired) and
// must not be used afterwards.
static int vmm_remap(const Context& c, Vmm& v, const std::vector<size_t>& order, Vmm& out) {
  HQ_HIP_CHECK(hipDeviceSynchronize());
  Vmm nv;
  nv.granule = v.granule;
  nv.size = v.size;
  HQ_HIP_CHECK(hipMemAddressReserve(&nv.va, nv.size, (size_t)1 << 21, nullptr, 0));
  hipError_t e = hipMemUnmap(v.va, v.size);
  if (e != hipSuccess) {
    (void)hipMemAddressFree(nv.va, nv.size);
    return fail(std::string("hipMemUnmap: ") + hipGetErrorString(e));
  }
  v.mapped = 0;
  nv.handles.swap(v.handles);  // from here on nv owns the physical granules: every failure below releases them
  const char* what = "hipMemMap";
  for (size_t i = 0; i < nv.handles.size() && e == hipSuccess; ++i) {
    e = hipMemMap(reinterpret_cast<unsigned char*>(nv.va) + order[i] * nv.granule, nv.granule, 0, nv.handles[i], 0);
    if (e == hipSuccess) { nv.mapped = i + 1; nv.touched = true; }
  }
  if (e == hipSuccess) {
    what = "hipMemSetAccess";
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = c.device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(nv.va, nv.size, &acc, 1);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    for (size_t i = 0; i < nv.mapped; ++i) (void)hipMemUnmap(reinterpret_cast<unsigned char*>(nv.va) + order[i] * nv.granule, nv.granule);
    nv.mapped = 0;
    vmm_destroy(nv);
    return fail(std::string(what) + ": " + hipGetErrorString(e));
  }
  out = nv;
  return 0;
}

// One state = one allocation holding both planes: re at the base, im `stride` elements later (2^n + a pad that keeps the
// two streams of every kernel out of step in the HBM channel hash, rounded to 32 bytes: U.h:34-36 wants that alignment).
struct StateAlloc {
  void* re = nullptr;
  void* im = nullptr;
  unsigned n = 0, float_bits = 0;
  size_t bytes = 0;
  bool tuned = false;
  Vmm vmm;            // tuned placements
  void* plain = nullptr;  // hipMalloc placements
  double probe_ms = 0;
  std::string layout, report;  // the report is the search's record and never changes afterwards
  bool from_pool = false;      // added to the report only when it is emitted (hq_state_info)
  std::vector<size_t> vmm_order_used;  // granule -> virtual slot of the mapping in use
};
struct StatePool {
  std::vector<StateAlloc> live, idle;
  std::string last_report = "{}";
  bool last_from_pool = false;
};
static std::string emit_report(const std::string& report, bool from_pool) {
  const size_t close = report.rfind('}');
  if (!from_pool || close == std::string::npos) return report;
  return report.substr(0, close) + (close > 1 ? ", " : "") + "\"from_pool\": true}";
}
static StatePool& state_pool() { static StatePool p; return p; }

static void state_release(StateAlloc& st) {
  if (st.tuned) vmm_destroy(st.vmm);
  else if (st.plain) (void)hipFree(st.plain);
  st = StateAlloc();
}

constexpr size_t kPlanePadBytes = 12288;     // measured at n = 30 (round 1 sweep; script in the history): +3..15 % for high targets
constexpr size_t kTunedMinBytes = 1u << 28;  // states below this stay on hipMalloc memory (HQ_STATE_TUNED_MIN_BYTES: tests)
static size_t tuned_min_bytes() {
  const char* e = getenv("HQ_STATE_TUNED_MIN_BYTES");
  return e && atoll(e) > 0 ? (size_t)atoll(e) : kTunedMinBytes;
}

static size_t state_stride(unsigned n, size_t itemsize) {
  const size_t pad = n >= 12 ? kPlanePadBytes / itemsize : 0;
  return ((((size_t)1 << n) + pad) * itemsize + 31) / 32 * 32 / itemsize;
}

// Average time of a gate application on the candidate planes (contents are overwritten): the probe of the search.
template <typename T>
static int state_probe(Context& c, T* re, T* im, unsigned n, double* ms) {
  const T h = (T)0.70710678118654752440;
  const T U1[8] = {h, 0, h, 0, h, 0, -h, 0};
  T U2[32];
  for (int r = 0; r < 4; ++r)
    for (int q = 0; q < 4; ++q) {
      U2[2 * (4 * r + q)] = (T)0.5 * (((r & q & 1) ^ ((r & q) >> 1)) ? (T)-1 : (T)1);  // H (x) H
      U2[2 * (4 * r + q) + 1] = 0;
    }
  const unsigned p1[3] = {3, n / 2, n - 1}, p2[2] = {5, n - 3};
  auto gates = [&]() -> int {
    for (unsigned p : p1) {
      const unsigned pos[1] = {p};
      if ((sizeof(T) == 4 ? apply_device_f32(c, (float*)re, (float*)im, (const float*)U1, pos, n, 1)
                          : apply_device_f64(c, (double*)re, (double*)im, (const double*)U1, pos, n, 1))) return 1;
    }
    return sizeof(T) == 4 ? apply_device_f32(c, (float*)re, (float*)im, (const float*)U2, p2, n, 2)
                          : apply_device_f64(c, (double*)re, (double*)im, (const double*)U2, p2, n, 2);
  };
  const uint64_t size = 1ull << n;
  const unsigned grid = (unsigned)std::min<uint64_t>((size + kBlock - 1) / kBlock, 256 * 32);
  hipLaunchKernelGGL((init_state_kernel<T>), dim3(grid), dim3(kBlock), 0, c.stream, re, im, size, 1, (uint64_t)0,
                     (T)std::pow(2.0, -0.5 * (double)n));
  if (gates()) return 1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HQ_HIP_CHECK(hipEventCreate(&e0));
  HQ_HIP_CHECK(hipEventCreate(&e1));
  HQ_HIP_CHECK(hipStreamSynchronize(c.stream));
  HQ_HIP_CHECK(hipEventRecord(e0, c.stream));
  int rc = gates() || gates();
  HQ_HIP_CHECK(hipEventRecord(e1, c.stream));
  HQ_HIP_CHECK(hipEventSynchronize(e1));
  float t = 0;
  HQ_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms = t / 8.0;
  return rc;
}

// flags: bit 0 (HQ_STATE_PLAIN) hipMalloc memory (what HIP IPC can export); bit 1 (HQ_STATE_NO_SEARCH) one mapped
// placement without probing; bit 2 (HQ_STATE_NO_POOL) never take a pooled placement
static int state_alloc(Context& c, unsigned n, int float_bits, int flags, void** out_re, void** out_im) {
  if (!out_re || !out_im) return fail("hq_alloc_state: null pointer");
  if (float_bits != 32 && float_bits != 64) return fail("hq_alloc_state: float_bits must be 32 or 64");
  if (n > 40) return fail("hq_alloc_state: n_qubits out of range");
  if (c.rec) return fail("hq_alloc_state: cannot allocate while recording a program");
  if (check_device(c)) return 1;
  const size_t itemsize = (size_t)float_bits / 8;
  const size_t stride = state_stride(n, itemsize);
  const size_t bytes = 2 * stride * itemsize;
  StatePool& pool = state_pool();
  StateAlloc st;
  st.n = n;
  st.float_bits = (unsigned)float_bits;
  st.bytes = bytes;
  const char* env_alloc = getenv("HQ_STATE_ALLOC");
  const bool env_plain = env_alloc && std::string(env_alloc) != "vmm";
  const bool tuned = !(flags & 1) && !env_plain && bytes >= tuned_min_bytes() && n >= 8;
  auto finish = [&](StateAlloc& s) {
    *out_re = s.re;
    *out_im = s.im;
    pool.live.push_back(s);
    pool.last_report = s.report;
    pool.last_from_pool = s.from_pool;
    return 0;
  };
  if (!tuned) {
    hipError_t e = hipMalloc(&st.plain, bytes);
    if (e != hipSuccess) {  // pooled placements are the first thing to give back
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
      for (auto& s : pool.idle) state_release(s);
      pool.idle.clear();
      e = hipMalloc(&st.plain, bytes);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(std::string("hq_alloc_state: hipMalloc: ") + hipGetErrorString(e)); }
    st.re = st.plain;
    st.im = reinterpret_cast<unsigned char*>(st.plain) + stride * itemsize;
    st.layout = "hipMalloc";
    st.report = "{\"n_qubits\": " + std::to_string(n) + ", \"chosen\": \"hipMalloc\", \"draws\": []}";
    return finish(st);
  }
  // a pooled winner of the same size: no search, no new virtual range
  if (!(flags & 4))
    for (size_t i = 0; i < pool.idle.size(); ++i)
      if (pool.idle[i].n == n && pool.idle[i].float_bits == (unsigned)float_bits) {
        st = pool.idle[i];
        pool.idle.erase(pool.idle.begin() + (long)i);
        st.from_pool = true;
        return finish(st);
      }
  // placements of other sizes are released first: the pool must never be what makes a new state not fit
  if (!pool.idle.empty()) {
    (void)hipDeviceSynchronize();
    for (auto& s : pool.idle) state_release(s);
    pool.idle.clear();
  }
  size_t free_b = 0, total_b = 0;
  HQ_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  int tries = bytes <= ((size_t)16 << 30) ? 8 : (bytes <= ((size_t)64 << 30) ? 3 : 1);
  if (const char* e = getenv("HQ_STATE_TRIES")) tries = std::max(1, atoi(e));
  if (flags & 2) tries = 1;
  tries = std::max(1, std::min<int>(tries, (int)(0.6 * (double)free_b / (double)bytes)));
  size_t gmin = 0;
  if (vmm_granule_min(c, &gmin)) return 1;
  // a draw that streams at least this fast ends the search early once the evidence is in (the fast family measures
  // 6.3-6.4 TB/s at n = 30; slower winners keep the search going to its limit)
  const char* env_good = getenv("HQ_STATE_GOOD_TBPS");
  const double good_tbps = env_good ? atof(env_good) : 6.25;
  std::vector<StateAlloc> cands;
  std::string draws;
  int rc = 0;
  for (int k = 0; k < tries; ++k) {
    if (k >= 3) {  // enough evidence?  a clear winner among slower draws, or draws that do not differ
      std::vector<double> ms;
      for (const auto& s : cands) ms.push_back(s.probe_ms);
      std::sort(ms.begin(), ms.end());
      const bool fast = 2.0 * bytes / ms[0] / 1e9 >= good_tbps;
      if (fast && (ms[0] < 0.93 * ms[ms.size() / 2] || ms.back() < 1.03 * ms[0])) break;
    }
    // families that were fast at least sometimes (profiles/r02_placement_3_vmm_layouts.txt): 2 MiB granules in creation
    // order, 4 / 8 / 16 MiB granules shuffled; which one wins differs from box to box and from draw to draw
    static const size_t shuffled[4] = {(size_t)8 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)8 << 20};
    size_t gran = (k % 2) ? ((size_t)2 << 20) : shuffled[(k / 2) % 4];
    gran = std::max(gran, gmin);
    gran = (gran + gmin - 1) / gmin * gmin;
    const uint64_t seed = (k % 2) ? 0 : 100 + (uint64_t)k;
    StateAlloc cand = st;
    cand.tuned = true;
    cand.vmm_order_used = vmm_order((bytes + gran - 1) / gran, seed);
    if (vmm_create(c, gran, cand.vmm_order_used, cand.vmm)) {
      if (cands.empty()) rc = 1;  // not even one placement: report the driver's error
      break;
    }
    cand.re = cand.vmm.va;
    cand.im = reinterpret_cast<unsigned char*>(cand.vmm.va) + stride * itemsize;
    cand.layout = std::to_string(gran >> 20) + " MiB granules, " + (seed ? "shuffled (seed " + std::to_string(seed) + ")" : "in creation order");
    if (tries > 1) {
      const int prc = float_bits == 32 ? state_probe<float>(c, (float*)cand.re, (float*)cand.im, n, &cand.probe_ms)
                                       : state_probe<double>(c, (double*)cand.re, (double*)cand.im, n, &cand.probe_ms);
      if (prc) { vmm_destroy(cand.vmm); rc = 1; break; }
    }
    draws += (draws.empty() ? "" : ", ") + std::string("{\"layout\": \"") + cand.layout + "\", \"probe_ms_per_gate\": " +
             (tries > 1 ? std::to_string(cand.probe_ms) : std::string("null")) + "}";
    cands.push_back(cand);  // held: the next draw must see other physical pages
  }
  if (cands.empty() || rc) {
    (void)hipDeviceSynchronize();
    for (auto& s : cands) vmm_destroy(s.vmm);
    return rc ? 1 : fail("hq_alloc_state: no placement could be created");
  }
  size_t best = 0;
  for (size_t i = 1; i < cands.size(); ++i)
    if (cands[i].probe_ms < cands[best].probe_ms) best = i;
  (void)hipDeviceSynchronize();
  for (size_t i = 0; i < cands.size(); ++i)
    if (i != best) vmm_destroy(cands[i].vmm);
  st = cands[best];
  // The rate belongs to the SET of physical granules, not to a random order of them (tools/placement_remap.py: any
  // shuffle of the same granules reproduces it to 0.2 %); only the monotone mapping differs, by a few per cent either
  // way.  So the winner's granules are also probed in creation order -- a remap costs no physical memory -- and the
  // faster of the two mappings is kept.
  if (tries > 1 && st.layout.find("shuffled") != std::string::npos && !(flags & 2)) {
    Vmm alt;
    std::vector<size_t> ident(st.vmm.handles.size());
    for (size_t i = 0; i < ident.size(); ++i) ident[i] = i;
    double ms_alt = 0;
    if (vmm_remap(c, st.vmm, ident, alt) != 0) {  // the winner's mapping may be gone: give everything back and report
      vmm_destroy(st.vmm);
      return 1;
    }
    {
      void* are = alt.va;
      void* aim = reinterpret_cast<unsigned char*>(alt.va) + stride * itemsize;
      const int prc = float_bits == 32 ? state_probe<float>(c, (float*)are, (float*)aim, n, &ms_alt)
                                       : state_probe<double>(c, (double*)are, (double*)aim, n, &ms_alt);
      draws += ", {\"layout\": \"the same granules in creation order\", \"probe_ms_per_gate\": " + std::to_string(ms_alt) + "}";
      if (prc == 0 && ms_alt < st.probe_ms) {
        st.vmm = alt;
        st.re = are;
        st.im = aim;
        st.probe_ms = ms_alt;
        st.layout += ", remapped in creation order";
      } else {  // back to the shuffled mapping (again a fresh range: the old one is retired)
        std::vector<size_t> order = st.vmm_order_used;
        Vmm back;
        if (vmm_remap(c, alt, order, back) != 0) {
          vmm_destroy(alt);
          return 1;
        }
        st.vmm = back;
        st.re = back.va;
        st.im = reinterpret_cast<unsigned char*>(back.va) + stride * itemsize;
      }
    }
  }
  const double tbps = st.probe_ms > 0 ? 2.0 * bytes / st.probe_ms / 1e9 : 0;
  st.report = "{\"n_qubits\": " + std::to_string(n) + ", \"draws\": [" + draws + "], \"chosen\": \"" + st.layout +
              "\", \"probe_ms_per_gate\": " + (st.probe_ms > 0 ? std::to_string(st.probe_ms) : std::string("null")) +
              ", \"probe_TBps\": " + (tbps > 0 ? std::to_string(tbps) : std::string("null")) + "}";
  return finish(st);
}

static int state_free(Context& c, void* re) {
  StatePool& pool = state_pool();
  for (size_t i = 0; i < pool.live.size(); ++i)
    if (pool.live[i].re == re) {
      StateAlloc st = pool.live[i];
      pool.live.erase(pool.live.begin() + (long)i);
      static const bool no_pool = env_int("HQ_STATE_POOL", 1) == 0;
      bool keep = st.tuned && !no_pool;
      for (const auto& s : pool.idle) keep = keep && !(s.n == st.n && s.float_bits == st.float_bits);  // one per size
      if (keep) {
        pool.idle.push_back(st);  // still mapped: the next state of this size takes it without a search
      } else {
        (void)hipDeviceSynchronize();
        state_release(st);
      }
      return 0;
    }
  return fail("hq_free_state: not a state of this library");
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

// ---- state memory --------------------------------------------------------------------------------------------------
// flags of hq_alloc: bit 0 = physically contiguous VRAM (hipDeviceMallocContiguous).
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

int hq_alloc_mapped(void** dev_ptr, uint64_t granule, uint64_t n_granules, const uint32_t* va_slot, uint64_t* granule_min) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  if (hq::check_device(c)) return 1;
  size_t gmin = 0;
  if (hq::vmm_granule_min(c, &gmin)) return 1;
  if (granule_min) *granule_min = gmin;
  if (!dev_ptr || !va_slot || !n_granules) return hq::fail("hq_alloc_mapped: bad arguments");
  if (granule % gmin) return hq::fail("hq_alloc_mapped: granule is not a multiple of the driver minimum " + std::to_string(gmin));
  std::vector<size_t> order(n_granules);
  for (size_t i = 0; i < n_granules; ++i) {
    if (va_slot[i] >= n_granules) return hq::fail("hq_alloc_mapped: slot out of range");
    order[i] = va_slot[i];
  }
  hq::Vmm v;
  if (hq::vmm_create(c, granule, order, v)) return 1;
  hq::vmm_registry().push_back(v);
  *dev_ptr = v.va;
  return 0;
}

int hq_alloc_scattered(void** dev_ptr, uint64_t bytes, uint64_t granule, uint64_t seed) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  if (!dev_ptr || !bytes) return hq::fail("hq_alloc_scattered: bad arguments");
  if (hq::check_device(c)) return 1;
  size_t gmin = 0;
  if (hq::vmm_granule_min(c, &gmin)) return 1;
  if (granule < gmin) granule = gmin;
  granule = (granule + gmin - 1) / gmin * gmin;
  hq::Vmm v;
  if (hq::vmm_create(c, granule, hq::vmm_order(((size_t)bytes + granule - 1) / granule, seed), v)) return 1;
  hq::vmm_registry().push_back(v);
  *dev_ptr = v.va;
  return 0;
}

int hq_free(void* dev_ptr) {
  if (!dev_ptr) return 0;
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  auto& reg = hq::vmm_registry();
  for (size_t i = 0; i < reg.size(); ++i)
    if (reg[i].va == dev_ptr) {
      (void)hipDeviceSynchronize();
      hq::vmm_destroy(reg[i]);
      reg.erase(reg.begin() + (long)i);
      return 0;
    }
  hipError_t e = hipFree(dev_ptr);
  if (e != hipSuccess) return hq::fail(std::string("hq_free: ") + hipGetErrorString(e));
  return 0;
}

int hq_alloc_state(unsigned int n_qubits, int float_bits, int flags, void** psi_re, void** psi_im) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::read_env(c);
  return hq::state_alloc(c, n_qubits, float_bits, flags, psi_re, psi_im);
}

int hq_free_state(void* psi_re) {
  if (!psi_re) return 0;
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  return hq::state_free(c, psi_re);
}

int hq_state_info(const void* psi_re, char* buf, uint64_t cap) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  const std::string* txt = &hq::state_pool().last_report;
  bool from_pool = hq::state_pool().last_from_pool;
  if (psi_re) {
    txt = nullptr;
    for (const auto& st : hq::state_pool().live)
      if (st.re == psi_re) { txt = &st.report; from_pool = st.from_pool; }
    if (!txt) return hq::fail("hq_state_info: not a state of this library");
  }
  if (!buf || cap == 0) return hq::fail("hq_state_info: no buffer");
  const std::string out = hq::emit_report(*txt, from_pool);
  // a report that does not fit is replaced by a valid stub, never cut in the middle of the JSON text
  const std::string stub = "{\"truncated\": true}";
  const std::string& use = out.size() + 1 <= (size_t)cap ? out : stub;
  if (use.size() + 1 > (size_t)cap) return hq::fail("hq_state_info: buffer too small");
  memcpy(buf, use.data(), use.size());
  buf[use.size()] = 0;
  return 0;
}

int hq_state_pool_trim(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  (void)hipDeviceSynchronize();
  for (auto& st : hq::state_pool().idle) hq::state_release(st);
  hq::state_pool().idle.clear();
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

// Diagnostics (not part of the reference boundary; tools/placement_remap.py): the SAME physical granules of a buffer from
// hq_alloc_mapped / hq_alloc_scattered mapped in another order into a fresh virtual range (granule i -> slot va_slot[i]).
// The old range is unmapped and retired; *new_ptr replaces dev_ptr (free it with hq_free).
extern "C" int hq_vmm_remap(void* dev_ptr, const uint32_t* va_slot, void** new_ptr) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  if (!dev_ptr || !va_slot || !new_ptr) return hq::fail("hq_vmm_remap: null pointer");
  for (auto& v : hq::vmm_registry())
    if (v.va == dev_ptr) {
      const size_t ng = v.handles.size();
      std::vector<size_t> order(ng);
      for (size_t i = 0; i < ng; ++i) {
        if (va_slot[i] >= ng) return hq::fail("hq_vmm_remap: slot out of range");
        order[i] = va_slot[i];
      }
      hq::Vmm nv;
      if (hq::vmm_remap(c, v, order, nv)) return 1;
      v = nv;
      *new_ptr = nv.va;
      return 0;
    }
  return hq::fail("hq_vmm_remap: not a mapped buffer of this library");
}
