// hq_swap.hip -- swap_* (the reference boundary of /root/reference/include/python_swap.cpp:31-99), hq_permute_bits_*
// and to_complex (python_U.cpp:114-123, 145-153).
#include "hq_common.h"
#include "hq_kernels_aux.h"
#include "hq_kernels_swap.h"
#include "hq_bitperm.h"

namespace hq {

// ---------------------------------------------------------------------------------
// swap
// ---------------------------------------------------------------------------------
// One in-place pass of tile_permute_kernel: the index bits `tile` (ascending, starting with the vector bits)
// are permuted inside LDS tiles, bit tile[i] of the destination index reading bit src_of[tile[i]] of the source.
template <typename E>
static int launch_tile_permute(Context& c, E* a, unsigned n, const std::vector<unsigned>& tile, const unsigned* src_of) {
  constexpr int VEC = 16 / (int)sizeof(E);
  TilePermArg ta;
  memset(&ta, 0, sizeof(ta));
  ta.tb = (unsigned)tile.size();
  bool identity = true;
  for (unsigned i = 0; i < ta.tb; ++i) {
    ta.apos[i] = tile[i];
    const unsigned sp = src_of[tile[i]];
    const unsigned li = (unsigned)(std::find(tile.begin(), tile.end(), sp) - tile.begin());
    if (li >= ta.tb) return fail("tile_permute: source bit outside the tile");
    ta.lp[i] = li;
    identity = identity && li == i;
  }
  if (identity) return 0;
  const uint64_t ntiles = 1ull << (n - ta.tb);
  for (unsigned i = 0; i < ta.tb; ++i)
    if (ta.apos[i] >= 32) return fail("tile_permute: tile bits must lie below bit 32");
  const size_t lds = ((((size_t)2 << ta.tb) + 15) & ~(size_t)15) + (((((size_t)4 << ta.tb) / VEC) + 15) & ~(size_t)15) +
                     ((size_t)sizeof(E) << ta.tb);
  static bool attr_done = false;
  if (!attr_done) {
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)tile_permute_kernel<uint32_t, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)tile_permute_kernel<uint64_t, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)tile_permute_kernel<uint32_t, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_done = true;
  }
  const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 256 * 8);
  // register prefetch of the next tile for the 32 KiB tiles of 4-byte elements (as in swap_lds_kernel; HQ_SWAP_PREF=0: off)
  static const int use_pref = env_int("HQ_SWAP_PREF", 1);
  if (use_pref && sizeof(E) == 4 && ((1u << ta.tb) / VEC) == 8u * kBlock) {
    if constexpr (sizeof(E) == 4) HQ_LAUNCH(c, (tile_permute_kernel<E, VEC, 8>), dim3(grid), dim3(kBlock), lds, a, ta, ntiles);
  } else {
    HQ_LAUNCH(c, (tile_permute_kernel<E, VEC, 0>), dim3(grid), dim3(kBlock), lds, a, ta, ntiles);
  }
  HQ_HIP_CHECK(hipGetLastError());
  return 0;
}

// A permutation pi of the low s index bits (destination bit i reads source bit pi[i]) that does not fit one
// LDS tile of t bits, as TWO in-place tile passes: the first permutes the bit set T1 = S \ O1, the second
// T2 = S \ O2 (first(x) = old[alpha(x)], new(x) = first[beta(x)], alpha o beta = pi).  O1 = the highest
// s - t bits, so that the first pass is the plain low-bit kernel; O2 = the highest bits outside
// O1, pi^-1(O1) and the vector bits.  Feasible up to s = 18 (4-byte) / 17 (8-byte elements).
template <typename E>
static bool plan_two_pass_swap(const unsigned* pi, unsigned s, unsigned t, std::vector<unsigned>& T1, std::vector<unsigned>& alpha,
                               std::vector<unsigned>& T2, std::vector<unsigned>& beta) {
  constexpr unsigned VB = sizeof(E) == 4 ? 2 : 1;
  if (s <= t || s > 31) return false;
  const unsigned no = s - t;
  std::vector<int> inv(s);
  for (unsigned i = 0; i < s; ++i) inv[pi[i]] = (int)i;
  std::vector<char> inO1(s, 0), inO2(s, 0), banned(s, 0);
  for (unsigned a = s - no; a < s; ++a) { inO1[a] = 1; banned[a] = 1; banned[inv[a]] = 1; }
  for (unsigned b = 0; b < VB; ++b) banned[b] = 1;
  unsigned got = 0;
  for (int cbit = (int)s - 1; cbit >= 0 && got < no; --cbit)
    if (!banned[cbit]) { inO2[cbit] = 1; ++got; }
  if (got < no) return false;
  // beta: fixes O2, sends pi^-1(a) to a for a in O1, identity wherever that is still free, the rest in order
  beta.assign(s, ~0u);
  std::vector<char> used(s, 0);
  for (unsigned cbit = 0; cbit < s; ++cbit)
    if (inO2[cbit]) { beta[cbit] = cbit; used[cbit] = 1; }
  for (unsigned a = 0; a < s; ++a)
    if (inO1[a]) { beta[inv[a]] = a; used[a] = 1; }
  for (unsigned i = 0; i < s; ++i)
    if (beta[i] == ~0u && !used[i]) { beta[i] = i; used[i] = 1; }
  unsigned nxt = 0;
  for (unsigned i = 0; i < s; ++i)
    if (beta[i] == ~0u) {
      while (used[nxt]) ++nxt;
      beta[i] = nxt;
      used[nxt] = 1;
    }
  std::vector<unsigned> binv(s);
  for (unsigned i = 0; i < s; ++i) binv[beta[i]] = i;
  alpha.assign(s, 0);
  for (unsigned x = 0; x < s; ++x) alpha[x] = pi[binv[x]];  // alpha = pi o beta^-1
  T1.clear();
  T2.clear();
  for (unsigned b = 0; b < s; ++b) {
    if (!inO1[b]) T1.push_back(b);
    if (!inO2[b]) T2.push_back(b);
    if (inO1[b] && alpha[b] != b) return false;
    if (inO2[b] && beta[b] != b) return false;
  }
  return T1.size() == t && T2.size() == t;
}

template <typename E>
static int swap_device(Context& c, E* a, const unsigned* pos, unsigned n, unsigned s) {
  SwapArg sa;
  memset(&sa, 0, sizeof(sa));
  sa.s = s;
  bool identity = true;
  for (unsigned i = 0; i < s; ++i) {
    sa.pos[i] = pos[i];
    identity &= pos[i] == i;
  }
  if (identity) return 0;
  const unsigned table_bits = sizeof(E) == 4 ? 13 : 12;  // 32 KiB of elements + index table
  const unsigned max_lds_bits = sizeof(E) == 4 ? 15 : 14;  // 128 KiB of elements, index computed inline
  static const bool two_pass = env_int("HQ_SWAP_TWO_PASS", 1) != 0;
  static const bool one_pass = env_int("HQ_PERM_TILE", 1) != 0;
  // s >= 8: ONE in-place pass through 128 KiB LDS tiles of bitperm_tile_kernel (the tile holds every moved bit; up to 15
  // moved bits for 4-byte, 14 for 8-byte elements).  Round 2 took two passes for s > 13 (2x the algorithmic traffic,
  // 2.6 TB/s); for 8 <= s <= 13 the tile kernel also beats the table-driven kernel below (n = 30: 5.5 vs 5.0-5.1 TB/s).
  static const unsigned tile_min = (unsigned)env_int("HQ_SWAP_TILE_MIN", 8);
  if ((s > table_bits || s >= tile_min) && one_pass && reinterpret_cast<uintptr_t>(a) % 16 == 0) {
    std::vector<unsigned> full(n);
    for (unsigned i = 0; i < n; ++i) full[i] = i < s ? pos[i] : i;
    BitPermPlan P;
    if (plan_bitperm<E>(full.data(), n, true, P)) {
      P.a.dst[0][0] = a;
      return launch_bitperm<E>(c, c.stream, true, (const E*)a, (const E*)nullptr, P);
    }
  }
  if (s > table_bits && two_pass && reinterpret_cast<uintptr_t>(a) % 16 == 0) {
    // 14 <= s <= 18 (17 for 8-byte elements): two in-place passes through 32 KiB LDS tiles, each at the rate of
    // the small-s kernel, instead of one 128 KiB-tile pass with the index computed inline (2.1 TB/s) or the
    // out-of-place gather + copy (1.7 TB/s)
    std::vector<unsigned> T1, T2, alpha, beta;
    if (plan_two_pass_swap<E>(pos, s, table_bits, T1, alpha, T2, beta)) {
      // T1 = the low `table_bits` bits: the first pass is a plain low-bit swap of its own
      if (swap_device<E>(c, a, alpha.data(), n, table_bits)) return 1;
      return launch_tile_permute<E>(c, a, n, T2, beta.data());
    }
  }
  if (s <= table_bits || (s <= max_lds_bits && reinterpret_cast<uintptr_t>(a) % 16 == 0)) {
    const bool table = s <= table_bits;
    const unsigned tile_bits = std::min<unsigned>(n, std::max<unsigned>(s, 11));
    const uint64_t ntiles = 1ull << (n - tile_bits);
    const size_t lds = (table ? ((((size_t)1 << s) * 2 + 15) & ~(size_t)15) : 0) + ((size_t)1 << tile_bits) * sizeof(E);
    const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 256 * 8);
    constexpr int VEC = 16 / sizeof(E);
    const bool vec = tile_bits >= 10 && reinterpret_cast<uintptr_t>(a) % 16 == 0;
    static bool attr_done = false;
    if (!attr_done) {
      HQ_HIP_CHECK(hipFuncSetAttribute((const void*)swap_lds_kernel<uint32_t, 4, false, 0>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HQ_HIP_CHECK(hipFuncSetAttribute((const void*)swap_lds_kernel<uint64_t, 2, false, 0>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done = true;
    }
    // register prefetch of the next tile: pays for the 32 KiB tiles of 4-byte elements only (s = 13 and the first
    // pass of the two-pass path: 4.39 -> 5.09 TB/s); smaller tiles already overlap through their many resident
    // workgroups and lose 5-9 % with it (tools/swap_rate.py).  HQ_SWAP_PREF=0 switches it off
    static const int use_pref = env_int("HQ_SWAP_PREF", 1);
    const unsigned npv = vec ? (1u << tile_bits) / (kBlock * VEC) : 0;
    if (!table)
      HQ_LAUNCH(c, (swap_lds_kernel<E, VEC, false, 0>), dim3(grid), dim3(kBlock), lds, a, sa, tile_bits, ntiles);
    else if (vec && use_pref && npv == 8 && sizeof(E) == 4)
      HQ_LAUNCH(c, (swap_lds_kernel<E, VEC, true, 8>), dim3(grid), dim3(kBlock), lds, a, sa, tile_bits, ntiles);
    else if (vec)
      HQ_LAUNCH(c, (swap_lds_kernel<E, VEC, true, 0>), dim3(grid), dim3(kBlock), lds, a, sa, tile_bits, ntiles);
    else
      HQ_LAUNCH(c, (swap_lds_kernel<E, 1, true, 0>), dim3(grid), dim3(kBlock), lds, a, sa, tile_bits, ntiles);
    HQ_HIP_CHECK(hipGetLastError());
    return 0;
  }
  HQ_NOT_RECORDABLE(c, "the out-of-place swap path");
  const uint64_t size = 1ull << n;
  void* tmp = nullptr;
  if (get_scratch(c, 2, size * sizeof(E), &tmp)) return 1;
  const unsigned grid = (unsigned)std::min<uint64_t>((size + kBlock - 1) / kBlock, 256 * 32);
  hipLaunchKernelGGL((swap_gather_kernel<E>), dim3(grid), dim3(kBlock), 0, c.stream,
                     (const E*)a, (E*)tmp, sa, size);
  HQ_HIP_CHECK(hipGetLastError());
  HQ_HIP_CHECK(hipMemcpyAsync(a, tmp, size * sizeof(E), hipMemcpyDeviceToDevice, c.stream));
  return 0;
}

template <typename E>
static int swap_entry(E* a, const unsigned* pos, unsigned n, unsigned s) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  if (s == 0) return 0;  // python_swap.cpp:35-36
  if (!a || !pos) return fail("swap: null pointer");
  if (s > n || s > 30 || n > 62) return fail("swap: invalid sizes");
  uint64_t seen = 0;
  for (unsigned i = 0; i < s; ++i) {
    if (pos[i] >= s || (seen >> pos[i]) & 1) return fail("swap: pos is not a permutation of 0..s-1");
    seen |= 1ull << pos[i];
  }
  if (is_device_pointer(a)) return swap_device<E>(c, a, pos, n, s);
  HQ_NOT_RECORDABLE(c, "a host-pointer call");
  const size_t bytes = ((size_t)1 << n) * sizeof(E);
  void* s0 = nullptr;
  if (get_scratch(c, 0, bytes, &s0)) return 1;
  HQ_HIP_CHECK(hipMemcpyAsync(s0, a, bytes, hipMemcpyHostToDevice, c.stream));
  if (swap_device<E>(c, (E*)s0, pos, n, s)) return 1;
  HQ_HIP_CHECK(hipMemcpyAsync(a, s0, bytes, hipMemcpyDeviceToHost, c.stream));
  HQ_HIP_CHECK(hipStreamSynchronize(c.stream));
  return 0;
}

// ---------------------------------------------------------------------------------
// permute_bits (device pointers only)
// ---------------------------------------------------------------------------------
template <typename E>
static int permute_bits_entry(const E* src, E* dst, const unsigned* perm, unsigned n) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  if (!src || !dst || !perm) return fail("permute_bits: null pointer");
  if (n > 62) return fail("permute_bits: n too large");
  if (src == dst) return fail("permute_bits: must be out of place");
  if (!is_device_pointer(src) || !is_device_pointer(dst)) return fail("permute_bits: device pointers only");
  uint64_t seen = 0;
  PermArg pa;
  memset(&pa, 0, sizeof(pa));
  for (unsigned i = 0; i < n; ++i) {
    if (perm[i] >= n || (seen >> perm[i]) & 1) return fail("permute_bits: perm is not a permutation of 0..n-1");
    seen |= 1ull << perm[i];
    if (perm[i] == i) pa.fixed_mask |= 1ull << i;
  }
  for (unsigned i = 0; i < n;) {  // moved bits -> fields (runs with consecutive sources)
    if (perm[i] == i) { ++i; continue; }
    unsigned len = 1;
    while (i + len < n && perm[i + len] == perm[i] + len && perm[i + len] != i + len) ++len;
    pa.from[pa.nfields] = (unsigned char)i;
    pa.to[pa.nfields] = (unsigned char)perm[i];
    pa.len[pa.nfields] = (unsigned char)len;
    ++pa.nfields;
    i += len;
  }
  static const bool one_pass = env_int("HQ_PERM_TILE", 1) != 0;
  // HQ_PERM_TILE=3: the gather kernel whenever the low 128 bytes of the index space stay in place (it then moves whole
  // cache lines too; measured 5.0-5.5 TB/s against 5.3-5.6 through the tiles, box to box: not the default)
  static const bool gather_low_fixed = env_int("HQ_PERM_TILE", 0) == 3;
  if (one_pass && !(gather_low_fixed && bitperm_low_run_fixed<E>(perm, n)) && reinterpret_cast<uintptr_t>(src) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(dst) % 16 == 0) {
    // one pass at full cache-line granularity on both sides, whatever bits move (bitperm_tile_kernel); the gather
    // kernel below stays for states smaller than a tile and unaligned pointers
    BitPermPlan P;
    if (plan_bitperm<E>(perm, n, false, P)) {
      P.a.dst[0][0] = dst;
      return launch_bitperm<E>(c, c.stream, true, src, (const E*)nullptr, P);
    }
  }
  const uint64_t size = 1ull << n;
  const bool vec16 = (pa.fixed_mask & 3) == 3 && n >= 2 && sizeof(E) == 4 &&
                     reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0;
  const bool vec16d = (pa.fixed_mask & 1) == 1 && n >= 1 && sizeof(E) == 8 &&
                      reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0;
  const uint64_t units = vec16 ? size / 4 : (vec16d ? size / 2 : size);
  const unsigned grid = (unsigned)std::min<uint64_t>((units + kBlock - 1) / kBlock, 256 * 64);
  if (vec16)
    HQ_LAUNCH(c, (permute_bits_kernel<E, 4>), dim3(grid), dim3(kBlock), 0, src, dst, pa, units);
  else if (vec16d)
    HQ_LAUNCH(c, (permute_bits_kernel<E, 2>), dim3(grid), dim3(kBlock), 0, src, dst, pa, units);
  else
    HQ_LAUNCH(c, (permute_bits_kernel<E, 1>), dim3(grid), dim3(kBlock), 0, src, dst, pa, units);
  HQ_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// to_complex
// ---------------------------------------------------------------------------------
template <typename T>
static int interleave_device(Context& c, const T* re, const T* im, T* out, uint64_t size) {
  if (size == 0) return 0;
  const bool vec = size % 4 == 0 && reinterpret_cast<uintptr_t>(re) % 32 == 0 &&
                   reinterpret_cast<uintptr_t>(im) % 32 == 0 &&
                   reinterpret_cast<uintptr_t>(out) % 32 == 0;
  if (vec) {
    const uint64_t nq = size / 4;
    // many short-lived workgroups: 5.42 TB/s against 5.27 with 256 x 32 (round 3 sweep; script in the history)
    const unsigned grid = (unsigned)std::min<uint64_t>((nq + kBlock - 1) / kBlock, (uint64_t)256 * 256);
    HQ_LAUNCH(c, (interleave4_kernel<T>), dim3(grid), dim3(kBlock), 0, re, im, out, nq);
  } else {
    const unsigned grid = (unsigned)std::min<uint64_t>((size + kBlock - 1) / kBlock, 256 * 32);
    HQ_LAUNCH(c, (interleave_kernel<T>), dim3(grid), dim3(kBlock), 0, re, im, out, size);
  }
  HQ_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename T>
static int to_complex_entry(T* re, T* im, T* out, uint64_t size) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  if (size == 0) return 0;
  if (!re || !im || !out) return fail("to_complex: null pointer");
  const bool d_in = is_device_pointer(re);
  if (d_in != is_device_pointer(im)) return fail("to_complex: mixed host/device planes");
  const bool d_out = is_device_pointer(out);
  const size_t bytes = size * sizeof(T);
  if (d_in && d_out) return interleave_device<T>(c, re, im, out, size);
  HQ_NOT_RECORDABLE(c, "a host-pointer call");
  // stage whatever lives on the host
  const T *sre = re, *sim = im;
  T* sout = out;
  if (!d_in) {
    void* s0 = nullptr;
    if (get_scratch(c, 0, 2 * bytes, &s0)) return 1;
    HQ_HIP_CHECK(hipMemcpyAsync(s0, re, bytes, hipMemcpyHostToDevice, c.stream));
    HQ_HIP_CHECK(hipMemcpyAsync((unsigned char*)s0 + bytes, im, bytes, hipMemcpyHostToDevice, c.stream));
    sre = (const T*)s0;
    sim = (const T*)((unsigned char*)s0 + bytes);
  }
  if (!d_out) {
    void* s1 = nullptr;
    if (get_scratch(c, 1, 2 * bytes, &s1)) return 1;
    sout = (T*)s1;
  }
  if (interleave_device<T>(c, sre, sim, sout, size)) return 1;
  if (!d_out) {
    HQ_HIP_CHECK(hipMemcpyAsync(out, sout, 2 * bytes, hipMemcpyDeviceToHost, c.stream));
    HQ_HIP_CHECK(hipStreamSynchronize(c.stream));
  }
  return 0;
}

}  // namespace hq

extern "C" {

int to_complex64(float* psi_re, float* psi_im, float* psi_out, unsigned int size) {
  return hq::to_complex_entry<float>(psi_re, psi_im, psi_out, size);
}

int to_complex128(double* psi_re, double* psi_im, double* psi_out, unsigned int size) {
  return hq::to_complex_entry<double>(psi_re, psi_im, psi_out, size);
}

int hq_to_complex64(float* psi_re, float* psi_im, float* psi_out, uint64_t size) {
  return hq::to_complex_entry<float>(psi_re, psi_im, psi_out, size);
}

int hq_to_complex128(double* psi_re, double* psi_im, double* psi_out, uint64_t size) {
  return hq::to_complex_entry<double>(psi_re, psi_im, psi_out, size);
}

int swap_float32(float* a, const unsigned int* pos, unsigned int n, unsigned int s) {
  return hq::swap_entry<uint32_t>(reinterpret_cast<uint32_t*>(a), pos, n, s);
}

int swap_float64(double* a, const unsigned int* pos, unsigned int n, unsigned int s) {
  return hq::swap_entry<uint64_t>(reinterpret_cast<uint64_t*>(a), pos, n, s);
}

int swap_int32(int* a, const unsigned int* pos, unsigned int n, unsigned int s) {
  return hq::swap_entry<uint32_t>(reinterpret_cast<uint32_t*>(a), pos, n, s);
}

int swap_int64(long* a, const unsigned int* pos, unsigned int n, unsigned int s) {
  return hq::swap_entry<uint64_t>(reinterpret_cast<uint64_t*>(a), pos, n, s);
}

int swap_uint32(unsigned int* a, const unsigned int* pos, unsigned int n, unsigned int s) {
  return hq::swap_entry<uint32_t>(reinterpret_cast<uint32_t*>(a), pos, n, s);
}

int swap_uint64(unsigned long* a, const unsigned int* pos, unsigned int n, unsigned int s) {
  return hq::swap_entry<uint64_t>(reinterpret_cast<uint64_t*>(a), pos, n, s);
}

int hq_permute_bits_32(const void* src, void* dst, const unsigned int* perm, unsigned int n) {
  return hq::permute_bits_entry<uint32_t>((const uint32_t*)src, (uint32_t*)dst, perm, n);
}

int hq_permute_bits_64(const void* src, void* dst, const unsigned int* perm, unsigned int n) {
  return hq::permute_bits_entry<uint64_t>((const uint64_t*)src, (uint64_t*)dst, perm, n);
}

}  // extern "C"
