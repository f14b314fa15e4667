// hq_hip.hip -- host side of libhq_hip.so: argument checking, kernel dispatch and the
// C ABI declared in include/hq_hip.h (the reference boundary of
// /root/reference/include/python_U.cpp:127-154 and python_swap.cpp:68-99).
#include "hq_kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen()ed (hq_shard_init_rccl)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/hq_hip.h"

namespace hq {

// ---------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------
enum class Mode { Auto, Direct, Mfma, Generic, Naive, Tile, Gemm };

// A recorded sequence of launches ("compiled circuit"): every matrix / operand table it needs
// lives in its own device buffer, so replaying it is pure kernel launches -- from a plain loop
// or, after the first run, from one hipGraph launch.
struct Program {
  std::vector<std::function<void(hipStream_t)>> ops;
  std::vector<unsigned char> host;  // staged tables, uploaded once by hq_program_end
  unsigned char* dev = nullptr;
  size_t cap = 0;
  bool finalized = false;
  bool use_graph = true;
  hipGraphExec_t exec = nullptr;
  hipStream_t graph_stream = nullptr;  // capture needs a non-default stream
};

struct Context {
  std::mutex mu;
  Program* rec = nullptr;  // non-null while hq_program_begin .. hq_program_end records
  hipStream_t stream = nullptr;
  unsigned log2_pack = 1;
  Mode mode = Mode::Auto;
  int nontemporal = -1;  // -1 auto, 0 never, 1 always
  int dummy_policy = -1;  // mfma identity dummies: 0 = free vector components first, 1 = lowest free bits >= 2, 2 = free bits >= 6, -1 = auto (= 2)
  std::string last_error = "";
  const char* last_kernel = "none";
  std::string last_desc = "none";  // full instantiation name of the last apply_U kernel
  // arena for matrices that do not fit kernel arguments (generic / naive kernels)
  unsigned char* arena_host = nullptr;  // pinned
  unsigned char* arena_dev = nullptr;
  size_t arena_size = 0, arena_used = 0;
  // scratch (host staging of planes, naive / swap_gather temporaries, norm2)
  void* scratch[3] = {nullptr, nullptr, nullptr};
  size_t scratch_size[3] = {0, 0, 0};
  bool attr_set = false;
  bool env_read = false;
  int device = -1;  // device that owns the arena / scratch buffers (one device per process)
};

static Context& ctx() {
  static Context c;
  return c;
}

static int fail(const std::string& msg) {
  ctx().last_error = msg;
  return 1;
}

#define HQ_HIP_CHECK(expr)                                                             \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return hq::fail(std::string(#expr) + ": " + hipGetErrorString(_e));              \
  } while (0)

// Launch now, or append to the program being recorded (arguments are captured by value).
#define HQ_LAUNCH(c_, kern, grid, block, lds, ...)                                              \
  do {                                                                                          \
    if ((c_).rec) {                                                                             \
      (c_).rec->ops.emplace_back(                                                               \
          [=](hipStream_t s_) { hipLaunchKernelGGL(kern, grid, block, lds, s_, __VA_ARGS__); });  \
    } else {                                                                                    \
      hipLaunchKernelGGL(kern, grid, block, lds, (c_).stream, __VA_ARGS__);                     \
    }                                                                                           \
  } while (0)

#define HQ_NOT_RECORDABLE(c_, what)                                                             \
  do {                                                                                          \
    if ((c_).rec) return fail(std::string(what) + " cannot be recorded into a program");        \
  } while (0)

static void read_env(Context& c) {
  if (c.env_read) return;
  c.env_read = true;
  if (const char* e = getenv("HQ_LOG2_PACK_SIZE")) {
    int v = atoi(e);
    if (v >= 1 && v <= 5) c.log2_pack = (unsigned)v;
  }
  if (const char* e = getenv("HQ_APPLY_MODE")) {
    std::string s(e);
    if (s == "direct") c.mode = Mode::Direct;
    else if (s == "mfma") c.mode = Mode::Mfma;
    else if (s == "generic") c.mode = Mode::Generic;
    else if (s == "naive") c.mode = Mode::Naive;
    else if (s == "tile") c.mode = Mode::Tile;
    else if (s == "gemm") c.mode = Mode::Gemm;
  }
  if (const char* e = getenv("HQ_NONTEMPORAL")) c.nontemporal = atoi(e) < 0 ? -1 : (atoi(e) != 0);
}

// The library keeps its upload arena and scratch buffers on ONE device: the design is one
// process per GPU (hybridq_amd.dist).  Using a second device from the same process is refused
// loudly instead of silently reading another device's memory.
static int check_device(Context& c) {
  int dev = -1;
  HQ_HIP_CHECK(hipGetDevice(&dev));
  if (c.device < 0) c.device = dev;
  if (dev != c.device)
    return fail("libhq_hip is bound to device " + std::to_string(c.device) + " but the current device is " +
                std::to_string(dev) + ": use one process per GPU");
  return 0;
}

static int get_scratch(Context& c, int slot, size_t bytes, void** out) {
  if (check_device(c)) return 1;
  if (c.scratch_size[slot] < bytes) {
    if (c.scratch[slot]) {
      HQ_HIP_CHECK(hipStreamSynchronize(c.stream));
      HQ_HIP_CHECK(hipFree(c.scratch[slot]));
      c.scratch[slot] = nullptr;
      c.scratch_size[slot] = 0;
    }
    HQ_HIP_CHECK(hipMalloc(&c.scratch[slot], bytes));
    c.scratch_size[slot] = bytes;
  }
  *out = c.scratch[slot];
  return 0;
}

// Copy `bytes` of host data into the arena; returns the device address in *dev.  The
// host copy is taken immediately (the caller's buffer may be a temporary), the H2D
// transfer is asynchronous on the stream.
static int arena_upload(Context& c, const void* host, size_t bytes, void** dev) {
  const size_t kArena = 64u << 20;
  if (check_device(c)) return 1;
  if (!c.arena_host) {
    HQ_HIP_CHECK(hipHostMalloc((void**)&c.arena_host, kArena, hipHostMallocDefault));
    HQ_HIP_CHECK(hipMalloc((void**)&c.arena_dev, kArena));
    c.arena_size = kArena;
  }
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (c.rec) {  // recording: the table goes into the program's own buffer
    const size_t off = c.rec->host.size();
    if (off + need > c.rec->cap) return fail("program table buffer exhausted (HQ_PROGRAM_MB)");
    c.rec->host.resize(off + need);
    memcpy(c.rec->host.data() + off, host, bytes);
    *dev = c.rec->dev + off;
    return 0;
  }
  if (need > c.arena_size) return fail("matrix too large for the upload arena");
  if (c.arena_used + need > c.arena_size) {
    HQ_HIP_CHECK(hipStreamSynchronize(c.stream));  // wrap: wait for in-flight users
    c.arena_used = 0;
  }
  memcpy(c.arena_host + c.arena_used, host, bytes);
  if (bytes >= (64u << 10)) {
    // Large tables (k >= 6 operand tables, up to 8 MiB): copied by a kernel that reads the pinned
    // host arena directly, on the SAME stream.  hipMemcpyAsync would go through an SDMA queue and
    // the cross-queue dependency costs sporadic ~75 ms host-side stalls on this platform.
    const size_t n16 = (bytes + 15) / 16;
    const unsigned grid = (unsigned)std::min<size_t>((n16 + kBlock - 1) / kBlock, 512);
    hipLaunchKernelGGL(upload_kernel, dim3(grid), dim3(kBlock), 0, c.stream,
                       reinterpret_cast<uint4*>(c.arena_dev + c.arena_used),
                       reinterpret_cast<const uint4*>(c.arena_host + c.arena_used), n16);
    HQ_HIP_CHECK(hipGetLastError());
  } else {
    HQ_HIP_CHECK(hipMemcpyAsync(c.arena_dev + c.arena_used, c.arena_host + c.arena_used, bytes,
                                hipMemcpyHostToDevice, c.stream));
  }
  *dev = c.arena_dev + c.arena_used;
  c.arena_used += need;
  return 0;
}

// true if `p` can be dereferenced by a kernel
static bool is_device_pointer(const void* p) {
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // unregistered host memory
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged ||
         attr.type == hipMemoryTypeUnified;
}

static int check_positions(const unsigned* pos, unsigned n, unsigned k) {
  if (k > n || n > 62) return 1;
  uint64_t seen = 0;
  for (unsigned i = 0; i < k; ++i) {
    if (pos[i] >= n) return 1;
    if (seen & (1ull << pos[i])) return 1;
    seen |= 1ull << pos[i];
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// apply_U dispatch (device pointers)
// ---------------------------------------------------------------------------------
template <typename T, int K, int VMASK>
static int launch_direct_kv(Context& c, T* re, T* im, const T* U, const unsigned* pos, unsigned n) {
  constexpr int VB = Vec<T>::VB;
  constexpr int KV = popc_c(VMASK), KR = K - KV, R = 1 << KR, D = 1 << K;
  constexpr int ILP = R >= 4 ? 1 : (R == 2 ? 2 : 4);
  // sort positions ascending, remember which original matrix bit each one is
  unsigned order[K];
  for (int j = 0; j < K; ++j) order[j] = j;
  std::sort(order, order + K, [&](unsigned a, unsigned b) { return pos[a] < pos[b]; });
  GateArg<T, K> g;
  for (int t = 0; t < D; ++t) {
    int to = 0;
    for (int j = 0; j < K; ++j) to |= ((t >> j) & 1) << order[j];
    for (int s = 0; s < D; ++s) {
      int so = 0;
      for (int j = 0; j < K; ++j) so |= ((s >> j) & 1) << order[j];
      g.re[t * D + s] = U[2 * (to * D + so)];
      g.im[t * D + s] = U[2 * (to * D + so) + 1];
    }
  }
  RegPos rp = {{0, 0, 0, 0}};
  for (int j = 0; j < KR; ++j) rp.p[j] = pos[order[KV + j]] - VB;
  const uint64_t nslots = 1ull << (n - VB - KR);
  const uint64_t nblocks = nslots / (ILP * kBlock);
  if (nblocks == 0 || nblocks > 0x7fffffffull) return fail("direct: grid out of range");
  // Non-temporal loads/stores: +7..10 % when every wave-level access is a contiguous run of
  // >= 512 B (all register targets at positions >= 7: measured 5.9 vs 5.5 TB/s at n=30),
  // but they bypass the cache-line merging that low targets rely on (pos 2: 2.7 vs 5.3
  // TB/s), so they are used only for high targets unless forced.
  bool nt = c.nontemporal > 0;
  if (c.nontemporal < 0) {
    nt = true;
    for (int j = 0; j < KR; ++j) nt = nt && (rp.p[j] + VB >= 7);
  }
  if (nt)
    HQ_LAUNCH(c, (apply_direct_kernel<T, K, VMASK, ILP, true>), dim3((unsigned)nblocks), dim3(kBlock), 0, re, im, g, rp);
  else
    HQ_LAUNCH(c, (apply_direct_kernel<T, K, VMASK, ILP, false>), dim3((unsigned)nblocks), dim3(kBlock), 0, re, im, g, rp);
  HQ_HIP_CHECK(hipGetLastError());
  c.last_kernel = "direct";
  c.last_desc = std::string("apply_direct_kernel<") + (sizeof(T) == 4 ? "float" : "double") + ", " +
                std::to_string(K) + ", " + std::to_string(VMASK) + ", " + std::to_string(ILP) + ", " +
                (nt ? "true" : "false") + ">";
  return 0;
}

template <typename T, int K>
static int launch_direct_k(Context& c, T* re, T* im, const T* U, const unsigned* pos, unsigned n,
                           int vmask) {
  constexpr int VB = Vec<T>::VB;
  switch (vmask) {
    case 0: return launch_direct_kv<T, K, 0>(c, re, im, U, pos, n);
    case 1: return launch_direct_kv<T, K, 1>(c, re, im, U, pos, n);
    case 2:
      if constexpr (VB >= 2) return launch_direct_kv<T, K, 2>(c, re, im, U, pos, n);
      break;
    case 3:
      if constexpr (VB >= 2 && K >= 2) return launch_direct_kv<T, K, 3>(c, re, im, U, pos, n);
      break;
  }
  return fail("direct: bad vmask");
}

// can the direct kernel run this call?
template <typename T>
static bool direct_ok(unsigned n, unsigned k, const unsigned* pos) {
  constexpr int VB = Vec<T>::VB;
  if (k < 1 || k > 3) return false;
  unsigned kv = 0;
  for (unsigned j = 0; j < k; ++j) kv += pos[j] < (unsigned)VB;
  const unsigned kr = k - kv;
  const unsigned R = 1u << kr;
  const unsigned ilp = R >= 4 ? 1 : (R == 2 ? 2 : 4);
  if (n < VB + kr) return false;
  const uint64_t nslots = 1ull << (n - VB - kr);
  return nslots >= (uint64_t)ilp * kBlock;
}

template <typename T>
static int launch_direct(Context& c, T* re, T* im, const T* U, const unsigned* pos, unsigned n,
                         unsigned k) {
  constexpr int VB = Vec<T>::VB;
  int vmask = 0;
  for (unsigned j = 0; j < k; ++j)
    if (pos[j] < (unsigned)VB) vmask |= 1 << pos[j];
  switch (k) {
    case 1: return launch_direct_k<T, 1>(c, re, im, U, pos, n, vmask);
    case 2: return launch_direct_k<T, 2>(c, re, im, U, pos, n, vmask);
    case 3: return launch_direct_k<T, 3>(c, re, im, U, pos, n, vmask);
  }
  return fail("direct: k out of range");
}

// U (interleaved, original bit order) -> planar re[D*D], im[D*D] with the matrix index
// bits re-ordered to ASCENDING target position; `sorted` receives the positions.
template <typename T>
static void sort_gate(const T* U, const unsigned* pos, unsigned k, std::vector<T>& out,
                      unsigned* sorted) {
  const unsigned D = 1u << k;
  unsigned order[kMaxK];
  for (unsigned j = 0; j < k; ++j) order[j] = j;
  std::sort(order, order + k, [&](unsigned a, unsigned b) { return pos[a] < pos[b]; });
  for (unsigned j = 0; j < k; ++j) sorted[j] = pos[order[j]];
  out.resize((size_t)2 * D * D);
  for (unsigned t = 0; t < D; ++t) {
    unsigned to = 0;
    for (unsigned j = 0; j < k; ++j) to |= ((t >> j) & 1u) << order[j];
    for (unsigned s = 0; s < D; ++s) {
      unsigned so = 0;
      for (unsigned j = 0; j < k; ++j) so |= ((s >> j) & 1u) << order[j];
      out[(size_t)t * D + s] = U[2 * ((size_t)to * D + so)];
      out[(size_t)D * D + (size_t)t * D + s] = U[2 * ((size_t)to * D + so) + 1];
    }
  }
}

// ---------------------------------------------------------------------------------
// matrix-core path (f32, k <= 4): role assignment + A-operand table (see hq_kernels.h)
// ---------------------------------------------------------------------------------
template <typename T>
struct MfmaPlan {
  int kbits = 0, vmask = 0, ilp = 1;
  bool nt = false;
  unsigned n_addr = 0;
  MfmaRoles ro;
  std::vector<T> A;
};

template <typename T>
static bool plan_mfma(const Context& c, const T* U, const unsigned* pos, unsigned n, unsigned k,
                      MfmaPlan<T>& P, bool for_tile = false) {
  constexpr unsigned CB = Vec<T>::VB;  // vector-component index bits: 2 (f32) / 1 (f64)
  if (k < 1 || k > (for_tile ? 4u : 6u)) return false;
  std::vector<T> Us;
  unsigned sp[kMaxK];
  sort_gate<T>(U, pos, k, Us, sp);
  const unsigned D = 1u << k;
  const T* Ur = Us.data();
  const T* Ui = Us.data() + (size_t)D * D;
  const unsigned k_eff = k <= 3 ? 3 : k;
  P.kbits = (int)k_eff + 1;
  // effective digits: real targets (tbit = sorted target index) + identity dummies (tbit = -1)
  struct Digit { unsigned pos; int tbit; };
  std::vector<Digit> E;
  uint64_t used = 0;
  for (unsigned j = 0; j < k; ++j) { E.push_back({sp[j], (int)j}); used |= 1ull << sp[j]; }
  // Where the identity dummies of a k < 3 gate go (measured at n = 30, tools/sweep_dummy.py):
  // in free index bits >= 6.  They become register digits whose 16-byte accesses are >= 256 B
  // apart (non-temporal policy applies) while the real low targets keep the q-digit role (a
  // permutation of one contiguous run).  Never slower than the two earlier placements
  // ("comp": free vector components, "low": lowest free bits >= 2) and 8 % faster for single
  // targets on bits 2-5.  Small states fall through to "low", then to the components.
  int dummy_low = c.dummy_policy;
  if (dummy_low < 0) dummy_low = for_tile ? ((sp[0] < CB || (k == 1 && sp[0] >= 5)) ? 1 : 0) : 2;  // LDS tiles: earlier rule (sweep_blocked)
  if (dummy_low == 2)
    for (unsigned p = 6; p < n && E.size() < k_eff; ++p)
      if (!((used >> p) & 1)) { E.push_back({p, -1}); used |= 1ull << p; }
  for (unsigned p = dummy_low >= 1 ? CB : 0; p < n && E.size() < k_eff; ++p)
    if (!((used >> p) & 1)) { E.push_back({p, -1}); used |= 1ull << p; }
  for (unsigned p = 0; p < CB && E.size() < k_eff; ++p)  // tiny n: fall back to the components
    if (!((used >> p) & 1)) { E.push_back({p, -1}); used |= 1ull << p; }
  if (E.size() < k_eff) return false;
  std::sort(E.begin(), E.end(), [](const Digit& a, const Digit& b) { return a.pos < b.pos; });
  int vmask = 0;
  std::vector<int> comp_digit, addr_digit;  // indices into E
  for (unsigned e = 0; e < k_eff; ++e) {
    if (E[e].pos < CB) { vmask |= 1 << E[e].pos; comp_digit.push_back((int)e); }
    else addr_digit.push_back((int)e);
  }
  P.vmask = vmask;
  const int KV = (int)comp_digit.size(), NS = P.kbits - 2, NR = NS - KV, NL = 1 << NR;
  const unsigned na = (unsigned)addr_digit.size();
  if (na < 1 || na > 6) return false;
  P.n_addr = na;
  P.ilp = std::max(1, 8 / NL);
  if (NR < 0 || n < CB + na) return false;
  const uint64_t nslots = 1ull << (n - CB - na);
  // tile mode and the k >= 5 kernel (grid-stride over wave iterations): one wave iteration
  if (nslots < ((for_tile || k_eff >= 5) ? 16u : (uint64_t)P.ilp * 64)) return false;
  // roles: -1 = plane, otherwise index into E.  q gets the low address digits first
  // (positions 2..5: a permutation of a contiguous run), then the plane, then the rest.
  constexpr int PLANE = -1;
  std::vector<int> order;
  for (int e : addr_digit) if (E[e].pos - CB <= 3) order.push_back(e);
  order.push_back(PLANE);
  for (int e : addr_digit) if (E[e].pos - CB > 3) order.push_back(e);
  const int qd[2] = {order[0], order[1]};
  std::vector<int> rd(order.begin() + 2, order.end());  // NR reg digits
  if ((int)rd.size() != NR) return false;
  MfmaRoles& ro = P.ro;
  for (int m = 0; m < 6; ++m) ro.pos[m] = 63;
  for (unsigned m = 0; m < na; ++m) ro.pos[m] = E[addr_digit[m]].pos - CB;
  ro.q_plane = -1;
  ro.r_plane = -1;
  for (int b = 0; b < 2; ++b) {
    ro.q_off[b] = qd[b] == PLANE ? 0u : (1u << (E[qd[b]].pos - CB));
    if (qd[b] == PLANE) ro.q_plane = b;
  }
  for (int b = 0; b < 5; ++b) ro.r_off[b] = 0;
  bool nt = true;
  for (int b = 0; b < NR; ++b) {
    if (rd[b] == PLANE) { ro.r_plane = b; continue; }
    if (E[rd[b]].pos - CB > 31) return false;  // offsets are 32-bit vec indices
    ro.r_off[b] = 1u << (E[rd[b]].pos - CB);
    nt = nt && E[rd[b]].pos >= 6;
  }
  for (int b = 0; b < 2; ++b)
    if (qd[b] != PLANE && E[qd[b]].pos - CB > 31) return false;
  P.nt = c.nontemporal < 0 ? nt : c.nontemporal > 0;
  // decode a K index (q, step) into (plane, effective amplitude index over E)
  auto decode = [&](unsigned q, unsigned st, unsigned& plane, unsigned& teff) {
    plane = 0;
    teff = 0;
    for (int b = 0; b < 2; ++b) {
      const unsigned bit = (q >> b) & 1u;
      if (qd[b] == PLANE) plane = bit; else teff |= bit << qd[b];
    }
    for (int cix = 0; cix < KV; ++cix) teff |= ((st >> cix) & 1u) << comp_digit[cix];
    for (int b = 0; b < NR; ++b) {
      const unsigned bit = (st >> (KV + b)) & 1u;
      if (rd[b] == PLANE) plane = bit; else teff |= bit << rd[b];
    }
  };
  auto split = [&](unsigned teff, unsigned& treal, unsigned& tdummy) {
    treal = 0;
    tdummy = 0;
    for (unsigned e = 0; e < k_eff; ++e) {
      const unsigned bit = (teff >> e) & 1u;
      if (E[e].tbit >= 0) treal |= bit << E[e].tbit; else tdummy |= bit << e;
    }
  };
  const int NSTEP = 1 << NS, NRB = 1 << (NS - 2);
  P.A.assign((size_t)NRB * NSTEP * 64, (T)0);
  for (int rb = 0; rb < NRB; ++rb)
    for (int st = 0; st < NSTEP; ++st)
      for (unsigned lane = 0; lane < 64; ++lane) {
        const unsigned i = lane & 15, q_in = lane >> 4;
        unsigned po, to, pi, ti, tor, tod, tir, tid;
        // D row of lane (q', j) register r: 4q'+r for the f32 MFMA, q'+4r for the f64 one
        const unsigned q_out = sizeof(T) == 4 ? (i >> 2) : (i & 3), r_out = sizeof(T) == 4 ? (i & 3) : (i >> 2);
        decode(q_out, r_out | ((unsigned)rb << 2), po, to);
        decode(q_in, (unsigned)st, pi, ti);
        split(to, tor, tod);
        split(ti, tir, tid);
        T val = 0;
        if (tod == tid) {
          const T ur = Ur[tor * D + tir], ui = Ui[tor * D + tir];
          val = po == pi ? ur : (po == 0 ? -ui : ui);
        }
        if (k_eff >= 5) {  // apply_mfma_big_kernel: G consecutive steps per 16-byte LDS read
          constexpr int G = 16 / (int)sizeof(T);
          P.A[((((size_t)rb * (NSTEP / G) + st / G) * 64 + lane) * G) + st % G] = val;
        } else {
          P.A[((size_t)rb * NSTEP + st) * 64 + lane] = val;
        }
      }
  return true;
}

template <typename T, int KBITS, int VMASK>
static void launch_mfma_kv(Context& c, T* re, T* im, const T* dA, const MfmaPlan<T>& P, unsigned nblocks) {
  constexpr int NS = KBITS - 2, KV = popc_c(VMASK), NL = 1 << (NS - KV);
  constexpr int ILP = NL >= 8 ? 1 : 8 / NL;
  const MfmaRoles ro = P.ro;  // captured by value when the launch is recorded
  if (P.nt)
    HQ_LAUNCH(c, (apply_mfma_kernel<T, KBITS, VMASK, ILP, true>), dim3(nblocks), dim3(kBlock), 0, re, im, dA, ro);
  else
    HQ_LAUNCH(c, (apply_mfma_kernel<T, KBITS, VMASK, ILP, false>), dim3(nblocks), dim3(kBlock), 0, re, im, dA, ro);
}

// k = 5, 6 role kernel, 512-thread workgroups; PHASED = the two halves of a workgroup alternate between their
// MFMA phase and their memory phase (see the kernel).  HQ_BIG_PHASED=0/1 forces one variant (experiments).
template <typename T, int KBITS, int VMASK, bool PHASED>
static int launch_mfma_big_var(Context& c, T* re, T* im, const T* dA, const MfmaPlan<T>& P, unsigned n,
                               const BigOffsets& tab) {
  constexpr unsigned CB = Vec<T>::VB;
  constexpr int BLOCK = 512;
  constexpr int NS = KBITS - 2, NRB = 1 << (NS - 2), NSTEP = 1 << NS;
  constexpr size_t lds = (size_t)NRB * NSTEP * 64 * sizeof(T);
  static bool attr_done = false;  // under the context mutex
  if (!attr_done) {
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)apply_mfma_big_kernel<T, KBITS, VMASK, true, BLOCK, PHASED>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)apply_mfma_big_kernel<T, KBITS, VMASK, false, BLOCK, PHASED>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  const uint64_t niter = (1ull << (n - CB - P.n_addr)) >> 4;  // 16 slots per wave iteration
  const uint64_t wgs = (niter + BLOCK / 64 - 1) / (BLOCK / 64);
  static const int grid_cap = getenv("HQ_BIG_GRID") ? atoi(getenv("HQ_BIG_GRID")) : 2048;
  const unsigned grid = (unsigned)std::min<uint64_t>(wgs, (uint64_t)grid_cap);
  const MfmaRoles ro = P.ro;
  if (P.nt)
    HQ_LAUNCH(c, (apply_mfma_big_kernel<T, KBITS, VMASK, true, BLOCK, PHASED>), dim3(grid), dim3(BLOCK), lds, re, im, dA, ro, tab, niter);
  else
    HQ_LAUNCH(c, (apply_mfma_big_kernel<T, KBITS, VMASK, false, BLOCK, PHASED>), dim3(grid), dim3(BLOCK), lds, re, im, dA, ro, tab, niter);
  return 0;
}

// measured at n = 30 / 29 (gpurun_out/r2e, r2f: means over 6 position patterns, phased vs free-running):
//   k = 5 f32 3.42 vs 3.61 ms, f64 3.40 vs 3.51;  k = 6 f32 4.65 vs 4.72;  k = 6 f64 5.35 vs 4.81 (the f64
//   k = 6 instantiation needs 218 of 256 registers and spills ~100 B/lane inside the MFMA phase: with the
//   partner wave parked at the barrier nothing covers the reloads)
static bool big_phased(int kbits, bool is_double) {
  static const int forced = getenv("HQ_BIG_PHASED") ? atoi(getenv("HQ_BIG_PHASED")) : -1;
  return forced >= 0 ? forced != 0 : !(kbits >= 7 && is_double);
}

template <typename T, int KBITS, int VMASK>
static int launch_mfma_big_kv(Context& c, T* re, T* im, const T* dA, const MfmaPlan<T>& P, unsigned n) {
  // byte offset of every register-digit load from the lane's base address (BigOffsets)
  constexpr int NS = KBITS - 2, KV = popc_c(VMASK), NR = NS - KV, NL = 1 << NR;
  BigOffsets tab;
  memset(&tab, 0, sizeof(tab));
  const int64_t plane_step = reinterpret_cast<const unsigned char*>(im) - reinterpret_cast<const unsigned char*>(re);
  for (int ld = 0; ld < NL; ++ld) {
    uint64_t o = 0;
    bool pl = false;
    for (int b = 0; b < NR; ++b)
      if ((ld >> b) & 1) { o |= P.ro.r_off[b]; pl = pl || P.ro.r_plane == b; }
    tab.off[ld] = (int64_t)(16 * o) + (pl ? plane_step : 0);
  }
  if (big_phased(KBITS, sizeof(T) == 8)) return launch_mfma_big_var<T, KBITS, VMASK, true>(c, re, im, dA, P, n, tab);
  return launch_mfma_big_var<T, KBITS, VMASK, false>(c, re, im, dA, P, n, tab);
}

template <typename T>
static int launch_mfma_big(Context& c, T* re, T* im, const T* A, const MfmaPlan<T>& P, unsigned n) {
  constexpr unsigned CB = Vec<T>::VB;
  int rc = -1;
  switch (P.kbits * 4 + P.vmask) {
    case 24: rc = launch_mfma_big_kv<T, 6, 0>(c, re, im, A, P, n); break;
    case 25: rc = launch_mfma_big_kv<T, 6, 1>(c, re, im, A, P, n); break;
    case 28: rc = launch_mfma_big_kv<T, 7, 0>(c, re, im, A, P, n); break;
    case 29: rc = launch_mfma_big_kv<T, 7, 1>(c, re, im, A, P, n); break;
    default:
      if constexpr (CB == 2) {
        switch (P.kbits * 4 + P.vmask) {
          case 26: rc = launch_mfma_big_kv<T, 6, 2>(c, re, im, A, P, n); break;
          case 27: rc = launch_mfma_big_kv<T, 6, 3>(c, re, im, A, P, n); break;
          case 30: rc = launch_mfma_big_kv<T, 7, 2>(c, re, im, A, P, n); break;
          case 31: rc = launch_mfma_big_kv<T, 7, 3>(c, re, im, A, P, n); break;
          default: break;
        }
      }
  }
  if (rc < 0) return fail("mfma: bad plan");
  if (rc) return rc;
  HQ_HIP_CHECK(hipGetLastError());
  c.last_kernel = "mfma";
  const bool phased = big_phased(P.kbits, sizeof(T) == 8);
  c.last_desc = std::string("apply_mfma_big_kernel<") + (sizeof(T) == 4 ? "float" : "double") + ", " +
                std::to_string(P.kbits) + ", " + std::to_string(P.vmask) + ", " + (P.nt ? "true" : "false") + ", " +
                (phased ? "512, true>" : "512, false>");
  return 0;
}

template <typename T>
static int launch_mfma(Context& c, T* re, T* im, const MfmaPlan<T>& P, unsigned n) {
  constexpr unsigned CB = Vec<T>::VB;
  void* dA = nullptr;
  if (arena_upload(c, P.A.data(), P.A.size() * sizeof(T), &dA)) return 1;
  if (P.kbits >= 6) return launch_mfma_big<T>(c, re, im, (const T*)dA, P, n);
  const uint64_t nslots = 1ull << (n - CB - P.n_addr);
  const uint64_t nblocks64 = nslots / ((uint64_t)P.ilp * 64);
  if (nblocks64 == 0 || nblocks64 > 0x7fffffffull) return fail("mfma: grid out of range");
  const unsigned nb = (unsigned)nblocks64;
  const T* A = (const T*)dA;
  bool ok = true;
  switch (P.kbits * 4 + P.vmask) {
    case 16: launch_mfma_kv<T, 4, 0>(c, re, im, A, P, nb); break;
    case 17: launch_mfma_kv<T, 4, 1>(c, re, im, A, P, nb); break;
    case 20: launch_mfma_kv<T, 5, 0>(c, re, im, A, P, nb); break;
    case 21: launch_mfma_kv<T, 5, 1>(c, re, im, A, P, nb); break;
    default:
      if constexpr (CB == 2) {
        switch (P.kbits * 4 + P.vmask) {
          case 18: launch_mfma_kv<T, 4, 2>(c, re, im, A, P, nb); break;
          case 19: launch_mfma_kv<T, 4, 3>(c, re, im, A, P, nb); break;
          case 22: launch_mfma_kv<T, 5, 2>(c, re, im, A, P, nb); break;
          case 23: launch_mfma_kv<T, 5, 3>(c, re, im, A, P, nb); break;
          default: ok = false;
        }
      } else {
        ok = false;
      }
  }
  if (!ok) return fail("mfma: bad plan");
  HQ_HIP_CHECK(hipGetLastError());
  c.last_kernel = "mfma";
  c.last_desc = std::string("apply_mfma_kernel<") + (sizeof(T) == 4 ? "float" : "double") + ", " +
                std::to_string(P.kbits) + ", " + std::to_string(P.vmask) + ", " + std::to_string(P.ilp) + ", " +
                (P.nt ? "true" : "false") + ">";
  return 0;
}

template <typename T>
static int launch_generic(Context& c, T* re, T* im, const T* U, const unsigned* pos, unsigned n,
                          unsigned k) {
  GenArg a;
  memset(&a, 0, sizeof(a));
  a.k = k;
  a.c = std::min<unsigned>(n - k, kTileBits - k);
  uint64_t tmask = 0;
  for (unsigned j = 0; j < k; ++j) {
    a.tpos[j] = pos[j];
    tmask |= 1ull << pos[j];
  }
  unsigned nc = 0;
  for (unsigned p = 0; p < n && nc < a.c; ++p)
    if (!((tmask >> p) & 1)) a.cpos[nc++] = p;
  std::vector<unsigned> all(a.tpos, a.tpos + k);
  all.insert(all.end(), a.cpos, a.cpos + a.c);
  std::sort(all.begin(), all.end());
  for (unsigned j = 0; j < k + a.c; ++j) a.apos[j] = all[j];
  a.vec_ok = (a.cpos[0] == 0 && a.cpos[1] == 1) ? 1u : 0u;
  const size_t D = (size_t)1 << k, C = (size_t)1 << a.c;
  void* dU = nullptr;
  if (arena_upload(c, U, 2 * D * D * sizeof(T), &dU)) return 1;
  const size_t lds = D * 8 + C * 4 + 2 * D * C * sizeof(T);
  if (!c.attr_set) {
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)apply_generic_kernel<float>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)apply_generic_kernel<double>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    c.attr_set = true;
  }
  const uint64_t nblocks = 1ull << (n - k - a.c);
  const unsigned grid = (unsigned)std::min<uint64_t>(nblocks, 256 * 16);
  HQ_LAUNCH(c, (apply_generic_kernel<T>), dim3(grid), dim3(kBlock), lds, re, im, (const T*)dU, a, nblocks);
  HQ_HIP_CHECK(hipGetLastError());
  c.last_kernel = "generic";
  c.last_desc = std::string("apply_generic_kernel<") + (sizeof(T) == 4 ? "float" : "double") + "> k=" + std::to_string(k);
  return 0;
}

// k = 7..10: tile GEMM on the matrix cores (apply_gemm_kernel)
// Tile bits of the GEMM kernel: 128 KiB of LDS (one workgroup per CU) for the largest k, where
// the U traffic per tile matters; 32-64 KiB for k = 7, 8 (two to four workgroups per CU, whose
// copy and MFMA phases overlap each other).  HQ_GEMM_TB overrides (experiments).
template <typename T>
static unsigned gemm_tile_bits(unsigned k) {
  static const int forced = getenv("HQ_GEMM_TB") ? atoi(getenv("HQ_GEMM_TB")) : 0;
  const unsigned big = sizeof(T) == 4 ? 14 : 13;
  if (forced) return std::min<unsigned>(big, std::max<unsigned>((unsigned)forced, k + 4));
  // measured at n = 30 / 29 (gpurun_out/sweep_gemm_tb.txt): 32 columns (f32) / 16 columns (f64)
  return std::min<unsigned>(big, k + (sizeof(T) == 4 ? 5 : 4));
}

template <typename T>
static bool gemm_ok(unsigned n, unsigned k, bool forced = false) {
  const unsigned big = sizeof(T) == 4 ? 14 : 13;
  const unsigned tb = gemm_tile_bits<T>(k);
  if (k < (forced ? 6u : 7u) || k + 4 > big || n < tb) return false;  // k = 6 only on request ("gemm" mode)
  return ((1u << k) >> 4) * ((1u << (tb - k)) >> 4) >= 8;              // one output block per wave at least
}

template <typename T, int RBW, int CBW>
static int launch_gemm_rc(Context& c, T* re, T* im, const T* dA, const unsigned* dOff, const GemmArg& a,
                          uint64_t ntiles) {
  const size_t lds = (size_t)2 * sizeof(T) << a.tb;
  static bool attr_done = false;  // under the context mutex
  if (!attr_done) {
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)apply_gemm_kernel<T, RBW, CBW, 0>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_done = true;
  }
  const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 2048);
  // register prefetch of the next tile where the tile has the usual size of this wave shape and the registers allow
  // (not the 8 x 1 shape of k = 10: 224 + 64 registers); HQ_GEMM_PREF=0 switches it off
  constexpr int CBv = sizeof(T) == 4 ? 2 : 1;
  constexpr int NPVx = sizeof(T) == 4 ? (CBW == 2 && RBW == 1 ? 2 : 0)  // f32: k = 7 only (2 x 2, k = 8: no gain, -3 % for some positions; 4 x 2 would spill)
                                     : (CBW == 1 ? (RBW == 1 ? 2 : (RBW == 2 ? 4 : (RBW == 4 ? 8 : 0))) : 0);
  static const int use_pref = getenv("HQ_GEMM_PREF") ? atoi(getenv("HQ_GEMM_PREF")) : 1;
  if constexpr (NPVx > 0) {
    if (use_pref && ((1u << (a.tb - CBv)) == (unsigned)NPVx * kGemmBlock)) {
      static bool attr2 = false;
      if (!attr2) {
        HQ_HIP_CHECK(hipFuncSetAttribute((const void*)apply_gemm_kernel<T, RBW, CBW, NPVx>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr2 = true;
      }
      HQ_LAUNCH(c, (apply_gemm_kernel<T, RBW, CBW, NPVx>), dim3(grid), dim3(kGemmBlock), lds, re, im, dA, dOff, a, ntiles);
      return 0;
    }
  }
  HQ_LAUNCH(c, (apply_gemm_kernel<T, RBW, CBW, 0>), dim3(grid), dim3(kGemmBlock), lds, re, im, dA, dOff, a, ntiles);
  return 0;
}

template <typename T>
static int launch_gemm(Context& c, T* re, T* im, const T* U, const unsigned* pos, unsigned n, unsigned k) {
  constexpr unsigned G = 16 / sizeof(T);
  const unsigned tb = gemm_tile_bits<T>(k);
  const unsigned D = 1u << k, cbits = tb - k;
  std::vector<T> Us;
  unsigned sp[kMaxK];
  sort_gate<T>(U, pos, k, Us, sp);  // matrix index bit j <-> sp[j], ascending
  const T* Ur = Us.data();
  const T* Ui = Us.data() + (size_t)D * D;
  GemmArg a;
  memset(&a, 0, sizeof(a));
  a.tb = tb;
  a.k = k;
  uint64_t tmask = 0;
  for (unsigned j = 0; j < k; ++j) tmask |= 1ull << sp[j];
  std::vector<unsigned> cpos;
  for (unsigned p = 0; p < n && cpos.size() < cbits; ++p)
    if (!((tmask >> p) & 1)) cpos.push_back(p);
  std::vector<unsigned> all(sp, sp + k);
  all.insert(all.end(), cpos.begin(), cpos.end());
  std::sort(all.begin(), all.end());
  for (unsigned m = 0; m < tb; ++m) a.apos[m] = all[m];
  auto local = [&](unsigned gpos) { return (unsigned)(std::find(all.begin(), all.end(), gpos) - all.begin()); };
  std::vector<unsigned> tl(k), cl(cbits);
  for (unsigned j = 0; j < k; ++j) tl[j] = local(sp[j]);
  for (unsigned j = 0; j < cbits; ++j) cl[j] = local(cpos[j]);
  for (int i = 0; i < 4; ++i) { a.tl[i] = tl[i]; a.cl[i] = cl[i]; }
  // swizzle: a half-wave's B read varies the element bits {tl[0], cl[0..3]}; ds_read_b32/b64
  // bank = element index mod 32.  Fold those of them that are >= 5 into free bits of [2,5)
  // (bits 0,1 stay: 16-byte vectors must remain contiguous for the copy phases).
  {
    const unsigned lb5[5] = {tl[0], cl[0], cl[1], cl[2], cl[3]};
    std::vector<unsigned> high, freeb;
    for (unsigned b : lb5) if (b >= 5) high.push_back(b);
    for (unsigned b = 2; b < 5; ++b)
      if (std::find(lb5, lb5 + 5, b) == lb5 + 5) freeb.push_back(b);
    std::sort(high.begin(), high.end());
    a.n_sw = (unsigned)std::min(high.size(), freeb.size());
    for (unsigned i = 0; i < a.n_sw; ++i) { a.sw_src[i] = high[i]; a.sw_dst[i] = freeb[i]; }
  }
  auto swz = [&](unsigned e) {
    for (unsigned i = 0; i < a.n_sw; ++i) e ^= ((e >> a.sw_src[i]) & 1u) << a.sw_dst[i];
    return e;
  };
  auto dep = [](unsigned v, const std::vector<unsigned>& p) {
    unsigned e = 0;
    for (size_t i = 0; i < p.size(); ++i) e |= ((v >> i) & 1u) << p[i];
    return e;
  };
  const unsigned D4 = D / 4, NRBT = D / 16, NCB = (1u << cbits) / 16;
  std::vector<unsigned> offs(D4 + NRBT + NCB);
  for (unsigned st = 0; st < D4; ++st) offs[st] = swz(dep(st << 2, tl));
  for (unsigned rb = 0; rb < NRBT; ++rb) offs[D4 + rb] = swz(dep(rb << 4, tl));
  for (unsigned cb = 0; cb < NCB; ++cb) offs[D4 + NRBT + cb] = swz(dep(cb << 4, cl));
  // A-operand table: [row block][step group][Ur | Ui][lane][G]: lane (i = lane & 15, q = lane >> 4)
  // holds M[16 rb + i][4 (G sg + s) + q]
  a.nsg = D4 / G;
  std::vector<T> A((size_t)2 * D * D);
  for (unsigned rb = 0; rb < NRBT; ++rb)
    for (unsigned sg = 0; sg < a.nsg; ++sg)
      for (unsigned lane = 0; lane < 64; ++lane) {
        const unsigned row = rb * 16 + (lane & 15);
        for (unsigned s2 = 0; s2 < G; ++s2) {
          const unsigned t = 4 * (sg * G + s2) + (lane >> 4);
          const size_t o = ((((size_t)rb * a.nsg + sg) * 2) * 64 + lane) * G + s2;
          A[o] = Ur[(size_t)row * D + t];
          A[o + (size_t)64 * G] = Ui[(size_t)row * D + t];
        }
      }
  void* dA = nullptr;
  void* dO = nullptr;
  if (arena_upload(c, A.data(), A.size() * sizeof(T), &dA)) return 1;
  if (arena_upload(c, offs.data(), offs.size() * sizeof(unsigned), &dO)) return 1;
  const uint64_t ntiles = 1ull << (n - tb);
  // 64 (f32) / 32 (f64) output blocks per tile over 8 waves: wave = RBW x CBW blocks
  const unsigned per_wave = (NRBT * NCB) / 8;
  const unsigned cbw = std::min(std::min(NCB, 4u), std::max(per_wave, 1u)), rbw = per_wave / cbw;
  int rc = -1;
  const T* Ap = (const T*)dA;
  const unsigned* Op = (const unsigned*)dO;
  switch (rbw * 16 + cbw) {
    case 1 * 16 + 1: rc = launch_gemm_rc<T, 1, 1>(c, re, im, Ap, Op, a, ntiles); break;
    case 1 * 16 + 2: rc = launch_gemm_rc<T, 1, 2>(c, re, im, Ap, Op, a, ntiles); break;
    case 2 * 16 + 1: rc = launch_gemm_rc<T, 2, 1>(c, re, im, Ap, Op, a, ntiles); break;
    case 1 * 16 + 4: rc = launch_gemm_rc<T, 1, 4>(c, re, im, Ap, Op, a, ntiles); break;
    case 2 * 16 + 4: rc = launch_gemm_rc<T, 2, 4>(c, re, im, Ap, Op, a, ntiles); break;
    case 2 * 16 + 2: rc = launch_gemm_rc<T, 2, 2>(c, re, im, Ap, Op, a, ntiles); break;
    case 4 * 16 + 2: rc = launch_gemm_rc<T, 4, 2>(c, re, im, Ap, Op, a, ntiles); break;
    case 4 * 16 + 1: rc = launch_gemm_rc<T, 4, 1>(c, re, im, Ap, Op, a, ntiles); break;
    case 8 * 16 + 1:  // k = 10 with 16 columns: float32 only (gemm_ok stops complex128 at k = 9; the f64 form would spill)
      if constexpr (sizeof(T) == 4) rc = launch_gemm_rc<T, 8, 1>(c, re, im, Ap, Op, a, ntiles);
      break;
    default: break;
  }
  if (rc < 0) return fail("gemm: unsupported shape");
  if (rc) return rc;
  HQ_HIP_CHECK(hipGetLastError());
  c.last_kernel = "gemm";
  c.last_desc = std::string("apply_gemm_kernel<") + (sizeof(T) == 4 ? "float" : "double") + ", " +
                std::to_string(rbw) + ", " + std::to_string(cbw) + "> k=" + std::to_string(k);
  return 0;
}

// k = 5, 6: LDS-staged tile GEMM on the matrix cores (apply_mfma_tile_kernel)
template <typename T>
static bool mfma_tile_ok(unsigned n, unsigned k) {
  const unsigned tile_bits = sizeof(T) == 4 ? 12 : 11;
  return (k == 5 || k == 6) && n >= tile_bits;
}

template <typename T>
static int launch_mfma_tile(Context& c, T* re, T* im, const T* U, const unsigned* pos, unsigned n,
                            unsigned k) {
  const unsigned tile_bits = sizeof(T) == 4 ? 12 : 11;
  GenArg a;
  memset(&a, 0, sizeof(a));
  a.k = k;
  a.c = tile_bits - k;
  uint64_t tmask = 0;
  for (unsigned j = 0; j < k; ++j) {
    a.tpos[j] = pos[j];
    tmask |= 1ull << pos[j];
  }
  unsigned nc = 0;
  for (unsigned p = 0; p < n && nc < a.c; ++p)
    if (!((tmask >> p) & 1)) a.cpos[nc++] = p;
  std::vector<unsigned> all(a.tpos, a.tpos + k);
  all.insert(all.end(), a.cpos, a.cpos + a.c);
  std::sort(all.begin(), all.end());
  for (unsigned j = 0; j < k + a.c; ++j) a.apos[j] = all[j];
  constexpr unsigned VB = Vec<T>::VB;
  a.vec_ok = 1;
  for (unsigned b = 0; b < VB; ++b) a.vec_ok &= (a.cpos[b] == b) ? 1u : 0u;
  // A-operand table: A[row block][step][lane] = M[16 rb + (lane & 15)][4 step + (lane >> 4)],
  // M = [[Ur,-Ui],[Ui,Ur]] with row/column index = plane * 2^k + t (t in the caller's bit order)
  const unsigned D = 1u << k, E = 2 * D, NSTEP = E / 4, NRBT = E / 16;
  std::vector<T> A((size_t)NRBT * NSTEP * 64);
  for (unsigned rb = 0; rb < NRBT; ++rb)
    for (unsigned st = 0; st < NSTEP; ++st)
      for (unsigned lane = 0; lane < 64; ++lane) {
        const unsigned row = 16 * rb + (lane & 15), col = 4 * st + (lane >> 4);
        const unsigned po = row / D, to = row % D, pi = col / D, ti = col % D;
        const T ur = U[2 * ((size_t)to * D + ti)], ui = U[2 * ((size_t)to * D + ti) + 1];
        A[((size_t)rb * NSTEP + st) * 64 + lane] = po == pi ? ur : (po == 0 ? -ui : ui);
      }
  void* dA = nullptr;
  if (arena_upload(c, A.data(), A.size() * sizeof(T), &dA)) return 1;
  const size_t C = (size_t)1 << a.c;
  const size_t lds = D * 8 + C * 4 + 2 * (size_t)D * C * sizeof(T);
  const uint64_t nblocks = 1ull << (n - tile_bits);
  const unsigned grid = (unsigned)std::min<uint64_t>(nblocks, 256 * 4);
  if (k == 5)
    HQ_LAUNCH(c, (apply_mfma_tile_kernel<T, 5>), dim3(grid), dim3(kBlock), lds, re, im, (const T*)dA, a, nblocks);
  else
    HQ_LAUNCH(c, (apply_mfma_tile_kernel<T, 6>), dim3(grid), dim3(kBlock), lds, re, im, (const T*)dA, a, nblocks);
  HQ_HIP_CHECK(hipGetLastError());
  c.last_kernel = "mfma_tile";
  c.last_desc = std::string("apply_mfma_tile_kernel<") + (sizeof(T) == 4 ? "float" : "double") + ", " +
                std::to_string(k) + ">";
  return 0;
}

template <typename T>
static int launch_naive(Context& c, T* re, T* im, const T* U, const unsigned* pos, unsigned n,
                        unsigned k) {
  HQ_NOT_RECORDABLE(c, "the tiny-state fallback kernel");
  NaiveArg a;
  memset(&a, 0, sizeof(a));
  a.k = k;
  for (unsigned j = 0; j < k; ++j) a.tpos[j] = pos[j];
  const uint64_t size = 1ull << n;
  const size_t D = (size_t)1 << k;
  void* dU = nullptr;
  if (arena_upload(c, U, 2 * D * D * sizeof(T), &dU)) return 1;
  void* tmp = nullptr;
  if (get_scratch(c, 2, 2 * size * sizeof(T), &tmp)) return 1;
  T* tre = (T*)tmp;
  T* tim = tre + size;
  HQ_HIP_CHECK(hipMemcpyAsync(tre, re, size * sizeof(T), hipMemcpyDeviceToDevice, c.stream));
  HQ_HIP_CHECK(hipMemcpyAsync(tim, im, size * sizeof(T), hipMemcpyDeviceToDevice, c.stream));
  const uint64_t nblocks = (size + kBlock - 1) / kBlock;
  if (nblocks > 0x7fffffffull) return fail("naive: state too large");
  hipLaunchKernelGGL((apply_naive_kernel<T>), dim3((unsigned)nblocks), dim3(kBlock), 0, c.stream,
                     (const T*)tre, (const T*)tim, re, im, (const T*)dU, a, size);
  HQ_HIP_CHECK(hipGetLastError());
  c.last_kernel = "naive";
  c.last_desc = "apply_naive_kernel";
  return 0;
}

template <typename T>
static int apply_device(Context& c, T* re, T* im, const T* U, const unsigned* pos, unsigned n,
                        unsigned k) {
  const bool can_direct = direct_ok<T>(n, k, pos);
  const bool can_generic = (n - k) >= 2 && k <= kMaxK;
  MfmaPlan<T> plan;
  bool can_mfma = false;
  if ((c.mode == Mode::Auto || c.mode == Mode::Mfma) && k <= 6) can_mfma = plan_mfma<T>(c, U, pos, n, k, plan);
  auto run_mfma = [&]() -> int { return launch_mfma<T>(c, re, im, plan, n); };
  switch (c.mode) {
    case Mode::Direct:
      if (can_direct) return launch_direct<T>(c, re, im, U, pos, n, k);
      break;
    case Mode::Mfma:
      if (can_mfma) return run_mfma();
      break;
    case Mode::Generic:
      if (can_generic) return launch_generic<T>(c, re, im, U, pos, n, k);
      break;
    case Mode::Naive:
      return launch_naive<T>(c, re, im, U, pos, n, k);
    case Mode::Tile:
      if (mfma_tile_ok<T>(n, k)) return launch_mfma_tile<T>(c, re, im, U, pos, n, k);
      break;
    case Mode::Gemm:
      if (gemm_ok<T>(n, k, true)) return launch_gemm<T>(c, re, im, U, pos, n, k);
      break;
    case Mode::Auto:
      break;
  }
  if (can_mfma) return run_mfma();
  if (can_direct) return launch_direct<T>(c, re, im, U, pos, n, k);
  if ((c.mode == Mode::Auto || c.mode == Mode::Mfma) && mfma_tile_ok<T>(n, k))
    return launch_mfma_tile<T>(c, re, im, U, pos, n, k);
  if ((c.mode == Mode::Auto || c.mode == Mode::Mfma) && gemm_ok<T>(n, k))
    return launch_gemm<T>(c, re, im, U, pos, n, k);
  if (can_generic) return launch_generic<T>(c, re, im, U, pos, n, k);
  return launch_naive<T>(c, re, im, U, pos, n, k);
}

template <typename T>
static int apply_U_entry(T* re, T* im, const T* U, const unsigned* pos, unsigned n, unsigned k) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  if (k == 0) return 0;  // python_U.cpp:38-39
  if (!re || !im || !U || !pos) return fail("apply_U: null pointer");
  if (k > kMaxK) return fail("apply_U: n_pos > 10 is not supported");
  if (check_positions(pos, n, k)) return fail("apply_U: invalid positions");
  if ((reinterpret_cast<uintptr_t>(re) % 32) || (reinterpret_cast<uintptr_t>(im) % 32))
    return fail("apply_U: planes must be 32-byte aligned");  // U.h:34-36
  const bool dre = is_device_pointer(re), dim_ = is_device_pointer(im);
  if (dre != dim_) return fail("apply_U: psi_re/psi_im must both be device or both host");
  if (dre) return apply_device<T>(c, re, im, U, pos, n, k);
  // host compatibility path: stage -> kernel -> copy back -> sync
  HQ_NOT_RECORDABLE(c, "a host-pointer call");
  const size_t bytes = ((size_t)1 << n) * sizeof(T);
  void* s0 = nullptr;
  if (get_scratch(c, 0, 2 * bytes, &s0)) return 1;
  T* dr = (T*)s0;
  T* di = (T*)((unsigned char*)s0 + bytes);
  HQ_HIP_CHECK(hipMemcpyAsync(dr, re, bytes, hipMemcpyHostToDevice, c.stream));
  HQ_HIP_CHECK(hipMemcpyAsync(di, im, bytes, hipMemcpyHostToDevice, c.stream));
  if (apply_device<T>(c, dr, di, U, pos, n, k)) return 1;
  HQ_HIP_CHECK(hipMemcpyAsync(re, dr, bytes, hipMemcpyDeviceToHost, c.stream));
  HQ_HIP_CHECK(hipMemcpyAsync(im, di, bytes, hipMemcpyDeviceToHost, c.stream));
  HQ_HIP_CHECK(hipStreamSynchronize(c.stream));
  return 0;
}

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
  static const int use_pref = getenv("HQ_SWAP_PREF") ? atoi(getenv("HQ_SWAP_PREF")) : 1;
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
  static const bool two_pass = !(getenv("HQ_SWAP_TWO_PASS") && atoi(getenv("HQ_SWAP_TWO_PASS")) == 0);
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
    static const int use_pref = getenv("HQ_SWAP_PREF") ? atoi(getenv("HQ_SWAP_PREF")) : 1;
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
    const unsigned grid = (unsigned)std::min<uint64_t>((nq + kBlock - 1) / kBlock, 256 * 32);
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

// ---------------------------------------------------------------------------------
// apply_blocked: a list of gates inside one LDS tile, one HBM pass (device pointers, f32)
// ---------------------------------------------------------------------------------
template <typename T>
static int apply_blocked_entry(T* re, T* im, unsigned n, const unsigned* tile_pos, unsigned tb,
                               unsigned n_gates, const T* U_all, const unsigned* pos_all,
                               const unsigned* k_all) {
  constexpr unsigned CB = Vec<T>::VB;
  const unsigned max_tb = sizeof(T) == 4 ? kBlockedMaxTileBits : kBlockedMaxTileBits - 1;  // 128 KiB of LDS
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  if (!re || !im || !tile_pos || (n_gates && (!U_all || !pos_all || !k_all))) return fail("apply_blocked: null pointer");
  if (n_gates == 0) return 0;
  if (n > 62 || tb > max_tb || tb < 10 || tb > n) return fail("apply_blocked: tile size out of range");
  if (!is_device_pointer(re) || !is_device_pointer(im)) return fail("apply_blocked: device pointers only");
  if ((reinterpret_cast<uintptr_t>(re) % 32) || (reinterpret_cast<uintptr_t>(im) % 32))
    return fail("apply_blocked: planes must be 32-byte aligned");
  BlockedArg ba;
  memset(&ba, 0, sizeof(ba));
  ba.tb = tb;
  int local_of[64];
  for (int i = 0; i < 64; ++i) local_of[i] = -1;
  for (unsigned i = 0; i < tb; ++i) {
    if (tile_pos[i] >= n || (i && tile_pos[i] <= tile_pos[i - 1])) return fail("apply_blocked: tile positions must be ascending and < n");
    ba.apos[i] = tile_pos[i];
    local_of[tile_pos[i]] = (int)i;
  }
  for (unsigned b = 0; b < CB; ++b)
    if (tile_pos[b] != b) return fail("apply_blocked: the tile must contain the vector-component index bits (0,1 for f32; 0 for f64)");
  std::vector<BlockedGate> gates(n_gates);
  std::vector<T> Atab;
  const T* Up = U_all;
  const unsigned* pp = pos_all;
  for (unsigned g = 0; g < n_gates; ++g) {
    const unsigned k = k_all[g];
    if (k < 1 || k > 4) return fail("apply_blocked: gates must have 1..4 targets");
    unsigned lp[4];
    for (unsigned j = 0; j < k; ++j) {
      if (pp[j] >= 64 || local_of[pp[j]] < 0) return fail("apply_blocked: gate target outside the tile");
      lp[j] = (unsigned)local_of[pp[j]];
    }
    if (check_positions(lp, tb, k)) return fail("apply_blocked: duplicate targets");
    // largest k that takes the register butterfly on the LDS tile (blocked_inner_gate_valu) instead
    // of the matrix-core form with identity dummies: measured, k = 1 wins (4x fewer flops), k = 2 does not
    static int valu_kmax = getenv("HQ_BLOCKED_VALU") ? atoi(getenv("HQ_BLOCKED_VALU")) : 1;
    if (k <= 2 && (int)k <= valu_kmax) {
      std::vector<T> Us;
      unsigned sp[kMaxK];
      sort_gate<T>(Up, lp, k, Us, sp);  // planar, matrix index bits in ascending local position
      BlockedGate& G = gates[g];
      memset(&G, 0, sizeof(G));
      unsigned vmask = 0, kr = 0;
      for (int m = 0; m < 6; ++m) G.ro.pos[m] = 31;
      for (unsigned j = 0; j < k; ++j) {
        if (sp[j] < CB) vmask |= 1u << sp[j];
        else G.ro.pos[kr++] = sp[j] - CB;
      }
      if (tb - CB < kr) return fail("apply_blocked: tile too small");
      G.a_off = (unsigned)Atab.size();
      G.kv = 64 + k * 4 + vmask;
      G.n_addr = kr;
      Atab.insert(Atab.end(), Us.begin(), Us.end());
      while (Atab.size() % 4) Atab.push_back((T)0);  // keep the next table 16-byte aligned
      Up += (size_t)2 << (2 * k);
      pp += k;
      continue;
    }
    MfmaPlan<T> P;
    if (!plan_mfma<T>(c, Up, lp, tb, k, P, true)) return fail("apply_blocked: cannot plan an inner gate");
    BlockedGate& G = gates[g];
    memset(&G, 0, sizeof(G));
    G.ro = P.ro;
    for (int m = 0; m < 4; ++m)
      if (G.ro.pos[m] >= 31) G.ro.pos[m] = 31;
    G.a_off = (unsigned)Atab.size();
    G.kv = (unsigned)(P.kbits * 4 + P.vmask);
    G.n_addr = P.n_addr;
    Atab.insert(Atab.end(), P.A.begin(), P.A.end());
    Up += (size_t)2 << (2 * k);
    pp += k;
  }
  const uint64_t ntiles = 1ull << (n - tb);
  const size_t tile_bytes = ((size_t)2 << tb) * sizeof(T);
  const size_t per_cu = std::min<size_t>(std::max<size_t>(1, (160 * 1024) / tile_bytes), 4);
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[] = {(const void*)apply_blocked_kernel<float, 256, false, false>, (const void*)apply_blocked_kernel<float, 512, false, false>,
                         (const void*)apply_blocked_kernel<double, 256, false, false>, (const void*)apply_blocked_kernel<double, 512, false, false>,
                         (const void*)apply_blocked_kernel<float, 512, true, false>, (const void*)apply_blocked_kernel<double, 512, true, false>,
                         (const void*)apply_blocked_kernel<float, 512, true, true>, (const void*)apply_blocked_kernel<float, 512, false, true>,
                         (const void*)apply_blocked_kernel<double, 512, true, true>};
    for (const void* f : fns) HQ_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 256 * per_cu);
  static int block_threads = getenv("HQ_BLOCKED_THREADS") ? atoi(getenv("HQ_BLOCKED_THREADS")) : 512;
  static int a_in_lds = getenv("HQ_BLOCKED_ALDS") ? atoi(getenv("HQ_BLOCKED_ALDS")) : 1;
  // LDS left per workgroup behind the tile when `per_cu` workgroups share a CU
  const size_t a_budget = (160 * 1024) / per_cu - tile_bytes > 2048 ? (160 * 1024) / per_cu - tile_bytes - 1024 : 0;
  // operand tables in LDS only when ALL of them fit behind the tile without costing a resident
  // workgroup (measured: -3 % per pass; splitting a pass to make them fit costs a whole HBM pass)
  const size_t tab_bytes = (((size_t)n_gates * kBlockedTabWords * sizeof(BlockedTabT)) + 15) & ~(size_t)15;  // per-gate address tables (built in-kernel)
  const bool fits = a_in_lds && block_threads != 256 && Atab.size() * sizeof(T) + tab_bytes <= a_budget;
  // register prefetch of the next tile (512 threads, 4 vectors per thread and plane = 13 (f32) / 12 (f64) tile bits;
  // f64 only with the table-driven gates: the computed-address variant has no registers left for it)
  static int use_pref = getenv("HQ_BLOCKED_PREF") ? atoi(getenv("HQ_BLOCKED_PREF")) : 1;
  const bool pref = use_pref && block_threads != 256 && tb == (sizeof(T) == 4 ? 13u : 12u) && (sizeof(T) == 4 || fits);
  if (fits) {
    void *dG = nullptr, *dA = nullptr;
    if (arena_upload(c, gates.data(), gates.size() * sizeof(BlockedGate), &dG)) return 1;
    if (arena_upload(c, Atab.data(), Atab.size() * sizeof(T), &dA)) return 1;
    const size_t lds = tile_bytes + Atab.size() * sizeof(T) + tab_bytes;
    if (pref) {
      HQ_LAUNCH(c, (apply_blocked_kernel<T, 512, true, true>), dim3(grid), dim3(512), lds, re, im, (const BlockedGate*)dG,
                n_gates, (const T*)dA, (unsigned)Atab.size(), ba, ntiles);
    } else {
      HQ_LAUNCH(c, (apply_blocked_kernel<T, 512, true, false>), dim3(grid), dim3(512), lds, re, im, (const BlockedGate*)dG,
                n_gates, (const T*)dA, (unsigned)Atab.size(), ba, ntiles);
    }
  } else {
    void *dG = nullptr, *dA = nullptr;
    if (arena_upload(c, gates.data(), gates.size() * sizeof(BlockedGate), &dG)) return 1;
    if (arena_upload(c, Atab.data(), Atab.size() * sizeof(T), &dA)) return 1;
    const size_t lds = tile_bytes;
    if (block_threads == 256)
      HQ_LAUNCH(c, (apply_blocked_kernel<T, 256, false, false>), dim3(grid), dim3(256), lds, re, im, (const BlockedGate*)dG, n_gates, (const T*)dA, 0u, ba, ntiles);
    else if (pref) {
      if constexpr (sizeof(T) == 4)
        HQ_LAUNCH(c, (apply_blocked_kernel<T, 512, false, true>), dim3(grid), dim3(512), lds, re, im, (const BlockedGate*)dG, n_gates, (const T*)dA, 0u, ba, ntiles);
    } else
      HQ_LAUNCH(c, (apply_blocked_kernel<T, 512, false, false>), dim3(grid), dim3(512), lds, re, im, (const BlockedGate*)dG, n_gates, (const T*)dA, 0u, ba, ntiles);
  }
  HQ_HIP_CHECK(hipGetLastError());
  c.last_kernel = "blocked";
  c.last_desc = std::string("apply_blocked_kernel<") + (sizeof(T) == 4 ? "float" : "double") + ", " +
                std::to_string(block_threads == 256 ? 256 : 512) + "> tb=" + std::to_string(tb) + " gates=" +
                std::to_string(n_gates);
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
// multi-GPU shard exchange (no reference counterpart: simulation.py:379-380 has no MPI for this path)
// ---------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct Shard {
  unsigned world = 1, rank = 0, g = 0;
  int transport = 0;  // 0 = none, 1 = RCCL send/recv, 2 = peer-to-peer stores through HIP IPC mappings
  RcclApi api;
  ncclComm_t comm = nullptr;
  bool own_comm = false;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  // p2p: local buffer address -> the same buffer on every rank (mapped into this process)
  struct Peers { const void* local; void* peer[kMaxShardRanks]; };
  std::vector<Peers> registry;
  double last_ms = 0;
};

static Shard& shard() {
  static Shard s;
  return s;
}

static int load_rccl(Shard& sh) {
  if (sh.api.lib) return 0;
  std::vector<std::string> names;
  if (const char* e = getenv("HQ_RCCL_LIBRARY")) names.push_back(e);
  names.insert(names.end(), {"librccl.so.1", "librccl.so"});
  void* h = nullptr;
  // the copy already mapped by the process first (torch bundles its own librccl next to its own HIP
  // runtime: a second RCCL on another runtime could not see this process's allocations)
  for (const auto& nm : names) if (!h) h = dlopen(nm.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const auto& nm : names) if (!h) h = dlopen(nm.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"));
  RcclApi& a = sh.api;
  a.lib = h;
#define HQ_SYM(field, name)                                                         \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                    \
  if (!a.field) { a.lib = nullptr; return fail(std::string("librccl lacks ") + name); }
  HQ_SYM(GetUniqueId, "ncclGetUniqueId")
  HQ_SYM(CommInitRank, "ncclCommInitRank")
  HQ_SYM(CommDestroy, "ncclCommDestroy")
  HQ_SYM(GroupStart, "ncclGroupStart")
  HQ_SYM(GroupEnd, "ncclGroupEnd")
  HQ_SYM(Send, "ncclSend")
  HQ_SYM(Recv, "ncclRecv")
  HQ_SYM(GetErrorString, "ncclGetErrorString")
#undef HQ_SYM
  return 0;
}

#define HQ_NCCL_CHECK(sh_, expr)                                                         \
  do {                                                                                   \
    ncclResult_t _r = (expr);                                                            \
    if (_r != ncclSuccess) return hq::fail(std::string(#expr) + ": " + (sh_).api.GetErrorString(_r)); \
  } while (0)

static int shard_common_init(Context& c, Shard& sh, unsigned world, unsigned rank) {
  if (world == 0 || (world & (world - 1)) || world > (unsigned)kMaxShardRanks) return fail("shard: the number of ranks must be a power of two <= 16");
  if (rank >= world) return fail("shard: rank out of range");
  if (check_device(c)) return 1;
  sh.world = world;
  sh.rank = rank;
  sh.g = 0;
  while ((1u << sh.g) < world) ++sh.g;
  if (!sh.comm_stream) HQ_HIP_CHECK(hipStreamCreateWithFlags(&sh.comm_stream, hipStreamNonBlocking));
  for (auto& e : sh.ev)
    if (!e) HQ_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return 0;
}

static int copy16(Context& c, hipStream_t s, void* dst, const void* src, size_t bytes) {
  const size_t n16 = bytes / 16;
  const unsigned grid = (unsigned)std::min<size_t>((n16 + kBlock - 1) / kBlock, 256 * 16);
  hipLaunchKernelGGL(upload_kernel, dim3(grid), dim3(kBlock), 0, s, reinterpret_cast<uint4*>(dst),
                     reinterpret_cast<const uint4*>(src), n16);
  HQ_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename E>
static int launch_pack(Context& c, hipStream_t s, const E* s0, const E* s1, const ExchArg& a) {
  const uint64_t size = 1ull << a.m;
  const bool al = reinterpret_cast<uintptr_t>(s0) % 16 == 0 && (!s1 || reinterpret_cast<uintptr_t>(s1) % 16 == 0);
  bool dal = true;
  for (unsigned j = 0; j < (1u << a.g); ++j)
    for (unsigned p = 0; p < a.planes; ++p) dal = dal && reinterpret_cast<uintptr_t>(a.dst[j][p]) % 16 == 0;
  constexpr int VEC = 16 / (int)sizeof(E);
  const bool lowfixed = (a.perm.fixed_mask & (VEC - 1)) == (uint64_t)(VEC - 1) && a.m - a.g >= 2;
  if (al && dal && lowfixed) {
    const uint64_t units = size / VEC;
    const unsigned grid = (unsigned)std::min<uint64_t>((units + kBlock - 1) / kBlock, 256 * 64);
    hipLaunchKernelGGL((exchange_pack_kernel<E, VEC>), dim3(grid), dim3(kBlock), 0, s, s0, s1, a, units);
  } else {
    const unsigned grid = (unsigned)std::min<uint64_t>((size + kBlock - 1) / kBlock, 256 * 64);
    hipLaunchKernelGGL((exchange_pack_kernel<E, 1>), dim3(grid), dim3(kBlock), 0, s, s0, s1, a, size);
  }
  HQ_HIP_CHECK(hipGetLastError());
  return 0;
}

// The exchange.  Logical effect on the shard viewed through the optional local bit permutation
// `perm` (dst bit i <- src bit perm[i], like hq_permute_bits) and cut into G chunks by its top g
// local index bits: chunk j of rank r becomes chunk r of rank j.  *result_in_src = 1 when the
// exchanged shard ends up in the src planes (RCCL transport with a permutation: pack src -> dst,
// transfer dst -> src), 0 when it is in the dst planes.
template <typename E>
static int exchange_entry(E* src_re, E* src_im, E* dst_re, E* dst_im, unsigned m, const unsigned* perm, int* result_in_src) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  Shard& sh = shard();
  HQ_NOT_RECORDABLE(c, "the shard exchange");
  if (!src_re || !src_im || !dst_re || !dst_im || !result_in_src) return fail("exchange: null pointer");
  if (sh.transport == 0 && sh.world > 1) return fail("exchange: no transport (call hq_shard_init_rccl / hq_shard_init_p2p)");
  if (m > 62 || m < 2 * sh.g + 2) return fail("exchange: shard too small for the number of ranks");
  if (!is_device_pointer(src_re) || !is_device_pointer(src_im) || !is_device_pointer(dst_re) || !is_device_pointer(dst_im))
    return fail("exchange: device pointers only");
  const unsigned G = sh.world, g = sh.g;
  const size_t chunk = (((size_t)1 << m) >> g) * sizeof(E);
  ExchArg a;
  memset(&a, 0, sizeof(a));
  a.g = g;
  a.m = m;
  a.perm.fixed_mask = ~0ull;
  bool has_perm = false;
  if (perm) {
    uint64_t seen = 0;
    a.perm.fixed_mask = ~((m < 64 ? (1ull << m) : 0ull) - 1);
    for (unsigned i = 0; i < m; ++i) {
      if (perm[i] >= m || (seen >> perm[i]) & 1) return fail("exchange: perm is not a permutation of 0..m-1");
      seen |= 1ull << perm[i];
      if (perm[i] == i) a.perm.fixed_mask |= 1ull << i; else has_perm = true;
    }
    for (unsigned i = 0; i < m;) {
      if (perm[i] == i) { ++i; continue; }
      unsigned len = 1;
      while (i + len < m && perm[i + len] == perm[i] + len) ++len;
      a.perm.from[a.perm.nfields] = (unsigned char)i;
      a.perm.to[a.perm.nfields] = (unsigned char)perm[i];
      a.perm.len[a.perm.nfields] = (unsigned char)len;
      ++a.perm.nfields;
      i += len;
    }
  }
  unsigned char* S[2] = {reinterpret_cast<unsigned char*>(src_re), reinterpret_cast<unsigned char*>(src_im)};
  unsigned char* D[2] = {reinterpret_cast<unsigned char*>(dst_re), reinterpret_cast<unsigned char*>(dst_im)};

  if (G == 1) {  // one rank: the exchange is the permutation alone
    if (!has_perm) { *result_in_src = 1; return 0; }
    a.planes = 2;
    a.dst[0][0] = D[0];
    a.dst[0][1] = D[1];
    if (launch_pack<E>(c, c.stream, src_re, src_im, a)) return 1;
    *result_in_src = 0;
    return 0;
  }

  if (sh.transport == 2) {
    // peer-to-peer: ONE pass.  Rank j's receive planes are mapped here; this rank's chunk j goes
    // straight into slot `rank` of them.  The caller brackets the call with barriers (header).
    const Shard::Peers *pr = nullptr, *pi = nullptr;
    for (const auto& e : sh.registry) {
      if (e.local == dst_re) pr = &e;
      if (e.local == dst_im) pi = &e;
    }
    if (!pr || !pi) return fail("exchange: dst planes are not registered for peer-to-peer access (hq_shard_p2p_register)");
    a.planes = 2;
    for (unsigned j = 0; j < G; ++j) {
      a.dst[j][0] = reinterpret_cast<unsigned char*>(pr->peer[j]) + (size_t)sh.rank * chunk;
      a.dst[j][1] = reinterpret_cast<unsigned char*>(pi->peer[j]) + (size_t)sh.rank * chunk;
    }
    if (launch_pack<E>(c, c.stream, src_re, src_im, a)) return 1;
    *result_in_src = 0;
    return 0;
  }

  // RCCL transport: every rank sends chunk j to rank j and receives chunk j from it, all 2(G-1)
  // transfers of a plane in ONE group so that the 7 xGMI links of a GPU run at the same time; the
  // self chunk never goes near RCCL (its self copy measured 180 GB/s; ours streams at HBM rate).
  hipStream_t cs = sh.comm_stream;
  auto transfer_plane = [&](unsigned char* from, unsigned char* to) -> int {
    for (unsigned j = 0; j < G; ++j) {
      if (j == sh.rank) continue;
      HQ_NCCL_CHECK(sh, sh.api.Send(from + (size_t)j * chunk, chunk, ncclChar, (int)j, sh.comm, cs));
      HQ_NCCL_CHECK(sh, sh.api.Recv(to + (size_t)j * chunk, chunk, ncclChar, (int)j, sh.comm, cs));
    }
    return 0;
  };
  if (!has_perm) {
    HQ_HIP_CHECK(hipEventRecord(sh.ev[0], c.stream));  // src is final once the stream reaches here
    HQ_HIP_CHECK(hipStreamWaitEvent(cs, sh.ev[0], 0));
    HQ_NCCL_CHECK(sh, sh.api.GroupStart());
    if (transfer_plane(S[0], D[0]) || transfer_plane(S[1], D[1])) { (void)sh.api.GroupEnd(); return 1; }
    HQ_NCCL_CHECK(sh, sh.api.GroupEnd());
    for (int p = 0; p < 2; ++p)
      if (copy16(c, c.stream, D[p] + (size_t)sh.rank * chunk, S[p] + (size_t)sh.rank * chunk, chunk)) return 1;
    HQ_HIP_CHECK(hipEventRecord(sh.ev[2], cs));
    HQ_HIP_CHECK(hipStreamWaitEvent(c.stream, sh.ev[2], 0));
    *result_in_src = 0;
    return 0;
  }
  // with a permutation: pack plane p into the dst planes (send layout) on the main stream, transfer
  // dst -> src on the communication stream; the transfer of the re plane overlaps the packing of im
  a.planes = 1;
  for (int p = 0; p < 2; ++p) {
    for (unsigned j = 0; j < G; ++j) a.dst[j][0] = D[p] + (size_t)j * chunk;
    if (launch_pack<E>(c, c.stream, reinterpret_cast<const E*>(S[p]), (const E*)nullptr, a)) return 1;
    HQ_HIP_CHECK(hipEventRecord(sh.ev[p], c.stream));
    HQ_HIP_CHECK(hipStreamWaitEvent(cs, sh.ev[p], 0));
    HQ_NCCL_CHECK(sh, sh.api.GroupStart());
    if (transfer_plane(D[p], S[p])) { (void)sh.api.GroupEnd(); return 1; }
    HQ_NCCL_CHECK(sh, sh.api.GroupEnd());
  }
  for (int p = 0; p < 2; ++p)  // self chunk: after BOTH packs (the main stream is ordered), src chunk `rank` is free
    if (copy16(c, c.stream, S[p] + (size_t)sh.rank * chunk, D[p] + (size_t)sh.rank * chunk, chunk)) return 1;
  HQ_HIP_CHECK(hipEventRecord(sh.ev[2], cs));
  HQ_HIP_CHECK(hipStreamWaitEvent(c.stream, sh.ev[2], 0));
  *result_in_src = 1;
  return 0;
}

}  // namespace hq

// -----------------------------------------------------------------------------------
// C ABI
// -----------------------------------------------------------------------------------
extern "C" {

unsigned int get_log2_pack_size(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::read_env(c);
  return c.log2_pack;
}

int apply_U_float32(float* psi_re, float* psi_im, const float* U, const unsigned int* pos,
                    unsigned int n_qubits, unsigned int n_pos) {
  return hq::apply_U_entry<float>(psi_re, psi_im, U, pos, n_qubits, n_pos);
}
int apply_U_float64(double* psi_re, double* psi_im, const double* U, const unsigned int* pos,
                    unsigned int n_qubits, unsigned int n_pos) {
  return hq::apply_U_entry<double>(psi_re, psi_im, U, pos, n_qubits, n_pos);
}

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

int hq_apply_blocked_float32(float* re, float* im, unsigned int n, const unsigned int* tile_pos,
                             unsigned int tile_bits, unsigned int n_gates, const float* U_all,
                             const unsigned int* pos_all, const unsigned int* k_all) {
  return hq::apply_blocked_entry<float>(re, im, n, tile_pos, tile_bits, n_gates, U_all, pos_all, k_all);
}
int hq_apply_blocked_float64(double* re, double* im, unsigned int n, const unsigned int* tile_pos,
                             unsigned int tile_bits, unsigned int n_gates, const double* U_all,
                             const unsigned int* pos_all, const unsigned int* k_all) {
  return hq::apply_blocked_entry<double>(re, im, n, tile_pos, tile_bits, n_gates, U_all, pos_all, k_all);
}

int hq_program_begin(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::read_env(c);
  if (c.rec) return hq::fail("hq_program_begin: already recording");
  if (hq::check_device(c)) return 1;
  hq::Program* p = new hq::Program();
  size_t mb = 64;
  if (const char* e = getenv("HQ_PROGRAM_MB")) mb = (size_t)std::max(1, atoi(e));
  p->cap = mb << 20;
  if (const char* e = getenv("HQ_PROGRAM_GRAPH")) p->use_graph = atoi(e) != 0;
  hipError_t err = hipMalloc((void**)&p->dev, p->cap);
  if (err != hipSuccess) {
    delete p;
    return hq::fail(std::string("hq_program_begin: hipMalloc: ") + hipGetErrorString(err));
  }
  c.rec = p;
  return 0;
}

int hq_program_end(void** handle) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  if (!c.rec) return hq::fail("hq_program_end: not recording");
  hq::Program* p = c.rec;
  c.rec = nullptr;
  if (!handle) {  // nobody could run or free it
    (void)hipFree(p->dev);
    delete p;
    return hq::fail("hq_program_end: null handle pointer");
  }
  if (!p->host.empty()) {
    hipError_t err = hipMemcpy(p->dev, p->host.data(), p->host.size(), hipMemcpyHostToDevice);
    if (err != hipSuccess) {
      (void)hipFree(p->dev);
      delete p;
      return hq::fail(std::string("hq_program_end: hipMemcpy: ") + hipGetErrorString(err));
    }
  }
  p->host.clear();
  p->host.shrink_to_fit();
  p->finalized = true;
  *handle = p;
  return 0;
}

int hq_program_size(void* handle) {
  return handle ? (int)reinterpret_cast<hq::Program*>(handle)->ops.size() : -1;
}

int hq_program_run(void* handle) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Program* p = reinterpret_cast<hq::Program*>(handle);
  if (!p || !p->finalized) return hq::fail("hq_program_run: invalid program");
  if (c.rec) return hq::fail("hq_program_run: cannot run while recording");
  if (!p->use_graph || p->ops.size() < 2) {
    for (auto& op : p->ops) op(c.stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hq::fail(std::string("hq_program_run: ") + hipGetErrorString(e));
    return 0;
  }
  // hipGraph replay.  The legacy default stream cannot be captured, so a null library stream
  // is replaced by a private BLOCKING stream: legacy-stream semantics order it with the work
  // around it on the default stream.
  hipStream_t s = c.stream;
  if (s == nullptr) {
    if (!p->graph_stream) {
      hipError_t e = hipStreamCreate(&p->graph_stream);
      if (e != hipSuccess) return hq::fail(std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    s = p->graph_stream;
  }
  if (!p->exec) {
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return hq::fail(std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
    for (auto& op : p->ops) op(s);
    e = hipStreamEndCapture(s, &graph);
    if (e != hipSuccess) return hq::fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(&p->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return hq::fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  }
  hipError_t e = hipGraphLaunch(p->exec, s);
  if (e != hipSuccess) return hq::fail(std::string("hipGraphLaunch: ") + hipGetErrorString(e));
  return 0;
}

int hq_program_free(void* handle) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Program* p = reinterpret_cast<hq::Program*>(handle);
  if (!p) return 0;
  if (c.rec == p) c.rec = nullptr;
  (void)hipStreamSynchronize(c.stream);
  if (p->graph_stream) {
    (void)hipStreamSynchronize(p->graph_stream);
    (void)hipStreamDestroy(p->graph_stream);
  }
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->dev) (void)hipFree(p->dev);
  delete p;
  return 0;
}

int hq_set_stream(void* hip_stream) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  if (s == c.stream) return 0;
  if (c.rec) return hq::fail("hq_set_stream: cannot change the stream while recording a program");
  // Work already enqueued on the old stream may still be reading the upload arena / scratch
  // buffers (they are recycled in issue order): the new stream waits for it on the DEVICE, the
  // host does not block.
  if (c.device >= 0) {
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, c.stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, ev, 0);
    if (ev) (void)hipEventDestroy(ev);
    if (e != hipSuccess) return hq::fail(std::string("hq_set_stream: ") + hipGetErrorString(e));
  }
  c.stream = s;
  return 0;
}

int hq_sync(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hipError_t e = hipStreamSynchronize(c.stream);
  if (e != hipSuccess) return hq::fail(std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  return 0;
}

int hq_set_log2_pack_size(unsigned int v) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::read_env(c);
  if (v < 1 || v > 5) return hq::fail("log2_pack_size must be in 1..5");
  c.log2_pack = v;
  return 0;
}

const char* hq_last_error(void) { return hq::ctx().last_error.c_str(); }

int hq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int hq_set_apply_mode(const char* name) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::read_env(c);
  std::string s(name ? name : "");
  if (s == "auto") c.mode = hq::Mode::Auto;
  else if (s == "direct") c.mode = hq::Mode::Direct;
  else if (s == "mfma") c.mode = hq::Mode::Mfma;
  else if (s == "generic") c.mode = hq::Mode::Generic;
  else if (s == "naive") c.mode = hq::Mode::Naive;
  else if (s == "tile") c.mode = hq::Mode::Tile;
  else if (s == "gemm") c.mode = hq::Mode::Gemm;
  else if (s == "nt=1") c.nontemporal = 1;
  else if (s == "nt=0") c.nontemporal = 0;
  else if (s == "nt=auto") c.nontemporal = -1;
  else if (s == "dummy=comp") c.dummy_policy = 0;
  else if (s == "dummy=low") c.dummy_policy = 1;
  else if (s == "dummy=high") c.dummy_policy = 2;
  else if (s == "dummy=auto") c.dummy_policy = -1;
  else return hq::fail("unknown apply mode: " + s);
  return 0;
}

const char* hq_last_kernel(void) { return hq::ctx().last_kernel; }
const char* hq_last_kernel_desc(void) { return hq::ctx().last_desc.c_str(); }

// ---- multi-GPU shard exchange ----------------------------------------------------------
int hq_shard_unique_id(void* id128) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  if (!id128) return hq::fail("hq_shard_unique_id: null pointer");
  if (hq::load_rccl(sh)) return 1;
  ncclUniqueId id;
  HQ_NCCL_CHECK(sh, sh.api.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int hq_shard_init_rccl(unsigned int world, unsigned int rank, const void* id128) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  if (hq::shard_common_init(c, sh, world, rank)) return 1;
  if (world == 1 && !id128) { sh.transport = 0; return 0; }
  if (!id128) return hq::fail("hq_shard_init_rccl: null id");
  if (hq::load_rccl(sh)) return 1;
  if (sh.comm && sh.own_comm) { (void)sh.api.CommDestroy(sh.comm); sh.comm = nullptr; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  HQ_NCCL_CHECK(sh, sh.api.CommInitRank(&sh.comm, (int)world, id, (int)rank));
  sh.own_comm = true;
  sh.transport = world > 1 ? 1 : 0;  // a one-rank communicator is legal (hq_shard_rccl_selftest); the exchange needs none
  return 0;
}

// Plumbing check of the RCCL transport that needs no second GPU: one grouped ncclSend + ncclRecv of
// `bytes` from `src` to `dst` with THIS rank as the peer, on the communication stream, ordered
// against the library stream with the same events the exchange uses.
int hq_shard_rccl_selftest(const void* src, void* dst, uint64_t bytes) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  if (!sh.comm) return hq::fail("hq_shard_rccl_selftest: no communicator");
  if (!src || !dst) return hq::fail("hq_shard_rccl_selftest: null pointer");
  HQ_HIP_CHECK(hipEventRecord(sh.ev[0], c.stream));
  HQ_HIP_CHECK(hipStreamWaitEvent(sh.comm_stream, sh.ev[0], 0));
  HQ_NCCL_CHECK(sh, sh.api.GroupStart());
  HQ_NCCL_CHECK(sh, sh.api.Send(src, (size_t)bytes, ncclChar, (int)sh.rank, sh.comm, sh.comm_stream));
  HQ_NCCL_CHECK(sh, sh.api.Recv(dst, (size_t)bytes, ncclChar, (int)sh.rank, sh.comm, sh.comm_stream));
  HQ_NCCL_CHECK(sh, sh.api.GroupEnd());
  HQ_HIP_CHECK(hipEventRecord(sh.ev[2], sh.comm_stream));
  HQ_HIP_CHECK(hipStreamWaitEvent(c.stream, sh.ev[2], 0));
  return 0;
}

int hq_shard_attach_rccl(void* nccl_comm, unsigned int world, unsigned int rank) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  if (!nccl_comm) return hq::fail("hq_shard_attach_rccl: null communicator");
  if (hq::shard_common_init(c, sh, world, rank)) return 1;
  if (hq::load_rccl(sh)) return 1;
  if (sh.comm && sh.own_comm) (void)sh.api.CommDestroy(sh.comm);
  sh.comm = reinterpret_cast<ncclComm_t>(nccl_comm);
  sh.own_comm = false;
  sh.transport = world > 1 ? 1 : 0;
  return 0;
}

int hq_shard_init_p2p(unsigned int world, unsigned int rank) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  const bool same = sh.transport == 2 && sh.world == world && sh.rank == rank;
  if (hq::shard_common_init(c, sh, world, rank)) return 1;
  if (!same) sh.registry.clear();  // a second state of the same job keeps the planes already registered
  sh.transport = world > 1 ? 2 : 0;
  return 0;
}

int hq_shard_p2p_register(const void* local_plane, void* const* peer_planes) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  if (!local_plane || !peer_planes) return hq::fail("hq_shard_p2p_register: null pointer");
  hq::Shard::Peers e;
  memset(&e, 0, sizeof(e));
  e.local = local_plane;
  for (unsigned j = 0; j < sh.world; ++j) {
    if (!peer_planes[j]) return hq::fail("hq_shard_p2p_register: null peer pointer");
    e.peer[j] = peer_planes[j];
  }
  for (auto& old : sh.registry)
    if (old.local == local_plane) { old = e; return 0; }
  sh.registry.push_back(e);
  return 0;
}

int hq_shard_info(unsigned int* world, unsigned int* rank, int* transport) {
  hq::Shard& sh = hq::shard();
  if (world) *world = sh.world;
  if (rank) *rank = sh.rank;
  if (transport) *transport = sh.transport;
  return 0;
}

int hq_shard_free(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  if (sh.comm_stream) (void)hipStreamSynchronize(sh.comm_stream);
  if (sh.comm && sh.own_comm && sh.api.CommDestroy) (void)sh.api.CommDestroy(sh.comm);
  sh.comm = nullptr;
  sh.own_comm = false;
  sh.registry.clear();
  sh.transport = 0;
  sh.world = 1;
  sh.rank = 0;
  sh.g = 0;
  return 0;
}

int hq_ipc_export(const void* dev_ptr, void* handle64, uint64_t* offset) {
  if (!dev_ptr || !handle64 || !offset) return hq::fail("hq_ipc_export: null pointer");
  void* base = nullptr;
  size_t size = 0;
  hipError_t e = hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, (hipDeviceptr_t)dev_ptr);
  if (e != hipSuccess) return hq::fail(std::string("hipMemGetAddressRange: ") + hipGetErrorString(e));
  hipIpcMemHandle_t h;
  e = hipIpcGetMemHandle(&h, base);
  if (e != hipSuccess) return hq::fail(std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
  static_assert(sizeof(h) == 64, "hipIpcMemHandle_t");
  memcpy(handle64, &h, sizeof(h));
  *offset = (uint64_t)(reinterpret_cast<const unsigned char*>(dev_ptr) - reinterpret_cast<const unsigned char*>(base));
  return 0;
}

int hq_ipc_open(const void* handle64, uint64_t offset, void** dev_ptr) {
  if (!handle64 || !dev_ptr) return hq::fail("hq_ipc_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  void* base = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return hq::fail(std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
  *dev_ptr = reinterpret_cast<unsigned char*>(base) + offset;
  return 0;
}

int hq_ipc_close(void* dev_ptr, uint64_t offset) {
  if (!dev_ptr) return 0;
  hipError_t e = hipIpcCloseMemHandle(reinterpret_cast<unsigned char*>(dev_ptr) - offset);
  if (e != hipSuccess) return hq::fail(std::string("hipIpcCloseMemHandle: ") + hipGetErrorString(e));
  return 0;
}

int hq_exchange_float32(float* src_re, float* src_im, float* dst_re, float* dst_im, unsigned int n_local,
                        const unsigned int* perm, int* result_in_src) {
  return hq::exchange_entry<uint32_t>((uint32_t*)src_re, (uint32_t*)src_im, (uint32_t*)dst_re, (uint32_t*)dst_im, n_local, perm, result_in_src);
}
int hq_exchange_float64(double* src_re, double* src_im, double* dst_re, double* dst_im, unsigned int n_local,
                        const unsigned int* perm, int* result_in_src) {
  return hq::exchange_entry<uint64_t>((uint64_t*)src_re, (uint64_t*)src_im, (uint64_t*)dst_re, (uint64_t*)dst_im, n_local, perm, result_in_src);
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

// Diagnostics (not part of the reference boundary): raw hipPointerGetAttributes result.
extern "C" int hq_pointer_info(const void* p, int* type, int* device, int* err) {
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (err) *err = (int)e;
  if (e != hipSuccess) (void)hipGetLastError();
  if (type) *type = (int)attr.type;
  if (device) *device = attr.device;
  return e == hipSuccess ? 0 : 1;
}

#ifdef HQ_EXP_TIMELINE
extern "C" int hq_debug_timeline(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hq::hq_timeline), sizeof(unsigned long long) * 512);
}
#endif
