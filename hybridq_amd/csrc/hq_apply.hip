// hq_apply.hip -- apply_U_float32/64 (the reference boundary of /root/reference/include/python_U.cpp:33-112, 131-143:
// k dispatch and kernel selection) and hq_apply_blocked_* (many gates per HBM pass).
#include <cmath>

#include "hq_common.h"
#include "hq_kernels_apply.h"
#include "hq_kernels_blocked.h"
#include "hq_kernels_blocked_r3.h"
#include "hq_kernels_gemm.h"

namespace hq {

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
  // Where the identity dummies of a k < 3 gate go (measured at n = 30 in round 2; the sweep script is in the history: tools/sweep_dummy.py):
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
// complex128 k = 6 (the only operand table above 64 KiB): HQ_BIG_TWOBASE=1 selects the round-5 form with a second LDS base
// address (no scratch, operand pipeline; compiled and emulated, never run on hardware), the default 0 the instantiation
// rounds 2-3 ran on hardware (plain indexing, ~52 B/lane of scratch).  One build, A/B by environment.
static bool big_two_bases() {
  static const bool on = env_int("HQ_BIG_TWOBASE", 0) != 0;
  return on;
}

template <typename T, int KBITS, int VMASK, bool PHASED, bool TWOB>
static int launch_mfma_big_tw(Context& c, T* re, T* im, const T* dA, const MfmaPlan<T>& P, unsigned n,
                              const BigOffsets& tab) {
  constexpr unsigned CB = Vec<T>::VB;
  constexpr int BLOCK = 512;
  constexpr int NS = KBITS - 2, NRB = 1 << (NS - 2), NSTEP = 1 << NS;
  constexpr size_t lds = (size_t)NRB * NSTEP * 64 * sizeof(T);
  static bool attr_done = false;  // under the context mutex
  if (!attr_done) {
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)apply_mfma_big_kernel<T, KBITS, VMASK, true, BLOCK, PHASED, TWOB>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)apply_mfma_big_kernel<T, KBITS, VMASK, false, BLOCK, PHASED, TWOB>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  const uint64_t niter = (1ull << (n - CB - P.n_addr)) >> 4;  // 16 slots per wave iteration
  const uint64_t wgs = (niter + BLOCK / 64 - 1) / (BLOCK / 64);
  static const int grid_cap = env_int("HQ_BIG_GRID", 2048);
  const unsigned grid = (unsigned)std::min<uint64_t>(wgs, (uint64_t)grid_cap);
  const MfmaRoles ro = P.ro;
  if (P.nt)
    HQ_LAUNCH(c, (apply_mfma_big_kernel<T, KBITS, VMASK, true, BLOCK, PHASED, TWOB>), dim3(grid), dim3(BLOCK), lds, re, im, dA, ro, tab, niter);
  else
    HQ_LAUNCH(c, (apply_mfma_big_kernel<T, KBITS, VMASK, false, BLOCK, PHASED, TWOB>), dim3(grid), dim3(BLOCK), lds, re, im, dA, ro, tab, niter);
  return 0;
}

template <typename T, int KBITS, int VMASK, bool PHASED>
static int launch_mfma_big_var(Context& c, T* re, T* im, const T* dA, const MfmaPlan<T>& P, unsigned n,
                               const BigOffsets& tab) {
  constexpr int NS = KBITS - 2, NRB = 1 << (NS - 2), NSTEP = 1 << NS;
  if constexpr ((size_t)NRB * NSTEP * 64 * sizeof(T) > 65536) {
    if (!big_two_bases()) return launch_mfma_big_tw<T, KBITS, VMASK, PHASED, false>(c, re, im, dA, P, n, tab);
  }
  return launch_mfma_big_tw<T, KBITS, VMASK, PHASED, true>(c, re, im, dA, P, n, tab);
}

// measured at n = 30 / 29 (gpurun_out/r2e, r2f: means over 6 position patterns, phased vs free-running):
//   k = 5 f32 3.42 vs 3.61 ms, f64 3.40 vs 3.51;  k = 6 f32 4.65 vs 4.72;  k = 6 f64 5.35 vs 4.81 (the f64
//   k = 6 instantiation needs 218 of 256 registers and spills ~100 B/lane inside the MFMA phase: with the
//   partner wave parked at the barrier nothing covers the reloads)
static bool big_phased(int kbits, bool is_double) {
  static const int forced = env_int("HQ_BIG_PHASED", -1);
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
                (phased ? "512, true>" : "512, false>") + (sizeof(T) == 8 && P.kbits >= 7 ? (big_two_bases() ? " twobase=1" : " twobase=0") : "");
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
  static const int forced = env_int("HQ_GEMM_TB", 0);
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
  const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 2048);
  // register prefetch of the next tile where the tile has the usual size of this wave shape and the registers allow
  // (not the 8 x 1 shape of k = 10: 224 + 64 registers); HQ_GEMM_PREF=0 switches it off
  constexpr int CBv = sizeof(T) == 4 ? 2 : 1;
  constexpr int NPVx = sizeof(T) == 4 ? (CBW == 2 && RBW == 1 ? 2 : 0)  // f32: k = 7 only (2 x 2, k = 8: no gain, -3 % for some positions; 4 x 2 would spill)
                                     : (CBW == 1 ? (RBW == 1 ? 2 : (RBW == 2 ? 4 : (RBW == 4 ? 8 : 0))) : 0);
  static const int use_pref = env_int("HQ_GEMM_PREF", 1);
  // HQ_GEMM_PIPE=1: operands requested ahead of the matrix cores (round 4, compiled and emulated only); the default 0 is the
  // K loop of rounds 1-3, the one hardware has run (defaults = what hardware has verified)
  static const int use_pipe = env_int("HQ_GEMM_PIPE", 0);
  constexpr bool kCanPipe = gemm_can_pipe<T, RBW, CBW>();
  const bool pipe = use_pipe && kCanPipe;
  static bool attr_done = false;  // under the context mutex (one flag per instantiation of this function; `pipe` is fixed per process)
  if (!attr_done) {  // only the instantiations this process can launch: a default run does not touch the opt-in ones
    const void* f0 = pipe ? (const void*)apply_gemm_kernel<T, RBW, CBW, 0, kCanPipe> : (const void*)apply_gemm_kernel<T, RBW, CBW, 0, false>;
    const void* f1 = pipe ? (const void*)apply_gemm_kernel<T, RBW, CBW, NPVx, kCanPipe> : (const void*)apply_gemm_kernel<T, RBW, CBW, NPVx, false>;
    HQ_HIP_CHECK(hipFuncSetAttribute(f0, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    HQ_HIP_CHECK(hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_done = true;
  }
  if constexpr (NPVx > 0) {
    if (use_pref && ((1u << (a.tb - CBv)) == (unsigned)NPVx * kGemmBlock)) {
      if (pipe) HQ_LAUNCH(c, (apply_gemm_kernel<T, RBW, CBW, NPVx, kCanPipe>), dim3(grid), dim3(kGemmBlock), lds, re, im, dA, dOff, a, ntiles);
      else HQ_LAUNCH(c, (apply_gemm_kernel<T, RBW, CBW, NPVx, false>), dim3(grid), dim3(kGemmBlock), lds, re, im, dA, dOff, a, ntiles);
      return 0;
    }
  }
  if (pipe) HQ_LAUNCH(c, (apply_gemm_kernel<T, RBW, CBW, 0, kCanPipe>), dim3(grid), dim3(kGemmBlock), lds, re, im, dA, dOff, a, ntiles);
  else HQ_LAUNCH(c, (apply_gemm_kernel<T, RBW, CBW, 0, false>), dim3(grid), dim3(kGemmBlock), lds, re, im, dA, dOff, a, ntiles);
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
  // north_star reserves the matrix cores for k >= 4.  Measured per position class on both placements
  // (profiles/r03_sweep_valu_vs_mfma.txt, n = 30): with every target at index bit >= 8 the VALU butterfly kernel ties with the
  // role kernel (+-1 %) and wins 1.5-4 % for pairs / triples of high targets, at the same power and clock; with a target
  // below bit 8 the role kernel wins 2-40 % (its q digits turn the stride into a permutation of a contiguous run).  So
  // Auto sends k <= 3 gates whose targets all sit at bit >= 8 to the VALU kernel and keeps the role kernel for the rest
  // (HQ_VALU_HIGH=0: role kernel everywhere, the round-2 default).
  static const bool valu_high = env_int("HQ_VALU_HIGH", 1) != 0;
  if (c.mode == Mode::Auto && valu_high && can_direct && k <= 3 && n >= 20 && sizeof(T) == 4) {  // measured for complex64 only
    unsigned lo = pos[0];
    for (unsigned j = 1; j < k; ++j) lo = std::min(lo, pos[j]);
    if (lo >= 8) return launch_direct<T>(c, re, im, U, pos, n, k);
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
// apply_blocked: a list of gates inside one LDS tile, one HBM pass (device pointers)
// ---------------------------------------------------------------------------------
// HQ_BLOCKED_*: read once per process.  `pipe`, `groups`, `direct` and `big` select kernel code of rounds 4-5, which was
// written without access to hardware: ALL FOUR DEFAULT TO 0 -- a default run launches the kernels the driver's GPU tests of
// round 2 saw -- and a process that opts in gets the self-check below, which may switch them off again at run time.
struct BlockedSwitches {
  int valu_kmax;  // largest k that takes the register butterfly on the LDS tile (blocked_inner_gate_valu) instead of the
                  // matrix-core form with identity dummies: measured, k = 1 wins (4x fewer flops), k = 2 does not
  int grid_cap;   // HQ_BLOCKED_GRID (a power of two) caps the resident workgroups: small states then walk several tiles per
                  // workgroup, which is how the tile loop of the prefetching kernels is exercised without a 2^22-amplitude state
  int alds, pref;
  int big;        // tiles of 2^14 (f32) / 2^13 (f64) amplitudes = 128 KiB, ONE workgroup of 1024 threads per CU (16 waves:
                  // the same four per SIMD as two 512-thread workgroups); opt-in until measured
  int direct;     // tile movement folded into the first gate (apply_blocked_direct_kernel); opt-in until measured
  int groups;     // barrier-free wave groups
  int pipe;       // LDS requests one wave-iteration ahead of the matrix cores, table words one gate ahead
  int selfcheck;  // how many of the first passes of the process are cross-checked (0: none)
  int r3;         // 1 (default): a pass that uses none of the opt-in variants is launched on the kernels of hq_kernels_blocked_r3.h
                  // -- the binaries of the last commit that ran on hardware; 0: on the PIPE = false instantiations of
                  // hq_kernels_blocked.h (the code the variants are built on; what the self-check compares against)
};
static BlockedSwitches& blocked_switches() {
  // (the direct first gate and the 1024-thread tiles exist with the pipelined inner gates only: asking for one of them
  // implies HQ_BLOCKED_PIPE=1 unless the environment says otherwise)
  static BlockedSwitches s = {env_int("HQ_BLOCKED_VALU", 1),   env_int("HQ_BLOCKED_GRID", 0),
                              env_int("HQ_BLOCKED_ALDS", 1),   env_int("HQ_BLOCKED_PREF", 1), env_int("HQ_BLOCKED_BIG", 0),
                              env_int("HQ_BLOCKED_DIRECT", 0), env_int("HQ_BLOCKED_GROUPS", 0),
                              env_int("HQ_BLOCKED_PIPE", (env_int("HQ_BLOCKED_DIRECT", 0) || env_int("HQ_BLOCKED_BIG", 0)) ? 1 : 0),
                              env_int("HQ_BLOCKED_SELFCHECK", 3), env_int("HQ_BLOCKED_R3", 1)};
  return s;
}
static int g_selfcheck_runs = 0, g_selfcheck_failures = 0;  // under the context mutex

// One pass, parsed: gates in tile-local coordinates, their operand tables, the positions they touch.
template <typename T>
struct BlockedPass {
  BlockedArg ba;
  unsigned tb = 0;
  std::vector<BlockedGate> gates;
  std::vector<unsigned> touched;  // tile-local positions a gate acts on
  std::vector<T> Atab;
};

template <typename T>
static int blocked_launch(Context& c, T* re, T* im, const unsigned n, BlockedPass<T> P /* a copy: reordered / annotated here */,
                          const BlockedSwitches& sw, const bool describe, int* moved_front = nullptr,
                          bool* variant_code = nullptr /* out: the launch would run kernel code of rounds 4-5 */,
                          const bool dry_run = false /* decide only: nothing is uploaded or launched */) {
  constexpr unsigned CB = Vec<T>::VB;
  const unsigned tb = P.tb, n_gates = (unsigned)P.gates.size();
  std::vector<BlockedGate>& gates = P.gates;
  std::vector<unsigned>& touched = P.touched;
  const std::vector<T>& Atab = P.Atab;
  const BlockedArg& ba = P.ba;
  const uint64_t ntiles = 1ull << (n - tb);
  const size_t tile_bytes = ((size_t)2 << tb) * sizeof(T);
  const size_t per_cu = std::min<size_t>(std::max<size_t>(1, (160 * 1024) / tile_bytes), 4);
  // LDS limits: the default family (the instantiations a default run can launch) once per process; the opt-in
  // instantiations only in a process that opted in -- a default run does not depend on anything about them
  static bool attr_done = false, variant_attr_done = false;  // under the context mutex
  if (!attr_done) {
    const void* fns[] = {(const void*)apply_blocked_kernel<float, 512, false, false, false>, (const void*)apply_blocked_kernel<double, 512, false, false, false>,
                         (const void*)apply_blocked_kernel<float, 512, true, false, false>, (const void*)apply_blocked_kernel<double, 512, true, false, false>,
                         (const void*)apply_blocked_kernel<float, 512, true, true, false>, (const void*)apply_blocked_kernel<double, 512, true, true, false>,
                         (const void*)apply_blocked_kernel<float, 512, false, true, false>};
    for (const void* f : fns) HQ_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  if (!variant_attr_done && (sw.pipe || sw.direct || sw.big)) {
    const void* fns[] = {(const void*)apply_blocked_kernel<float, 512, true, false, true>, (const void*)apply_blocked_kernel<double, 512, true, false, true>,
                         (const void*)apply_blocked_kernel<float, 512, true, true, true>, (const void*)apply_blocked_kernel<double, 512, true, true, true>,
                         (const void*)apply_blocked_direct_kernel<float, 512>, (const void*)apply_blocked_direct_kernel<double, 512>,
                         (const void*)apply_blocked_kernel<float, 1024, true, true, true>, (const void*)apply_blocked_kernel<double, 1024, true, true, true>,
                         (const void*)apply_blocked_direct_kernel<float, 1024>, (const void*)apply_blocked_direct_kernel<double, 1024>};
    for (const void* f : fns) HQ_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    variant_attr_done = true;
  }
  unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 256 * per_cu);
  if (sw.grid_cap > 0 && (sw.grid_cap & (sw.grid_cap - 1)) == 0) grid = std::min<unsigned>(grid, (unsigned)sw.grid_cap);
  const int a_in_lds = sw.alds;
  // LDS left per workgroup behind the tile when `per_cu` workgroups share a CU
  const size_t a_budget = (160 * 1024) / per_cu - tile_bytes > 2048 ? (160 * 1024) / per_cu - tile_bytes - 1024 : 0;
  auto table_bytes = [&](unsigned words) { return (((size_t)n_gates * words * sizeof(BlockedTabT)) + 15) & ~(size_t)15; };
  // all or nothing: the 1024-thread kernel exists only with the tables in LDS, the register prefetch and the pipelined gates
  const bool big = sw.big && sw.pref && sw.pipe && a_in_lds && tb == (sizeof(T) == 4 ? 14u : 13u) &&
                   Atab.size() * sizeof(T) + table_bytes(BlockedTab<1024>::kWords) <= a_budget;
  const unsigned wave_bits_n = big ? 4u : 3u;
  const size_t tab_bytes = table_bytes(big ? BlockedTab<1024>::kWords : BlockedTab<512>::kWords);  // per-gate address tables (built in-kernel)
  // a gate's wave-iterations must fit the ITER part of its address table: 128 KiB tiles reach 128 of them (complex64 k <= 3
  // gate with both vector-component bits among its targets); the 512-thread kernel's tables hold 64 -- such a pass computes
  // its addresses (found under emulation in round 4: the table read ran into the next gate's entries)
  bool iter_ok = true;
  for (const BlockedGate& G : gates)
    if (G.kv < 64 && ((1u << (tb - CB - G.n_addr)) >> 4) > (big ? BlockedTab<1024>::kNIter : BlockedTab<512>::kNIter)) iter_ok = false;
  // operand tables in LDS only when ALL of them fit behind the tile without costing a resident
  // workgroup (measured: -3 % per pass; splitting a pass to make them fit costs a whole HBM pass)
  const bool fits = a_in_lds && iter_ok && Atab.size() * sizeof(T) + tab_bytes <= a_budget;
  // register prefetch of the next tile (512 threads, 4 vectors per thread and plane = 13 (f32) / 12 (f64) tile bits;
  // f64 only with the table-driven gates: the computed-address variant has no registers left for it)
  const bool pref = sw.pref && ((tb == (sizeof(T) == 4 ? 13u : 12u) && (sizeof(T) == 4 || fits)) || big);
  // Tile movement folded into the first gate (apply_blocked_direct_kernel): the pass needs a
  // matrix-core gate (k <= 4) whose register digits lie above tile-local vector bit 2 -- every wave-level HBM
  // access of the gate's own addressing is then a set of whole 128-byte lines -- that may run first: the earliest such gate
  // that shares no position with the gates in front of it is moved to the front (disjoint gates commute exactly).
  bool direct = false;
  const size_t gtab_bytes = (big ? blocked_gtab_words<1024>() : blocked_gtab_words<512>()) * sizeof(uint64_t);
  if (sw.direct && sw.pipe && pref && n_gates >= 2 && a_in_lds && fits && Atab.size() * sizeof(T) + tab_bytes + gtab_bytes <= a_budget) {
    auto eligible = [&](const BlockedGate& G) {
      if (G.kv < 16 || G.kv > 23) return false;  // matrix-core gates: KBITS = 4 (k <= 3), 5 (k = 4)
      const unsigned nr = (G.kv >> 2) - 2u - (unsigned)__builtin_popcount(G.kv & 3u);
      for (unsigned b = 0; b < nr; ++b)
        if (G.ro.r_plane != (int)b && G.ro.r_off[b] < 8u) return false;
      return true;
    };
    unsigned before = 0;
    for (unsigned g = 0; g < n_gates; ++g) {
      if (eligible(gates[g]) && !(touched[g] & before)) {
        std::rotate(gates.begin(), gates.begin() + g, gates.begin() + g + 1);
        std::rotate(touched.begin(), touched.begin() + g, touched.begin() + g + 1);
        direct = true;
        if (moved_front) *moved_front = (int)g;
        break;
      }
      before |= touched[g];
    }
  }
  unsigned n_barriers = n_gates;
  if (fits && sw.groups) {
    // Barrier-free groups (round 4).  A matrix-core inner gate splits the tile among the 8 waves by three tile-local
    // vector bits that are not address digits of the gate; consecutive gates that can agree on those three bits hand
    // the tile on wave by wave, and the workgroup barrier between them goes ("barrier only when a
    // gate crosses waves").  Greedy over the given order; the register-butterfly gates (k = 1) and gates with fewer
    // than 8 wave-iterations keep their barriers.
    const unsigned tvb = tb - CB, all = (1u << tvb) - 1;
    auto free_bits = [&](const BlockedGate& G) -> unsigned {
      if (G.kv >= 64 || ((1u << (tvb - G.n_addr)) >> 4) < (1u << wave_bits_n)) return 0;
      unsigned d = 0;
      for (int m = 0; m < 4; ++m)
        if (G.ro.pos[m] < 31) d |= 1u << G.ro.pos[m];
      const unsigned f = all & ~d;
      return (unsigned)__builtin_popcount(f) >= 4 + wave_bits_n ? f : 0;  // 4 slot bits + the wave bits
    };
    unsigned g0 = 0;
    while (g0 < n_gates) {
      unsigned common = free_bits(gates[g0]), g1 = g0 + 1;
      if (common) {
        while (g1 < n_gates) {
          const unsigned f = free_bits(gates[g1]);
          if (!f || (unsigned)__builtin_popcount(common & f) < wave_bits_n) break;
          common &= f;
          ++g1;
        }
      }
      if (g1 - g0 >= 2) {
        unsigned w = 0, rest = common;
        for (unsigned i = 0; i < wave_bits_n; ++i) {  // the HIGHEST common bits: the slot bits stay the lowest free ones (bank behaviour)
          const unsigned top = 31 - (unsigned)__builtin_clz(rest);
          w |= 1u << top;
          rest &= ~(1u << top);
        }
        for (unsigned g = g0; g < g1; ++g) gates[g].wave_bits = w | (g + 1 < g1 ? kBlockedNoBarrier : 0u);
        n_barriers -= g1 - g0 - 1;
      }
      g0 = g1;
    }
  }
  if (variant_code) *variant_code = fits && (sw.pipe || n_barriers < n_gates || direct || big);
  if (dry_run) return 0;
  void *dG = nullptr, *dA = nullptr;
  if (arena_upload(c, gates.data(), gates.size() * sizeof(BlockedGate), &dG)) return 1;
  if (arena_upload(c, Atab.data(), Atab.size() * sizeof(T), &dA)) return 1;
  const BlockedGate* const pG = (const BlockedGate*)dG;
  const T* const pA = (const T*)dA;
  const unsigned a_elems = (unsigned)Atab.size();
  // A pass that uses none of the opt-in variants: the kernels of the last commit that ran on hardware (hq_kernels_blocked_r3.h),
  // launched as that commit launched them -- same grid, same LDS bytes, same arguments (BlockedGate / BlockedArg have the same
  // layout in both namespaces; wave_bits = 0 is that commit's padding word).
  const bool use_r3 = sw.r3 && !sw.pipe && !direct && !big && n_barriers == n_gates;
  if (use_r3) {
    static_assert(sizeof(r3::BlockedGate) == sizeof(BlockedGate) && sizeof(r3::BlockedArg) == sizeof(BlockedArg), "layout");
    static_assert(r3::kBlockedTabWords == BlockedTab<512>::kWords, "address-table layout");
    static bool r3_attr_done = false;  // under the context mutex
    if (!r3_attr_done) {
      const void* fns[] = {(const void*)r3::apply_blocked_kernel<float, 512, false, false>, (const void*)r3::apply_blocked_kernel<double, 512, false, false>,
                           (const void*)r3::apply_blocked_kernel<float, 512, true, false>, (const void*)r3::apply_blocked_kernel<double, 512, true, false>,
                           (const void*)r3::apply_blocked_kernel<float, 512, true, true>, (const void*)r3::apply_blocked_kernel<float, 512, false, true>,
                           (const void*)r3::apply_blocked_kernel<double, 512, true, true>};
      for (const void* f : fns) HQ_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      r3_attr_done = true;
    }
    const r3::BlockedGate* const rG = reinterpret_cast<const r3::BlockedGate*>(dG);
    r3::BlockedArg rba;
    memcpy(&rba, &ba, sizeof(rba));
#define HQ_R3_LAUNCH(kern, lds_, ae_) HQ_LAUNCH(c, kern, dim3(grid), dim3(512), lds_, re, im, rG, n_gates, pA, ae_, rba, ntiles)
    if (fits) {
      const size_t lds = tile_bytes + Atab.size() * sizeof(T) + tab_bytes;
      if (pref) HQ_R3_LAUNCH((r3::apply_blocked_kernel<T, 512, true, true>), lds, a_elems);
      else HQ_R3_LAUNCH((r3::apply_blocked_kernel<T, 512, true, false>), lds, a_elems);
    } else if (pref && sizeof(T) == 4) {
      if constexpr (sizeof(T) == 4) HQ_R3_LAUNCH((r3::apply_blocked_kernel<T, 512, false, true>), tile_bytes, 0u);
    } else {
      HQ_R3_LAUNCH((r3::apply_blocked_kernel<T, 512, false, false>), tile_bytes, 0u);
    }
#undef HQ_R3_LAUNCH
    HQ_HIP_CHECK(hipGetLastError());
    if (describe) {
      c.last_kernel = "blocked";
      c.last_desc = std::string("r3::apply_blocked_kernel<") + (sizeof(T) == 4 ? "float" : "double") + ", 512> pipe=0 tb=" + std::to_string(tb) +
                    " gates=" + std::to_string(n_gates) + " barriers=" + std::to_string(n_barriers);
    }
    return 0;
  }
#define HQ_BLOCKED_LAUNCH(kern, threads_, lds_, ae_) \
  HQ_LAUNCH(c, kern, dim3(grid), dim3(threads_), lds_, re, im, pG, n_gates, pA, ae_, ba, ntiles)
  if (fits) {
    const size_t lds = tile_bytes + Atab.size() * sizeof(T) + tab_bytes + (direct ? gtab_bytes : 0);
    if (big) {
      if (direct) HQ_BLOCKED_LAUNCH((apply_blocked_direct_kernel<T, 1024>), 1024, lds, a_elems);
      else HQ_BLOCKED_LAUNCH((apply_blocked_kernel<T, 1024, true, true, true>), 1024, lds, a_elems);
    } else if (direct) {
      HQ_BLOCKED_LAUNCH((apply_blocked_direct_kernel<T, 512>), 512, lds, a_elems);
    } else if (pref) {
      if (sw.pipe) HQ_BLOCKED_LAUNCH((apply_blocked_kernel<T, 512, true, true, true>), 512, lds, a_elems);
      else HQ_BLOCKED_LAUNCH((apply_blocked_kernel<T, 512, true, true, false>), 512, lds, a_elems);
    } else {
      if (sw.pipe) HQ_BLOCKED_LAUNCH((apply_blocked_kernel<T, 512, true, false, true>), 512, lds, a_elems);
      else HQ_BLOCKED_LAUNCH((apply_blocked_kernel<T, 512, true, false, false>), 512, lds, a_elems);
    }
  } else {
    const size_t lds = tile_bytes;
    if (pref && sizeof(T) == 4) {
      if constexpr (sizeof(T) == 4) HQ_BLOCKED_LAUNCH((apply_blocked_kernel<T, 512, false, true, false>), 512, lds, 0u);
    } else {
      HQ_BLOCKED_LAUNCH((apply_blocked_kernel<T, 512, false, false, false>), 512, lds, 0u);
    }
  }
#undef HQ_BLOCKED_LAUNCH
  HQ_HIP_CHECK(hipGetLastError());
  if (describe) {
    c.last_kernel = "blocked";
    c.last_desc = std::string("apply_blocked_kernel<") + (sizeof(T) == 4 ? "float" : "double") + ", " +
                  std::to_string(big ? 1024 : 512) + "> pipe=" + (fits && sw.pipe ? "1" : "0") + " tb=" + std::to_string(tb) +
                  " gates=" + std::to_string(n_gates) + " barriers=" + std::to_string(n_barriers) + (direct ? " direct" : "");
  }
  return 0;
}

// The cross-check of the kernel code that was written without hardware (rounds 3-5: pipelined inner gates, barrier-free
// wave groups, direct first gate, 1024-thread tiles; all opt-in).  The first `HQ_BLOCKED_SELFCHECK` (default 3) passes of a
// process whose launch WOULD run any of that code are also run -- same gates, same tile shape, a scratch state of 64 tiles
// of pseudo-random amplitudes walked by 16 workgroups -- through the kernels of round 2 (one workgroup barrier per gate,
// LDS reads where the compiler puts them), whose results hardware tests have compared with the oracle.  All variants
// perform the same floating-point operations in the same order on every amplitude (the direct kernel may move a gate to
// the front of the pass: the reference run then takes the same order); the comparison still allows a few units in the
// last place (two template instantiations of one source need not contract a*b+c alike; -0.0 == +0.0), a wrong index or a
// stale operand is off by O(1).  If they disagree, the switches that survive the same comparison on their own stay on, the
// others go off for the rest of the process, and a warning names them.
// What this is NOT: a race detector (a missing barrier shows only under the timing of a full grid; tests/test_gpu_determinism.py
// and the wave-order runs of the emulation look for those) and not a reason for a caller's apply to fail: any runtime
// error in here (out of memory for the 4 MiB scratch state, a stream under capture, ...) counts the check as SKIPPED.
// Costs ~1 ms per checked pass.  Not recorded into programs; hq_blocked_selfcheck() reports runs / failures, the library
// warns once about a skip.
static int g_selfcheck_skipped = 0;  // under the context mutex

template <typename T>
static void blocked_selfcheck(Context& c, const BlockedPass<T>& P, const unsigned n, BlockedSwitches& sw) {
  constexpr unsigned CB = Vec<T>::VB;
  auto skipped = [&](const char* why) {
    (void)hipGetLastError();  // the caller's launch must not inherit this check's error
    if (!g_selfcheck_skipped++)
      fprintf(stderr, "libhq_hip: note: self-check of the opt-in cache-blocked kernel variants skipped (%s)\n", why);
  };
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(c.stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return skipped("the stream is being captured");
  const unsigned tb = P.tb, nc = std::min(n, tb + 6);
  BlockedPass<T> Q = P;
  for (unsigned i = CB; i < tb; ++i)  // the contiguous low run stays, the scattered bits move to the top of the small state
    if (P.ba.apos[i] != i) Q.ba.apos[i] = nc - tb + i;
  const size_t amps = (size_t)1 << nc, plane = amps * sizeof(T);
  std::vector<T> host(2 * amps), ref(2 * amps), got(2 * amps);
  uint64_t x = 0x9E3779B97F4A7C15ull ^ (uint64_t)P.gates.size();
  for (T& v : host) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    v = (T)((double)(int64_t)(x >> 11) * (1.0 / 4503599627370496.0) - 1.0);  // [-1, 1)
  }
  unsigned char* buf = nullptr;
  if (hipMalloc((void**)&buf, 2 * plane) != hipSuccess) return skipped("no device memory for the scratch state");
  auto run = [&](const BlockedPass<T>& pass, const BlockedSwitches& s, std::vector<T>& out, int* moved) -> int {
    hipError_t e = hipMemcpyAsync(buf, host.data(), 2 * plane, hipMemcpyHostToDevice, c.stream);
    if (e == hipSuccess && blocked_launch<T>(c, (T*)buf, (T*)(buf + plane), nc, pass, s, false, moved)) return 1;
    if (e == hipSuccess) e = hipMemcpyAsync(out.data(), buf, 2 * plane, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    return e != hipSuccess;
  };
  BlockedSwitches base = sw;
  base.pipe = base.groups = base.direct = base.big = 0;  // (with base.r3 at its default: the hardware-run kernels are the reference)
  base.grid_cap = 16;
  // a handful of units in the last place of the largest amplitude a pass of <= 64 gates on |x| < 1 produces
  const double ulps = 16.0 * (sizeof(T) == 4 ? 1.1920929e-7 : 2.220446049250313e-16);
  // 0: the switches agree with the round-2 kernels on this pass, 1: they do not, -1: a runtime call failed
  auto differs = [&](BlockedSwitches s) -> int {
    s.grid_cap = 16;
    int moved = -1;
    if (run(Q, s, got, &moved)) return -1;
    BlockedPass<T> R = Q;
    if (moved > 0) {
      std::rotate(R.gates.begin(), R.gates.begin() + moved, R.gates.begin() + moved + 1);
      std::rotate(R.touched.begin(), R.touched.begin() + moved, R.touched.begin() + moved + 1);
    }
    if (run(R, base, ref, nullptr)) return -1;
    if (!memcmp(ref.data(), got.data(), 2 * plane)) return 0;
    double scale = 1.0;
    for (const T v : ref) scale = std::max(scale, std::fabs((double)v));
    for (size_t i = 0; i < ref.size(); ++i)
      if (!(std::fabs((double)ref[i] - (double)got[i]) <= ulps * scale)) return 1;  // (NaN counts as a difference)
    return 0;
  };
  ++g_selfcheck_runs;
  int d = differs(sw);
  if (d > 0) {
    ++g_selfcheck_failures;
    std::string off;
    struct { const char* name; int BlockedSwitches::*field; } parts[] = {{"HQ_BLOCKED_PIPE", &BlockedSwitches::pipe}, {"HQ_BLOCKED_GROUPS", &BlockedSwitches::groups},
                                                                         {"HQ_BLOCKED_DIRECT", &BlockedSwitches::direct}, {"HQ_BLOCKED_BIG", &BlockedSwitches::big}};
    const BlockedSwitches asked = sw;
    for (auto& part : parts) {
      if (!(asked.*(part.field))) continue;
      BlockedSwitches one = base;
      one.*(part.field) = asked.*(part.field);
      if (part.field == &BlockedSwitches::direct || part.field == &BlockedSwitches::big) one.pipe = 1;  // they exist with the pipelined gates only
      if ((d = differs(one)) < 0) break;
      if (d) {
        sw.*(part.field) = 0;
        off += std::string(off.empty() ? "" : ", ") + part.name;
      }
    }
    if (d >= 0 && (d = differs(sw)) > 0) {  // what is left, together
      sw.pipe = sw.groups = sw.direct = sw.big = 0;
      off = "HQ_BLOCKED_PIPE, HQ_BLOCKED_GROUPS, HQ_BLOCKED_DIRECT, HQ_BLOCKED_BIG";
    }
    if (d >= 0)
      fprintf(stderr, "libhq_hip: WARNING: the cache-blocked kernel variants disagree on this device (pass of %zu gates, tile of 2^%u); "
              "switched off for this process: %s\n", P.gates.size(), tb, off.c_str());
  }
  (void)hipFree(buf);
  if (d < 0) {
    --g_selfcheck_runs;
    skipped("a runtime call of the check failed");
  }
}

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
  BlockedSwitches& sw = blocked_switches();
  BlockedPass<T> P;
  BlockedArg& ba = P.ba;
  memset(&ba, 0, sizeof(ba));
  ba.tb = P.tb = tb;
  int local_of[64];
  for (int i = 0; i < 64; ++i) local_of[i] = -1;
  for (unsigned i = 0; i < tb; ++i) {
    if (tile_pos[i] >= n || (i && tile_pos[i] <= tile_pos[i - 1])) return fail("apply_blocked: tile positions must be ascending and < n");
    ba.apos[i] = tile_pos[i];
    local_of[tile_pos[i]] = (int)i;
  }
  for (unsigned b = 0; b < CB; ++b)
    if (tile_pos[b] != b) return fail("apply_blocked: the tile must contain the vector-component index bits (0,1 for f32; 0 for f64)");
  std::vector<BlockedGate>& gates = P.gates;
  std::vector<unsigned>& touched = P.touched;
  std::vector<T>& Atab = P.Atab;
  gates.resize(n_gates);
  touched.assign(n_gates, 0u);
  const T* Up = U_all;
  const unsigned* pp = pos_all;
  for (unsigned g = 0; g < n_gates; ++g) {
    const unsigned k = k_all[g];
    if (k < 1 || k > 4) return fail("apply_blocked: gates must have 1..4 targets");
    unsigned lp[4];
    for (unsigned j = 0; j < k; ++j) {
      if (pp[j] >= 64 || local_of[pp[j]] < 0) return fail("apply_blocked: gate target outside the tile");
      lp[j] = (unsigned)local_of[pp[j]];
      touched[g] |= 1u << lp[j];
    }
    if (check_positions(lp, tb, k)) return fail("apply_blocked: duplicate targets");
    if (k <= 2 && (int)k <= sw.valu_kmax) {
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
    MfmaPlan<T> M;
    if (!plan_mfma<T>(c, Up, lp, tb, k, M, true)) return fail("apply_blocked: cannot plan an inner gate");
    BlockedGate& G = gates[g];
    memset(&G, 0, sizeof(G));
    G.ro = M.ro;
    for (int m = 0; m < 4; ++m)
      if (G.ro.pos[m] >= 31) G.ro.pos[m] = 31;
    G.a_off = (unsigned)Atab.size();
    G.kv = (unsigned)(M.kbits * 4 + M.vmask);
    G.n_addr = M.n_addr;
    Atab.insert(Atab.end(), M.A.begin(), M.A.end());
    Up += (size_t)2 << (2 * k);
    pp += k;
  }
  if (sw.selfcheck > 0 && !c.rec && (sw.pipe || sw.groups || sw.direct || sw.big) && sw.alds) {
    bool variant = false;  // the budget is spent on passes whose launch selects variant code only
    if (!blocked_launch<T>(c, re, im, n, P, sw, false, nullptr, &variant, true) && variant) {
      --sw.selfcheck;
      blocked_selfcheck<T>(c, P, n, sw);  // never fails the caller's apply
    }
  }
  return blocked_launch<T>(c, re, im, n, std::move(P), sw, true);
}

int apply_device_f32(Context& c, float* re, float* im, const float* U, const unsigned* pos, unsigned n, unsigned k) {
  return apply_device<float>(c, re, im, U, pos, n, k);
}
int apply_device_f64(Context& c, double* re, double* im, const double* U, const unsigned* pos, unsigned n, unsigned k) {
  return apply_device<double>(c, re, im, U, pos, n, k);
}

}  // namespace hq

extern "C" {

int apply_U_float32(float* psi_re, float* psi_im, const float* U, const unsigned int* pos,
                    unsigned int n_qubits, unsigned int n_pos) {
  return hq::apply_U_entry<float>(psi_re, psi_im, U, pos, n_qubits, n_pos);
}

int apply_U_float64(double* psi_re, double* psi_im, const double* U, const unsigned int* pos,
                    unsigned int n_qubits, unsigned int n_pos) {
  return hq::apply_U_entry<double>(psi_re, psi_im, U, pos, n_qubits, n_pos);
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

int hq_blocked_selfcheck(int* runs, int* failures, int* switches) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  const hq::BlockedSwitches& sw = hq::blocked_switches();
  if (runs) *runs = hq::g_selfcheck_runs;
  if (failures) *failures = hq::g_selfcheck_failures;
  if (switches) *switches = (sw.pipe ? 1 : 0) | (sw.groups ? 2 : 0) | (sw.direct ? 4 : 0) | (sw.big ? 8 : 0);
  return 0;
}

}  // extern "C"

