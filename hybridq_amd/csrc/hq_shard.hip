// hq_shard.hip -- multi-GPU: high-qubit shards and the qubit exchange (hq_shard_*, hq_exchange_*, hq_ipc_*).
#include "hq_common.h"
#include "hq_kernels_swap.h"
#include "hq_bitperm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen()ed (hq_shard_init_rccl)

namespace hq {

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
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;  // optional (diagnostics)
};

struct Shard {
  unsigned world = 1, rank = 0, g = 0;
  int transport = 0;  // 0 = none, 1 = RCCL send/recv, 2 = peer-to-peer stores through HIP IPC mappings
  RcclApi api;
  ncclComm_t comm = nullptr;
  bool own_comm = false;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> round_ev;  // hq_exchange_rounds_*: one per round, recorded on the communication stream
  unsigned rounds_pending = 0;       // rounds of the last hq_exchange_rounds_* call (hq_exchange_round_wait checks its argument)
  // p2p: local buffer address -> the same buffer on every rank (mapped into this process)
  struct Peers { const void* local; void* peer[kMaxShardRanks]; };
  std::vector<Peers> registry;
  double last_ms = 0;
  unsigned epoch = 0;  // bumped by hq_shard_free: a communicator creation that returns afterwards is discarded
};

static Shard& shard() {
  static Shard s;
  return s;
}

static int load_rccl(Shard& sh) {
  if (sh.api.lib) return 0;
  std::vector<std::string> names;
  const char* named = getenv("HQ_RCCL_LIBRARY");  // a library named by the user is the only candidate
  if (named && *named) names.push_back(named);
  else names.insert(names.end(), {"librccl.so.1", "librccl.so"});
  void* h = nullptr;
  // the copy already mapped by the process first (torch bundles its own librccl next to its own HIP
  // runtime: a second RCCL on another runtime could not see this process's allocations)
  for (const auto& nm : names) if (!h) h = dlopen(nm.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const auto& nm : names) if (!h) h = dlopen(nm.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h && !(named && *named)) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    const char* err = dlerror();  // one call: dlerror() clears the message it returns
    return fail(std::string("cannot load librccl: ") + (err ? err : "?"));
  }
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
  a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
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

// `perm`: the local bit permutation of the pack pass (m entries) or nullptr
template <typename E>
static int launch_pack(Context& c, hipStream_t s, const E* s0, const E* s1, const ExchArg& a, const unsigned* perm) {
  const uint64_t size = 1ull << a.m;
  const bool al = reinterpret_cast<uintptr_t>(s0) % 16 == 0 && (!s1 || reinterpret_cast<uintptr_t>(s1) % 16 == 0);
  bool dal = true;
  for (unsigned j = 0; j < (1u << a.g); ++j)
    for (unsigned p = 0; p < a.planes; ++p) dal = dal && reinterpret_cast<uintptr_t>(a.dst[j][p]) % 16 == 0;
  static const bool one_pass = env_int("HQ_PERM_TILE", 1) != 0;
  static const bool gather_low_fixed = env_int("HQ_PERM_TILE", 0) == 3;
  if (one_pass && al && dal && perm && !(gather_low_fixed && bitperm_low_run_fixed<E>(perm, a.m))) {
    // the eviction permutation at full cache-line granularity on both sides (bitperm_tile_kernel); the gather kernel
    // below stays for shards smaller than a tile
    BitPermPlan P;
    if (plan_bitperm<E>(perm, a.m, false, P)) {
      P.a.cbits = a.m - a.g;
      P.a.planes = a.planes;
      for (unsigned j = 0; j < (1u << a.g); ++j)
        for (unsigned p = 0; p < a.planes; ++p) P.a.dst[j][p] = a.dst[j][p];
      return launch_bitperm<E>(c, s, false, s0, s1, P);
    }
  }
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
//
// `sub_bits` / `n_rounds` (hq_exchange_rounds_*): the same exchange moved in 2^sub_bits ROUNDS -- round s carries piece s
// (of 2^sub_bits) of every chunk, both planes, all 2(G-1) transfers of a round in one group -- with an event per round
// on the communication stream and NO wait on the library stream: the caller waits round by round
// (hq_exchange_round_wait) and works on the pieces that have landed while the later rounds are on the wire.  The pack
// pass stays folded in and stays whole-plane: the transfers of a plane write into the src plane the pack has just read
// (there is no third buffer), so a plane's first round starts when that plane's pack is done -- the rounds of the re
// plane overlap the pack of im -- and the rounds then alternate between the planes.  The peer-to-peer transport (stores
// into the peers' planes, bracketed by the caller's barriers) and a single rank run as ONE round.
template <typename E>
static int exchange_entry(E* src_re, E* src_im, E* dst_re, E* dst_im, unsigned m, const unsigned* perm, int* result_in_src,
                          const unsigned sub_bits = 0, unsigned* n_rounds = nullptr) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  read_env(c);
  Shard& sh = shard();
  HQ_NOT_RECORDABLE(c, "the shard exchange");
  if (!src_re || !src_im || !dst_re || !dst_im || !result_in_src) return fail("exchange: null pointer");
  if (sh.transport == 0 && sh.world > 1) return fail("exchange: no transport (call hq_shard_init_rccl / hq_shard_init_p2p)");
  if (m > 62 || m < 2 * sh.g + 2) return fail("exchange: shard too small for the number of ranks");
  if (sub_bits > 6 || (sub_bits && m < 2 * sh.g + 2 + sub_bits)) return fail("exchange: too many rounds for this shard");
  const bool rounds = n_rounds != nullptr;
  // No shard state changes before every argument has been validated, and a call that fails leaves NO round to wait for:
  // rounds_pending is set when (and to what) the call has really recorded (hq_exchange_round_wait refuses the rest).
  if (rounds) { *n_rounds = 1; sh.rounds_pending = 0; }
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
  unsigned char* S_[2] = {reinterpret_cast<unsigned char*>(src_re), reinterpret_cast<unsigned char*>(src_im)};
  unsigned char* D[2] = {reinterpret_cast<unsigned char*>(dst_re), reinterpret_cast<unsigned char*>(dst_im)};
  // one-round transports (single rank, peer-to-peer): round 0 = "everything this call has issued on the library stream"
  auto one_round_done = [&]() -> int {
    if (!rounds) return 0;
    if (sh.round_ev.empty()) {
      hipEvent_t e0 = nullptr;
      HQ_HIP_CHECK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
      sh.round_ev.push_back(e0);
    }
    HQ_HIP_CHECK(hipEventRecord(sh.round_ev[0], c.stream));
    sh.rounds_pending = 1;
    return 0;
  };

  if (G == 1) {  // one rank: the exchange is the permutation alone
    if (!has_perm) { *result_in_src = 1; return one_round_done(); }
    a.planes = 2;
    a.dst[0][0] = D[0];
    a.dst[0][1] = D[1];
    if (launch_pack<E>(c, c.stream, src_re, src_im, a, has_perm ? perm : nullptr)) return 1;
    *result_in_src = 0;
    return one_round_done();
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
    if (launch_pack<E>(c, c.stream, src_re, src_im, a, has_perm ? perm : nullptr)) return 1;
    *result_in_src = 0;
    return one_round_done();
  }

  // RCCL transport: every rank sends chunk j to rank j and receives chunk j from it, all 2(G-1)
  // transfers of a plane in ONE group so that the 7 xGMI links of a GPU run at the same time; the
  // self chunk never goes near RCCL (its self copy measured 180 GB/s; ours streams at HBM rate).
  hipStream_t cs = sh.comm_stream;
  const unsigned S = rounds ? 1u << sub_bits : 1u;
  const size_t piece = chunk >> (rounds ? sub_bits : 0);
  // piece s of every chunk of one plane (S = 1: the whole chunks)
  auto transfer_plane = [&](unsigned char* from, unsigned char* to, unsigned s_ = 0) -> int {
    for (unsigned j = 0; j < G; ++j) {
      if (j == sh.rank) continue;
      HQ_NCCL_CHECK(sh, sh.api.Send(from + (size_t)j * chunk + s_ * piece, piece, ncclChar, (int)j, sh.comm, cs));
      HQ_NCCL_CHECK(sh, sh.api.Recv(to + (size_t)j * chunk + s_ * piece, piece, ncclChar, (int)j, sh.comm, cs));
    }
    return 0;
  };
  if (rounds) {
    while (sh.round_ev.size() < S) {
      hipEvent_t e = nullptr;
      HQ_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      sh.round_ev.push_back(e);
    }
    *n_rounds = S;
    unsigned char** from = has_perm ? D : S_;
    unsigned char** to = has_perm ? S_ : D;
    if (has_perm) {
      // pack plane p into the dst planes (send layout) on the library stream; ev[p] releases that plane's rounds
      a.planes = 1;
      for (int p = 0; p < 2; ++p) {
        for (unsigned j = 0; j < G; ++j) a.dst[j][0] = D[p] + (size_t)j * chunk;
        if (launch_pack<E>(c, c.stream, reinterpret_cast<const E*>(S_[p]), (const E*)nullptr, a, perm)) return 1;
        HQ_HIP_CHECK(hipEventRecord(sh.ev[p], c.stream));
      }
    } else {
      HQ_HIP_CHECK(hipEventRecord(sh.ev[0], c.stream));  // src is final once the stream reaches here
    }
    // One ncclGroup per round carrying BOTH planes = 4(G-1) transfers, so that every xGMI link has a send and a receive of
    // each plane queued in every round (include/hq_hip.h).  The one exception is round 0 of an exchange with a folded
    // permutation: the re plane's transfers go in a group of their own as soon as re is packed, while im is still being
    // packed on the library stream (one group would make the re transfers wait for the im pack).
    for (unsigned s_ = 0; s_ < S; ++s_) {
      if (s_ == 0 && has_perm) {
        for (int p = 0; p < 2; ++p) {
          HQ_HIP_CHECK(hipStreamWaitEvent(cs, sh.ev[p], 0));
          HQ_NCCL_CHECK(sh, sh.api.GroupStart());
          if (transfer_plane(from[p], to[p], 0)) { (void)sh.api.GroupEnd(); return 1; }
          HQ_NCCL_CHECK(sh, sh.api.GroupEnd());
        }
      } else {
        if (s_ == 0) HQ_HIP_CHECK(hipStreamWaitEvent(cs, sh.ev[0], 0));
        HQ_NCCL_CHECK(sh, sh.api.GroupStart());
        if (transfer_plane(from[0], to[0], s_) || transfer_plane(from[1], to[1], s_)) { (void)sh.api.GroupEnd(); return 1; }
        HQ_NCCL_CHECK(sh, sh.api.GroupEnd());
      }
      HQ_HIP_CHECK(hipEventRecord(sh.round_ev[s_], cs));
      sh.rounds_pending = s_ + 1;  // an error further down leaves exactly the recorded rounds waitable
    }
    for (int p = 0; p < 2; ++p)  // self chunk, all its pieces (with a permutation: after BOTH packs, src chunk `rank` is free)
      if (copy16(c, c.stream, to[p] + (size_t)sh.rank * chunk, from[p] + (size_t)sh.rank * chunk, chunk)) return 1;
    *result_in_src = has_perm ? 1 : 0;
    return 0;
  }
  if (!has_perm) {
    HQ_HIP_CHECK(hipEventRecord(sh.ev[0], c.stream));  // src is final once the stream reaches here
    HQ_HIP_CHECK(hipStreamWaitEvent(cs, sh.ev[0], 0));
    HQ_NCCL_CHECK(sh, sh.api.GroupStart());
    if (transfer_plane(S_[0], D[0]) || transfer_plane(S_[1], D[1])) { (void)sh.api.GroupEnd(); return 1; }
    HQ_NCCL_CHECK(sh, sh.api.GroupEnd());
    for (int p = 0; p < 2; ++p)
      if (copy16(c, c.stream, D[p] + (size_t)sh.rank * chunk, S_[p] + (size_t)sh.rank * chunk, chunk)) return 1;
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
    if (launch_pack<E>(c, c.stream, reinterpret_cast<const E*>(S_[p]), (const E*)nullptr, a, perm)) return 1;
    HQ_HIP_CHECK(hipEventRecord(sh.ev[p], c.stream));
    HQ_HIP_CHECK(hipStreamWaitEvent(cs, sh.ev[p], 0));
    HQ_NCCL_CHECK(sh, sh.api.GroupStart());
    if (transfer_plane(D[p], S_[p])) { (void)sh.api.GroupEnd(); return 1; }
    HQ_NCCL_CHECK(sh, sh.api.GroupEnd());
  }
  for (int p = 0; p < 2; ++p)  // self chunk: after BOTH packs (the main stream is ordered), src chunk `rank` is free
    if (copy16(c, c.stream, S_[p] + (size_t)sh.rank * chunk, D[p] + (size_t)sh.rank * chunk, chunk)) return 1;
  HQ_HIP_CHECK(hipEventRecord(sh.ev[2], cs));
  HQ_HIP_CHECK(hipStreamWaitEvent(c.stream, sh.ev[2], 0));
  *result_in_src = 1;
  return 0;
}

}  // namespace hq

extern "C" {

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

int hq_shard_load_rccl(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  return hq::load_rccl(hq::shard());
}

int hq_shard_init_rccl(unsigned int world, unsigned int rank, const void* id128) {
  hq::Context& c = hq::ctx();
  hq::Shard& sh = hq::shard();
  ncclUniqueId id;
  unsigned epoch = 0;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    if (hq::shard_common_init(c, sh, world, rank)) return 1;
    if (world == 1 && !id128) { sh.transport = 0; return 0; }
    if (!id128) return hq::fail("hq_shard_init_rccl: null id");
    if (hq::load_rccl(sh)) return 1;
    if (sh.comm && sh.own_comm) { (void)sh.api.CommDestroy(sh.comm); sh.comm = nullptr; }
    memcpy(&id, id128, sizeof(id));
    epoch = sh.epoch;
  }
  // ncclCommInitRank is collective and blocks until every rank has arrived: the context lock is NOT held meanwhile,
  // so that a caller who gave up waiting (hybridq_amd.dist runs this in a helper thread with a timeout) can still use
  // the library -- and cancel this creation with hq_shard_free
  ncclComm_t comm = nullptr;
  const ncclResult_t r = sh.api.CommInitRank(&comm, (int)world, id, (int)rank);
  std::lock_guard<std::mutex> lock(c.mu);
  if (r != ncclSuccess) return hq::fail(std::string("ncclCommInitRank: ") + sh.api.GetErrorString(r));
  if (epoch != sh.epoch) {  // cancelled while we were inside
    (void)sh.api.CommDestroy(comm);
    return hq::fail("hq_shard_init_rccl: cancelled by hq_shard_free");
  }
  sh.comm = comm;
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

int hq_shard_comm_count(int* count) {
  if (!count) return hq::fail("hq_shard_comm_count: null pointer");
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  *count = 0;
  if (sh.transport != 1 || !sh.comm) return 0;
  *count = (int)sh.world;
  if (sh.api.CommCount) HQ_NCCL_CHECK(sh, sh.api.CommCount(sh.comm, count));
  return 0;
}

int hq_shard_free(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  // a transfer that never completed (the reason the caller gives the transport up) must not hang the cleanup too: the
  // communication stream is only waited for, and the communicator only destroyed, when nothing is pending on it
  bool idle = true;
  if (sh.comm_stream) {
    idle = hipStreamQuery(sh.comm_stream) == hipSuccess;
    if (!idle) (void)hipGetLastError();
  }
  if (idle && sh.comm && sh.own_comm && sh.api.CommDestroy) (void)sh.api.CommDestroy(sh.comm);
  sh.comm = nullptr;
  sh.own_comm = false;
  sh.registry.clear();
  sh.rounds_pending = 0;  // (the round events themselves are kept for the next transport: they are re-recorded before use)
  ++sh.epoch;
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

int hq_exchange_rounds_float32(float* src_re, float* src_im, float* dst_re, float* dst_im, unsigned int n_local,
                               const unsigned int* perm, unsigned int sub_bits, int* result_in_src, unsigned int* n_rounds) {
  if (!n_rounds) return hq::fail("exchange: null pointer");
  return hq::exchange_entry<uint32_t>((uint32_t*)src_re, (uint32_t*)src_im, (uint32_t*)dst_re, (uint32_t*)dst_im, n_local, perm, result_in_src,
                                      sub_bits, n_rounds);
}

int hq_exchange_rounds_float64(double* src_re, double* src_im, double* dst_re, double* dst_im, unsigned int n_local,
                               const unsigned int* perm, unsigned int sub_bits, int* result_in_src, unsigned int* n_rounds) {
  if (!n_rounds) return hq::fail("exchange: null pointer");
  return hq::exchange_entry<uint64_t>((uint64_t*)src_re, (uint64_t*)src_im, (uint64_t*)dst_re, (uint64_t*)dst_im, n_local, perm, result_in_src,
                                      sub_bits, n_rounds);
}

int hq_exchange_round_wait(unsigned int round) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Shard& sh = hq::shard();
  if (round >= sh.rounds_pending || round >= sh.round_ev.size()) return hq::fail("hq_exchange_round_wait: no such round");
  HQ_HIP_CHECK(hipStreamWaitEvent(c.stream, sh.round_ev[round], 0));
  return 0;
}

}  // extern "C"
