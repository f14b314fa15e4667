// hq_kernels_blocked.h -- the cache-blocked pass: MANY gates in ONE HBM pass over LDS tiles (apply_blocked_kernel, its
// inner gates, apply_blocked_direct_kernel).  No reference counterpart for the schedule (the reference applies one fused
// gate per pass, simulation.py:522-646); each inner gate is U::apply (U.h:28-102) on the tile.  Split from hq_kernels_apply.h
// in round 5.
#pragma once
#include "hq_kernels_apply.h"

namespace hq {

// ---------------------------------------------------------------------------------
// apply_blocked (f32 and f64): MANY gates in ONE HBM pass.
//
// The per-gate kernels above sit at the memory system's ceiling (~3 ms per pass at n = 30),
// so the remaining lever is fewer passes.  A workgroup stages a tile of 2^TB amplitudes
// (TB = 13 for f32, 12 for f64: 2 x 32 KiB of LDS, two workgroups per CU) spanned by TB chosen index bits --
// the low bits (coalescing) plus any others -- applies a whole LIST of gates whose targets
// all lie inside those bits with the same role-assigned MFMA scheme as apply_mfma, now
// reading/writing LDS (ds_read_b128 / ds_write_b128, one workgroup barrier per gate), and
// streams the tile back.  HBM traffic is one read + one write of the state for the whole
// list; each inner gate costs MFMA time only (~0.45 ms for k <= 3, ~0.9 ms for k = 4 at
// n = 30, vs ~3.1 ms for a pass of its own).  The host planner (hybridq_amd/blocking.py)
// picks the tile bits and the gate lists from the circuit's dependency DAG.
// ---------------------------------------------------------------------------------
constexpr int kBlockedMaxTileBits = 14;
// LDS layout of a tile plane: 16-byte vector v lives at slot v ^ ((v >> 4) & 15).  The XOR
// spreads the stride-2/4/8/16 vector patterns that inner gates with low tile-local targets
// produce over all 16 vector slots of a 256-byte bank row (PMC before: 37-47 % of the LDS
// cycles of the blocked kernel were bank conflicts).
__device__ __forceinline__ unsigned blocked_swz(unsigned v) { return v ^ ((v >> 4) & 15u); }
struct BlockedArg {
  unsigned tb;                          // tile bits
  unsigned apos[kBlockedMaxTileBits];   // their global index positions, ascending (component bits first: 0,1 / 0)
};
struct BlockedGate {
  MfmaRoles ro;      // roles in TILE-LOCAL coordinates (vec position = local bit - #component bits; unused = 31)
  unsigned a_off;    // offset (elements) of this gate's A table
  unsigned kv;       // kbits * 4 + vmask
  unsigned n_addr;   // number of address digits
  unsigned wave_bits;  // bits 0..30: the three tile-local vector bits that carry the WAVE index of this gate's iterations (0: the
                       // three lowest free bits above the slot bits, as round 1-3); bit 31: the gate after this one works on the
                       // same per-wave partition of the tile, so no workgroup barrier is needed between them (host: hq_apply.hip)
};
constexpr unsigned kBlockedNoBarrier = 1u << 31;

template <typename T, int KBITS, int VMASK, int BLOCK>
__device__ __forceinline__ void blocked_inner_gate(T* __restrict__ xr, T* __restrict__ xi,
                                                   const BlockedGate& G, const T* __restrict__ A,
                                                   const unsigned tile_vec_bits) {
  using V = typename Vec<T>::type;
  using Acc = typename Mfma<T>::acc;
  constexpr int CB = Vec<T>::VB, NCOMP = 1 << CB;
  constexpr int NS = KBITS - 2, KV = popc_c(VMASK), NR = NS - KV, NL = 1 << NR;
  constexpr int NRB = 1 << (NS - 2), NCB = 1 << (CB - KV), NSTEP = 1 << NS;
  constexpr int FMASK = ~VMASK & (NCOMP - 1);
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned q = lane >> 4, j = lane & 15;
  const MfmaRoles& ro = G.ro;
  T a[NRB][NSTEP];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) a[rb][s] = A[(rb * NSTEP + s) * 64 + lane];
  // Slot addressing.  The SIMD issues about one instruction per 4 cycles, i.e. 8 per 32-cycle MFMA, and this loop
  // runs only 2 iterations per gate and wave: the address arithmetic IS the budget (PMC before: 4.2 VALU + 1.4 SALU
  // per MFMA, matrix pipe 58 % busy).  Everything is XOR-linear -- the zero-bit deposit moves every index bit on its
  // own, the bank swizzle XORs bits 4..7 into bits 0..3, the digits occupy disjoint bits -- so
  //   address(iteration t, register digit ld) = L ^ S(t) ^ OFF[ld]
  // with L per lane and gate (deposit of wave/slot bits, q digits, plane), S(t) and OFF[ld] wave-uniform:
  // one v_xor per vector and iteration instead of a deposit and a swizzle each.  Byte units throughout.
  auto deposit = [&](unsigned v) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned lo = (1u << ro.pos[m]) - 1;  // unused digits carry 31: no-op
      v = ((v & ~lo) << 1) | (v & lo);
    }
    return v;
  };
  constexpr unsigned WB = BLOCK == 512 ? 3 : 4;  // log2(waves per workgroup)
  static_assert(BLOCK == 64u << WB, "workgroup size");
  const unsigned lane_off = ((q & 1) ? ro.q_off[0] : 0u) | ((q & 2) ? ro.q_off[1] : 0u);
  const unsigned lane_plane = ro.q_plane >= 0 ? ((q >> ro.q_plane) & 1u) : 0u;
  const unsigned L = (blocked_swz(deposit((wave << 4) | j) | lane_off) | (lane_plane << tile_vec_bits)) << 4;
  unsigned OFF[NL];
#pragma unroll
  for (int ld = 0; ld < NL; ++ld) {
    unsigned o = 0;
#pragma unroll
    for (int b = 0; b < NR; ++b)
      if ((ld >> b) & 1) o |= ro.r_off[b];
    const unsigned pl = ro.r_plane >= 0 ? ((ld >> ro.r_plane) & 1u) : 0u;
    OFF[ld] = (blocked_swz(o) | (pl << tile_vec_bits)) << 4;
  }
  unsigned char* const tile = reinterpret_cast<unsigned char*>(xr);  // xi = xr + one plane: the plane is bit tile_vec_bits
  const unsigned niter = (1u << (tile_vec_bits - G.n_addr)) >> 4;  // 16 slots per wave iteration
  for (unsigned t = 0; (t << WB) + wave < niter; ++t) {
    const unsigned Lt = L ^ (blocked_swz(deposit(t << (4 + WB))) << 4);
    unsigned addr[NL];
    V x[NL];
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) {
      addr[ld] = Lt ^ OFF[ld];
      x[ld] = *reinterpret_cast<V*>(tile + addr[ld]);
    }
    // all requests of the iteration stay in front of its first MFMA (left alone the compiler sinks each read to its
    // consumers and waits for it there: see blocked_inner_gate_tab); the 8-vector shape keeps the compiler's order --
    // it has no registers for more
    if constexpr (NL <= 4) __builtin_amdgcn_sched_barrier(0);
    Acc acc[NCB][NRB];
#pragma unroll
    for (int cf = 0; cf < NCB; ++cf)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) acc[cf][rb] = Acc{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int ck = s & ((1 << KV) - 1), ld = s >> KV;
#pragma unroll
      for (int cf = 0; cf < NCB; ++cf) {
        const int comp = pdep_c(ck, VMASK) | pdep_c(cf, FMASK);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
          acc[cf][rb] = Mfma<T>::run(a[rb][s], x[ld][comp], acc[cf][rb]);
      }
    }
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) {
      V y;
#pragma unroll
      for (int comp = 0; comp < NCOMP; ++comp) {
        const int ck = pext_c(comp, VMASK), cf = pext_c(comp, FMASK);
        const int so = ck | (ld << KV);
        y[comp] = acc[cf][so >> 2][so & 3];
      }
      *reinterpret_cast<V*>(tile + addr[ld]) = y;
    }
  }
}

// Table-driven form of blocked_inner_gate (the default: tables of a pass in LDS next to its A operands).
// s_memtime stamps through one gate (tools/blocked_timeline.py) showed where the 42 % idle matrix pipe comes from:
// a SIMD issues roughly one instruction per 4 cycles for ALL its waves, and with only two 16-MFMA bursts per gate
// and wave the ~190 scalar + vector instructions of descriptor decoding and address arithmetic around them
// (x 4 waves) cost as much issue time as the MFMAs cost pipe time -- segments of 15-30 instructions took 900-1300
// cycles.  So every per-gate quantity that does not depend on the data is read from a table the workgroup builds
// ONCE per kernel: address(lane, iteration it, register digit ld) = LANE[lane] ^ ITER[it] ^ OFF[ld] (see the XOR
// argument in blocked_inner_gate), one ds_read_b32 + one v_xor3 per vector.
typedef unsigned BlockedTabT;  // 16-bit entries were tried: more passes fit their tables, each gate 7 % slower
// Layout of a gate's table: LANE[64], ITER[number of wave-iterations a gate can have: 64 for tiles of 2^11 vectors, 128 for
// the 2^12-vector tiles of the 1024-thread kernels], OFF[8].
template <int BLOCK> struct BlockedTab {
  static constexpr unsigned kLane = 0, kIter = 64, kNIter = BLOCK == 1024 ? 128 : 64, kOff = kIter + kNIter, kWords = kOff + 8;
};
constexpr unsigned kBlockedTabWords = BlockedTab<512>::kWords;

// bits of (iteration << 4 | slot) -> the tile-local vector bits that are not address digits of a gate, in ascending
// order; with `wmask` (BlockedGate::wave_bits) the three bits that number the waves (iteration bits 0..2) go to those
// positions instead, so that every gate of a barrier-free group gives wave w the SAME part of the tile
__device__ __forceinline__ unsigned blocked_digits(const MfmaRoles& ro) {
  unsigned digits = 0;
  for (int m = 0; m < 4; ++m)
    if (ro.pos[m] < 31) digits |= 1u << ro.pos[m];
  return digits;
}
__device__ __forceinline__ unsigned blocked_deposit(const unsigned v, const unsigned digits, const unsigned wmask,
                                                    const unsigned tile_vec_bits) {
  unsigned rest = ~(digits | wmask) & ((1u << tile_vec_bits) - 1), out = 0, b = 0;
  if (wmask) {
    for (int i = 0; i < 4 && rest; ++i, ++b) { out |= ((v >> b) & 1u) << __builtin_ctz(rest); rest &= rest - 1; }
    for (unsigned w = wmask; w; w &= w - 1, ++b) out |= ((v >> b) & 1u) << __builtin_ctz(w);
  }
  for (; rest; rest &= rest - 1, ++b) out |= ((v >> b) & 1u) << __builtin_ctz(rest);
  return out;
}

template <typename T, int BLOCK>
__device__ __forceinline__ void blocked_build_tables(BlockedTabT* __restrict__ tabs, const BlockedGate* __restrict__ gates,
                                                     const unsigned ngates, const unsigned tile_vec_bits,
                                                     const unsigned lds_base) {
  // the tile's LDS address is folded into the lane entries: XOR = ADD needs it aligned to the two planes (it is 0:
  // the tile opens the dynamic LDS segment and the kernel has no static one)
  if (lds_base & ((2u << (tile_vec_bits + 4)) - 1)) __builtin_trap();
  const unsigned tid = threadIdx.x;
  for (unsigned g = 0; g < ngates; ++g) {
    const MfmaRoles& ro = gates[g].ro;
    const unsigned wmask = gates[g].wave_bits & ~kBlockedNoBarrier;
    const unsigned digits = blocked_digits(ro);
    auto deposit = [&](unsigned v) { return blocked_deposit(v, digits, wmask, tile_vec_bits); };
    BlockedTabT* tb = tabs + g * BlockedTab<BLOCK>::kWords;
    for (unsigned e = tid; e < BlockedTab<BLOCK>::kWords; e += BLOCK) {
      unsigned val;
      if (e < BlockedTab<BLOCK>::kIter) {  // lane part: slot bits j, q digits, plane
        const unsigned q = e >> 4, j = e & 15;
        const unsigned lane_off = ((q & 1) ? ro.q_off[0] : 0u) | ((q & 2) ? ro.q_off[1] : 0u);
        const unsigned lane_plane = ro.q_plane >= 0 ? ((q >> ro.q_plane) & 1u) : 0u;
        val = ((blocked_swz(deposit(j) | lane_off) | (lane_plane << tile_vec_bits)) << 4) | lds_base;
      } else if (e < BlockedTab<BLOCK>::kOff) {  // wave-iteration part
        val = blocked_swz(deposit((e - BlockedTab<BLOCK>::kIter) << 4)) << 4;
      } else {  // register-digit part
        const unsigned ld = e - BlockedTab<BLOCK>::kOff;
        unsigned o = 0;
        for (int b = 0; b < 3; ++b)
          if ((ld >> b) & 1) o |= ro.r_off[b];
        const unsigned pl = ro.r_plane >= 0 ? ((ld >> ro.r_plane) & 1u) : 0u;
        val = (blocked_swz(o) | (pl << tile_vec_bits)) << 4;
      }
      tb[e] = (BlockedTabT)val;
    }
  }
}

// (BlockedPre, further down: the table words a gate needs first, requested one gate early; BlockedNoPre: read them here)
struct BlockedNoPre {};
template <typename T, int KBITS, int VMASK, int BLOCK, bool PIPE, typename PRE>
__device__ __forceinline__ void blocked_inner_gate_tab(const T* __restrict__ A,
                                                       const BlockedTabT* __restrict__ tab, const unsigned niter, const PRE& P) {
  using V = typename Vec<T>::type;
  using Acc = typename Mfma<T>::acc;
  constexpr int CB = Vec<T>::VB, NCOMP = 1 << CB;
  constexpr int NS = KBITS - 2, KV = popc_c(VMASK), NR = NS - KV, NL = 1 << NR;
  constexpr int NRB = 1 << (NS - 2), NCB = 1 << (CB - KV), NSTEP = 1 << NS;
  constexpr int FMASK = ~VMASK & (NCOMP - 1);
  constexpr unsigned WB = BLOCK == 512 ? 3 : 4;
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  T a[NRB][NSTEP];
  unsigned L;
  unsigned OFF[NL];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) a[rb][s] = A[(rb * NSTEP + s) * 64 + lane];
  constexpr bool kPre = !__is_same(PRE, BlockedNoPre);
  unsigned t_first, t_second;
  if constexpr (kPre) {  // requested one gate early
    L = P.L;
    t_first = P.t0;
    t_second = P.t1;
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) OFF[ld] = ld < 4 ? P.off[ld < 4 ? ld : 0] : tab[BlockedTab<BLOCK>::kOff + ld];
  } else {
    L = tab[BlockedTab<BLOCK>::kLane + lane];
    t_first = tab[BlockedTab<BLOCK>::kIter + wave];
    t_second = tab[BlockedTab<BLOCK>::kIter + wave + (1u << WB)];
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) OFF[ld] = tab[BlockedTab<BLOCK>::kOff + ld];
  }
  typedef __attribute__((address_space(3))) V LdsV;  // addresses are absolute LDS byte addresses (base folded in LANE)
  // multiply-accumulate of one wave-iteration whose vectors are in x, results to the slots they came from (Lt ^ OFF[ld]).
  // Component c of vector ld sits in accumulator block (cf, so >> 2), register so & 3: a 16-byte store wants 4
  // consecutive registers, i.e. a transpose by ~14 v_mov per iteration.
  auto store_results = [&](Acc (&acc)[NCB][NRB], const unsigned Lt) {
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) {
      V y;
#pragma unroll
      for (int comp = 0; comp < NCOMP; ++comp) {
        const int ck = pext_c(comp, VMASK), cf = pext_c(comp, FMASK);
        const int so = ck | (ld << KV);
        y[comp] = acc[cf][so >> 2][so & 3];
      }
      *reinterpret_cast<LdsV*>((uintptr_t)(Lt ^ OFF[ld])) = y;
    }
  };
  if constexpr (!PIPE) {
  // (PIPE = false, HQ_BLOCKED_PIPE=0 at run time: the loop of rounds 2-4a -- the compiler sinks every ds_read_b128 to just
  // in front of the MFMAs that consume it and waits for it there, 4 to 8 exposed LDS latencies per wave-iteration)
  for (unsigned it = wave; it < niter; it += 1u << WB) {
    const unsigned Lt = L ^ tab[BlockedTab<BLOCK>::kIter + it];
    V x[NL];
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) x[ld] = *reinterpret_cast<LdsV*>((uintptr_t)(Lt ^ OFF[ld]));
    Acc acc[NCB][NRB];
#pragma unroll
    for (int cf = 0; cf < NCB; ++cf)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) acc[cf][rb] = Acc{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int ck = s & ((1 << KV) - 1), ld = s >> KV;
#pragma unroll
      for (int cf = 0; cf < NCB; ++cf) {
        const int comp = pdep_c(ck, VMASK) | pdep_c(cf, FMASK);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[cf][rb] = Mfma<T>::run(a[rb][s], x[ld][comp], acc[cf][rb]);
      }
    }
    store_results(acc, Lt);
  }
  } else {
  // LDS reads ahead of the matrix cores (round 4, from the assembly: left to itself the compiler sinks every
  // ds_read_b128 to just in front of the 4-8 MFMAs that consume it, `s_waitcnt lgkmcnt(0)` in between -- a wave then
  // feeds the matrix pipe for 128-256 cycles, waits ~100+ for LDS, feeds it again: the 1276 cycles that the 16 MFMAs
  // (512 cycles of pipe) of one iteration took in round 2's s_memtime timeline).  Now the vectors of the NEXT
  // wave-iteration are requested before the MFMAs of the current one start (two register sets, ping-pong; a
  // scheduling barrier keeps the requests where they are written), so that only the first iteration of a gate waits
  // for LDS; the one shape without registers for a second set (k = 4 without a component target: 8 vectors) runs its
  // requests one vector ahead of the MFMAs inside the iteration.  Same LDS operations, same arithmetic, same order of
  // every accumulation: results are bit-identical to the loop above.
  constexpr unsigned STEP = 1u << WB;
  auto request = [&](V (&x)[NL], const unsigned Lt) {
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) x[ld] = *reinterpret_cast<LdsV*>((uintptr_t)(Lt ^ OFF[ld]));
  };
  auto multiply = [&](V (&x)[NL], const unsigned Lt) {
    Acc acc[NCB][NRB];
#pragma unroll
    for (int cf = 0; cf < NCB; ++cf)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) acc[cf][rb] = Acc{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int ck = s & ((1 << KV) - 1), ld = s >> KV;
#pragma unroll
      for (int cf = 0; cf < NCB; ++cf) {
        const int comp = pdep_c(ck, VMASK) | pdep_c(cf, FMASK);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[cf][rb] = Mfma<T>::run(a[rb][s], x[ld][comp], acc[cf][rb]);
      }
    }
    store_results(acc, Lt);
  };
  // (k = 4: 16 operand + 32 accumulator registers beside the 32 of the tile prefetch leave no room for a second set)
  constexpr bool kTwoSets = NL <= 4 && KBITS == 4;
  if constexpr (kTwoSets) {
    unsigned it = wave;
    if (it < niter) {
      // the table entry of an iteration is read one phase before the requests that need it (its latency used to sit in
      // front of every iteration's first request); the index is clamped, an entry read past the last iteration is unused
      auto entry = [&](const unsigned i) { return tab[BlockedTab<BLOCK>::kIter + (i < niter ? i : wave)]; };
      V x0[NL], x1[NL];
      unsigned Lt0 = L ^ t_first, Lt1 = 0, t_next = t_second;
      request(x0, Lt0);
      // The requests of the next iteration are UNCONDITIONAL: on a path without them the compiler's wait counts for the
      // set being multiplied are those of "nothing requested since" (lgkmcnt counts in order), and merged over both
      // paths the multiply would wait for the requests just issued -- the latency this loop is there to hide.  Past the
      // last iteration every lane requests the same 16 bytes (the head of this gate's table: a broadcast, no bank
      // traffic to speak of) into the set that is never multiplied.
      const unsigned idle = (unsigned)reinterpret_cast<uintptr_t>(tab);
      auto request_next = [&](V (&x)[NL], const unsigned Lt, const bool more) {
#pragma unroll
        for (int ld = 0; ld < NL; ++ld) x[ld] = *reinterpret_cast<LdsV*>((uintptr_t)(more ? (Lt ^ OFF[ld]) : idle));
      };
      for (;;) {
        bool more = it + STEP < niter;  // wave-uniform
        Lt1 = L ^ t_next;
        request_next(x1, Lt1, more);
        t_next = entry(it + 2 * STEP);
        __builtin_amdgcn_sched_barrier(0);
        multiply(x0, Lt0);
        if (!more) break;
        it += STEP;
        more = it + STEP < niter;
        Lt0 = L ^ t_next;
        request_next(x0, Lt0, more);
        t_next = entry(it + 2 * STEP);
        __builtin_amdgcn_sched_barrier(0);
        multiply(x1, Lt1);
        if (!more) break;
        it += STEP;
      }
    }
  } else if constexpr (NL <= 4) {  // all requests of the iteration in front of its first MFMA
    for (unsigned it = wave; it < niter; it += STEP) {
      V x[NL];
      const unsigned Lt = L ^ (it == wave ? t_first : tab[BlockedTab<BLOCK>::kIter + it]);
      request(x, Lt);
      __builtin_amdgcn_sched_barrier(0);
      multiply(x, Lt);
    }
  } else {
    static_assert(KV == 0, "eight vectors per wave-iteration: no component target");
    for (unsigned it = wave; it < niter; it += STEP) {
      const unsigned Lt = L ^ (it == wave ? t_first : tab[BlockedTab<BLOCK>::kIter + it]);
      Acc acc[NCB][NRB];
#pragma unroll
      for (int cf = 0; cf < NCB; ++cf)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[cf][rb] = Acc{0, 0, 0, 0};
      V xa = *reinterpret_cast<LdsV*>((uintptr_t)(Lt ^ OFF[0]));
#pragma unroll
      for (int ld = 0; ld < NL; ++ld) {  // step s = ld
        V xb = xa;
        if (ld + 1 < NL) xb = *reinterpret_cast<LdsV*>((uintptr_t)(Lt ^ OFF[ld + 1]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cf = 0; cf < NCB; ++cf) {
          const int comp = pdep_c(0, VMASK) | pdep_c(cf, FMASK);
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) acc[cf][rb] = Mfma<T>::run(a[rb][ld], xa[comp], acc[cf][rb]);
        }
        __builtin_amdgcn_sched_barrier(0);
        xa = xb;
      }
      store_results(acc, Lt);
    }
  }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Pin a wave-uniform value to SGPRs (the optimiser does not always prove uniformity of loads).
__device__ __forceinline__ float hq_uniform(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
__device__ __forceinline__ double hq_uniform(double x) {
  const uint64_t b = __builtin_bit_cast(uint64_t, x);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// k = 1, 2 inner gates on the VALU: the real-embedded MFMA form needs k_eff = 3, i.e. a 1- or
// 2-qubit gate pays for identity dummies (4x / 2x the matrix-core time); a register butterfly
// on the LDS tile costs 2^k complex MACs per amplitude and the same LDS traffic.  A lane owns
// the 2^KR partner vectors of both planes (KR = targets that are not vector components);
// U (planar, ascending target order, 2 * 4^K elements at A, always in GLOBAL memory) is read
// with uniform addresses: scalar loads, the matrix lives in SGPRs.
template <typename T, int K, int VMASK, int BLOCK>
__device__ __forceinline__ void blocked_inner_gate_valu(T* __restrict__ xr, T* __restrict__ xi,
                                                        const BlockedGate& G, const T* __restrict__ A,
                                                        const unsigned tile_vec_bits) {
  using V = typename Vec<T>::type;
  constexpr int VB = Vec<T>::VB, VE = 1 << VB;
  constexpr int KV = popc_c(VMASK), KR = K - KV, R = 1 << KR, D = 1 << K;
  T ur[D * D], ui[D * D];
#pragma unroll
  for (int e = 0; e < D * D; ++e) { ur[e] = hq_uniform(A[e]); ui[e] = hq_uniform(A[D * D + e]); }
  unsigned off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned o = 0;
#pragma unroll
    for (int jj = 0; jj < KR; ++jj) o |= (unsigned)((r >> jj) & 1) << G.ro.pos[jj];
    off[r] = o;
  }
  const unsigned nfree = 1u << (tile_vec_bits - KR);
#pragma unroll 1
  for (unsigned v0 = threadIdx.x; v0 < nfree; v0 += BLOCK) {
    unsigned v = v0;
#pragma unroll
    for (int jj = 0; jj < KR; ++jj) {
      const unsigned lo = (1u << G.ro.pos[jj]) - 1;
      v = ((v & ~lo) << 1) | (v & lo);
    }
    V pr[R], pi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      pr[r] = reinterpret_cast<V*>(xr)[blocked_swz(v | off[r])];
      pi[r] = reinterpret_cast<V*>(xi)[blocked_swz(v | off[r])];
    }
#pragma unroll
    for (int ro = 0; ro < R; ++ro) {
      V yr, yi;
#pragma unroll
      for (int co = 0; co < VE; ++co) {
        const int to = pext_c(co, VMASK) | (ro << KV);
        const int cfree = co & ~VMASK;
        T ar = 0, ai = 0;
#pragma unroll
        for (int ti = 0; ti < D; ++ti) {
          const int ci = pdep_c(ti & ((1 << KV) - 1), VMASK) | cfree;
          const int ri = ti >> KV;
          ar = hq_fma(ur[to * D + ti], pr[ri][ci], ar);
          ar = hq_fma(-ui[to * D + ti], pi[ri][ci], ar);
          ai = hq_fma(ur[to * D + ti], pi[ri][ci], ai);
          ai = hq_fma(ui[to * D + ti], pr[ri][ci], ai);
        }
        yr[co] = ar;
        yi[co] = ai;
      }
      reinterpret_cast<V*>(xr)[blocked_swz(v | off[ro])] = yr;
      reinterpret_cast<V*>(xi)[blocked_swz(v | off[ro])] = yi;
    }
  }
}

// The four descriptor words of a gate that the gate loop itself needs (BlockedGate::a_off .. wave_bits: one 16-byte scalar
// load).  The loop requests the NEXT gate's descriptor before it runs the current gate: read at the top of a gate, the
// scalar-cache round trip and the kind dispatch behind it stood in front of every gate's first LDS request.
struct BlockedDesc {
  unsigned a_off, kv, n_addr, wave_bits;
};
__device__ __forceinline__ BlockedDesc blocked_desc(const BlockedGate* __restrict__ gates, const unsigned gi) {
  const BlockedGate& G = gates[gi];
  return BlockedDesc{G.a_off, G.kv, G.n_addr, G.wave_bits};
}

// The table words every matrix-core gate needs before it can request its first vectors -- the lane part, the entries of
// this wave's first two iterations and the first four register-digit offsets -- requested from LDS one gate EARLY (at the
// top of the previous gate, next to the descriptor): a gate then starts with its first vector requests instead of with an
// LDS round trip for their addresses.  (Entries past a gate's last iteration are read but unused: wave + STEP is always
// inside the ITER part of the table.)
struct BlockedPre {
  unsigned L, t0, t1, off[4];
};
template <int BLOCK, typename PRE>
__device__ __forceinline__ PRE blocked_pre(const BlockedTabT* __restrict__ tabs, const unsigned gi) {
  if constexpr (__is_same(PRE, BlockedNoPre)) return BlockedNoPre{};
  else {
  constexpr unsigned WB = BLOCK == 512 ? 3 : 4;
  const BlockedTabT* __restrict__ tab = tabs + gi * BlockedTab<BLOCK>::kWords;
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  BlockedPre P;
  P.L = tab[BlockedTab<BLOCK>::kLane + lane];
  P.t0 = tab[BlockedTab<BLOCK>::kIter + wave];
  P.t1 = tab[BlockedTab<BLOCK>::kIter + wave + (1u << WB)];
#pragma unroll
  for (int ld = 0; ld < 4; ++ld) P.off[ld] = tab[BlockedTab<BLOCK>::kOff + ld];
  return P;
  }
}

// One inner gate of a pass by its kind (G.kv: KBITS * 4 + VMASK for the matrix-core form, 64 + k * 4 + VMASK for the
// register butterflies).
template <typename T, int BLOCK, bool ALDS, bool PIPE, typename PRE>
__device__ __forceinline__ void blocked_dispatch_gate(const BlockedGate& G, const BlockedDesc D, const PRE& P, const unsigned gi,
                                                      T* __restrict__ xr, T* __restrict__ xi, const T* __restrict__ als,
                                                      const T* __restrict__ Atab, const BlockedTabT* __restrict__ tabs,
                                                      const unsigned tvb) {
  constexpr unsigned CB = Vec<T>::VB;
  const T* A = ALDS ? als + D.a_off : Atab + D.a_off;
#define HQ_BLOCKED_MFMA_GATE(KB, VM)                                                                    \
  do {                                                                                                  \
    if constexpr (ALDS)                                                                                 \
      blocked_inner_gate_tab<T, KB, VM, BLOCK, PIPE>(A, tabs + gi * BlockedTab<BLOCK>::kWords, (1u << (tvb - D.n_addr)) >> 4, P); \
    else                                                                                                \
      blocked_inner_gate<T, KB, VM, BLOCK>(xr, xi, G, A, tvb);                                          \
  } while (0)
  switch (D.kv) {
    case 16: HQ_BLOCKED_MFMA_GATE(4, 0); break;
    case 17: HQ_BLOCKED_MFMA_GATE(4, 1); break;
    case 20: HQ_BLOCKED_MFMA_GATE(5, 0); break;
    case 21: HQ_BLOCKED_MFMA_GATE(5, 1); break;
    case 64 + 4 + 0: blocked_inner_gate_valu<T, 1, 0, BLOCK>(xr, xi, G, Atab + D.a_off, tvb); break;
    case 64 + 4 + 1: blocked_inner_gate_valu<T, 1, 1, BLOCK>(xr, xi, G, Atab + D.a_off, tvb); break;
    case 64 + 8 + 0: blocked_inner_gate_valu<T, 2, 0, BLOCK>(xr, xi, G, Atab + D.a_off, tvb); break;
    case 64 + 8 + 1: blocked_inner_gate_valu<T, 2, 1, BLOCK>(xr, xi, G, Atab + D.a_off, tvb); break;
    default:
      if constexpr (CB == 2) {
        switch (D.kv) {
          case 18: HQ_BLOCKED_MFMA_GATE(4, 2); break;
          case 19: HQ_BLOCKED_MFMA_GATE(4, 3); break;
          case 22: HQ_BLOCKED_MFMA_GATE(5, 2); break;
          case 23: HQ_BLOCKED_MFMA_GATE(5, 3); break;
          case 64 + 4 + 2: blocked_inner_gate_valu<T, 1, 2, BLOCK>(xr, xi, G, Atab + D.a_off, tvb); break;
          case 64 + 8 + 2: blocked_inner_gate_valu<T, 2, 2, BLOCK>(xr, xi, G, Atab + D.a_off, tvb); break;
          case 64 + 8 + 3: blocked_inner_gate_valu<T, 2, 3, BLOCK>(xr, xi, G, Atab + D.a_off, tvb); break;
          default: break;
        }
      }
      break;
  }
#undef HQ_BLOCKED_MFMA_GATE
}

// ALDS: the A-operand tables of all gates of the pass (a_elems elements) are staged once per
// (persistent) workgroup in LDS behind the tile; a table read from global memory puts an L2 round
// trip (~1500 clk, as long as the gate's MFMAs) in front of every gate of every tile.
// PREF (tiles of exactly 4 * BLOCK vectors per plane): serial phases -- load a tile (one HBM round trip), run the
// gates, store -- run in step on the whole chip, so HBM idles while the gates run and the matrix cores idle while
// tiles move: a pass costs HBM time PLUS gate time.  With PREF the next tile's vectors are requested into registers
// before the gates of the current tile start and dropped into LDS after its stores were issued.  Needs the
// no-scratch register budget: a scratch reload is a vector-memory load and would queue (vmcnt is in order) behind
// the prefetch it was supposed to overlap.
// PIPE (with ALDS; HQ_BLOCKED_PIPE, default 1): the inner gates request their LDS vectors one wave-iteration ahead of the
// matrix cores and the table words of the next gate one gate early (blocked_inner_gate_tab, BlockedPre); false = the
// loops of rounds 2-4a, kept as a run-time alternative so that one lease can time and bisect both.
template <typename T, int BLOCK, bool ALDS, bool PREF, bool PIPE>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4)))
apply_blocked_kernel(T* __restrict__ re, T* __restrict__ im, const BlockedGate* __restrict__ gates,
                     const unsigned ngates, const T* __restrict__ Atab, const unsigned a_elems,
                     const BlockedArg ba, const uint64_t ntiles) {
  using V = typename Vec<T>::type;
  constexpr unsigned CB = Vec<T>::VB;
  HQ_DYN_LDS(smem);
  T* xr = reinterpret_cast<T*>(smem);
  T* xi = xr + (1u << ba.tb);
  T* als = xi + (1u << ba.tb);
  const unsigned tid = threadIdx.x;
  const unsigned tvb = ba.tb - CB, nvec = 1u << tvb;
  BlockedTabT* const tabs = reinterpret_cast<BlockedTabT*>(als + a_elems);  // ALDS: address tables of all gates (built here)
  if (ALDS) {
    for (unsigned i = tid; i < a_elems; i += BLOCK) als[i] = Atab[i];
    blocked_build_tables<T, BLOCK>(tabs, gates, ngates, tvb, (unsigned)reinterpret_cast<uintptr_t>(xr));
    __syncthreads();
  }
  V* __restrict__ vre = reinterpret_cast<V*>(re);
  V* __restrict__ vim = reinterpret_cast<V*>(im);
  constexpr unsigned NPV = 4;  // PREF: vectors per thread and plane
  // constant trip count: the positions are read from the kernel arguments once (a runtime loop re-fetches
  // ba.apos[m] with a scalar load + wait per digit, twice per tile, in every wave)
  auto tile_base = [&](uint64_t tile) {
    uint64_t base = tile;  // in 16-byte vector units: tile positions minus the component bits
#pragma unroll
    for (unsigned m = CB; m < (PREF ? CB + (BLOCK == 1024 ? 12u : 11u) : (unsigned)kBlockedMaxTileBits); ++m) {  // PREF: exactly 4 * BLOCK vectors
      const uint64_t lo = (PREF || m < ba.tb) ? (1ull << (ba.apos[m] - CB)) - 1 : ~0ull;  // ~0: no-op
      base = ((base & ~lo) << 1) | (base & lo);
    }
    return base;
  };
  auto vec_off = [&](unsigned e) {  // OR-linear in e
    uint64_t g = 0;
    for (unsigned m = CB; m < ba.tb; ++m) g |= (uint64_t)((e >> (m - CB)) & 1u) << (ba.apos[m] - CB);
    return g;
  };
  V pr[PREF ? NPV : 1], pi[PREF ? NPV : 1];
  const uint64_t off_tid = vec_off(tid);
  uint64_t off_blk[NPV];  // wave-uniform
#pragma unroll
  for (unsigned i = 0; i < NPV; ++i) off_blk[i] = vec_off(i * BLOCK);
  // unconditional (callers clamp the tile): a conditional request merges "new" and "old" register values and the
  // compiler then copies every vector right after its load, i.e. waits for HBM on the spot
  // (tile base | uniform offset) is pinned to scalar registers: left alone the compiler hoists off_tid | off_blk[i]
  // out of the tile loop -- 8 more vector registers alive across the gates, i.e. spills inside the loop
  auto prefetch = [&](const uint64_t b) {  // b = tile_base(tile)
#pragma unroll
    for (unsigned i = 0; i < (PREF ? NPV : 1); ++i) {
      uint64_t sb = b | off_blk[i];
      HQ_PIN_SGPR(sb);
      pr[i] = __builtin_nontemporal_load(vre + (sb | off_tid));
      pi[i] = __builtin_nontemporal_load(vim + (sb | off_tid));
    }
  };
  const unsigned fs = blocked_swz(tid);  // the swizzle only touches bits 0..3: swz(tid + i * BLOCK) = fs + i * BLOCK
  auto fill = [&]() {
#pragma unroll
    for (unsigned i = 0; i < (PREF ? NPV : 1); ++i) {
      reinterpret_cast<V*>(xr)[fs + i * BLOCK] = pr[i];
      reinterpret_cast<V*>(xi)[fs + i * BLOCK] = pi[i];
    }
  };
  const uint64_t stride = gridDim.x;
  // PREF walks its tiles by `stride` (a power of two: min(ntiles, 512)): in deposited coordinates that is
  // next = ((cur | ~M) + D) & M with M = the index bits outside the tile and D = deposit(stride) -- the carry runs
  // through the filled tile bits -- three 64-bit scalar operations instead of an 11-digit deposit twice per tile
  // (70 scalar instructions with spilled masks each)
  const uint64_t dep_mask = tile_base(~0ull), dep_stride = tile_base(stride);
  auto next_base = [&](uint64_t b) { return ((b | ~dep_mask) + dep_stride) & dep_mask; };
  if constexpr (PREF) {
    if (blockIdx.x >= ntiles) return;
    // the tile is filled at the END of the loop body, right after the stores of the previous tile were issued: on
    // every path the wait for the prefetched vectors then sees "8 loads, then 8 stores" in the (in-order) vmcnt
    // queue and does not drain the stores
    {
      const uint64_t b = tile_base(blockIdx.x) | off_tid;  // first tile: straight into LDS, one vector pair at a time
#pragma unroll 1
      for (unsigned i = 0; i < NPV; ++i) {
        const uint64_t g = b | vec_off(i * BLOCK);
        reinterpret_cast<V*>(xr)[fs + i * BLOCK] = __builtin_nontemporal_load(vre + g);
        reinterpret_cast<V*>(xi)[fs + i * BLOCK] = __builtin_nontemporal_load(vim + g);
      }
    }
    prefetch(blockIdx.x + stride < ntiles ? tile_base(blockIdx.x + stride) : tile_base(blockIdx.x));
  }
  uint64_t base_cur = tile_base(blockIdx.x);
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += stride) {
    const uint64_t base = PREF ? base_cur : tile_base(tile);
    if constexpr (!PREF) {
      for (unsigned e = tid; e < nvec; e += BLOCK) {
        const uint64_t g = base | vec_off(e);
        reinterpret_cast<V*>(xr)[blocked_swz(e)] = __builtin_nontemporal_load(vre + g);
        reinterpret_cast<V*>(xi)[blocked_swz(e)] = __builtin_nontemporal_load(vim + g);
      }
    }
    __syncthreads();
    BlockedDesc D = blocked_desc(gates, 0);
    // (complex128: no registers left beside the tile prefetch for the words of the next gate)
    using Pre = typename std::conditional<ALDS && PIPE && sizeof(T) == 4, BlockedPre, BlockedNoPre>::type;
    Pre P = blocked_pre<BLOCK, Pre>(tabs, 0);
    for (unsigned gi = 0; gi < ngates; ++gi) {
      const BlockedDesc Dn = blocked_desc(gates, gi + 1 < ngates ? gi + 1 : gi);  // in flight while this gate runs
      const Pre Pn = blocked_pre<BLOCK, Pre>(tabs, gi + 1 < ngates ? gi + 1 : gi);
      blocked_dispatch_gate<T, BLOCK, ALDS, PIPE>(gates[gi], D, P, gi, xr, xi, als, Atab, tabs, tvb);
      P = Pn;
      // gates of one barrier-free group touch, wave by wave, the same part of the tile (same `wave_bits`): a wave only
      // needs its OWN stores to have landed (LDS operations of a wave complete in order; the gate ends with lgkmcnt(0))
      if (!(ALDS && (D.wave_bits & kBlockedNoBarrier))) __syncthreads();
      D = Dn;
    }
    if constexpr (PREF) {
      V sr[NPV], si[NPV];  // all LDS reads in flight before the first store (the gates' registers are free here)
#pragma unroll
      for (unsigned i = 0; i < NPV; ++i) {
        sr[i] = reinterpret_cast<V*>(xr)[fs + i * BLOCK];
        si[i] = reinterpret_cast<V*>(xi)[fs + i * BLOCK];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (unsigned i = 0; i < NPV; ++i) {
        uint64_t sb = base | off_blk[i];
        HQ_PIN_SGPR(sb);
        const uint64_t g = sb | off_tid;
        __builtin_nontemporal_store(sr[i], vre + g);
        __builtin_nontemporal_store(si[i], vim + g);
      }
    } else {
      for (unsigned e = tid; e < nvec; e += BLOCK) {
        const uint64_t g = base | vec_off(e);
        __builtin_nontemporal_store(reinterpret_cast<V*>(xr)[blocked_swz(e)], vre + g);
        __builtin_nontemporal_store(reinterpret_cast<V*>(xi)[blocked_swz(e)], vim + g);
      }
    }
    // no barrier between the store phase and the fill with PREF: a thread refills exactly the LDS slots it has just
    // read for its stores (fs + i * BLOCK both times), in its own program order
    if constexpr (!PREF) __syncthreads();
    if constexpr (PREF) {
      fill();  // tile + stride (a repeat of a finished tile past the end: never used)
      base_cur = next_base(base);
      prefetch(tile + 2 * stride < ntiles ? next_base(base_cur) : base);
    }
  }
}

// ---------------------------------------------------------------------------------
// apply_blocked_direct_kernel: the cache-blocked pass with the tile movement folded into its FIRST gate (round 4;
// opt-in through HQ_BLOCKED_DIRECT until it has been measured).
//
// apply_blocked_kernel<.., PREF> moves a tile HBM -> registers -> LDS (fill), runs the gates LDS -> registers -> LDS and
// streams it back LDS -> registers -> HBM (store phase): per tile that is two LDS passes, two workgroup barriers and a
// BURST of 8 stores + 8 loads per lane issued into a memory system that is already saturated -- the waves stall at
// issue and the matrix cores see only the other workgroup of the CU meanwhile (~1.3 ms of every 4.9 ms pass at
// n = 30).  Here the first gate of the pass (a matrix-core gate: KBITS = 4 for k <= 3, KBITS = 5 for k = 4) does the
// movement in ITS OWN addressing:
//   * the prefetch requests the next tile's vectors from HBM as that gate's B operands (lane (q, j), wave-iteration,
//     register digit -> global address through two 64-bit tables built once per kernel: GLANE[lane] ^ GWAVE[wave][i]);
//   * the gate multiplies straight from the prefetch registers and writes its results into the LDS tile -- no fill pass;
//   * just before it overwrites a part of the LDS tile, the wave reads what is there -- the finished amplitudes of the
//     PREVIOUS tile -- and stores them to HBM through the same tables: the stores trickle out between the gate's
//     MFMA groups, wave by wave, instead of in one burst, and the store phase with its barrier is gone.
// Every wave-level access stays a set of whole 128-byte lines (the host only takes a first gate whose register digits
// lie above tile-local vector bit 2; q digits and slot bits fill the lines).  The other gates of the pass are the ones
// of apply_blocked_kernel, barrier-free groups included; the last tile of a workgroup leaves through a linear store.
// ---------------------------------------------------------------------------------
constexpr unsigned kBlockedGTabLane = 0, kBlockedGTabWave = 64;  // 64-bit words: GLANE[64], GWAVE[waves][8]
template <int BLOCK> constexpr unsigned blocked_gtab_words() { return kBlockedGTabWave + (BLOCK / 64) * 8; }
constexpr unsigned kBlockedGTabWords = blocked_gtab_words<512>();
constexpr uint64_t kBlockedPlaneBit = 1ull << 63;  // of a table entry: the vector lives in the imaginary plane
// This lane's entry of the GLANE part.  The lane index is recomputed HERE from the thread index, opaquely: one shared
// address register kept alive across all the gates of a tile is what the complex128 kernel (128 registers, 32 of them the
// prefetch) spilled -- and a scratch reload is a vector-memory load that queues behind the prefetch (vmcnt is in order).
__device__ __forceinline__ uint64_t blocked_gtab_lane(const uint64_t* __restrict__ gt) {
  unsigned l = threadIdx.x & 63u;
#ifndef HQ_ASAN
  asm volatile("" : "+v"(l));
#endif
  return gt[kBlockedGTabLane + l];
}

template <typename T, int BLOCK>
__device__ __forceinline__ void blocked_build_direct_tables(uint64_t* __restrict__ gt, const BlockedGate& G,
                                                            const BlockedArg& ba, const unsigned tile_vec_bits) {
  constexpr unsigned CB = Vec<T>::VB;
  constexpr unsigned WB = BLOCK == 512 ? 3 : 4;
  const MfmaRoles& ro = G.ro;
  const unsigned digits = blocked_digits(ro), wmask = G.wave_bits & ~kBlockedNoBarrier;
  const unsigned nr = (G.kv >> 2) - 2u - (unsigned)__builtin_popcount(G.kv & 3u);  // register digits of the first gate (KBITS = kv >> 2)
  auto vec_off = [&](unsigned e) {  // tile-local vector index -> global vector offset (OR-linear)
    uint64_t g = 0;
    for (unsigned m = CB; m < ba.tb; ++m) g |= (uint64_t)((e >> (m - CB)) & 1u) << (ba.apos[m] - CB);
    return g;
  };
  for (unsigned e = threadIdx.x; e < blocked_gtab_words<BLOCK>(); e += BLOCK) {
    uint64_t val;
    if (e < kBlockedGTabWave) {  // lane part: slot bits j, q digits, plane
      const unsigned q = e >> 4, j = e & 15;
      const unsigned lane_off = ((q & 1) ? ro.q_off[0] : 0u) | ((q & 2) ? ro.q_off[1] : 0u);
      const unsigned lane_plane = ro.q_plane >= 0 ? ((q >> ro.q_plane) & 1u) : 0u;
      val = vec_off(blocked_deposit(j, digits, wmask, tile_vec_bits) | lane_off) | (lane_plane ? kBlockedPlaneBit : 0ull);
    } else {  // wave w, prefetch register i = (local iteration, register digit)
      const unsigned w = (e - kBlockedGTabWave) >> 3, i = (e - kBlockedGTabWave) & 7;
      const unsigned it = w + ((i >> nr) << WB), ld = i & ((1u << nr) - 1);
      unsigned o = 0;
      for (unsigned b = 0; b < nr; ++b)
        if ((ld >> b) & 1) o |= ro.r_off[b];
      const unsigned pl = ro.r_plane >= 0 ? ((ld >> ro.r_plane) & 1u) : 0u;
      val = vec_off(blocked_deposit(it << 4, digits, wmask, tile_vec_bits) | o) | (pl ? kBlockedPlaneBit : 0ull);
    }
    gt[e] = val;
  }
}

// The first gate of a direct pass: blocked_inner_gate_tab<T, KBITS, VMASK, BLOCK> with its B operands in `pf` and the
// store-out of the previous tile in front of every overwrite.  KBITS = 4 (k <= 3) or 5 (k = 4, round 5).
//
// A wave owns 8 vectors per lane of the tile (both planes): NITL wave-iterations of NL vectors.  The work is cut into
// HALVES of NLH = 4 >> KV result vectors -- one output row block of one wave-iteration each (KBITS = 4: NRB = 1, a half
// IS an iteration; KBITS = 5: NRB = 2 row blocks per iteration).  A half multiplies ALL NL prefetched vectors of its
// iteration by its row block of the operand table (the inputs live in registers, so the LDS slots of the tile are free
// to be overwritten half by half), and its results are exactly the vectors ld = rb * NLH .. rb * NLH + NLH - 1 of the
// iteration: only one row block of accumulators is live at a time (k = 4 without a component target: 16 instead of 32
// registers beside the 32 of the prefetch), and the store-out moves NLH vectors at a time.
template <typename T, int KBITS, int VMASK, int BLOCK>
__device__ __forceinline__ void blocked_gate0_direct(const T* __restrict__ A, const BlockedTabT* __restrict__ tab,
                                                     const uint64_t* __restrict__ gt,
                                                     typename Vec<T>::type (&pf)[8], const bool have_prev,
                                                     const uint64_t base_prev,
                                                     typename Vec<T>::type* __restrict__ vre,
                                                     typename Vec<T>::type* __restrict__ vim) {
  using V = typename Vec<T>::type;
  using Acc = typename Mfma<T>::acc;
  static_assert(KBITS == 4 || KBITS == 5, "first gates of a direct pass: k <= 4");
  constexpr int CB = Vec<T>::VB, NCOMP = 1 << CB;
  constexpr int NS = KBITS - 2, KV = popc_c(VMASK), NR = NS - KV, NL = 1 << NR;
  constexpr int NRB = 1 << (NS - 2), NCB = 1 << (CB - KV), NSTEP = 1 << NS, NITL = 8 / NL;
  constexpr int NLH = NL / NRB, NH = NITL * NRB;  // result vectors per half, halves per wave
  constexpr int FMASK = ~VMASK & (NCOMP - 1);
  static_assert(NLH * NH == 8 && NLH == (4 >> KV), "eight vectors per lane");
  constexpr unsigned WB = BLOCK == 512 ? 3 : 4;
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // the operand rows of both row blocks up front (KBITS = 5, f32: 16 registers) -- or one row block at a time, re-read at
  // the top of each half (f64: 2 x 16 registers do not fit beside the prefetch; the read hides behind 32 f64 MFMAs)
  constexpr bool kRowsUpFront = NRB == 1 || sizeof(T) == 4;
  T a[kRowsUpFront ? NRB : 1][NSTEP];
  if constexpr (kRowsUpFront) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) a[rb][s] = A[(rb * NSTEP + s) * 64 + lane];
  }
  const unsigned L = tab[BlockedTab<BLOCK>::kLane + lane];
  unsigned OFF[NL];
#pragma unroll
  for (int ld = 0; ld < NL; ++ld) OFF[ld] = tab[BlockedTab<BLOCK>::kOff + ld];
  const uint64_t gl = blocked_gtab_lane(gt);
  typedef __attribute__((address_space(3))) V LdsV;
  // The store-out reads run one half AHEAD of the MFMAs (left in program order -- read, store, multiply -- every
  // half had an LDS round trip in front of its MFMAs; see blocked_inner_gate_tab): the slots of half h + 1 are
  // requested before the MFMAs of half h, the stores of half h are issued behind its MFMAs.
  unsigned Ltv[NITL];
#pragma unroll
  for (int itl = 0; itl < NITL; ++itl) Ltv[itl] = L ^ tab[BlockedTab<BLOCK>::kIter + wave + ((unsigned)itl << WB)];
  // (complex128: no registers for a second set -- the slots of half h are requested in front of ITS OWN MFMAs, which
  // still hides the round trip: the data is first needed by the stores behind them)
  constexpr int TD = sizeof(T) == 8 ? 1 : 2;
  V t[TD][NLH];
  auto request_out = [&](V (&dst)[NLH], const int h) {  // h = itl * NRB + rb
#pragma unroll
    for (int l = 0; l < NLH; ++l) dst[l] = *reinterpret_cast<LdsV*>((uintptr_t)(Ltv[h / NRB] ^ OFF[(h % NRB) * NLH + l]));
  };
  if (TD == 2 && have_prev) request_out(t[0], 0);  // uniform
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int itl = h / NRB, rb = h % NRB;
    const unsigned Lt = Ltv[itl];
    if constexpr (TD == 2) {
      if (have_prev && h + 1 < NH) request_out(t[(h + 1) & 1], h + 1);
    } else {
      if (have_prev) request_out(t[0], h);
    }
    if constexpr (!kRowsUpFront) {
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) a[0][s] = A[(rb * NSTEP + s) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    Acc acc[NCB];
#pragma unroll
    for (int cf = 0; cf < NCB; ++cf) acc[cf] = Acc{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int ck = s & ((1 << KV) - 1), ld = s >> KV;
#pragma unroll
      for (int cf = 0; cf < NCB; ++cf) {
        const int comp = pdep_c(ck, VMASK) | pdep_c(cf, FMASK);
        acc[cf] = Mfma<T>::run(a[kRowsUpFront ? rb : 0][s], pf[itl * NL + ld][comp], acc[cf]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (have_prev) {  // the finished amplitudes of the previous tile leave from the slots this half overwrites
#pragma unroll
      for (int l = 0; l < NLH; ++l) {
        const uint64_t o = gl ^ gt[kBlockedGTabWave + wave * 8 + itl * NL + rb * NLH + l];
        V* const p = (o & kBlockedPlaneBit) ? vim : vre;
        __builtin_nontemporal_store(t[h & (TD - 1)][l], p + (base_prev | (o & ~kBlockedPlaneBit)));
      }
    }
#pragma unroll
    for (int l = 0; l < NLH; ++l) {
      const int ld = rb * NLH + l;  // so = ck | (ld << KV) lies in row block rb: so >> 2 == rb
      V y;
#pragma unroll
      for (int comp = 0; comp < NCOMP; ++comp) {
        const int ck = pext_c(comp, VMASK), cf = pext_c(comp, FMASK);
        const int so = ck | (ld << KV);
        y[comp] = acc[cf][so & 3];
      }
      *reinterpret_cast<LdsV*>((uintptr_t)(Lt ^ OFF[ld])) = y;
    }
  }
  // the prefetch registers stay allocated to the end of the gate: were the store-out data of a later half to reuse
  // them, the next prefetch (which overwrites them right after this gate) would have to wait for those stores to drain
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(pf[i]));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <typename T, int BLOCK>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4)))
apply_blocked_direct_kernel(T* __restrict__ re, T* __restrict__ im, const BlockedGate* __restrict__ gates,
                            const unsigned ngates, const T* __restrict__ Atab, const unsigned a_elems,
                            const BlockedArg ba, const uint64_t ntiles) {
  using V = typename Vec<T>::type;
  constexpr unsigned CB = Vec<T>::VB;
  HQ_DYN_LDS(smem);
  T* xr = reinterpret_cast<T*>(smem);
  T* xi = xr + (1u << ba.tb);
  T* als = xi + (1u << ba.tb);
  const unsigned tid = threadIdx.x;
  const unsigned tvb = ba.tb - CB, nvec = 1u << tvb;  // host: nvec == 4 * BLOCK
  BlockedTabT* const tabs = reinterpret_cast<BlockedTabT*>(als + a_elems);
  // host: the address tables end on a 16-byte boundary
  uint64_t* const gt = reinterpret_cast<uint64_t*>(tabs + ((ngates * BlockedTab<BLOCK>::kWords + 3u) & ~3u));
  for (unsigned i = tid; i < a_elems; i += BLOCK) als[i] = Atab[i];
  blocked_build_tables<T, BLOCK>(tabs, gates, ngates, tvb, (unsigned)reinterpret_cast<uintptr_t>(xr));
  blocked_build_direct_tables<T, BLOCK>(gt, gates[0], ba, tvb);
  __syncthreads();
  if (blockIdx.x >= ntiles) return;
  V* __restrict__ vre = reinterpret_cast<V*>(re);
  V* __restrict__ vim = reinterpret_cast<V*>(im);
  auto tile_base = [&](uint64_t tile) {  // in 16-byte vector units: exactly log2(4 * BLOCK) vector bits inside the tile
    uint64_t base = tile;
#pragma unroll
    for (unsigned m = CB; m < CB + (BLOCK == 1024 ? 12u : 11u); ++m) {
      const uint64_t lo = (1ull << (ba.apos[m] - CB)) - 1;
      base = ((base & ~lo) << 1) | (base & lo);
    }
    return base;
  };
  const unsigned lane = tid & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  V pf[8];
  auto prefetch = [&](uint64_t b) {  // unconditional (see apply_blocked_kernel); b = tile_base(tile), wave-uniform
    HQ_PIN_SGPR(b);
    const uint64_t gl = blocked_gtab_lane(gt);
#pragma unroll
    for (unsigned i = 0; i < 8; ++i) {
      const uint64_t o = gl ^ gt[kBlockedGTabWave + wave * 8 + i];
      const V* const p = (o & kBlockedPlaneBit) ? vim : vre;
      pf[i] = __builtin_nontemporal_load(p + (b | (o & ~kBlockedPlaneBit)));
    }
  };
  const uint64_t stride = gridDim.x;  // a power of two (host)
  const uint64_t dep_mask = tile_base(~0ull), dep_stride = tile_base(stride);
  auto next_base = [&](uint64_t b) { return ((b | ~dep_mask) + dep_stride) & dep_mask; };
  const BlockedGate& G0 = gates[0];
  const T* const A0 = als + G0.a_off;
  uint64_t base = tile_base(blockIdx.x), base_prev = 0;
  bool have_prev = false;
  prefetch(base);
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += stride) {
    // every wave is past the last gate of the previous tile here (that gate ends with a workgroup barrier)
    // The prefetched operands were requested a whole tile ago: one wait for ALL of them, here and on every path, costs
    // nothing -- and keeps the compiler from placing its own waits for the individual registers further down, behind the
    // first gate's stores (vmcnt counts loads and stores in order: on the path without stores -- the first tile -- the wait
    // for the last prefetched register is vmcnt(0), and merged over both paths that drained the eight stores just issued;
    // a wait in front of the NEXT prefetch, for the old contents of its registers, would do the same).  A real S_WAITCNT
    // (not inline assembly), so that the compiler's wait-count insertion sees the queue empty.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt, lgkmcnt untouched
#define HQ_GATE0(KB, VM) blocked_gate0_direct<T, KB, VM, BLOCK>(A0, tabs, gt, pf, have_prev, base_prev, vre, vim)
    switch (G0.kv) {
      case 16: HQ_GATE0(4, 0); break;
      case 17: HQ_GATE0(4, 1); break;
      case 20: HQ_GATE0(5, 0); break;
      case 21: HQ_GATE0(5, 1); break;
      default:
        if constexpr (CB == 2) {
          switch (G0.kv) {
            case 18: HQ_GATE0(4, 2); break;
            case 19: HQ_GATE0(4, 3); break;
            case 22: HQ_GATE0(5, 2); break;
            default: HQ_GATE0(5, 3); break;
          }
        }
        break;
    }
#undef HQ_GATE0
    const uint64_t nb = next_base(base);
    prefetch(tile + stride < ntiles ? nb : base);  // past the end: a repeat of this tile, never used
    if (!(G0.wave_bits & kBlockedNoBarrier)) __syncthreads();
    BlockedDesc D = blocked_desc(gates, 1);  // (the host takes passes of at least two gates)
    using Pre = typename std::conditional<sizeof(T) == 4, BlockedPre, BlockedNoPre>::type;
    Pre P = blocked_pre<BLOCK, Pre>(tabs, 1);
    for (unsigned gi = 1; gi < ngates; ++gi) {
      const BlockedDesc Dn = blocked_desc(gates, gi + 1 < ngates ? gi + 1 : gi);
      const Pre Pn = blocked_pre<BLOCK, Pre>(tabs, gi + 1 < ngates ? gi + 1 : gi);
      blocked_dispatch_gate<T, BLOCK, true, true>(gates[gi], D, P, gi, xr, xi, als, Atab, tabs, tvb);
      P = Pn;
      if (!(D.wave_bits & kBlockedNoBarrier)) __syncthreads();
      D = Dn;
    }
    base_prev = base;
    have_prev = true;
    base = nb;
  }
  // the last tile of this workgroup: linear store (16-byte vectors of contiguous runs)
  for (unsigned e = tid; e < nvec; e += BLOCK) {
    uint64_t g = base_prev;
    for (unsigned m = CB; m < ba.tb; ++m) g |= (uint64_t)((e >> (m - CB)) & 1u) << (ba.apos[m] - CB);
    __builtin_nontemporal_store(reinterpret_cast<V*>(xr)[blocked_swz(e)], vre + g);
    __builtin_nontemporal_store(reinterpret_cast<V*>(xi)[blocked_swz(e)], vim + g);
  }
}


}  // namespace hq
