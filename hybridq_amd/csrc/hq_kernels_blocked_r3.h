// hq_kernels_blocked_r3.h -- the cache-blocked kernel family EXACTLY as the last commit that ran on hardware compiled it
// ("round 3: VERDICT", af36621: csrc/hq_kernels_apply.h lines 605-1316 of that commit with its compile-time experiment
// branches -- all off in that build -- removed).  Namespace hq::r3.  This is what a DEFAULT run launches for a cache-blocked pass:
// tools/isa_vs_round.py shows these instantiations instruction for instruction identical to that commit's binaries.  The
// kernels of hq_kernels_blocked.h descend from this code (rounds 4-5: in-kernel table building for barrier-free wave groups,
// operand-ahead inner gates, direct first gate, 1024-thread tiles) and take over as soon as one of their switches is on
// (HQ_BLOCKED_PIPE / GROUPS / DIRECT / BIG, or HQ_BLOCKED_R3=0); the first hardware run that has seen them retires this file.
// Same host-side data: BlockedArg and BlockedGate have the layout of hq_kernels_blocked.h (`pad_` is `wave_bits` there, 0 here),
// the operand tables, the 136-word address table per gate and the LDS layout are unchanged.
#pragma once
#include "hq_kernels_common.h"
#include "hq_kernels_apply.h"

namespace hq {
namespace r3 {

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
  unsigned pad_;
};

template <typename T, int KBITS, int VMASK, int BLOCK>
__device__ __forceinline__ void blocked_inner_gate(T* __restrict__ xr, T* __restrict__ xi,
                                                   const BlockedGate& G, const T* __restrict__ A,
                                                   const unsigned tile_vec_bits
) {
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
  constexpr unsigned WB = BLOCK == 512 ? 3 : (BLOCK == 256 ? 2 : 4);  // log2(waves per workgroup)
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
// argument in blocked_inner_gate), one ds_read_b32 + one v_xor3 per vector, and the result rows go back with
// ds_write2_b32 pairs straight from the accumulators instead of 15 v_mov + 4 ds_write_b128.
typedef unsigned BlockedTabT;  // 16-bit entries were tried: more passes fit their tables, each gate 7 % slower
constexpr unsigned kBlockedTabLane = 0, kBlockedTabIter = 64, kBlockedTabOff = 128, kBlockedTabWords = 136;

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
    auto deposit = [&](unsigned v) {
      for (int m = 0; m < 4; ++m) {
        const unsigned lo = (1u << ro.pos[m]) - 1;
        v = ((v & ~lo) << 1) | (v & lo);
      }
      return v;
    };
    BlockedTabT* tb = tabs + g * kBlockedTabWords;
    for (unsigned e = tid; e < kBlockedTabWords; e += BLOCK) {
      unsigned val;
      if (e < kBlockedTabIter) {  // lane part: slot bits j, q digits, plane
        const unsigned q = e >> 4, j = e & 15;
        const unsigned lane_off = ((q & 1) ? ro.q_off[0] : 0u) | ((q & 2) ? ro.q_off[1] : 0u);
        const unsigned lane_plane = ro.q_plane >= 0 ? ((q >> ro.q_plane) & 1u) : 0u;
        val = ((blocked_swz(deposit(j) | lane_off) | (lane_plane << tile_vec_bits)) << 4) | lds_base;
      } else if (e < kBlockedTabOff) {  // wave-iteration part
        val = blocked_swz(deposit((e - kBlockedTabIter) << 4)) << 4;
      } else {  // register-digit part
        const unsigned ld = e - kBlockedTabOff;
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

// The prologue of the NEXT gate (4 operand values and the lane entry of a k <= 3 gate: the reads whose address hangs on the
// gate descriptor's scalar load) requested while the current gate runs; the 4 register-digit entries are read at the
// gate's start (holding them too spills: 128 registers is the budget of four waves per SIMD).  Unconditional, whatever the next
// gate's kind (wider gates load the rest themselves): a conditional request would merge old and new register values.
template <typename T> struct BlockedPre {
  T a[4];
  unsigned L;
};

template <typename T, int KBITS, int VMASK, int BLOCK, bool USEPRE = false>
__device__ __forceinline__ void blocked_inner_gate_tab(const T* __restrict__ A,
                                                       const BlockedTabT* __restrict__ tab, const unsigned niter,
                                                       const BlockedPre<T>& pre
) {
  using V = typename Vec<T>::type;
  using Acc = typename Mfma<T>::acc;
  constexpr int CB = Vec<T>::VB, NCOMP = 1 << CB;
  constexpr int NS = KBITS - 2, KV = popc_c(VMASK), NR = NS - KV, NL = 1 << NR;
  constexpr int NRB = 1 << (NS - 2), NCB = 1 << (CB - KV), NSTEP = 1 << NS;
  constexpr int FMASK = ~VMASK & (NCOMP - 1);
  constexpr unsigned WB = BLOCK == 512 ? 3 : (BLOCK == 256 ? 2 : 4);
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  T a[NRB][NSTEP];
  unsigned L;
  unsigned OFF[NL];
  if constexpr (USEPRE && KBITS == 4) {  // requested one gate ahead (apply_blocked_kernel): nothing to wait for here
    static_assert(NRB == 1 && NSTEP == 4 && NL <= 4, "k <= 3 shape");
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) a[0][s] = pre.a[s];
    L = pre.L;
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) OFF[ld] = tab[kBlockedTabOff + ld];
  } else {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) a[rb][s] = A[(rb * NSTEP + s) * 64 + lane];
    L = tab[kBlockedTabLane + lane];
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) OFF[ld] = tab[kBlockedTabOff + ld];
  }
  typedef __attribute__((address_space(3))) V LdsV;  // addresses are absolute LDS byte addresses (base folded in LANE)
  for (unsigned it = wave; it < niter; it += 1u << WB) {
    const unsigned Lt = L ^ tab[kBlockedTabIter + it];
    unsigned addr[NL];
    V x[NL];
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) {
      addr[ld] = Lt ^ OFF[ld];
      x[ld] = *reinterpret_cast<LdsV*>((uintptr_t)addr[ld]);
    }
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
    // component c of vector ld sits in accumulator block (cf, so >> 2), register so & 3.  A 16-byte store wants 4
    // consecutive registers, i.e. a transpose by 15 v_mov per iteration (and the compiler re-vectorises element
    // stores into exactly that); ds_write2 takes its two elements from any two registers.  Inline assembly: the
    // compiler neither counts these stores (lgkmcnt is drained by hand after the loop) nor pads the
    // MFMA-result -> LDS-read hazard in front of them (s_nop by hand: 8-pass MFMA, 16 wait states cover it).
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) {
      V y;
#pragma unroll
      for (int comp = 0; comp < NCOMP; ++comp) {
        const int ck = pext_c(comp, VMASK), cf = pext_c(comp, FMASK);
        const int so = ck | (ld << KV);
        y[comp] = acc[cf][so >> 2][so & 3];
      }
      *reinterpret_cast<LdsV*>((uintptr_t)addr[ld]) = y;
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

// ALDS: the A-operand tables of all gates of the pass (a_elems elements) are staged once per
// (persistent) workgroup in LDS behind the tile; a table read from global memory puts an L2 round
// trip (~1500 clk, as long as the gate's MFMAs) in front of every gate of every tile.
// PREF (tiles of exactly 4 * BLOCK vectors per plane): serial phases -- load a tile (one HBM round trip), run the
// gates, store -- run in step on the whole chip, so HBM idles while the gates run and the matrix cores idle while
// tiles move: a pass costs HBM time PLUS gate time.  With PREF the next tile's vectors are requested into registers
// before the gates of the current tile start and dropped into LDS after its stores were issued.  Needs the
// no-scratch register budget: a scratch reload is a vector-memory load and would queue (vmcnt is in order) behind
// the prefetch it was supposed to overlap.
template <typename T, int BLOCK, bool ALDS, bool PREF, bool GPRE = false>
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
    for (unsigned m = CB; m < (PREF ? CB + 11u : (unsigned)kBlockedMaxTileBits); ++m) {  // PREF: exactly 4 * 512 vectors
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
    BlockedPre<T> pre_next;
    auto request = [&](unsigned gn) {  // prologue of gate gn (clamped by the caller): LDS reads only, waited for when used
      const T* An = als + gates[gn].a_off;
      const BlockedTabT* tn = tabs + gn * kBlockedTabWords;
      const unsigned lane = tid & 63;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) pre_next.a[s2] = An[s2 * 64 + lane];
      pre_next.L = tn[kBlockedTabLane + lane];
    };
    if constexpr (GPRE && ALDS) request(0);
    for (unsigned gi = 0; gi < ngates; ++gi) {
      const BlockedGate& G = gates[gi];
      const T* A = ALDS ? als + G.a_off : Atab + G.a_off;
      const BlockedPre<T>& pre = pre_next;  // consumed in the gate's first instructions; re-requested after its last
#define HQ_BLOCKED_MFMA_GATE(KB, VM)                                                                    \
  do {                                                                                                  \
    if constexpr (ALDS)                                                                                 \
      blocked_inner_gate_tab<T, KB, VM, BLOCK, GPRE>(A, tabs + gi * kBlockedTabWords, (1u << (tvb - G.n_addr)) >> 4, pre); \
    else                                                                                                \
      blocked_inner_gate<T, KB, VM, BLOCK>(xr, xi, G, A, tvb);                                          \
  } while (0)
      switch (G.kv) {
        case 16: HQ_BLOCKED_MFMA_GATE(4, 0); break;
        case 17: HQ_BLOCKED_MFMA_GATE(4, 1); break;
        case 20: HQ_BLOCKED_MFMA_GATE(5, 0); break;
        case 21: HQ_BLOCKED_MFMA_GATE(5, 1); break;
        case 64 + 4 + 0: blocked_inner_gate_valu<T, 1, 0, BLOCK>(xr, xi, G, Atab + G.a_off, tvb); break;
        case 64 + 4 + 1: blocked_inner_gate_valu<T, 1, 1, BLOCK>(xr, xi, G, Atab + G.a_off, tvb); break;
        case 64 + 8 + 0: blocked_inner_gate_valu<T, 2, 0, BLOCK>(xr, xi, G, Atab + G.a_off, tvb); break;
        case 64 + 8 + 1: blocked_inner_gate_valu<T, 2, 1, BLOCK>(xr, xi, G, Atab + G.a_off, tvb); break;
        default:
          if constexpr (CB == 2) {
            switch (G.kv) {
              case 18: HQ_BLOCKED_MFMA_GATE(4, 2); break;
              case 19: HQ_BLOCKED_MFMA_GATE(4, 3); break;
              case 22: HQ_BLOCKED_MFMA_GATE(5, 2); break;
              case 23: HQ_BLOCKED_MFMA_GATE(5, 3); break;
              case 64 + 4 + 2: blocked_inner_gate_valu<T, 1, 2, BLOCK>(xr, xi, G, Atab + G.a_off, tvb); break;
              case 64 + 8 + 2: blocked_inner_gate_valu<T, 2, 2, BLOCK>(xr, xi, G, Atab + G.a_off, tvb); break;
              case 64 + 8 + 3: blocked_inner_gate_valu<T, 2, 3, BLOCK>(xr, xi, G, Atab + G.a_off, tvb); break;
              default: break;
            }
          }
          break;
      }
      // the next gate's prologue reads are issued in front of the barrier and land while the workgroup gathers at it
      // (requested at the gate's START they cost 5 more live registers through the MFMA phase: spills)
      if constexpr (GPRE && ALDS) request(gi + 1 < ngates ? gi + 1 : gi);
      __syncthreads();
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


}  // namespace r3
}  // namespace hq
