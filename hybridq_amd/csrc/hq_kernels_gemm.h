// hq_kernels_gemm.h -- k = 7..10 on the matrix cores: the tile GEMM (apply_gemm_kernel; reference: the runtime-k loop
// U.h:123-202).  Split from hq_kernels_apply.h in round 5.
#pragma once
#include "hq_kernels_apply.h"

namespace hq {

// ---------------------------------------------------------------------------------
// k = 7..10 on the matrix cores: apply_gemm_kernel (reference: the runtime-k loop U.h:123-202).
//
// A workgroup (8 waves) owns a tile of 2^TB amplitudes (TB = 14 f32 / 13 f64: both planes =
// 128 KiB of LDS) spanned by the k targets + the lowest TB-k non-target bits ("columns",
// always including index bits 0/1 so that every HBM access is a 16-byte vector of a
// contiguous run).  The tile is the B operand X[2^k rows][C columns] of a complex GEMM
// out = U . X, done as 4 real MFMA streams (Ur.xr, -Ui.xi -> re; Ui.xr, Ur.xi -> im; the minus
// sign is applied to the B register).  Wave (wr, wc) accumulates RBW x CBW 16x16 blocks of the
// output in registers (64 accumulator VGPRs for every k); A operands (its own rows of Ur, Ui)
// come straight from global/L2 as one 16-byte load per 4 (f32) / 2 (f64) K-steps from a table
// the host lays out in operand order; B operands are one ds_read_b32/b64 per K-step and column
// block.  LDS holds the tile in its natural tile-local order with an XOR swizzle chosen by
// the host per gate so that the B reads of a half-wave hit 32 distinct banks whatever the
// target positions.  After the K loop the results replace the tile in LDS and stream back.
// ---------------------------------------------------------------------------------
constexpr int kGemmBlock = 512;
template <typename T, int RBW, int CBW> constexpr bool gemm_can_pipe() { return !(sizeof(T) == 8 && RBW * CBW >= 8); }
constexpr int kGemmMaxTileBits = 14;
struct GemmArg {
  unsigned tb, k;                    // tile bits, target bits
  unsigned apos[kGemmMaxTileBits];   // global index positions of the tile-local bits, ascending
  unsigned tl[4], cl[4];             // tile-local bit of the 4 lowest target / column digits
  unsigned n_sw, sw_src[4], sw_dst[4];  // LDS swizzle: element bit src is XORed into bit dst
  unsigned nsg;                      // A-load groups = 2^k / (4 G), G = 16 / sizeof(T)
};

// NPV > 0 (tiles of exactly NPV * 512 vectors per plane): the next tile is requested into registers before the
// MFMA phase of the current one and dropped into LDS after its results were stored (the same recipe, for the same
// reason, as apply_blocked_kernel's PREF: copy-in, MFMA and copy-out phases run in step on the whole chip, so HBM
// idled while the matrix cores worked and vice versa -- k = 7 measured 10.5 ms = 7.0 ms of MFMA + 2.9 ms of HBM).
// PIPE (HQ_GEMM_PIPE, default 1): operands requested ahead of the matrix cores (below); false = the loop of rounds 1-4a.
template <typename T, int RBW, int CBW, int NPV, bool PIPE>
__global__ void __launch_bounds__(kGemmBlock)
apply_gemm_kernel(T* __restrict__ re, T* __restrict__ im, const T* __restrict__ Atab,
                  const unsigned* __restrict__ offs, const GemmArg a, const uint64_t ntiles) {
  using V = typename Vec<T>::type;
  using Acc = typename Mfma<T>::acc;
  HQ_DYN_LDS(hq_gemm_smem);
  constexpr int CB = Vec<T>::VB, G = 16 / (int)sizeof(T);
  T* __restrict__ xr = reinterpret_cast<T*>(hq_gemm_smem);
  T* __restrict__ xi = xr + ((size_t)1 << a.tb);
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned q = lane >> 4, j = lane & 15;
  const unsigned D4 = (1u << a.k) >> 2, NRBT = (1u << a.k) >> 4, NCB = (1u << (a.tb - a.k)) >> 4;
  const unsigned WC = NCB / CBW, wr = wave / WC, wc = wave % WC;
  const unsigned* __restrict__ toff = offs;           // [D4]   swizzled offset of K-step (t = 4 step)
  const unsigned* __restrict__ rboff = offs + D4;     // [NRBT] ... of output row block (t = 16 rb)
  const unsigned* __restrict__ coff = rboff + NRBT;   // [NCB]  ... of column block (col = 16 cb)
  auto swz = [&](unsigned e) {
    for (unsigned i = 0; i < a.n_sw; ++i) e ^= ((e >> a.sw_src[i]) & 1u) << a.sw_dst[i];
    return e;
  };
  auto dep4 = [](unsigned v, const unsigned* p) {
    return ((v & 1u) << p[0]) | (((v >> 1) & 1u) << p[1]) | (((v >> 2) & 1u) << p[2]) | (((v >> 3) & 1u) << p[3]);
  };
  const unsigned colpart = dep4(j, a.cl);
  unsigned lane_cb[CBW];  // B operand: row digits 0,1 = q, column digits 0..3 = j
  {
    const unsigned lb = swz(((q & 1u) << a.tl[0]) | ((q >> 1) << a.tl[1]) | colpart);
#pragma unroll
    for (int c = 0; c < CBW; ++c) lane_cb[c] = lb ^ coff[wc * CBW + c];
  }
  unsigned lane_w[4];  // D operand register r: row 4q+r (f32 MFMA) / q+4r (f64 MFMA) of the block
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned rowin = sizeof(T) == 4 ? (4 * q + r) : (q + 4 * r);
    lane_w[r] = swz(dep4(rowin, a.tl) | colpart);
  }
  V* __restrict__ pre = reinterpret_cast<V*>(re);
  V* __restrict__ pim = reinterpret_cast<V*>(im);
  const unsigned nvec = 1u << (a.tb - CB);
  const V* __restrict__ Av = reinterpret_cast<const V*>(Atab) + lane;

  constexpr bool PREF = NPV > 0;
  constexpr int NP = PREF ? NPV : 1;
  auto tile_base = [&](uint64_t tile) {  // vec index with zeros at the tile's (non-component) positions
    uint64_t base = tile;
#pragma unroll
    for (unsigned m = CB; m < (unsigned)kGemmMaxTileBits; ++m) {  // constant trip count: positions read from the arguments once
      const uint64_t lo = m < a.tb ? (1ull << (a.apos[m] - CB)) - 1 : ~0ull;
      base = ((base & ~lo) << 1) | (base & lo);
    }
    return base;
  };
  auto vec_off = [&](unsigned v) {  // OR-linear in v
    uint64_t g = 0;
    for (unsigned m = CB; m < a.tb; ++m) g |= (uint64_t)((v >> (m - CB)) & 1u) << (a.apos[m] - CB);
    return g;
  };
  const uint64_t stride = gridDim.x;
  // deposited-coordinate increment: next = ((cur | ~M) + D) & M, M = the index bits outside the tile
  const uint64_t dep_mask = tile_base(~0ull), dep_stride = tile_base(stride);
  auto next_base = [&](uint64_t b) { return ((b | ~dep_mask) + dep_stride) & dep_mask; };
  V pr[NP], pi[NP];
  unsigned slot[NP];
  uint64_t off_blk[NP];
  const uint64_t off_tid = PREF ? vec_off(tid) : 0;
  if constexpr (PREF) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      slot[i] = swz((tid + i * kGemmBlock) << CB) >> CB;
      off_blk[i] = vec_off(i * kGemmBlock);  // wave-uniform
    }
  }
  auto prefetch = [&](const uint64_t b) {  // unconditional, uniform address part pinned to SGPRs (see apply_blocked_kernel)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      uint64_t sb = b | off_blk[i];
      HQ_PIN_SGPR(sb);
      pr[i] = __builtin_nontemporal_load(pre + (sb | off_tid));
      pi[i] = __builtin_nontemporal_load(pim + (sb | off_tid));
    }
  };
  uint64_t base_cur = 0;
  if constexpr (PREF) {
    if (blockIdx.x >= ntiles) return;
    base_cur = tile_base(blockIdx.x);
    {
      const uint64_t b = base_cur | off_tid;  // first tile: straight into LDS
#pragma unroll 1
      for (int i = 0; i < NP; ++i) {
        const uint64_t g = b | vec_off(i * kGemmBlock);
        const unsigned sl = swz((tid + i * kGemmBlock) << CB) >> CB;
        reinterpret_cast<V*>(xr)[sl] = __builtin_nontemporal_load(pre + g);
        reinterpret_cast<V*>(xi)[sl] = __builtin_nontemporal_load(pim + g);
      }
    }
    prefetch(blockIdx.x + stride < ntiles ? next_base(base_cur) : base_cur);
  }

  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += stride) {
    const uint64_t base = PREF ? base_cur : tile_base(tile);
    if constexpr (!PREF) {
      for (unsigned v = tid; v < nvec; v += kGemmBlock) {
        const uint64_t g = base | vec_off(v);
        const unsigned sl = swz(v << CB) >> CB;
        reinterpret_cast<V*>(xr)[sl] = __builtin_nontemporal_load(pre + g);
        reinterpret_cast<V*>(xi)[sl] = __builtin_nontemporal_load(pim + g);
      }
    }
    __syncthreads();
    Acc accr[RBW][CBW], acci[RBW][CBW];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
      for (int c = 0; c < CBW; ++c) { accr[rb][c] = Acc{0, 0, 0, 0}; acci[rb][c] = Acc{0, 0, 0, 0}; }
    // the loop of rounds 1-4a: still what complex128 with 128 accumulator registers runs (no registers for a second
    // operand set), and what PIPE = false runs everywhere
    constexpr bool kPipe = PIPE && gemm_can_pipe<T, RBW, CBW>();
    if constexpr (!kPipe) {
    for (unsigned sg = 0; sg < a.nsg; ++sg) {
        V ur[RBW], ui[RBW];
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
          const V* __restrict__ pA = Av + ((size_t)((wr * RBW + rb) * a.nsg + sg) * 2) * 64;
          ur[rb] = pA[0];
          ui[rb] = pA[64];
        }
#pragma unroll
        for (int s = 0; s < G; ++s) {
          const unsigned to = toff[sg * G + s];
#pragma unroll
          for (int c = 0; c < CBW; ++c) {
            const unsigned e = lane_cb[c] ^ to;
            const T br = xr[e], bi = xi[e], nbi = -bi;
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) accr[rb][c] = Mfma<T>::run(ur[rb][s], br, accr[rb][c]);
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) acci[rb][c] = Mfma<T>::run(ui[rb][s], br, acci[rb][c]);
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) accr[rb][c] = Mfma<T>::run(ui[rb][s], nbi, accr[rb][c]);
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) acci[rb][c] = Mfma<T>::run(ur[rb][s], bi, acci[rb][c]);
          }
        }
      }
    } else {
    // Operands ahead of the matrix cores (round 4, from the assembly of the loop above: the B operands of a K-step were
    // requested from LDS right in front of the step's 4 RBW CBW MFMAs and the A operands of a step group from L2 at its
    // top -- one LDS round trip per K-step and one L2 round trip per step group in front of the matrix pipe, most of
    // the 23-29 % this kernel stayed below the MFMA peak).  Two register sets each: the A operands of step group
    // sg + 1 are requested before the MFMAs of group sg start, the B operands of K-step s + 1 before those of step s;
    // requests are unconditional (past the end: a repeat of the last one) and scheduling barriers keep them where
    // they are written.  Same MFMAs in the same order on every accumulator: bit-identical results.
    auto request_a = [&](V (&ur)[RBW], V (&ui)[RBW], const unsigned sg) {
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb) {
        const V* __restrict__ pA = Av + ((size_t)((wr * RBW + rb) * a.nsg + sg) * 2) * 64;
        ur[rb] = pA[0];
        ui[rb] = pA[64];
      }
    };
    auto request_b = [&](T (&br)[CBW], T (&bi)[CBW], const unsigned step) {
      const unsigned to = toff[step];
#pragma unroll
      for (int c = 0; c < CBW; ++c) {
        const unsigned e = lane_cb[c] ^ to;
        br[c] = xr[e];
        bi[c] = xi[e];
      }
    };
    auto multiply = [&](V (&ur)[RBW], V (&ui)[RBW], const int s, T (&br)[CBW], T (&bi)[CBW]) {
#pragma unroll
      for (int c = 0; c < CBW; ++c) {
        const T nbi = -bi[c];
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) accr[rb][c] = Mfma<T>::run(ur[rb][s], br[c], accr[rb][c]);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acci[rb][c] = Mfma<T>::run(ui[rb][s], br[c], acci[rb][c]);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) accr[rb][c] = Mfma<T>::run(ui[rb][s], nbi, accr[rb][c]);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acci[rb][c] = Mfma<T>::run(ur[rb][s], bi[c], acci[rb][c]);
      }
    };
    static_assert(G % 2 == 0, "the B sets alternate by K-step parity");
    const unsigned last_step = a.nsg * G - 1;
    V ua0[RBW], ub0[RBW], ua1[RBW], ub1[RBW];
    T br0[CBW], bi0[CBW], br1[CBW], bi1[CBW];
    request_a(ua0, ub0, 0);
    request_b(br0, bi0, 0);
    // one step group: its K-steps alternate between the two B sets, each step requesting the next one's operands first
    auto group = [&](V (&ur)[RBW], V (&ui)[RBW], const unsigned sg) {
#pragma unroll
      for (int s = 0; s < G; s += 2) {
        const unsigned st = sg * G + s;
        request_b(br1, bi1, st + 1);  // st + 1 <= last_step: G is even
        __builtin_amdgcn_sched_barrier(0);
        multiply(ur, ui, s, br0, bi0);
        request_b(br0, bi0, st + 2 <= last_step ? st + 2 : last_step);
        __builtin_amdgcn_sched_barrier(0);
        multiply(ur, ui, s + 1, br1, bi1);
      }
    };
    // the second A set only where the registers are there (512 threads = two waves per SIMD = 256 registers: the widest
    // wave tiles hold 64 / 128 of them in accumulators and run the A requests of a group at its top as before)
    constexpr int kAccRegs = RBW * CBW * 2 * (sizeof(T) == 4 ? 4 : 8) + (NPV > 0 ? NPV * 8 : 0);
    constexpr bool kTwoA = kAccRegs < (sizeof(T) == 4 ? 64 : 128);
    if constexpr (kTwoA) {
      for (unsigned sg = 0; sg < a.nsg; sg += 2) {
        request_a(ua1, ub1, sg + 1 < a.nsg ? sg + 1 : sg);
        group(ua0, ub0, sg);
        if (sg + 1 >= a.nsg) break;
        request_a(ua0, ub0, sg + 2 < a.nsg ? sg + 2 : sg + 1);
        group(ua1, ub1, sg + 1);
      }
    } else {
      for (unsigned sg = 0; sg < a.nsg; ++sg) {
        if (sg) request_a(ua0, ub0, sg);
        group(ua0, ub0, sg);
      }
    }
    }
    __syncthreads();  // every wave is done reading the tile: replace it with the results
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
      const unsigned ro = rboff[wr * RBW + rb];
#pragma unroll
      for (int c = 0; c < CBW; ++c) {
        const unsigned rc = ro ^ coff[wc * CBW + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xr[lane_w[r] ^ rc] = accr[rb][c][r];
          xi[lane_w[r] ^ rc] = acci[rb][c][r];
        }
      }
    }
    __syncthreads();
    if constexpr (PREF) {
      V sr[NP], si[NP];  // all LDS reads in flight before the first store
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        sr[i] = reinterpret_cast<V*>(xr)[slot[i]];
        si[i] = reinterpret_cast<V*>(xi)[slot[i]];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        uint64_t sb = base | off_blk[i];
        HQ_PIN_SGPR(sb);
        __builtin_nontemporal_store(sr[i], pre + (sb | off_tid));
        __builtin_nontemporal_store(si[i], pim + (sb | off_tid));
      }
      // (no barrier: a thread refills exactly the slots it has just read for its stores)
      // the fill follows the stores on every path: the in-order vmcnt wait for the prefetched vectors sees
      // "2 NPV loads, then 2 NPV stores" and never drains the stores
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        reinterpret_cast<V*>(xr)[slot[i]] = pr[i];
        reinterpret_cast<V*>(xi)[slot[i]] = pi[i];
      }
      base_cur = next_base(base);
      prefetch(tile + 2 * stride < ntiles ? next_base(base_cur) : base);
    } else {
      for (unsigned v = tid; v < nvec; v += kGemmBlock) {
        const uint64_t g = base | vec_off(v);
        const unsigned sl = swz(v << CB) >> CB;
        __builtin_nontemporal_store(reinterpret_cast<V*>(xr)[sl], pre + g);
        __builtin_nontemporal_store(reinterpret_cast<V*>(xi)[sl], pim + g);
      }
      __syncthreads();
    }
  }
}


}  // namespace hq
