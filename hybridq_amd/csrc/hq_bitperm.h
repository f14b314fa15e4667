// hq_bitperm.h -- host side of bitperm_tile_kernel (hq_kernels_swap.h): tile choice, LDS swizzle and launch.
// Included by hq_swap.hip (hq_permute_bits_*, low-bit swaps of 14 / 15 bits in place) and hq_shard.hip (the pack
// pass of the qubit exchange).  Semantics: dst[x] = src[pi(x)], dst bit i <-> src bit perm[i]
// (the low-bit form is the gather of /root/reference/include/swap.h:61-95).
#pragma once
#include "hq_common.h"
#include "hq_kernels_swap.h"

namespace hq {

struct BitPermPlan {
  BitPermArg a;
  bool vread = false;   // the vector-component bits stay: 16-byte LDS reads
  unsigned block = 256, nv = 4;
  int split = 0;        // 1..3: one moved bit more than the 128 KiB tile holds (bitperm_tile_kernel SPLIT mode)
  size_t lds = 0;
};

// Tile bits for the out-of-place form (HQ_PERM_TB overrides).  Measured at n = 30 (profiles/r03_perm_rate.txt): longer
// runs win -- 2^13 float32 elements (32 KiB, four 256-thread workgroups per CU) 4.6-5.5 TB/s, 2^12 4.2-4.8, 2^11 2.5-3.4
template <typename E>
static unsigned bitperm_default_tb() {
  static const int forced = env_int("HQ_PERM_TB", 0);
  const unsigned lo = sizeof(E) == 4 ? 10 : 9, hi = sizeof(E) == 4 ? 15 : 14;
  if (forced) return std::min(hi, std::max(lo, (unsigned)forced - (sizeof(E) == 4 ? 0u : 1u)));
  return sizeof(E) == 4 ? 13 : 12;
}

// true when the permutation keeps the low 128 bytes of the index space in place: the plain gather kernels then move
// whole cache lines on both sides too (HQ_PERM_TILE=3 routes those to them; not the default, see hq_swap.hip)
template <typename E>
static bool bitperm_low_run_fixed(const unsigned* perm, unsigned m) {
  const unsigned c = sizeof(E) == 4 ? 5 : 4;
  if (m < c) return false;
  for (unsigned b = 0; b < c; ++b)
    if (perm[b] != b) return false;
  return true;
}

// perm: m entries.  `inplace`: every moved bit must lie inside the tile (the caller passes src == dst); the tile then
// has exactly the size the moved bits need (>= 2^12 / 2^11 elements), up to 128 KiB.  Returns false when this kernel
// cannot run the permutation (state smaller than a tile, too many moved bits for an in-place tile): the callers keep
// their older paths for those.
template <typename E>
static bool plan_bitperm(const unsigned* perm, unsigned m, bool inplace, BitPermPlan& P) {
  constexpr unsigned VB = sizeof(E) == 4 ? 2 : 1;
  const unsigned c = sizeof(E) == 4 ? 5 : 4;  // 128-byte runs on both sides
  if (m > 62 || m < 2 * c) return false;
  std::vector<unsigned> inv(m);
  for (unsigned i = 0; i < m; ++i) inv[perm[i]] = i;
  uint64_t T = 0;
  unsigned tb;
  unsigned split_b = 0, split_rank = 0;
  bool split_b_set = false;
  if (inplace) {
    for (unsigned b = 0; b < c; ++b) T |= 1ull << b;
    for (unsigned i = 0; i < m; ++i)
      if (perm[i] != i) T |= 1ull << i;  // moved dst bits (the set of moved src bits is the same set)
    const unsigned need = (unsigned)__builtin_popcountll(T);
    const unsigned max_tb = sizeof(E) == 4 ? 15 : 14;  // 128 KiB
    // the largest tile the state allows: one 1024-thread workgroup per CU on 128 KiB tiles measured 5.1-5.3 TB/s, two
    // on 64 KiB tiles 4.2-4.9 (HQ_PERM_INPLACE_TB overrides)
    static const int forced_ip = env_int("HQ_PERM_INPLACE_TB", 0);
    tb = std::max(need, std::min(m, forced_ip ? (unsigned)forced_ip - (sizeof(E) == 4 ? 0u : 1u) : max_tb));
    static const bool allow_split = env_int("HQ_PERM_SPLIT", 1) != 0;
    if (need == max_tb + 1 && allow_split) {
      // one moved bit too many for the LDS tile: take a moved bit b out of the tile (SPLIT mode of the kernel).  b must be
      // one of the three highest tile bits whose source side rank is an iteration bit, and neither b nor perm[b] may be
      // one of the contiguous run bits
      std::vector<unsigned> Tb;
      for (unsigned i = 0; i < m; ++i) if ((T >> i) & 1) Tb.push_back(i);
      for (int cand = (int)Tb.size() - 1; cand >= (int)Tb.size() - 3 && !split_b_set; --cand) {
        const unsigned b = Tb[cand], pb = perm[b];
        if (pb == b || b < c || pb < c) continue;
        unsigned rank = 0;  // rank of b among the source bits of the reduced tile: S' = T minus {pb}
        for (unsigned v : Tb) if (v < b && v != pb) ++rank;
        const unsigned tbr = max_tb;
        if (rank + 3 < tbr) continue;
        split_b = b; split_rank = rank; split_b_set = true;
      }
      if (!split_b_set) return false;
      T &= ~(1ull << split_b);
      tb = max_tb;
    } else {
      if (need > max_tb || tb > m || tb > max_tb) return false;
      for (unsigned b = 0; b < m && (unsigned)__builtin_popcountll(T) < tb; ++b) T |= 1ull << b;  // lowest fixed bits
    }
  } else {
    tb = bitperm_default_tb<E>();
    if (tb > m) return false;
    for (unsigned b = 0; b < c; ++b) T |= (1ull << b) | (1ull << inv[b]);
    if ((unsigned)__builtin_popcountll(T) > tb) return false;
    // longer runs on both sides, alternately: the next dst bit, the dst bit feeding the next src bit
    unsigned nd = c, ns = c;
    bool turn = false;
    while ((unsigned)__builtin_popcountll(T) < tb) {
      if (turn) {
        while (ns < m && ((T >> inv[ns]) & 1)) ++ns;
        if (ns < m) T |= 1ull << inv[ns];
      } else {
        while (nd < m && ((T >> nd) & 1)) ++nd;
        if (nd < m) T |= 1ull << nd;
      }
      turn = !turn;
    }
  }
  BitPermArg& a = P.a;
  memset(&a, 0, sizeof(a));
  a.tb = tb;
  a.m = m;
  a.cbits = m;
  a.planes = 1;
  std::vector<unsigned> tpos, spos;
  for (unsigned b = 0; b < m; ++b)
    if ((T >> b) & 1) { tpos.push_back(b); spos.push_back(perm[b]); }
  std::sort(spos.begin(), spos.end());
  for (unsigned k = 0; k < tb; ++k) {
    a.tpos[k] = (unsigned char)tpos[k];
    a.spos[k] = (unsigned char)spos[k];
    a.sigma[k] = (unsigned char)(std::find(spos.begin(), spos.end(), perm[tpos[k]]) - spos.begin());
  }
  for (unsigned b = 0; b < c; ++b)
    if (a.tpos[b] != b || a.spos[b] != b) return false;  // both index spaces start with the contiguous run bits
  // LDS swizzle: the element reads of a half-wave vary the dst-local bits VB..VB+4, i.e. the source-local bits
  // sigma[VB..VB+4]; those of them that are >= 5 are folded into bank bits of [VB, 5) no other one occupies
  {
    bool taken[5] = {false, false, false, false, false};
    std::vector<unsigned> high;
    for (unsigned b = VB; b < VB + 5 && b < tb; ++b) {
      if (a.sigma[b] < 5) taken[a.sigma[b]] = true; else high.push_back(a.sigma[b]);
    }
    unsigned tgt = VB;
    for (unsigned u : high) {
      while (tgt < 5 && taken[tgt]) ++tgt;
      if (tgt >= 5) break;
      a.sw_hi[a.nsw] = (unsigned char)u;
      a.sw_lo[a.nsw] = (unsigned char)tgt;
      ++a.nsw;
      taken[tgt++] = true;
    }
  }
  // tile base: runs of non-tile dst bits whose sources are consecutive too
  for (unsigned i = 0; i < m;) {
    if ((T >> i) & 1) { ++i; continue; }
    unsigned len = 1;
    while (i + len < m && !((T >> (i + len)) & 1) && perm[i + len] == perm[i] + len) ++len;
    if (a.nfields >= 48) return false;
    a.f_from[a.nfields] = (unsigned char)i;
    a.f_to[a.nfields] = (unsigned char)perm[i];
    a.f_len[a.nfields] = (unsigned char)len;
    ++a.nfields;
    i += len;
  }
  // bits the tile number skips when it is deposited into the index
  a.nb = 0;
  for (unsigned b = 0; b < m; ++b)
    if (((T >> b) & 1) || (split_b_set && b == split_b)) a.bpos[a.nb++] = (unsigned char)b;
  P.split = 0;
  if (split_b_set) {
    a.half_x = 1ull << split_b;
    a.half_y = 1ull << perm[split_b];
    P.split = (int)(split_rank - (tb - 3)) + 1;
  }
  P.vread = true;
  for (unsigned b = 0; b < VB; ++b) P.vread = P.vread && perm[b] == b;
  const unsigned nvec = (1u << tb) >> VB;
  // (a register prefetch of the next tile, the recipe of apply_blocked_kernel, measured SLOWER here -- 4.19 vs 4.89 TB/s on
  // 64 KiB tiles, 5.13 vs 5.32 on 128 KiB tiles, round 3 -- and left the library in round 5)
  if (nvec <= 8 * 256) {
    P.block = 256;
    P.nv = nvec / 256;
  } else {  // 64 / 128 KiB tiles: two / one 1024-thread workgroups per CU
    P.block = 1024;
    P.nv = nvec / 1024;
  }
  if (P.nv < 1 || P.nv > 8 || (P.nv & (P.nv - 1))) return false;
  P.lds = (size_t)sizeof(E) << tb;
  return true;
}

template <typename E, int BLOCK, int NV, bool VREAD, int SPLIT = 0>
static int launch_bitperm_inst(Context& c, hipStream_t s, bool on_lib_stream, const E* s0, const E* s1, const BitPermPlan& P,
                               uint64_t ntiles) {
  static bool attr_done = false;  // per instantiation, under the context mutex
  if (!attr_done) {
    HQ_HIP_CHECK(hipFuncSetAttribute((const void*)bitperm_tile_kernel<E, BLOCK, NV, VREAD, SPLIT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024));
    attr_done = true;
  }
  const uint64_t total = ntiles * P.a.planes;
  // one resident set of workgroups looping over the tiles measured best (n = 30: grid x1 5.35-5.85 TB/s, x4 4.86-5.69, x32 3.5-4.3)
  static const int grid_mult = std::max(1, env_int("HQ_PERM_GRID", 1));
  const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>((160 * 1024) / (P.lds + 512), 2048 / BLOCK));
  const unsigned grid = (unsigned)std::min<uint64_t>(total, 256 * per_cu * (BLOCK == 256 ? (uint64_t)grid_mult : 1));
  const BitPermArg a = P.a;
  if (on_lib_stream) {
    HQ_LAUNCH(c, (bitperm_tile_kernel<E, BLOCK, NV, VREAD, SPLIT>), dim3(grid), dim3(BLOCK), P.lds, s0, s1, a, ntiles);
  } else {
    hipLaunchKernelGGL((bitperm_tile_kernel<E, BLOCK, NV, VREAD, SPLIT>), dim3(grid), dim3(BLOCK), P.lds, s, s0, s1, a, ntiles);
  }
  HQ_HIP_CHECK(hipGetLastError());
  return 0;
}

// s = stream to launch on; on_lib_stream = it is the library stream (the launch may then be recorded into a program)
template <typename E>
static int launch_bitperm(Context& c, hipStream_t s, bool on_lib_stream, const E* s0, const E* s1, const BitPermPlan& P) {
  const uint64_t ntiles = 1ull << (P.a.m - P.a.nb);
  if (P.split) {
    if (P.block != 1024 || P.nv != 8 || P.a.planes != 1) return fail("bitperm: bad split plan");
#define HQ_BPS(V, S) return launch_bitperm_inst<E, 1024, 8, V, S>(c, s, on_lib_stream, s0, s1, P, ntiles)
    if (P.vread) { switch (P.split) { case 1: HQ_BPS(true, 1); case 2: HQ_BPS(true, 2); case 3: HQ_BPS(true, 3); } }
    else { switch (P.split) { case 1: HQ_BPS(false, 1); case 2: HQ_BPS(false, 2); case 3: HQ_BPS(false, 3); } }
#undef HQ_BPS
    return fail("bitperm: bad split plan");
  }
#define HQ_BP(B, N, V) return launch_bitperm_inst<E, B, N, V>(c, s, on_lib_stream, s0, s1, P, ntiles)
  if (P.block == 256) {
    if (P.vread) {
      switch (P.nv) { case 1: HQ_BP(256, 1, true); case 2: HQ_BP(256, 2, true); case 4: HQ_BP(256, 4, true); case 8: HQ_BP(256, 8, true); }
    } else {
      switch (P.nv) { case 1: HQ_BP(256, 1, false); case 2: HQ_BP(256, 2, false); case 4: HQ_BP(256, 4, false); case 8: HQ_BP(256, 8, false); }
    }
  } else {
    if (P.vread) { switch (P.nv) { case 4: HQ_BP(1024, 4, true); case 8: HQ_BP(1024, 8, true); } }
    else { switch (P.nv) { case 4: HQ_BP(1024, 4, false); case 8: HQ_BP(1024, 8, false); } }
  }
#undef HQ_BP
  return fail("bitperm: unsupported tile shape");
}

}  // namespace hq
