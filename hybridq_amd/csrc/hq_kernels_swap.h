// hq_kernels_swap.h -- index-bit permutations (reference: /root/reference/include/swap.h:28-95): low-bit swap through
// LDS, tile permutation, out-of-place gather, arbitrary bit permutation, the pack pass of the multi-GPU exchange.
#pragma once
#include "hq_kernels_common.h"

namespace hq {

// ---------------------------------------------------------------------------------
// swap (low-bit permutation)
// ---------------------------------------------------------------------------------
struct SwapArg {
  unsigned s;
  unsigned pos[32];
};

// One workgroup permutes 2^tile_bits contiguous elements (>= one 2^s chunk) through LDS:
// 16-byte loads into LDS, permuted LDS reads, 16-byte stores (VEC elements per access).
// TABLE = false (large s: the whole LDS budget goes to the tile) computes the permuted index
// inline instead of reading it from a 2^s-entry table.
// NPV > 0 (tiles of exactly NPV * kBlock vectors): the next tile is requested into registers while this one is
// permuted and stored, and dropped into LDS after the stores were issued (the recipe of apply_blocked_kernel's PREF:
// load / permute-store phases of a tile no longer alternate in step on the whole chip).
template <typename E, int VEC, bool TABLE, int NPV>
__global__ void __launch_bounds__(kBlock)
swap_lds_kernel(E* __restrict__ a, const SwapArg sa, const unsigned tile_bits,
                const uint64_t ntiles) {
  struct alignas(sizeof(E) * VEC) Pack { E e[VEC]; };
  typedef E PackV __attribute__((ext_vector_type(VEC)));
  HQ_DYN_LDS(smem);
  const unsigned S = 1u << sa.s, TILE = 1u << tile_bits;
  uint16_t* src = reinterpret_cast<uint16_t*>(smem);          // S entries (TABLE only)
  E* buf = reinterpret_cast<E*>(smem + (TABLE ? (((size_t)S * 2 + 15) & ~(size_t)15) : 0));
  const unsigned tid = threadIdx.x;
  if (TABLE)
    for (unsigned x = tid; x < S; x += kBlock) {
      unsigned y = 0;
      for (unsigned i = 0; i < sa.s; ++i) y |= ((x >> i) & 1u) << sa.pos[i];
      src[x] = (uint16_t)y;
    }
  auto permute_store = [&](E* g) {
    for (unsigned x = tid * VEC; x < TILE; x += kBlock * VEC) {
      Pack p;
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        const unsigned xx = x + c;
        unsigned y;
        if (TABLE) {
          y = src[xx & (S - 1)];
        } else {
          y = 0;
          for (unsigned i = 0; i < sa.s; ++i) y |= ((xx >> i) & 1u) << sa.pos[i];
        }
        p.e[c] = buf[(xx & ~(S - 1)) | y];
      }
      *reinterpret_cast<Pack*>(g + x) = p;
    }
  };
  if constexpr (NPV > 0 && VEC > 1) {
    if (blockIdx.x >= ntiles) return;
    const uint64_t stride = gridDim.x;
    PackV pr[NPV];
    auto prefetch = [&](uint64_t tb) {  // unconditional (callers clamp): see apply_blocked_kernel
      const PackV* g = reinterpret_cast<const PackV*>(a + tb * TILE) + tid;
#pragma unroll
      for (int i = 0; i < NPV; ++i) pr[i] = __builtin_nontemporal_load(g + i * kBlock);
    };
    {
      const PackV* g = reinterpret_cast<const PackV*>(a + (uint64_t)blockIdx.x * TILE) + tid;
#pragma unroll 1
      for (int i = 0; i < NPV; ++i) reinterpret_cast<PackV*>(buf)[tid + i * kBlock] = __builtin_nontemporal_load(g + i * kBlock);
    }
    prefetch(blockIdx.x + stride < ntiles ? blockIdx.x + stride : blockIdx.x);
    for (uint64_t tb = blockIdx.x; tb < ntiles; tb += stride) {
      __syncthreads();
      permute_store(a + tb * TILE);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NPV; ++i) reinterpret_cast<PackV*>(buf)[tid + i * kBlock] = pr[i];
      prefetch(tb + 2 * stride < ntiles ? tb + 2 * stride : tb);
    }
  } else {
    for (uint64_t tb = blockIdx.x; tb < ntiles; tb += gridDim.x) {
      E* g = a + tb * TILE;
      __syncthreads();
      for (unsigned x = tid * VEC; x < TILE; x += kBlock * VEC) {
        *reinterpret_cast<Pack*>(buf + x) = *reinterpret_cast<const Pack*>(g + x);
      }
      __syncthreads();
      permute_store(g);
    }
  }
}

// In-place permutation of the TB index bits `apos` (ascending; the lowest ones are the index bits
// 0..log2(VEC)-1, so every global access is a 16-byte vector of a contiguous run) inside tiles staged
// through LDS: new tile element x = old tile element whose tile-local index has bit i of x at bit
// lp[i].  A low-bit permutation of up to 16 bits (swap_*, transpose()) that does not fit one LDS
// tile is the product of TWO such passes over different bit sets (host: plan_two_pass_swap).
constexpr int kTilePermBits = 13;
struct TilePermArg {
  unsigned tb;
  unsigned apos[kTilePermBits];  // global positions of the tile-local bits
  unsigned lp[kTilePermBits];    // tile-local bit i of the destination index -> tile-local bit of the source index
};

template <typename E, int VEC> struct PackOf { typedef E type __attribute__((ext_vector_type(VEC))); };
template <typename E> struct PackOf<E, 1> { typedef E type; };

// NPV > 0 (tiles of exactly NPV * kBlock vectors): register prefetch of the next tile (see swap_lds_kernel).
template <typename E, int VEC, int NPV>
__global__ void __launch_bounds__(kBlock)
tile_permute_kernel(E* __restrict__ a, const TilePermArg ta, const uint64_t ntiles) {
  using Pack = typename PackOf<E, VEC>::type;
  HQ_DYN_LDS(smem);
  constexpr unsigned VBITS = VEC == 4 ? 2 : (VEC == 2 ? 1 : 0);
  const unsigned TILE = 1u << ta.tb, NV = TILE >> VBITS;
  uint16_t* src = reinterpret_cast<uint16_t*>(smem);                                        // TILE entries: permuted tile-local index
  uint32_t* goff = reinterpret_cast<uint32_t*>(smem + (((size_t)TILE * 2 + 15) & ~(size_t)15));  // NV entries: element offset of vector v
  E* buf = reinterpret_cast<E*>(reinterpret_cast<unsigned char*>(goff) + (((size_t)NV * 4 + 15) & ~(size_t)15));
  const unsigned tid = threadIdx.x;
  for (unsigned x = tid; x < TILE; x += kBlock) {
    unsigned y = 0;
    for (unsigned i = 0; i < ta.tb; ++i) y |= ((x >> i) & 1u) << ta.lp[i];
    src[x] = (uint16_t)y;
  }
  for (unsigned v = tid; v < NV; v += kBlock) {  // the tile's bits all sit below bit 32 (swap: s <= 18)
    uint32_t g = 0;
    for (unsigned m = VBITS; m < ta.tb; ++m) g |= ((v >> (m - VBITS)) & 1u) << ta.apos[m];
    goff[v] = g;
  }
  auto tile_ptr = [&](uint64_t t) {
    uint64_t base = t;  // element index with zeros at the tile's positions
    for (unsigned m = 0; m < ta.tb; ++m) {
      const uint64_t lo = (1ull << ta.apos[m]) - 1;
      base = ((base & ~lo) << 1) | (base & lo);
    }
    return a + base;
  };
  auto permute_store = [&](E* __restrict__ at) {
    for (unsigned v = tid; v < NV; v += kBlock) {
      Pack p;
      if constexpr (VEC == 1) {
        p = buf[src[v]];
      } else {
#pragma unroll
        for (int c = 0; c < VEC; ++c) p[c] = buf[src[v * VEC + c]];
      }
      __builtin_nontemporal_store(p, reinterpret_cast<Pack*>(at + goff[v]));
    }
  };
  if constexpr (NPV > 0 && VEC > 1) {
    if (blockIdx.x >= ntiles) return;
    __syncthreads();  // goff is read below by other threads than its writers
    const uint64_t stride = gridDim.x;
    Pack pr[NPV];
    uint32_t go[NPV];
#pragma unroll
    for (int i = 0; i < NPV; ++i) go[i] = goff[tid + i * kBlock];
    auto prefetch = [&](uint64_t t) {  // unconditional (callers clamp)
      const E* __restrict__ at = tile_ptr(t);
#pragma unroll
      for (int i = 0; i < NPV; ++i) pr[i] = __builtin_nontemporal_load(reinterpret_cast<const Pack*>(at + go[i]));
    };
    {
      const E* __restrict__ at = tile_ptr(blockIdx.x);
#pragma unroll 1
      for (int i = 0; i < NPV; ++i)
        *reinterpret_cast<Pack*>(buf + (tid + i * kBlock) * VEC) = __builtin_nontemporal_load(reinterpret_cast<const Pack*>(at + goff[tid + i * kBlock]));
    }
    prefetch(blockIdx.x + stride < ntiles ? blockIdx.x + stride : blockIdx.x);
    for (uint64_t t = blockIdx.x; t < ntiles; t += stride) {
      __syncthreads();
      permute_store(tile_ptr(t));
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NPV; ++i) *reinterpret_cast<Pack*>(buf + (tid + i * kBlock) * VEC) = pr[i];
      prefetch(t + 2 * stride < ntiles ? t + 2 * stride : t);
    }
  } else {
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
      E* __restrict__ at = tile_ptr(t);
      __syncthreads();
      for (unsigned v = tid; v < NV; v += kBlock)
        *reinterpret_cast<Pack*>(buf + v * VEC) = __builtin_nontemporal_load(reinterpret_cast<const Pack*>(at + goff[v]));
      __syncthreads();
      permute_store(at);
    }
  }
}

// Out-of-place gather for large s: out[x] = in[(x & ~(S-1)) | perm(x & (S-1))].
template <typename E>
__global__ void __launch_bounds__(kBlock)
swap_gather_kernel(const E* __restrict__ in, E* __restrict__ out, const SwapArg sa,
                   const uint64_t size) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const uint64_t S = 1ull << sa.s;
  for (uint64_t x = (uint64_t)blockIdx.x * kBlock + threadIdx.x; x < size; x += stride) {
    uint64_t y = x & ~(S - 1);
    for (unsigned i = 0; i < sa.s; ++i) y |= ((x >> i) & 1ull) << sa.pos[i];
    out[x] = in[y];
  }
}

// ---------------------------------------------------------------------------------
// permute_bits: out-of-place permutation of ARBITRARY index bits, dst[x] = src[pi(x)],
// where bit i of x moves to bit perm[i] of pi(x).  Generalises swap (which only moves the
// low s bits) to the whole index; used to bring qubits into the exchange slots of the
// multi-GPU shard exchange and to restore the canonical order.  Only the moved bits cost
// index arithmetic; if bits 0..1 are fixed the copy runs on 16-byte vectors.
// ---------------------------------------------------------------------------------
// Moved bits are grouped into FIELDS: runs of consecutive destination bits whose sources are
// consecutive too (a rotation of a block of qubits is one field, whatever its width), so the
// index arithmetic is one shift + mask per run and any permutation of up to 62 bits fits.
constexpr int kPermMaxFields = 62;
struct PermArg {
  unsigned nfields;
  unsigned char from[kPermMaxFields];   // lowest destination-index bit of the field ...
  unsigned char to[kPermMaxFields];     // ... lands at this source-index bit
  unsigned char len[kPermMaxFields];    // field width in bits
  uint64_t fixed_mask; // bits that stay where they are
};

template <typename E, int VEC>
__global__ void __launch_bounds__(kBlock)
permute_bits_kernel(const E* __restrict__ src, E* __restrict__ dst, const PermArg pa,
                    const uint64_t nunits /* 2^n / VEC */) {
  struct alignas(sizeof(E) * VEC) Pack { E e[VEC]; };
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t u = (uint64_t)blockIdx.x * kBlock + threadIdx.x; u < nunits; u += stride) {
    const uint64_t x = u * VEC;
    uint64_t y = x & pa.fixed_mask;
#pragma unroll 4
    for (unsigned i = 0; i < pa.nfields; ++i) y |= ((x >> pa.from[i]) & ((1ull << pa.len[i]) - 1)) << pa.to[i];
    *reinterpret_cast<Pack*>(dst + x) = *reinterpret_cast<const Pack*>(src + y);
  }
}

// ---------------------------------------------------------------------------------
// exchange_pack: the local half of the multi-GPU qubit exchange.  The shard (2^m elements per
// plane) is cut into G = 2^g chunks by its top g LOCAL index bits; chunk j belongs to rank j after
// the exchange.  One pass reads the plane(s) through an optional local bit permutation (the
// eviction that brings the outgoing qubits to the top g bits -- folded in here instead of a pass
// of its own) and writes every chunk to its own destination base:
//   * RCCL transport: dst[j] = slot j of the local send buffer (then ncclSend / ncclRecv);
//   * peer-to-peer transport: dst[j] = slot `rank` of rank j's receive buffer, mapped through HIP
//     IPC -- the stores travel over xGMI and no second pass exists at all.
// Both planes in one launch (planes = 2) or one plane per launch (so that the transfer of the first
// plane overlaps the packing of the second).
// ---------------------------------------------------------------------------------
constexpr int kMaxShardRanks = 16;
struct ExchArg {
  unsigned g, m, planes;
  PermArg perm;                       // identity: nfields = 0
  void* dst[kMaxShardRanks][2];       // [chunk][plane]
};

template <typename E, int VEC>
__global__ void __launch_bounds__(kBlock)
exchange_pack_kernel(const E* __restrict__ src0, const E* __restrict__ src1, const ExchArg a,
                     const uint64_t nunits /* 2^m / VEC */) {
  struct alignas(sizeof(E) * VEC) Pack { E e[VEC]; };
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const unsigned cbits = a.m - a.g;
  const uint64_t wmask = (1ull << cbits) - 1;
  for (uint64_t u = (uint64_t)blockIdx.x * kBlock + threadIdx.x; u < nunits; u += stride) {
    const uint64_t x = u * VEC;
    uint64_t y = x & a.perm.fixed_mask;
#pragma unroll 4
    for (unsigned i = 0; i < a.perm.nfields; ++i)
      y |= ((x >> a.perm.from[i]) & ((1ull << a.perm.len[i]) - 1)) << a.perm.to[i];
    const unsigned j = (unsigned)(x >> cbits);
    const uint64_t w = x & wmask;
    *reinterpret_cast<Pack*>(reinterpret_cast<E*>(a.dst[j][0]) + w) = *reinterpret_cast<const Pack*>(src0 + y);
    if (a.planes == 2)
      *reinterpret_cast<Pack*>(reinterpret_cast<E*>(a.dst[j][1]) + w) = *reinterpret_cast<const Pack*>(src1 + y);
  }
}

// ---------------------------------------------------------------------------------
// bitperm_tile: ANY permutation of index bits in ONE HBM pass at full line granularity (round 3).
//
// dst[x] = src[pi(x)], dst bit i <-> src bit perm[i] (the convention of hq_permute_bits / swap_*).  The gather kernels
// above read 16-byte vectors wherever the permutation sends them, so a moved low bit costs partial cache lines
// (random permutation above bit 3: 2x the algorithmic read traffic, 3.15 TB/s), and low-bit swaps of more than 13 bits
// took two full passes.  Here a workgroup owns a TILE of 2^tb elements spanned by the dst bits
//     T = {0..c-1}  u  perm^-1({0..c-1})  (+ the next lowest dst bits / sources of the next lowest src bits up to tb),
// c = 5 (4-byte) / 4 (8-byte elements): the tile is a union of >= 128-byte runs on BOTH sides.  It is read in SOURCE
// order (tile-local source index u: bit k of u <-> src bit spos[k], 16-byte vectors of contiguous runs), dropped into LDS
// linearly, read back in DESTINATION order (t: bit a <-> dst bit tpos[a]; u = sigma(t) is a bit permutation, so
// every address is an XOR of per-thread, per-iteration and per-component terms computed once per kernel) and stored
// as 16-byte vectors of contiguous runs.  A host-chosen XOR swizzle (source-index bits >= 5 folded into free bits
// of [VB, 5)) spreads the element reads of a half-wave over the banks.  Out of place for any permutation; IN PLACE
// (src == dst) whenever every moved bit lies inside the tile (low-bit swaps up to 15 bits: 128 KiB of LDS).
// The destination may be split into 2^g chunks by its top g bits, each with its own base pointer (the multi-GPU
// exchange: local send slots or the peers' receive buffers).
// ---------------------------------------------------------------------------------
constexpr int kBitPermMaxTile = 16;
struct BitPermArg {
  unsigned tb, m, cbits, planes;            // tile bits, index bits, m - g (chunk-local bits), planes per launch
  unsigned char tpos[kBitPermMaxTile];      // dst positions of the tile bits, ascending
  unsigned char spos[kBitPermMaxTile];      // src positions of the tile-local source index bits, ascending
  unsigned char sigma[kBitPermMaxTile];     // dst-local bit a -> source-local bit
  unsigned nsw;
  unsigned char sw_hi[4], sw_lo[4];         // LDS element address ^= bit(u, sw_hi) << sw_lo
  unsigned nfields;                         // tile base: runs of non-tile dst bits -> src bits
  unsigned char f_from[48], f_to[48], f_len[48];
  unsigned nb;                              // bits the tile number skips: the tile bits (+ the half bit in SPLIT mode), ascending
  unsigned char bpos[kBitPermMaxTile + 1];
  uint64_t half_x, half_y;                  // SPLIT: 1 << b (dst side), 1 << perm[b] (src side)
  void* dst[kMaxShardRanks][2];             // [chunk][plane]
};

// SPLIT = 1, 2, 3 (in place, BLOCK = 1024, NV = 8): ONE bit more than 128 KiB of LDS hold.  The 2^(tb+1)-element block is
// cut by a moved dst bit b (x_b = v) on the destination side and by p = perm[b] (y_p = v) on the source side; the four
// address quarters Q(b, p) are handled in the order  read S0 = Q(0,0) u Q(1,0) -> LDS;  read Q(0,1) -> REGISTERS;
// write D0 = Q(0,0) u Q(0,1);  LDS <- S1 = Q(0,1) (from the registers) u Q(1,1) (from memory, still untouched);
// write D1 = Q(1,0) u Q(1,1): every element is read once before its address is overwritten, one HBM pass.  On the source
// side b is iteration bit SPLIT-1 of a thread's eight vectors, so the register-held quarter is a static half of them.
template <typename E, int BLOCK, int NV, bool VREAD, int SPLIT = 0>
__global__ void __launch_bounds__(BLOCK)
bitperm_tile_kernel(const E* __restrict__ src0, const E* __restrict__ src1, const BitPermArg a, const uint64_t ntiles) {
  constexpr int VEC = 16 / (int)sizeof(E);
  constexpr unsigned VB = VEC == 4 ? 2 : 1;
  typedef E PackV __attribute__((ext_vector_type(VEC)));
  HQ_DYN_LDS(smem);
  E* buf = reinterpret_cast<E*>(smem);
  HQ_LDS unsigned char* dptr[kMaxShardRanks][2];
  const unsigned tid = threadIdx.x;
  if (tid < 2 * kMaxShardRanks) dptr[tid >> 1][tid & 1] = reinterpret_cast<unsigned char*>(a.dst[tid >> 1][tid & 1]);
  // ---- per-thread address terms (every map below is a bit permutation or an XOR of them: terms of disjoint parts of
  // the thread's vector index combine by OR / XOR)
  auto swz = [&](unsigned u) {
    for (unsigned k = 0; k < a.nsw; ++k) u ^= ((u >> a.sw_hi[k]) & 1u) << a.sw_lo[k];
    return u;
  };
  auto src_off = [&](unsigned u) {  // source-local index -> element offset inside the source plane
    uint64_t y = 0;
    for (unsigned k = VB; k < a.tb; ++k) y |= (uint64_t)((u >> k) & 1u) << a.spos[k];
    return y;
  };
  auto dst_off = [&](unsigned t) {  // dst-local index -> dst index bits
    uint64_t x = 0;
    for (unsigned k = VB; k < a.tb; ++k) x |= (uint64_t)((t >> k) & 1u) << a.tpos[k];
    return x;
  };
  auto sig = [&](unsigned t) {  // dst-local index -> (swizzled) LDS element address of its source
    unsigned u = 0;
    for (unsigned k = 0; k < a.tb; ++k) u |= ((t >> k) & 1u) << a.sigma[k];
    return swz(u);
  };
  // thread part (VGPRs) and iteration part (wave-uniform: SGPRs) of every address
  const unsigned e_tid = tid << VB;  // first element of the thread's vector 0, in either order
  const uint64_t y_tid = src_off(e_tid), x_tid = dst_off(e_tid);
  const unsigned w_tid = swz(e_tid), r_tid = sig(e_tid);
  uint64_t y_it[NV], x_it[NV];
  unsigned w_it[NV], r_it[NV], rc[VEC];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const unsigned e = ((unsigned)i * BLOCK) << VB;
    y_it[i] = src_off(e);
    w_it[i] = swz(e);
    x_it[i] = dst_off(e);
    r_it[i] = sig(e);
  }
#pragma unroll
  for (int c = 0; c < VEC; ++c) rc[c] = sig((unsigned)c);
  const uint64_t wmask = (1ull << a.cbits) - 1;
  auto bases = [&](uint64_t h, uint64_t& xb, uint64_t& yb) {
    xb = h;  // deposit the tile number into the non-tile dst bits
    for (unsigned k = 0; k < a.nb; ++k) {
      const uint64_t lo = (1ull << a.bpos[k]) - 1;
      xb = ((xb & ~lo) << 1) | (xb & lo);
    }
    yb = 0;
    for (unsigned f = 0; f < a.nfields; ++f) yb |= ((xb >> a.f_from[f]) & ((1ull << a.f_len[f]) - 1)) << a.f_to[f];
  };
  const uint64_t total = ntiles * a.planes;
  auto load_tile = [&](uint64_t ht, PackV (&v)[NV]) {
    const uint64_t h = ht < ntiles ? ht : ht - ntiles;
    const E* __restrict__ sp = ht < ntiles ? src0 : src1;
    uint64_t xb, yb;
    bases(h, xb, yb);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const PackV*>(sp + (yb | y_tid | y_it[i])));
  };
  auto fill = [&](const PackV (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<PackV*>(buf + (w_tid ^ w_it[i])) = v[i];
  };
  auto permute_store = [&](uint64_t ht) {
    const uint64_t h = ht < ntiles ? ht : ht - ntiles;
    const unsigned plane = ht < ntiles ? 0u : 1u;
    uint64_t xb, yb;
    bases(h, xb, yb);
    constexpr int CH = NV < 4 ? NV : 4;  // LDS reads and stores in batches of four vectors (16 live result registers)
#pragma unroll
    for (int i0 = 0; i0 < NV; i0 += CH) {
      PackV o[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int i = i0 + j;
        if constexpr (VREAD) {
          o[j] = *reinterpret_cast<const PackV*>(buf + (r_tid ^ r_it[i]));
        } else {
#pragma unroll
          for (int c = 0; c < VEC; ++c) o[j][c] = buf[r_tid ^ r_it[i] ^ rc[c]];
        }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const uint64_t x = xb | x_tid | x_it[i0 + j];
        E* dp = reinterpret_cast<E*>(dptr[x >> a.cbits][plane]) + (x & wmask);
        __builtin_nontemporal_store(o[j], reinterpret_cast<PackV*>(dp));
      }
    }
  };
  if constexpr (SPLIT > 0) {
    static_assert(NV == 8, "split mode: eight vectors per thread");
    constexpr int UB = SPLIT - 1;  // iteration bit that is address bit b on the source side
    auto store_half = [&](uint64_t xb) {
      constexpr int CH = VREAD ? 4 : 2;  // element gathers: two vectors at a time keep the kernel inside 128 registers
#pragma unroll
      for (int i0 = 0; i0 < NV; i0 += CH) {
        PackV o[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const int i = i0 + j;
          if constexpr (VREAD) {
            o[j] = *reinterpret_cast<const PackV*>(buf + (r_tid ^ r_it[i]));
          } else {
#pragma unroll
            for (int c = 0; c < VEC; ++c) o[j][c] = buf[r_tid ^ r_it[i] ^ rc[c]];
          }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const uint64_t x = xb | x_tid | x_it[i0 + j];
          __builtin_nontemporal_store(o[j], reinterpret_cast<PackV*>(reinterpret_cast<E*>(dptr[0][0]) + x));
        }
      }
    };
    for (uint64_t h = blockIdx.x; h < ntiles; h += gridDim.x) {  // ntiles = number of 2^(tb+1) blocks
      uint64_t xb, yb;
      bases(h, xb, yb);
      PackV v[NV], q[NV / 2];
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const PackV*>(src0 + (yb | y_tid | y_it[i])));
      __syncthreads();  // the previous block's LDS reads are done (and dptr is visible)
      fill(v);
#pragma unroll
      for (int i = 0, k = 0; i < NV; ++i)
        if (!((i >> UB) & 1)) q[k++] = __builtin_nontemporal_load(reinterpret_cast<const PackV*>(src0 + (yb | a.half_y | y_tid | y_it[i])));
      // Q(0,1) must have been READ by every thread before any thread overwrites it with D0: the loads are waited for HERE,
      // in front of the barrier (a load still in flight across the barrier could see another wave's store)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      store_half(xb);
      __syncthreads();  // also keeps the loads below from being scheduled into the gathers above (register pressure)
#pragma unroll
      for (int i = 0, k = 0; i < NV; ++i) {
        if (!((i >> UB) & 1)) v[i] = q[k++];
        else v[i] = __builtin_nontemporal_load(reinterpret_cast<const PackV*>(src0 + (yb | a.half_y | y_tid | y_it[i])));
      }
      fill(v);
      __syncthreads();
      store_half(xb | a.half_x);
    }
  } else {
    for (uint64_t ht = blockIdx.x; ht < total; ht += gridDim.x) {
      PackV v[NV];
      load_tile(ht, v);
      __syncthreads();  // the previous tile's LDS reads are done (and dptr is visible)
      fill(v);
      __syncthreads();
      permute_store(ht);
    }
  }
}

}  // namespace hq
