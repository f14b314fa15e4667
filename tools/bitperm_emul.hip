// Host-only emulation of bitperm_tile_kernel's index arithmetic with the REAL planner (hq_bitperm.h): runs in the build
// container (no GPU).  For random permutations and low-bit swaps it walks every tile / thread / iteration exactly as the
// kernel does (source-order load -> swizzled LDS slot -> destination-order read -> store address) and checks
//   * dst[x] == src[pi(x)] for every x, every element written exactly once,
//   * every 16-byte access is aligned and the LDS slots of a tile are a bijection,
//   * (report) the worst bank multiplicity of a half-wave's element reads.
// Build + run:  hipcc -std=c++17 -O1 tools/bitperm_emul.hip -o /tmp/bitperm_emul && /tmp/bitperm_emul
#include "../hybridq_amd/csrc/hq_bitperm.h"

#include <cstdint>
#include <random>

namespace hq {  // the planner's only external dependencies
Context& ctx() { static Context c; return c; }
int fail(const std::string& m) { fprintf(stderr, "fail: %s\n", m.c_str()); return 1; }
}

template <typename E>
static int emulate(const std::vector<unsigned>& perm, bool inplace, int* worst_conflict) {
  using namespace hq;
  constexpr unsigned VB = sizeof(E) == 4 ? 2 : 1, VEC = 1u << VB;
  const unsigned m = (unsigned)perm.size();
  BitPermPlan P;
  if (!plan_bitperm<E>(perm.data(), m, inplace, P)) return -1;
  const BitPermArg& a = P.a;
  const unsigned BLOCK = P.block, NV = P.nv;
  const uint64_t size = 1ull << m, ntiles = 1ull << (m - a.tb);
  std::vector<uint64_t> dst(size, ~0ull), lds(1ull << a.tb);
  auto swz = [&](unsigned u) { for (unsigned k = 0; k < a.nsw; ++k) u ^= ((u >> a.sw_hi[k]) & 1u) << a.sw_lo[k]; return u; };
  auto src_off = [&](unsigned u) { uint64_t y = 0; for (unsigned k = VB; k < a.tb; ++k) y |= (uint64_t)((u >> k) & 1u) << a.spos[k]; return y; };
  auto dst_off = [&](unsigned t) { uint64_t x = 0; for (unsigned k = VB; k < a.tb; ++k) x |= (uint64_t)((t >> k) & 1u) << a.tpos[k]; return x; };
  auto sig = [&](unsigned t) { unsigned u = 0; for (unsigned k = 0; k < a.tb; ++k) u |= ((t >> k) & 1u) << a.sigma[k]; return swz(u); };
  if (BLOCK * NV * VEC != (1u << a.tb)) { printf("tile shape mismatch\n"); return 1; }
  for (uint64_t h = 0; h < ntiles; ++h) {
    uint64_t xb = h;
    for (unsigned k = 0; k < a.tb; ++k) { const uint64_t lo = (1ull << a.tpos[k]) - 1; xb = ((xb & ~lo) << 1) | (xb & lo); }
    uint64_t yb = 0;
    for (unsigned f = 0; f < a.nfields; ++f) yb |= ((xb >> a.f_from[f]) & ((1ull << a.f_len[f]) - 1)) << a.f_to[f];
    std::fill(lds.begin(), lds.end(), ~0ull);
    for (unsigned tid = 0; tid < BLOCK; ++tid)
      for (unsigned i = 0; i < NV; ++i) {
        const unsigned e_tid = tid << VB, e_it = (i * BLOCK) << VB;
        const uint64_t y = yb | src_off(e_tid) | src_off(e_it);
        if (y % VEC) { printf("unaligned load\n"); return 1; }
        const unsigned w = swz(e_tid) ^ swz(e_it);
        if (w % VEC) { printf("unaligned LDS write\n"); return 1; }
        for (unsigned c = 0; c < VEC; ++c) {
          if (lds[w + c] != ~0ull) { printf("LDS slot written twice\n"); return 1; }
          lds[w + c] = y + c;  // "value" = source index
        }
      }
    for (unsigned tid = 0; tid < BLOCK; ++tid)
      for (unsigned i = 0; i < NV; ++i) {
        const unsigned e_tid = tid << VB, e_it = (i * BLOCK) << VB;
        const uint64_t x = xb | dst_off(e_tid) | dst_off(e_it);
        if (x % VEC) { printf("unaligned store\n"); return 1; }
        for (unsigned c = 0; c < VEC; ++c) {
          const unsigned r = sig(e_tid) ^ sig(e_it) ^ sig(c);
          if (P.vread && r != ((sig(e_tid) ^ sig(e_it)) + c)) { printf("vread: components not contiguous\n"); return 1; }
          if (dst[x + c] != ~0ull) { printf("dst written twice\n"); return 1; }
          dst[x + c] = lds[r];
        }
      }
    if (h == 0 && worst_conflict) {  // bank multiplicity of the element reads of each half-wave, component 0
      for (unsigned wv = 0; wv < BLOCK / 32; ++wv) {
        int cnt[32] = {0};
        for (unsigned l = 0; l < 32; ++l) {
          const unsigned r = sig((wv * 32 + l) << VB);
          cnt[(sizeof(E) == 4 ? r : r) & 31]++;
        }
        for (int b = 0; b < 32; ++b) *worst_conflict = std::max(*worst_conflict, cnt[b]);
      }
    }
  }
  for (uint64_t x = 0; x < size; ++x) {
    uint64_t y = 0;
    for (unsigned i = 0; i < m; ++i) y |= ((x >> i) & 1ull) << perm[i];
    if (dst[x] != y) { printf("wrong element at %llu: got %llu want %llu\n", (unsigned long long)x, (unsigned long long)dst[x], (unsigned long long)y); return 1; }
  }
  return 0;
}

// SPLIT mode (in place, one moved bit more than the tile holds): the kernel's exact order of reads and writes on ONE array
template <typename E>
static int emulate_split(const std::vector<unsigned>& perm) {
  using namespace hq;
  constexpr unsigned VB = sizeof(E) == 4 ? 2 : 1, VEC = 1u << VB;
  const unsigned m = (unsigned)perm.size();
  BitPermPlan P;
  if (!plan_bitperm<E>(perm.data(), m, true, P)) return -1;
  if (!P.split) return -2;
  const BitPermArg& a = P.a;
  const unsigned BLOCK = P.block, NV = P.nv, UB = (unsigned)P.split - 1;
  if (BLOCK != 1024 || NV != 8) { printf("split: bad shape\n"); return 1; }
  const uint64_t size = 1ull << m, nblocks = 1ull << (m - a.nb);
  std::vector<uint64_t> mem(size), lds(1ull << a.tb);
  for (uint64_t i = 0; i < size; ++i) mem[i] = i;
  auto swz = [&](unsigned u) { for (unsigned k = 0; k < a.nsw; ++k) u ^= ((u >> a.sw_hi[k]) & 1u) << a.sw_lo[k]; return u; };
  auto src_off = [&](unsigned u) { uint64_t y = 0; for (unsigned k = VB; k < a.tb; ++k) y |= (uint64_t)((u >> k) & 1u) << a.spos[k]; return y; };
  auto dst_off = [&](unsigned t) { uint64_t x = 0; for (unsigned k = VB; k < a.tb; ++k) x |= (uint64_t)((t >> k) & 1u) << a.tpos[k]; return x; };
  auto sig = [&](unsigned t) { unsigned u = 0; for (unsigned k = 0; k < a.tb; ++k) u |= ((t >> k) & 1u) << a.sigma[k]; return swz(u); };
  for (uint64_t h = 0; h < nblocks; ++h) {
    uint64_t xb = h;
    for (unsigned k = 0; k < a.nb; ++k) { const uint64_t lo = (1ull << a.bpos[k]) - 1; xb = ((xb & ~lo) << 1) | (xb & lo); }
    uint64_t yb = 0;
    for (unsigned f = 0; f < a.nfields; ++f) yb |= ((xb >> a.f_from[f]) & ((1ull << a.f_len[f]) - 1)) << a.f_to[f];
    std::vector<std::vector<uint64_t>> v(BLOCK * NV, std::vector<uint64_t>(VEC)), q(BLOCK * NV, std::vector<uint64_t>(VEC));
    auto ld = [&](uint64_t base, unsigned tid, unsigned i, std::vector<uint64_t>& out) {
      const uint64_t y = base | src_off(tid << VB) | src_off((i * BLOCK) << VB);
      for (unsigned c = 0; c < VEC; ++c) out[c] = mem[y + c];
    };
    auto fill = [&]() {
      for (unsigned tid = 0; tid < BLOCK; ++tid) for (unsigned i = 0; i < NV; ++i) {
        const unsigned w = swz(tid << VB) ^ swz((i * BLOCK) << VB);
        for (unsigned c = 0; c < VEC; ++c) lds[w + c] = v[tid * NV + i][c];
      }
    };
    auto store_half = [&](uint64_t xbase) {
      for (unsigned tid = 0; tid < BLOCK; ++tid) for (unsigned i = 0; i < NV; ++i) {
        const uint64_t x = xbase | dst_off(tid << VB) | dst_off((i * BLOCK) << VB);
        for (unsigned c = 0; c < VEC; ++c) mem[x + c] = lds[sig(tid << VB) ^ sig((i * BLOCK) << VB) ^ sig(c)];
      }
    };
    for (unsigned tid = 0; tid < BLOCK; ++tid) for (unsigned i = 0; i < NV; ++i) ld(yb, tid, i, v[tid * NV + i]);
    for (unsigned tid = 0; tid < BLOCK; ++tid) for (unsigned i = 0; i < NV; ++i) if (!((i >> UB) & 1)) ld(yb | a.half_y, tid, i, q[tid * NV + i]);
    fill();
    store_half(xb);
    for (unsigned tid = 0; tid < BLOCK; ++tid) for (unsigned i = 0; i < NV; ++i) {
      if (!((i >> UB) & 1)) v[tid * NV + i] = q[tid * NV + i]; else ld(yb | a.half_y, tid, i, v[tid * NV + i]);
    }
    fill();
    store_half(xb | a.half_x);
  }
  for (uint64_t x = 0; x < size; ++x) {
    uint64_t y = 0;
    for (unsigned i = 0; i < m; ++i) y |= ((x >> i) & 1ull) << perm[i];
    if (mem[x] != y) { printf("split: wrong element at %llu\n", (unsigned long long)x); return 1; }
  }
  return 0;
}

int main() {
  std::mt19937 rng(7);
  int bad = 0, ran = 0, skipped = 0, worst = 0;
  for (int trial = 0; trial < 60; ++trial) {
    const unsigned m = 14 + trial % 6;
    std::vector<unsigned> perm(m);
    for (unsigned i = 0; i < m; ++i) perm[i] = i;
    const int kind = trial % 5;
    if (kind == 0) std::shuffle(perm.begin(), perm.end(), rng);                       // everything moves
    else if (kind == 1) std::shuffle(perm.begin() + 4, perm.end(), rng);              // bits 0-3 stay
    else if (kind == 2) std::reverse(perm.begin(), perm.end());                        // bit reversal (worst banks)
    else if (kind == 3) { std::swap(perm[1], perm[m - 1]); std::swap(perm[3], perm[m - 2]); std::swap(perm[7], perm[m - 3]); }  // eviction
    else std::rotate(perm.begin(), perm.begin() + 3, perm.end());
    int w = 0;
    int r = emulate<uint32_t>(perm, false, &w);
    if (r < 0) ++skipped; else { ++ran; bad += r; worst = std::max(worst, w); }
    r = emulate<uint64_t>(perm, false, &w);
    if (r < 0) ++skipped; else { ++ran; bad += r; }
  }
  for (unsigned s : {13u, 14u, 15u}) {  // in-place low-bit swaps: the tile holds every moved bit
    std::vector<unsigned> perm(s + 3);
    for (unsigned i = 0; i < perm.size(); ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.begin() + s, rng);
    int r = emulate<uint32_t>(perm, true, nullptr);
    if (r < 0) ++skipped; else { ++ran; bad += r; }
    if (s <= 14) { r = emulate<uint64_t>(perm, true, nullptr); if (r < 0) ++skipped; else { ++ran; bad += r; } }
  }
  int split_ran = 0, split_skipped = 0;  // skipped: no suitable half bit among the three highest tile bits -> the two-pass path
  for (int trial = 0; trial < 8; ++trial) {  // 16 (4-byte) / 15 (8-byte) moved low bits, in place, a few fixed bits above
    for (int wide = 0; wide < 2; ++wide) {
      const unsigned s = wide ? 15u : 16u;
      std::vector<unsigned> perm(s + 1 + trial % 2);
      for (unsigned i = 0; i < perm.size(); ++i) perm[i] = i;
      if (trial == 0) std::rotate(perm.begin(), perm.begin() + 3, perm.begin() + s);
      else if (trial == 1) std::reverse(perm.begin(), perm.begin() + s);
      else {  // random derangement-ish: shuffle until no fixed point
        do { std::shuffle(perm.begin(), perm.begin() + s, rng); } while ([&] { for (unsigned i = 0; i < s; ++i) if (perm[i] == i) return true; return false; }());
      }
      const int r = wide ? emulate_split<uint64_t>(perm) : emulate_split<uint32_t>(perm);
      if (r == -1 || r == -2) ++split_skipped; else { ++ran; ++split_ran; bad += r; }
    }
  }
  printf("split mode: %d cases ran, %d left to the two-pass path\n", split_ran, split_skipped);
  printf("bitperm emulation: %d cases ran, %d skipped by the planner, %d failed; worst half-wave bank multiplicity %d\n", ran, skipped, bad, worst);
  return bad != 0;
}
