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
  printf("bitperm emulation: %d cases ran, %d skipped by the planner, %d failed; worst half-wave bank multiplicity %d\n", ran, skipped, bad, worst);
  return bad != 0;
}
