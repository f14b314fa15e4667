// hq_plan.hip -- host-side planning behind the C ABI (no device code): the cache-blocked schedule of a run of matrix
// gates (hq_plan_blocked) and the greedy fusion it is built on.  Same algorithms as hybridq_amd/blocking.py /
// hybridq_amd/fusion.py (which stay as the readable statement and the cross-check of the tests), moved out of
// per-gate Python because the caller waits for them: at n = 30 the Python planner needed 50-114 ms for a 137 ms gate
// loop.  The fusion rule is the reference's (hybridq/circuit/utils.py:606-669: walk the layers from the
// newest to the oldest, remember the oldest layer the gate may merge into, stop at the first layer it neither avoids nor
// commutes with), here always with matrix commutation and with the "commutes to rounding" tolerance of the planner.
#include <complex>

#include "hq_common.h"

namespace hq {
namespace plan {

using cd = std::complex<double>;

struct Gate {
  std::vector<unsigned> q;  // q[0] = most significant bit of the matrix index (gate.qubits order)
  std::vector<cd> U;        // row-major 2^k x 2^k
};

static std::vector<unsigned> sorted_union(const std::vector<unsigned>& a, const std::vector<unsigned>& b) {
  std::vector<unsigned> u(a);
  u.insert(u.end(), b.begin(), b.end());
  std::sort(u.begin(), u.end());
  u.erase(std::unique(u.begin(), u.end()), u.end());
  return u;
}
static uint64_t mask_of(const std::vector<unsigned>& q) {
  uint64_t m = 0;
  for (unsigned x : q) m |= 1ull << x;
  return m;
}

// matrix of (U, qs) on the ordered qubit list Q (Q[0] = most significant bit)
static std::vector<cd> embed(const std::vector<cd>& U, const std::vector<unsigned>& qs, const std::vector<unsigned>& Q) {
  const unsigned k = (unsigned)Q.size(), kg = (unsigned)qs.size();
  if (qs == Q) return U;
  const size_t D = (size_t)1 << k, Dg = (size_t)1 << kg;
  std::vector<unsigned> gate_bit(kg);
  uint64_t used = 0;
  for (unsigned j = 0; j < kg; ++j) {
    const unsigned a = (unsigned)(std::find(Q.begin(), Q.end(), qs[j]) - Q.begin());
    gate_bit[j] = k - 1 - a;
    used |= 1ull << gate_bit[j];
  }
  std::vector<unsigned> rest_bit;
  for (unsigned b = 0; b < k; ++b)
    if (!((used >> b) & 1)) rest_bit.push_back(b);
  std::vector<size_t> gpart(Dg, 0);
  for (size_t g = 0; g < Dg; ++g)
    for (unsigned j = 0; j < kg; ++j) gpart[g] |= ((g >> (kg - 1 - j)) & 1) << gate_bit[j];
  std::vector<cd> M(D * D, cd(0, 0));
  const size_t R = (size_t)1 << (k - kg);
  for (size_t r = 0; r < R; ++r) {
    size_t rp = 0;
    for (unsigned j = 0; j < rest_bit.size(); ++j) rp |= ((r >> j) & 1) << rest_bit[j];
    for (size_t a = 0; a < Dg; ++a)
      for (size_t b = 0; b < Dg; ++b) M[(rp | gpart[a]) * D + (rp | gpart[b])] = U[a * Dg + b];
  }
  return M;
}

static std::vector<cd> matmul(const std::vector<cd>& A, const std::vector<cd>& B, size_t D) {
  std::vector<cd> C(D * D, cd(0, 0));
  for (size_t i = 0; i < D; ++i)
    for (size_t l = 0; l < D; ++l) {
      const cd a = A[i * D + l];
      if (a == cd(0, 0)) continue;  // embedded matrices are mostly zero
      const cd* b = &B[l * D];
      cd* c = &C[i * D];
      for (size_t j = 0; j < D; ++j) c[j] += a * b[j];
    }
  return C;
}

// widest union of two gates whose commutator is still evaluated (2^24 complex entries per operand)
constexpr size_t kMaxCommuteQubits = 12;

// |AB - BA| <= tol + tol |BA| entry by entry (np.allclose's test); one row first: generic gates fail there
static bool commute(const std::vector<cd>& U1, const std::vector<unsigned>& q1, const std::vector<cd>& U2,
                    const std::vector<unsigned>& q2, double tol) {
  if (!(mask_of(q1) & mask_of(q2))) return true;
  const std::vector<unsigned> Q = sorted_union(q1, q2);
  if (Q.size() > kMaxCommuteQubits) return false;  // not tested (no reordering): three 2^(2|Q|) matrices would be needed
  const size_t D = (size_t)1 << Q.size();
  const std::vector<cd> A = embed(U1, q1, Q), B = embed(U2, q2, Q);
  for (size_t j = 0; j < D; ++j) {
    cd ab(0, 0), ba(0, 0);
    for (size_t l = 0; l < D; ++l) { ab += A[l] * B[l * D + j]; ba += B[l] * A[l * D + j]; }
    if (std::abs(ab - ba) > tol + tol * std::abs(ba)) return false;
  }
  const std::vector<cd> AB = matmul(A, B, D), BA = matmul(B, A, D);
  for (size_t e = 0; e < D * D; ++e)
    if (std::abs(AB[e] - BA[e]) > tol + tol * std::abs(BA[e])) return false;
  return true;
}

// B == inv(A) within np.allclose(inv(A), B, atol): Gauss-Jordan with partial pivoting (a singular A has no inverse gate)
static bool is_inverse(const std::vector<cd>& A, const std::vector<cd>& B, size_t D, double atol) {
  std::vector<cd> M(A), I(D * D, cd(0, 0));
  for (size_t i = 0; i < D; ++i) I[i * D + i] = cd(1, 0);
  for (size_t c = 0; c < D; ++c) {
    size_t piv = c;
    for (size_t r = c + 1; r < D; ++r)
      if (std::abs(M[r * D + c]) > std::abs(M[piv * D + c])) piv = r;
    if (std::abs(M[piv * D + c]) < 1e-300) return false;
    if (piv != c)
      for (size_t j = 0; j < D; ++j) { std::swap(M[piv * D + j], M[c * D + j]); std::swap(I[piv * D + j], I[c * D + j]); }
    const cd inv = cd(1, 0) / M[c * D + c];
    for (size_t j = 0; j < D; ++j) { M[c * D + j] *= inv; I[c * D + j] *= inv; }
    for (size_t r = 0; r < D; ++r) {
      if (r == c) continue;
      const cd f = M[r * D + c];
      if (f == cd(0, 0)) continue;
      for (size_t j = 0; j < D; ++j) { M[r * D + j] -= f * M[c * D + j]; I[r * D + j] -= f * I[c * D + j]; }
    }
  }
  for (size_t e = 0; e < D * D; ++e)
    if (!(std::abs(I[e] - B[e]) <= atol + 1e-5 * std::abs(B[e]))) return false;
  return true;
}

// fusion.simplify (hybridq/circuit/utils.py:825-866 with insert_from_left, :122-208): identity gates dropped, the circuit
// rebuilt from its last gate backwards, every gate sliding right through the gates it commutes with (no shared qubit, or
// commuting matrices within the reference's fixed 1e-5) and cancelling against the first gate that is its inverse.
// Returns the indices of the surviving gates in their new order.
static std::vector<unsigned> simplify(const std::vector<Gate>& gates, double atol, bool use_mc, unsigned max_nqm, bool remove_id) {
  std::vector<unsigned> keep;
  for (unsigned g = 0; g < gates.size(); ++g) {
    const Gate& G = gates[g];
    bool drop = false;
    if (remove_id && G.q.size() <= max_nqm) {
      const size_t D = (size_t)1 << G.q.size();
      drop = true;
      for (size_t i = 0; i < D && drop; ++i)
        for (size_t j = 0; j < D; ++j) {
          const cd want = i == j ? cd(1, 0) : cd(0, 0);
          if (!(std::abs(G.U[i * D + j] - want) <= atol + (i == j ? 1e-5 : 0.0))) { drop = false; break; }
        }
    }
    if (!drop) keep.push_back(g);
  }
  std::vector<unsigned> out;  // circuit order; gates are inserted from the left
  std::vector<uint64_t> qm(gates.size());
  for (unsigned g : keep) qm[g] = mask_of(gates[g].q);
  for (size_t r = keep.size(); r-- > 0;) {
    const unsigned g = keep[r];
    const Gate& Gg = gates[g];
    bool placed = false;
    for (size_t p = 0; p < out.size(); ++p) {
      const unsigned v = out[p];
      const Gate& Gv = gates[v];
      if (qm[v] == qm[g]) {
        std::vector<unsigned> Q(Gg.q);
        std::sort(Q.begin(), Q.end());
        if (is_inverse(embed(Gg.U, Gg.q, Q), embed(Gv.U, Gv.q, Q), (size_t)1 << Q.size(), atol)) {
          out.erase(out.begin() + (long)p);
          placed = true;
          break;
        }
      }
      bool ok = false;
      if (Gv.q.size() <= max_nqm) {
        ok = !(qm[g] & qm[v]);
        if (!ok && use_mc) ok = commute(Gg.U, Gg.q, Gv.U, Gv.q, 1e-5);  // commutes_with's fixed tolerance (property.py:573)
      }
      if (!ok) {
        out.insert(out.begin() + (long)p, g);
        placed = true;
        break;
      }
    }
    if (!placed) out.push_back(g);
  }
  return out;
}

struct Layer {
  std::vector<unsigned> q;  // sorted
  std::vector<cd> U;
};

struct FuseOptions {
  unsigned max_n = 4;
  bool use_mc = true;           // use_matrix_commutation
  unsigned max_nqm = 10;        // max_n_qubits_matrix
  uint64_t exclude = 0;         // gates touching these qubits are never compressed (utils.py:615-617)
  double tol = 1e-5;            // commutes_with's fixed tolerance, or the planner's "commutes to rounding"
};

static std::vector<Layer> build_layers(const std::vector<const Gate*>& gates, const FuseOptions& o) {
  struct Meta { bool compress, has_matrix; };
  std::vector<Layer> layers;
  std::vector<Meta> meta;
  for (const Gate* g : gates) {
    const uint64_t qm = mask_of(g->q);
    const unsigned nq = (unsigned)g->q.size();
    const bool can = !(qm & o.exclude);
    const bool gate_matrix = o.use_mc && nq <= o.max_nqm;  // utils.py:606-611
    size_t merge_to = layers.size();
    for (size_t i = layers.size(); i-- > 0;) {
      const Layer& L = layers[i];
      const uint64_t cm = mask_of(L.q);
      const unsigned nu = (unsigned)__builtin_popcountll(qm | cm);
      if (can && meta[i].compress && nu <= std::max<unsigned>(o.max_n, std::max<unsigned>((unsigned)L.q.size(), nq))) merge_to = i;  // utils.py:626-630
      if (o.use_mc) {  // utils.py:633-646
        if (!(qm & cm)) continue;
        if (gate_matrix && meta[i].has_matrix && commute(g->U, g->q, L.U, L.q, o.tol)) continue;
      }
      break;
    }
    if (merge_to < layers.size()) {
      Layer& L = layers[merge_to];
      const std::vector<unsigned> Q = sorted_union(L.q, g->q);
      // the new gate acts AFTER everything already in the layer
      L.U = matmul(embed(g->U, g->q, Q), embed(L.U, L.q, Q), (size_t)1 << Q.size());
      L.q = Q;
      meta[merge_to].has_matrix = meta[merge_to].has_matrix && gate_matrix && Q.size() <= o.max_nqm;  // utils.py:660-669
    } else {
      std::vector<unsigned> Q(g->q);
      std::sort(Q.begin(), Q.end());
      layers.push_back(Layer{Q, embed(g->U, g->q, Q)});
      meta.push_back(Meta{can, gate_matrix});
    }
  }
  return layers;
}
static std::vector<Layer> build_layers(const std::vector<const Gate*>& gates, unsigned max_n, double tol) {
  FuseOptions o;
  o.max_n = max_n;
  o.tol = tol;
  return build_layers(gates, o);
}

// the grouping build_layers would produce on qubit sets alone (sliding through disjoint layers only)
static std::vector<uint64_t> dry_layers(const std::vector<uint64_t>& qsets, unsigned kmax) {
  std::vector<uint64_t> layers;
  for (uint64_t q : qsets) {
    size_t merge_to = layers.size();
    for (size_t i = layers.size(); i-- > 0;) {
      const uint64_t cq = layers[i];
      const unsigned nu = (unsigned)__builtin_popcountll(q | cq);
      if (nu <= std::max<unsigned>(kmax, std::max<unsigned>((unsigned)__builtin_popcountll(cq), (unsigned)__builtin_popcountll(q)))) merge_to = i;
      if (!(q & cq)) continue;
      break;
    }
    if (merge_to < layers.size()) layers[merge_to] |= q;
    else layers.push_back(q);
  }
  return layers;
}

static double inner_cost(unsigned k) { return k >= 4 ? 1.9 : 1.0; }  // measured, n = 30 complex64 (blocking.INNER_COST)

struct Rng {  // deterministic, seeded per call
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
  uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s * 0x2545F4914F6CDD1Dull; }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  size_t below(size_t n) { return (size_t)(next() % n); }
};

struct Result {
  unsigned tile_bits = 0;
  std::vector<unsigned> op_kind;        // 0 = plain gate, 1 = blocked pass
  std::vector<unsigned> op_first_gate;  // n_ops + 1
  std::vector<unsigned> op_tile;        // n_ops * tile_bits (zeros for plain gates)
  std::vector<Gate> gates;              // q = POSITIONS, q[0] = most significant matrix bit
  double plan_ms = 0;
};

struct Options {
  unsigned n, tile_bits, low_bits, inner_max /* 0 none, 255 auto */, min_gates, tries, fusion_orders, elem_bytes;
  uint64_t seed;
  double tol;
};

// LDS a pass's operand and address tables need, and what is left beside the tile (hq_apply.hip: a_budget).  128 KiB
// tiles (2^14 complex64 / 2^13 complex128 amplitudes: one 1024-thread workgroup per CU) have 128 wave-iterations per
// gate instead of 64 in their address tables and 31 KiB instead of 15 KiB beside them.
static bool big_tile(const Options& o) { return ((size_t)2 << o.tile_bits) * o.elem_bytes == 128 * 1024; }
static size_t lds_bytes(const std::vector<Layer>& gl, const Options& o) {
  size_t b = 0;
  for (const auto& L : gl) {
    static const size_t a_elems[5] = {0, 0, 256, 256, 1024};
    b += a_elems[std::min<size_t>(L.q.size(), 4)] * o.elem_bytes + (big_tile(o) ? 800 : 544);
  }
  return b;
}
static size_t lds_budget(const Options& o) { return (big_tile(o) ? 31 : 15) * 1024; }

static void plan_blocked(const std::vector<Gate>& gates, const Options& o, Result& out) {
  const size_t G = gates.size();
  std::vector<uint64_t> gp(G);
  std::vector<unsigned> gk(G);
  std::vector<std::vector<unsigned>> qlist(o.n);
  for (size_t g = 0; g < G; ++g) {
    gp[g] = mask_of(gates[g].q);
    gk[g] = (unsigned)gates[g].q.size();
    for (unsigned p : gates[g].q) qlist[p].push_back((unsigned)g);
  }
  std::vector<unsigned> ptr(o.n, 0);
  const uint64_t low = o.low_bits >= 64 ? ~0ull : ((1ull << o.low_bits) - 1);
  Rng rnd(o.seed);
  out.tile_bits = o.tile_bits;
  out.op_first_gate.push_back(0);
  std::vector<int> at_head(G);
  std::vector<double> prio(G);

  struct Cand { std::vector<unsigned> chosen; uint64_t S; std::vector<unsigned> ptr; };
  auto grow = [&](bool program_order) {
    Cand c{{}, low, ptr};
    std::fill(at_head.begin(), at_head.end(), 0);
    std::vector<unsigned> ready;
    auto key_less = [&](unsigned a, unsigned b) { return program_order ? a < b : (prio[a] < prio[b] || (prio[a] == prio[b] && a < b)); };
    for (unsigned p = 0; p < o.n; ++p)
      if (c.ptr[p] < qlist[p].size()) {
        const unsigned g = qlist[p][c.ptr[p]];
        if (++at_head[g] == (int)gk[g] && gk[g] <= 4) ready.push_back(g);
      }
    auto take = [&](unsigned g) {
      c.chosen.push_back(g);
      ready.erase(std::find(ready.begin(), ready.end(), g));
      for (unsigned p : gates[g].q) {
        if (++c.ptr[p] < qlist[p].size()) {
          const unsigned h = qlist[p][c.ptr[p]];
          if (++at_head[h] == (int)gk[h] && gk[h] <= 4) ready.push_back(h);
        }
      }
    };
    while (!ready.empty()) {
      std::vector<unsigned> fit;
      for (unsigned g : ready)
        if (!(gp[g] & ~c.S)) fit.push_back(g);
      if (!fit.empty()) {  // everything ready that already fits
        std::sort(fit.begin(), fit.end(), key_less);
        for (unsigned g : fit) take(g);
        continue;
      }
      std::vector<unsigned> order(ready);
      std::sort(order.begin(), order.end(), key_less);
      int best = -1;
      unsigned best_new = 0;
      for (unsigned g : order) {  // spend spare capacity on the ready gate that needs the fewest new positions
        const unsigned nn = (unsigned)__builtin_popcountll(gp[g] & ~c.S);
        if ((unsigned)__builtin_popcountll(c.S) + nn <= o.tile_bits && (best < 0 || nn < best_new)) { best = (int)g; best_new = nn; }
      }
      if (best < 0) break;
      c.S |= gp[best];
      take((unsigned)best);
    }
    return c;
  };

  auto emit_plain = [&](unsigned g) {
    out.op_kind.push_back(0);
    out.op_tile.insert(out.op_tile.end(), o.tile_bits, 0u);
    out.gates.push_back(gates[g]);
    out.op_first_gate.push_back((unsigned)out.gates.size());
  };

  size_t done = 0;
  while (done < G) {
    Cand best;
    bool have = false;
    for (unsigned t = 0; t < std::max(1u, o.tries); ++t) {
      if (t) for (size_t g = 0; g < G; ++g) prio[g] = rnd.uniform();
      Cand c = grow(t == 0);
      if (!have || c.chosen.size() > best.chosen.size()) { best = std::move(c); have = true; }
    }
    if (best.chosen.empty()) {  // a gate that fits no tile (k > 4): on its own
      unsigned gi = ~0u;
      for (unsigned p = 0; p < o.n; ++p)
        if (ptr[p] < qlist[p].size()) {
          const unsigned g = qlist[p][ptr[p]];
          bool heads_all = true;
          for (unsigned x : gates[g].q) heads_all = heads_all && qlist[x][ptr[x]] == g;
          if (heads_all) gi = std::min(gi, g);
        }
      for (unsigned x : gates[gi].q) ++ptr[x];
      emit_plain(gi);
      ++done;
      continue;
    }
    ptr = best.ptr;
    done += best.chosen.size();
    if (best.chosen.size() < o.min_gates) {
      for (unsigned g : best.chosen) emit_plain(g);
      continue;
    }
    uint64_t S = best.S;
    for (unsigned p = 0; (unsigned)__builtin_popcountll(S) < o.tile_bits; ++p) S |= 1ull << p;  // pad with the lowest unused positions
    std::vector<unsigned> chosen = best.chosen;
    std::vector<Layer> inner;
    auto as_ptrs = [&](const std::vector<unsigned>& idx) {
      std::vector<const Gate*> v;
      for (unsigned g : idx) v.push_back(&gates[g]);
      return v;
    };
    auto cost = [&](const std::vector<Layer>& gl) {
      double c = 0;
      for (const auto& L : gl) c += inner_cost((unsigned)L.q.size());
      return c * (lds_bytes(gl, o) <= lds_budget(o) ? 1.0 : 1.25);
    };
    auto widen = [&](const std::vector<Layer>& in, unsigned kmax) {
      std::vector<Gate> tmp;
      for (const auto& L : in) tmp.push_back(Gate{L.q, L.U});
      std::vector<const Gate*> v;
      for (const auto& g : tmp) v.push_back(&g);
      return build_layers(v, kmax, o.tol);
    };
    if (o.inner_max == 255) {
      // any topological order of the pass's gates is allowed and the greedy fusion depends on it: a few random ones are
      // scored on qubit sets alone and the cheapest is fused for real
      if (o.fusion_orders > 1 && chosen.size() >= 3) {
        auto score = [&](const std::vector<unsigned>& order) {
          std::vector<uint64_t> qs;
          for (unsigned g : order) qs.push_back(gp[g]);
          const std::vector<uint64_t> l3 = dry_layers(qs, 3), l4 = dry_layers(l3, 4);
          double c3 = 0, c4 = 0;
          for (uint64_t x : l3) c3 += inner_cost((unsigned)__builtin_popcountll(x));
          for (uint64_t x : l4) c4 += inner_cost((unsigned)__builtin_popcountll(x));
          return std::min(c3, c4);
        };
        Rng ro(o.seed * 7919 + out.op_kind.size());
        std::vector<unsigned> best_order = chosen;
        double best_cost = score(chosen);
        std::vector<std::vector<unsigned>> ql(o.n);
        for (unsigned g : chosen)
          for (unsigned p : gates[g].q) ql[p].push_back(g);
        for (unsigned trial = 1; trial < o.fusion_orders; ++trial) {
          std::vector<unsigned> pp(o.n, 0), ready, order;
          std::vector<int> heads(G, 0);
          for (unsigned p = 0; p < o.n; ++p)
            if (!ql[p].empty() && ++heads[ql[p][0]] == (int)gk[ql[p][0]]) ready.push_back(ql[p][0]);
          std::sort(ready.begin(), ready.end());
          while (!ready.empty()) {
            const size_t pick = ro.below(ready.size());
            const unsigned g = ready[pick];
            ready.erase(ready.begin() + (long)pick);
            order.push_back(g);
            for (unsigned p : gates[g].q)
              if (++pp[p] < ql[p].size()) {
                const unsigned h = ql[p][pp[p]];
                if (++heads[h] == (int)gk[h]) ready.push_back(h);
              }
          }
          const double c = score(order);
          if (c < best_cost - 1e-9) { best_cost = c; best_order = order; }
        }
        chosen = best_order;
      }
      inner = build_layers(as_ptrs(chosen), 3, o.tol);
      std::vector<Layer> wider = widen(inner, 4);
      if (cost(wider) < cost(inner)) inner.swap(wider);
    } else if (o.inner_max) {
      inner = build_layers(as_ptrs(chosen), o.inner_max, o.tol);
    } else {
      for (unsigned g : chosen) inner.push_back(Layer{gates[g].q, gates[g].U});  // as given (q in gate order, not sorted)
    }
    out.op_kind.push_back(1);
    for (unsigned p = 0; p < 64; ++p)
      if ((S >> p) & 1) out.op_tile.push_back(p);
    for (auto& L : inner) out.gates.push_back(Gate{L.q, L.U});
    out.op_first_gate.push_back((unsigned)out.gates.size());
  }
}

}  // namespace plan
}  // namespace hq

namespace hq { namespace plan {
// nothing thrown inside a planner (std::bad_alloc from a matrix a caller's limits made huge, std::length_error) may cross
// the C ABI: it becomes an error code and a message
template <typename F>
static int guarded(const char* what, F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return hq::fail(std::string(what) + ": out of memory");
  } catch (const std::exception& e) {
    return hq::fail(std::string(what) + ": " + e.what());
  } catch (...) {
    return hq::fail(std::string(what) + ": unknown exception");
  }
}
}}  // namespace hq::plan

extern "C" {

int hq_plan_blocked(unsigned int n_qubits, unsigned int n_gates, const unsigned int* k, const unsigned int* positions,
                    const double* U, unsigned int tile_bits, unsigned int low_bits, unsigned int inner_max,
                    unsigned int min_gates, unsigned int tries, unsigned int fusion_orders, unsigned int elem_bytes,
                    uint64_t seed, double commute_tol, void** plan) {
  return hq::plan::guarded("hq_plan_blocked", [&]() -> int {
  using namespace hq::plan;
  if (!plan) return hq::fail("hq_plan_blocked: null output");
  *plan = nullptr;
  if (n_qubits == 0 || n_qubits > 62) return hq::fail("hq_plan_blocked: n_qubits must be in 1..62");
  if (n_gates && (!k || !positions || !U)) return hq::fail("hq_plan_blocked: null input");
  if (!(commute_tol > 0)) return hq::fail("hq_plan_blocked: commute_tol must be positive");
  if (inner_max != 255 && inner_max > 4) return hq::fail("hq_plan_blocked: inner gates are limited to 4 qubits");
  Options o{n_qubits, std::min(tile_bits, n_qubits), 0, inner_max, min_gates, tries, fusion_orders, elem_bytes, seed, commute_tol};
  o.low_bits = std::min(low_bits, o.tile_bits);
  std::vector<Gate> gates(n_gates);
  size_t po = 0, uo = 0;
  for (unsigned g = 0; g < n_gates; ++g) {
    if (k[g] == 0 || k[g] > hq::kMaxK) return hq::fail("hq_plan_blocked: gates act on 1..10 qubits");
    gates[g].q.assign(positions + po, positions + po + k[g]);
    if (hq::check_positions(positions + po, n_qubits, k[g])) return hq::fail("hq_plan_blocked: invalid positions");
    const size_t e = (size_t)1 << (2 * k[g]);
    gates[g].U.resize(e);
    for (size_t i = 0; i < e; ++i) gates[g].U[i] = cd(U[2 * (uo + i)], U[2 * (uo + i) + 1]);
    po += k[g];
    uo += e;
  }
  Result* r = new Result();
  plan_blocked(gates, o, *r);
  *plan = r;
  return 0;
  });
}

int hq_plan_simplify(unsigned int n_qubits, unsigned int n_gates, const unsigned int* k, const unsigned int* qubits, const double* U,
                     double atol, int use_matrix_commutation, unsigned int max_n_qubits_matrix, int remove_id_gates,
                     unsigned int* out_index, unsigned int* out_count) {
  return hq::plan::guarded("hq_plan_simplify", [&]() -> int {
  using namespace hq::plan;
  if (!out_index || !out_count) return hq::fail("hq_plan_simplify: null output");
  if (n_gates && (!k || !qubits || !U)) return hq::fail("hq_plan_simplify: null input");
  if (n_qubits == 0 || n_qubits > 62) return hq::fail("hq_plan_simplify: qubit ids must be below 62");
  max_n_qubits_matrix = std::min<unsigned>(max_n_qubits_matrix, (unsigned)kMaxCommuteQubits);
  std::vector<Gate> gates(n_gates);
  size_t po = 0, uo = 0;
  for (unsigned g = 0; g < n_gates; ++g) {
    if (k[g] == 0 || k[g] > hq::kMaxK) return hq::fail("hq_plan_simplify: gates act on 1..10 qubits");
    if (hq::check_positions(qubits + po, n_qubits, k[g])) return hq::fail("hq_plan_simplify: invalid qubit ids");
    gates[g].q.assign(qubits + po, qubits + po + k[g]);
    const size_t e = (size_t)1 << (2 * k[g]);
    gates[g].U.resize(e);
    for (size_t i = 0; i < e; ++i) gates[g].U[i] = cd(U[2 * (uo + i)], U[2 * (uo + i) + 1]);
    po += k[g];
    uo += e;
  }
  const std::vector<unsigned> out = simplify(gates, atol, use_matrix_commutation != 0, max_n_qubits_matrix, remove_id_gates != 0);
  std::copy(out.begin(), out.end(), out_index);
  *out_count = (unsigned)out.size();
  return 0;
  });
}

int hq_plan_fuse(unsigned int n_qubits, unsigned int n_gates, const unsigned int* k, const unsigned int* qubits, const double* U,
                 unsigned int max_n_qubits, int use_matrix_commutation, unsigned int max_n_qubits_matrix, uint64_t exclude_mask,
                 double commute_tol, void** plan) {
  return hq::plan::guarded("hq_plan_fuse", [&]() -> int {
  using namespace hq::plan;
  if (!plan) return hq::fail("hq_plan_fuse: null output");
  *plan = nullptr;
  if (n_gates && (!k || !qubits || !U)) return hq::fail("hq_plan_fuse: null input");
  if (n_qubits == 0 || n_qubits > 62) return hq::fail("hq_plan_fuse: qubit ids must be below 62");
  if (!(commute_tol > 0)) return hq::fail("hq_plan_fuse: commute_tol must be positive");
  if (max_n_qubits > hq::kMaxK) return hq::fail("hq_plan_fuse: fused gates are limited to 10 qubits (max_n_qubits)");
  max_n_qubits_matrix = std::min<unsigned>(max_n_qubits_matrix, (unsigned)kMaxCommuteQubits);
  std::vector<Gate> gates(n_gates);
  size_t po = 0, uo = 0;
  for (unsigned g = 0; g < n_gates; ++g) {
    if (k[g] == 0 || k[g] > hq::kMaxK) return hq::fail("hq_plan_fuse: gates act on 1..10 qubits");
    if (hq::check_positions(qubits + po, n_qubits, k[g])) return hq::fail("hq_plan_fuse: invalid qubit ids");
    gates[g].q.assign(qubits + po, qubits + po + k[g]);
    const size_t e = (size_t)1 << (2 * k[g]);
    gates[g].U.resize(e);
    for (size_t i = 0; i < e; ++i) gates[g].U[i] = cd(U[2 * (uo + i)], U[2 * (uo + i) + 1]);
    po += k[g];
    uo += e;
  }
  FuseOptions o;
  o.max_n = max_n_qubits;
  o.use_mc = use_matrix_commutation != 0;
  o.max_nqm = max_n_qubits_matrix;
  o.exclude = exclude_mask;
  o.tol = commute_tol;
  std::vector<const Gate*> ptrs;
  for (const auto& g : gates) ptrs.push_back(&g);
  Result* r = new Result();
  r->op_first_gate.push_back(0);
  for (auto& L : build_layers(ptrs, o)) {
    r->op_kind.push_back(0);
    r->gates.push_back(Gate{L.q, L.U});
    r->op_first_gate.push_back((unsigned)r->gates.size());
  }
  *plan = r;
  return 0;
  });
}

int hq_plan_counts(const void* plan, unsigned int* n_ops, unsigned int* n_gates, uint64_t* n_positions,
                   uint64_t* n_matrix_elems, unsigned int* tile_bits) {
  return hq::plan::guarded("hq_plan_counts", [&]() -> int {
  if (!plan) return hq::fail("hq_plan_counts: null plan");
  const hq::plan::Result& r = *static_cast<const hq::plan::Result*>(plan);
  uint64_t np = 0, ne = 0;
  for (const auto& g : r.gates) { np += g.q.size(); ne += g.U.size(); }
  if (n_ops) *n_ops = (unsigned)r.op_kind.size();
  if (n_gates) *n_gates = (unsigned)r.gates.size();
  if (n_positions) *n_positions = np;
  if (n_matrix_elems) *n_matrix_elems = ne;
  if (tile_bits) *tile_bits = r.tile_bits;
  return 0;
  });
}

int hq_plan_read(const void* plan, unsigned int* op_kind, unsigned int* op_first_gate, unsigned int* op_tile,
                 unsigned int* gate_k, unsigned int* gate_positions, double* U) {
  return hq::plan::guarded("hq_plan_read", [&]() -> int {
  if (!plan) return hq::fail("hq_plan_read: null plan");
  const hq::plan::Result& r = *static_cast<const hq::plan::Result*>(plan);
  if (op_kind) std::copy(r.op_kind.begin(), r.op_kind.end(), op_kind);
  if (op_first_gate) std::copy(r.op_first_gate.begin(), r.op_first_gate.end(), op_first_gate);
  if (op_tile) std::copy(r.op_tile.begin(), r.op_tile.end(), op_tile);
  size_t po = 0, uo = 0;
  for (size_t g = 0; g < r.gates.size(); ++g) {
    const auto& G = r.gates[g];
    if (gate_k) gate_k[g] = (unsigned)G.q.size();
    if (gate_positions) std::copy(G.q.begin(), G.q.end(), gate_positions + po);
    if (U)
      for (size_t i = 0; i < G.U.size(); ++i) { U[2 * (uo + i)] = G.U[i].real(); U[2 * (uo + i) + 1] = G.U[i].imag(); }
    po += G.q.size();
    uo += G.U.size();
  }
  return 0;
  });
}

int hq_plan_free(void* plan) {
  return hq::plan::guarded("hq_plan_free", [&]() -> int {
  delete static_cast<hq::plan::Result*>(plan);
  return 0;
  });
}

}  // extern "C"
