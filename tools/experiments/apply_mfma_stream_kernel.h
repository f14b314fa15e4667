// Not part of the product (moved out of hybridq_amd/csrc/hq_kernels_apply.h in round 4): the k = 5, 6 kernel with ONE
// wave per SIMD, two register sets and the memory operations interleaved into the MFMA stream.  Measured in round 3:
// 4.67-4.75 ms against 4.74 for the phased kernel at k = 6 (profiles/r03_sweep_k56_stream.txt); bit-identical results.
// It needs the surrounding header (MfmaRoles, BigOffsets, Mfma<T>, Vec<T>) to compile; launch code: git show 90757e8.
// ---------------------------------------------------------------------------------
// apply_mfma_stream_kernel: k = 5, 6 with ONE wave per SIMD and TWO register sets (round 3).
//
// The barrier-phased form above alternates two waves per SIMD: while one multiplies, the other stores its 2^NR result
// vectors and requests its next 2^NR input vectors in one burst.  At k = 6 the matrix-core phase (3.69 ms alone at
// n = 30) and the memory phases (3.6 ms alone) are the same length, so every late vector of a burst stalls the pipe
// (4.65 ms together).  Here a wave owns 256 threads' worth of registers (launch bounds 256, one workgroup per CU) and
// keeps TWO sets of 2^NR vectors: while it multiplies set X (wave-iteration p) it stores set Y (the results of p-1) and
// re-fills Y with the inputs of p+1 -- one store / one load placed between the MFMA groups of the stream, so that the
// memory traffic of a CU is a steady trickle instead of a burst and the matrix pipe of a SIMD is fed by one wave without
// a phase switch.  vmcnt is in order: every load is issued after the store of the register it overwrites and is waited
// for (by the compiler's own counters) a whole phase later.  No barriers after the operand table is staged.
// ---------------------------------------------------------------------------------
template <typename T, int KBITS, int VMASK, bool NT, int SPANQ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
apply_mfma_stream_kernel(T* __restrict__ re, T* __restrict__ im, const T* __restrict__ A,
                         const MfmaRoles ro, const BigOffsets tab, const uint64_t niter) {
  using V = typename Vec<T>::type;
  using Acc = typename Mfma<T>::acc;
  HQ_DYN_LDS(hq_big_smem);
  constexpr int BLOCK = 256;
  constexpr int CB = Vec<T>::VB, NCOMP = 1 << CB, G = 16 / (int)sizeof(T);
  constexpr int NS = KBITS - 2, KV = popc_c(VMASK), NR = NS - KV, NL = 1 << NR, NA = KBITS - 1 - KV;
  constexpr int NRB = 1 << (NS - 2), NCB = 1 << (CB - KV), NSTEP = 1 << NS, NG = NSTEP / G;
  constexpr int FMASK = ~VMASK & (NCOMP - 1);
  constexpr int NP = NRB / 2, NGRP = NG * NP, NPG = NGRP * NCB;  // pair-groups of 2 G MFMAs per wave-iteration
  static_assert(NL <= 32 && NRB >= 2, "shape");
  V* __restrict__ As = reinterpret_cast<V*>(hq_big_smem);
  {
    const V* __restrict__ Ag = reinterpret_cast<const V*>(A);
    for (int e = threadIdx.x; e < NRB * NG * 64; e += BLOCK) As[e] = Ag[e];
  }
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned q = lane >> 4, j = lane & 15;
  const V* __restrict__ Al = As + lane;
  auto spread = [&](uint64_t v) {
#pragma unroll
    for (int m = 0; m < NA; ++m) {
      const uint64_t lo = (1ull << ro.pos[m]) - 1;
      v = ((v & ~lo) << 1) | (v & lo);
    }
    return v;
  };
  const unsigned lane_plane = ro.q_plane >= 0 ? ((q >> ro.q_plane) & 1u) : 0u;
  const uint64_t lane_vec = spread((uint64_t)j) | ((q & 1) ? (uint64_t)ro.q_off[0] : 0ull) | ((q & 2) ? (uint64_t)ro.q_off[1] : 0ull);
  unsigned char* const lane_base = reinterpret_cast<unsigned char*>(lane_plane ? im : re) + 16 * lane_vec;
  auto vptr = [&](int64_t it_off, int ld) { return reinterpret_cast<V*>(lane_base + (it_off + tab.off[ld])); };
  auto it_offset = [&](uint64_t it) { return (int64_t)(16 * spread(it * 16)); };
  auto load1 = [&](V& x, int64_t off, int ld) { x = NT ? __builtin_nontemporal_load(vptr(off, ld)) : *vptr(off, ld); };
  auto store1 = [&](const V& x, int64_t off, int ld) {
    if (NT) __builtin_nontemporal_store(x, vptr(off, ld));
    else *vptr(off, ld) = x;
  };
  // one wave-iteration on X; between its MFMA pair-groups: store Y[ld] (results of the previous iteration, at st_off) and
  // re-fill Y[ld] (inputs of the next one, at ld_off).  The 2 NL memory operations sit in the first SPANQ quarters of the
  // pair-groups (compile time: register indices must be static).
  constexpr int span = NPG * SPANQ / 4;
  auto phase = [&](V (&x)[NL], V (&y)[NL], auto overlap_tag, int64_t st_off, int64_t ld_off) {
    constexpr bool overlap = decltype(overlap_tag)::value;
    HQ_PIN_SGPR(st_off);
    HQ_PIN_SGPR(ld_off);
    V a0 = Al[0], a1 = Al[NG * 64];
#pragma unroll
    for (int cf = 0; cf < NCB; ++cf) {
      Acc acc[NRB];
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) acc[rb] = Acc{0, 0, 0, 0};
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        const int sg = g / NP, rb = 2 * (g % NP);
        const int gn = (g + 1) % NGRP, sgn = gn / NP, rbn = 2 * (gn % NP);
        V n0 = a0, n1 = a1;
        if (g + 1 < NGRP || cf + 1 < NCB) {
          n0 = Al[(rbn * NG + sgn) * 64];
          n1 = Al[((rbn + 1) * NG + sgn) * 64];
        }
        if constexpr (overlap) {
          const int gg = cf * NGRP + g;  // 0 .. NPG-1
          // memory operation number mo = 0 .. 2 NL - 1 (even: store ld = mo / 2, odd: load ld = mo / 2) goes in front of
          // pair-group floor(mo * span / (2 NL))
          const int m_lo = (gg * 2 * NL + span - 1) / span, m_hi = ((gg + 1) * 2 * NL + span - 1) / span;
#pragma unroll
          for (int mo = m_lo; mo < m_hi && mo < 2 * NL; ++mo) {
            if (mo & 1) load1(y[mo >> 1], ld_off, mo >> 1);
            else store1(y[mo >> 1], st_off, mo >> 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < G; ++t) {
          const int s = sg * G + t;
          const int ck = s & ((1 << KV) - 1), ld = s >> KV;
          const int comp = pdep_c(ck, VMASK) | pdep_c(cf, FMASK);
          acc[rb] = Mfma<T>::run(a0[t], x[ld][comp], acc[rb]);
          acc[rb + 1] = Mfma<T>::run(a1[t], x[ld][comp], acc[rb + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = n0;
        a1 = n1;
      }
#pragma unroll
      for (int ld = 0; ld < NL; ++ld) {
#pragma unroll
        for (int ck = 0; ck < (1 << KV); ++ck) {
          const int comp = pdep_c(ck, VMASK) | pdep_c(cf, FMASK);
          const int so = ck | (ld << KV);
          x[ld][comp] = acc[so >> 2][so & 3];
        }
        if (NCB > 1) asm volatile("" : "+v"(x[ld]));
      }
    }
  };
  const uint64_t stride = (uint64_t)gridDim.x * (BLOCK / 64);
  const uint64_t first = (uint64_t)blockIdx.x * (BLOCK / 64) + wave;
  if (first >= niter) return;
  const uint64_t m = (niter - first + stride - 1) / stride;  // wave-iterations of this wave
  auto it_of = [&](uint64_t p) { return first + (p < m ? p : m - 1) * stride; };  // clamped: a wasted, harmless load at the end
  V xa[NL], xb[NL];
  {
    const int64_t o0 = it_offset(it_of(0)), o1 = it_offset(it_of(1));
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) load1(xa[ld], o0, ld);
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) load1(xb[ld], o1, ld);
  }
  phase(xa, xb, std::false_type{}, 0, 0);  // wave-iteration 0: nothing to store yet
  uint64_t p = 1;
  while (true) {
    if (p >= m) {
      const int64_t o = it_offset(it_of(p - 1));
#pragma unroll
      for (int ld = 0; ld < NL; ++ld) store1(xa[ld], o, ld);
      break;
    }
    phase(xb, xa, std::true_type{}, it_offset(it_of(p - 1)), it_offset(it_of(p + 1)));
    ++p;
    if (p >= m) {
      const int64_t o = it_offset(it_of(p - 1));
#pragma unroll
      for (int ld = 0; ld < NL; ++ld) store1(xb[ld], o, ld);
      break;
    }
    phase(xa, xb, std::true_type{}, it_offset(it_of(p - 1)), it_offset(it_of(p + 1)));
    ++p;
  }
}

