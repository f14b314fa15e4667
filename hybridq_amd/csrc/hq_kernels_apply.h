// hq_kernels_apply.h -- apply_U kernels (reference: /root/reference/include/U.h:28-102, 123-202): VALU butterflies,
// matrix-core role kernels (k <= 4, k = 5/6), the cache-blocked many-gates-per-pass kernel, the k >= 7 tile GEMM,
// the generic LDS-tile kernel and the tiny-state fallback.  Overview in hq_kernels_common.h.
#pragma once
#include <type_traits>
#include "hq_kernels_common.h"

// Pins a wave-uniform value to scalar registers (stops the optimiser from hoisting per-lane copies of it out of a loop).
// Under AddressSanitizer (-DHQ_ASAN: tools/asan_smoke.sh) the instrumented code computes it per lane and the constraint
// cannot be met: the pin is dropped there, it only matters for speed.
#ifdef HQ_ASAN
#define HQ_PIN_SGPR(x) ((void)0)
#else
#define HQ_PIN_SGPR(x) asm volatile("" : "+s"(x))
#endif

namespace hq {

// ---------------------------------------------------------------------------------
// apply_direct
// ---------------------------------------------------------------------------------
template <typename T, int K> struct GateArg {
  T re[1 << (2 * K)];  // row-major, matrix index bits in ASCENDING position order
  T im[1 << (2 * K)];
};
struct RegPos { unsigned p[4]; };  // register-target positions minus VB, ascending

template <typename T, int K, int VMASK, int ILP, bool NT>
__global__ void __launch_bounds__(kBlock)
apply_direct_kernel(T* __restrict__ re, T* __restrict__ im, const GateArg<T, K> U,
                    const RegPos rp) {
  using V = typename Vec<T>::type;
  constexpr int VB = Vec<T>::VB, VE = 1 << VB;
  constexpr int KV = popc_c(VMASK), KR = K - KV, R = 1 << KR, D = 1 << K;
  V* __restrict__ vre = reinterpret_cast<V*>(re);
  V* __restrict__ vim = reinterpret_cast<V*>(im);

  uint64_t off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint64_t o = 0;
#pragma unroll
    for (int j = 0; j < KR; ++j) o |= (uint64_t)((r >> j) & 1) << rp.p[j];
    off[r] = o;
  }

  const uint64_t g0 = (uint64_t)blockIdx.x * (ILP * kBlock) + threadIdx.x;
  uint64_t vb[ILP];
  V xr[ILP][R], xi[ILP][R];
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
    uint64_t v = g0 + (uint64_t)i * kBlock;
#pragma unroll
    for (int j = 0; j < KR; ++j) {
      const uint64_t lo = (1ull << rp.p[j]) - 1;
      v = ((v & ~lo) << 1) | (v & lo);
    }
    vb[i] = v;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (NT) {
        xr[i][r] = __builtin_nontemporal_load(&vre[v | off[r]]);
        xi[i][r] = __builtin_nontemporal_load(&vim[v | off[r]]);
      } else {
        xr[i][r] = vre[v | off[r]];
        xi[i][r] = vim[v | off[r]];
      }
    }
  }

#pragma unroll
  for (int i = 0; i < ILP; ++i) {
#pragma unroll
    for (int ro = 0; ro < R; ++ro) {
      V yr, yi;
#pragma unroll
      for (int co = 0; co < VE; ++co) {
        constexpr int dummy = 0;
        (void)dummy;
        const int to = pext_c(co, VMASK) | (ro << KV);
        const int cfree = co & ~VMASK;
        T ar = 0, ai = 0;
#pragma unroll
        for (int ti = 0; ti < D; ++ti) {
          const int ci = pdep_c(ti & ((1 << KV) - 1), VMASK) | cfree;
          const int ri = ti >> KV;
          const T ur = U.re[to * D + ti], ui = U.im[to * D + ti];
          const T pr = xr[i][ri][ci], pi = xi[i][ri][ci];
          ar = hq_fma(ur, pr, ar);
          ar = hq_fma(-ui, pi, ar);
          ai = hq_fma(ur, pi, ai);
          ai = hq_fma(ui, pr, ai);
        }
        yr[co] = ar;
        yi[co] = ai;
      }
      if (NT) {
        __builtin_nontemporal_store(yr, &vre[vb[i] | off[ro]]);
        __builtin_nontemporal_store(yi, &vim[vb[i] | off[ro]]);
      } else {
        vre[vb[i] | off[ro]] = yr;
        vim[vb[i] | off[ro]] = yi;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// apply_mfma (f32 and f64, k <= 4): the gate on the matrix cores, no LDS, no cross-lane traffic.
//
// The complex 2^k x 2^k gate acts as the REAL 2^(k+1) x 2^(k+1) matrix [[Ur,-Ui],[Ui,Ur]]
// on [re; im] (the same 8*2^k flops per amplitude as complex arithmetic).  The f32-input
// MFMA is an exact k-ordered f32 fma chain, so parity with the VALU path is at rounding
// level.  v_mfma_f32_16x16x4_f32:  D[16x16] += A[16x4] B[4x16] with lane l holding
// A[l&15][l>>4], B[l>>4][l&15] and D[4*(l>>4)+r][l&15] in register r.
//
// The K index of the embedded matrix has KBITS = k_eff+1 binary digits (k_eff = 3 or 4
// "effective" targets: real targets plus identity dummies, and the re/im plane).  Each
// digit is given one of three ROLES, the same on the input (B) and output (D) side:
//   * q digit    (2 of them): lane>>4.  If it is an index bit, the lane's address carries
//                that bit, so low targets (positions 2..5) become a PERMUTATION of a
//                contiguous run instead of a stride; if it is the plane digit, the lane
//                reads/writes that plane only.
//   * comp digit (VMASK):     component of the 16-byte vector (index bits 0..1 for f32,
//                bit 0 for f64); on the
//                input side it selects the component fed to the MFMA step, on the
//                output side the accumulator register lands in that component.
//   * reg digit  (the rest):  a separate 16-byte load/store per value.
// Step s = (comp digits, reg digits) walks the K dimension; D register r (and the
// row-block for k_eff = 4) enumerates the same digits, so every result register goes
// back to exactly the address/component it came from.  Free vector components are
// independent column blocks.  Host side (hq_hip.hip: plan_mfma) picks the roles and
// builds the A-operand table A[row-block][step][lane].
// ---------------------------------------------------------------------------------
struct MfmaRoles {
  unsigned pos[6];    // vec positions (index bit - #component bits) of all address digits, ascending; 63 = unused
  unsigned q_off[2];  // vec offset carried by q bit b (0 if that digit is the plane)
  int q_plane;        // q bit that selects the plane, -1 if the plane is a reg digit
  unsigned r_off[5];  // vec offset carried by reg digit b (0 if plane)
  int r_plane;        // reg digit that selects the plane, -1 if the plane is a q digit
};

template <typename T> struct Mfma;
template <> struct Mfma<float> {
  using acc = f32x4;
  static __device__ __forceinline__ acc run(float a, float b, acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma<double> {  // v_mfma_f64_16x16x4_f64: D row = (l>>4) + 4r (host table differs)
  using acc = f64x4;
  static __device__ __forceinline__ acc run(double a, double b, acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
};

template <typename T, int KBITS, int VMASK, int ILP, bool NT>
__global__ void __launch_bounds__(kBlock)
apply_mfma_kernel(T* __restrict__ re, T* __restrict__ im, const T* __restrict__ A,
                  const MfmaRoles ro) {
  using V = typename Vec<T>::type;
  using Acc = typename Mfma<T>::acc;
  constexpr int CB = Vec<T>::VB, NCOMP = 1 << CB;  // vector components: 4 (f32) / 2 (f64)
  constexpr int NS = KBITS - 2, KV = popc_c(VMASK), NR = NS - KV, NL = 1 << NR;
  constexpr int NRB = 1 << (NS - 2), NCB = 1 << (CB - KV), NSTEP = 1 << NS;
  constexpr int FMASK = ~VMASK & (NCOMP - 1);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, j = lane & 15;
  V* __restrict__ pre = reinterpret_cast<V*>(re);
  V* __restrict__ pim = reinterpret_cast<V*>(im);

  T a[NRB][NSTEP];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) a[rb][s] = A[(rb * NSTEP + s) * 64 + lane];

  const uint64_t lane_off = ((q & 1) ? (uint64_t)ro.q_off[0] : 0ull) | ((q & 2) ? (uint64_t)ro.q_off[1] : 0ull);
  const unsigned lane_plane = ro.q_plane >= 0 ? ((q >> ro.q_plane) & 1u) : 0u;
  uint64_t off[NL];
  unsigned pl[NL];
#pragma unroll
  for (int ld = 0; ld < NL; ++ld) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < NR; ++b)
      if ((ld >> b) & 1) o |= ro.r_off[b];
    off[ld] = o;
    pl[ld] = ro.r_plane >= 0 ? ((ld >> ro.r_plane) & 1u) : 0u;
  }

  V x[ILP][NL];
  V* ptr[ILP][NL];
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
    uint64_t v = (((uint64_t)blockIdx.x * ILP + i) * (kBlock / 64) + wave) * 16 + j;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint64_t lo = (1ull << ro.pos[m]) - 1;
      v = ((v & ~lo) << 1) | (v & lo);
    }
    v |= lane_off;
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) {
      ptr[i][ld] = ((lane_plane | pl[ld]) ? pim : pre) + (v | off[ld]);
      x[i][ld] = NT ? __builtin_nontemporal_load(ptr[i][ld]) : *ptr[i][ld];
    }
  }
  // every load of the workgroup is in flight before the first MFMA is scheduled
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
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
          acc[cf][rb] = Mfma<T>::run(a[rb][s], x[i][ld][comp], acc[cf][rb]);
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
      if (NT) __builtin_nontemporal_store(y, ptr[i][ld]);
      else *ptr[i][ld] = y;
    }
  }
}

// ---------------------------------------------------------------------------------
// k = 5, 6 with the same role scheme (KBITS = 6, 7).  The A-operand table no longer fits
// in registers (k = 6: 8 row-blocks x 32 steps), so a workgroup stages it ONCE in LDS
// (16 / 64 KiB f32, 32 / 128 KiB f64) and loops over wave iterations (grid-stride);
// per MFMA group one conflict-free ds_read_b128 fetches 4 (f32) / 2 (f64) consecutive
// steps.  Column blocks (free vector components) are processed one after the other and
// each result overwrites the input component it replaces (dead by then), so the register
// budget is the 2^NR loaded vectors + one set of accumulators.  Targets in index bits 0/1
// are component digits exactly as for k <= 4: every access stays a 16-byte vector of a
// contiguous run, whatever the positions.
//   table layout: A4[(rb * NSTEP/G + s/G) * 64 + lane][s % G], G = 16 / sizeof(T).
// ---------------------------------------------------------------------------------
// Byte offset of register-digit load `ld` from the lane's base address: the OR of the digit
// offsets it selects plus (im - re) when the plane is one of those digits.  Wave-uniform, built by
// the host (it knows both plane pointers) and read through the scalar cache: the kernel keeps no
// per-load address state in registers.
struct BigOffsets { int64_t off[32]; };

// PHASED = false: every wave loads / multiplies / stores in turn and the other resident waves cover
//   its memory phases (k = 5: HBM-bound, 4 waves per SIMD fit).
// PHASED = true (k = 6, matrix-core bound; BLOCK = 512 = two waves per SIMD): the two halves of the
//   workgroup run in ANTI-PHASE, separated by workgroup barriers: while waves 0-3 (one per SIMD) are
//   in their MFMA phase, waves 4-7 store their results and issue the loads of their next
//   wave-iteration, then the roles swap.  A SIMD's matrix pipe always has exactly one wave feeding it
//   and that wave's operands arrived a whole phase earlier.  Why it is needed (n = 30 knock-outs):
//   MFMA phase alone 3.69 ms, memory phases alone 3.6 ms, two free-running waves 4.7 ms -- left
//   alone the two waves of a SIMD share the pipe, finish together and then BOTH wait for HBM with
//   the pipe idle (s_setprio did not break the symmetry: 4.72 vs 4.73 ms; two register sets in one
//   wave did not either, 4.67 ms: vmcnt counts loads and stores in order, so a prefetch issued
//   before the stores cannot be waited for without waiting for the stores' acknowledgement, and
//   the register allocator copies the second set around the loop).
template <typename T, int KBITS, int VMASK, bool NT, int BLOCK, bool PHASED>
__global__ void __launch_bounds__(BLOCK)
apply_mfma_big_kernel(T* __restrict__ re, T* __restrict__ im, const T* __restrict__ A,
                      const MfmaRoles ro, const BigOffsets tab, const uint64_t niter) {
  using V = typename Vec<T>::type;
  using Acc = typename Mfma<T>::acc;
  HQ_DYN_LDS(hq_big_smem);
  constexpr int CB = Vec<T>::VB, NCOMP = 1 << CB, G = 16 / (int)sizeof(T);
  constexpr int NS = KBITS - 2, KV = popc_c(VMASK), NR = NS - KV, NL = 1 << NR, NA = KBITS - 1 - KV;
  constexpr int NRB = 1 << (NS - 2), NCB = 1 << (CB - KV), NSTEP = 1 << NS, NG = NSTEP / G;
  constexpr int FMASK = ~VMASK & (NCOMP - 1);
  constexpr int NP = NRB / 2, NGRP = NG * NP;
  static_assert(NL <= 32 && NRB >= 2, "shape");
  V* __restrict__ As = reinterpret_cast<V*>(hq_big_smem);
  {
    const V* __restrict__ Ag = reinterpret_cast<const V*>(A);
    for (int e = threadIdx.x; e < NRB * NG * 64; e += BLOCK) As[e] = Ag[e];
  }
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // scalar for the optimiser
  const unsigned q = lane >> 4, j = lane & 15;
  const V* __restrict__ Al = As + lane;
  // A ds_read carries a 16-bit byte offset: with a table above 64 KiB (complex128 k = 6: 128 KiB) the compiler
  // materialised one address register per operand read in the upper half and SPILLED them to reuse in the second column
  // block -- 10 of the 13 dwords of scratch that instantiation had in rounds 2-4, which had been put down to the 32 input
  // vectors + 8 accumulator blocks and had cost it its operand pipelining.  A second, opaque base address 64 KiB up keeps
  // every read at base + immediate (32-bit LDS addresses: a pointer through inline assembly loses its address space):
  // 223 registers with the operand pipeline, no scratch.
  constexpr int kHalf = 65536 / 16;  // vectors
  constexpr bool kTwoBases = (size_t)NRB * NG * 64 > (size_t)kHalf;
  typedef __attribute__((address_space(3))) const V LdsCV;
  const unsigned al_lo = (unsigned)reinterpret_cast<uintptr_t>(Al);  // LDS byte address of this lane's slot of row 0
  unsigned al_hi = al_lo + (kTwoBases ? 65536u : 0u);
  if constexpr (kTwoBases) asm volatile("" : "+v"(al_hi));
  auto Aop = [&](const int idx) -> V {
    if constexpr (!kTwoBases) return Al[idx];
    else return *reinterpret_cast<LdsCV*>((uintptr_t)((idx >= kHalf ? al_hi : al_lo) + 16u * (unsigned)(idx & (kHalf - 1))));
  };
  // address of load ld in iteration it = lane base (loop-invariant, per lane: plane of the q digit,
  // q-digit offsets, the slot bits j spread over the non-digit index bits) + 16 * spread(it * 16)
  // (wave-uniform, a handful of scalar ops per iteration) + tab.off[ld] (wave-uniform, scalar load)
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

  auto load_x = [&](V (&x)[NL], const uint64_t it) {
    const int64_t it_off = (int64_t)(16 * spread(it * 16));
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) {
      V* ptr = reinterpret_cast<V*>(lane_base + (it_off + tab.off[ld]));
      x[ld] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
    }
  };
  // the store addresses are computed from an opaque copy of the iteration offset: shared with the
  // loads they would be 2^NR live 64-bit values across the whole MFMA phase (spills at k = 6)
  auto store_x = [&](V (&x)[NL], const uint64_t it) {
    int64_t st_off = (int64_t)(16 * spread(it * 16));
    HQ_PIN_SGPR(st_off);
#pragma unroll
    for (int ld = 0; ld < NL; ++ld) {
      V* ptr = reinterpret_cast<V*>(lane_base + (st_off + tab.off[ld]));
      if (NT) __builtin_nontemporal_store(x[ld], ptr);
      else *ptr = x[ld];
    }
  };
#define HQ_BIG_MFMA(a_, b_, c_) Mfma<T>::run(a_, b_, c_)
  // MFMA phase, software-pipelined: the operand reads of pair-group g+1 (two ds_read_b128: row blocks
  // 2p and 2p+1 of one step group) are issued BEFORE the 2*G MFMAs of pair-group g, which alternate
  // between the two accumulators (a 16x16x4 MFMA issues every 32 cycles but its result is only
  // available after 40: back-to-back MFMAs on ONE accumulator lose a fifth of the pipe).  The
  // pipeline runs across column blocks (the operand sequence repeats): only the first read of an
  // iteration is exposed.  Alone this phase runs at 149 of 157 TFLOP/s (k = 6 knock-out).
  auto compute = [&](V (&x)[NL]) {
    V a0 = Aop(0), a1 = Aop(NG * 64);
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
          n0 = Aop((rbn * NG + sgn) * 64);
          n1 = Aop(((rbn + 1) * NG + sgn) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);  // the reads of the NEXT pair-group stay in front of ...
#pragma unroll
        for (int t = 0; t < G; ++t) {
          const int s = sg * G + t;
          const int ck = s & ((1 << KV) - 1), ld = s >> KV;
          const int comp = pdep_c(ck, VMASK) | pdep_c(cf, FMASK);
          acc[rb] = HQ_BIG_MFMA(a0[t], x[ld][comp], acc[rb]);
          acc[rb + 1] = HQ_BIG_MFMA(a1[t], x[ld][comp], acc[rb + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);  // ... this pair-group's MFMAs, which wait for nothing
        a0 = n0;
        a1 = n1;
      }
      // the inputs of this column block are dead: overwrite them with its results.  The empty asm
      // pins the merged vector in registers HERE: left alone, the optimiser keeps the results of every
      // column block in their accumulator registers until the stores (4 x the accumulators live).
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
#undef HQ_BIG_MFMA
  const uint64_t stride = (uint64_t)gridDim.x * (BLOCK / 64);
  uint64_t it = (uint64_t)blockIdx.x * (BLOCK / 64) + wave;
  if constexpr (!PHASED) {
    for (; it < niter; it += stride) {
      V x[NL];
      load_x(x, it);
      compute(x);
      store_x(x, it);
    }
  } else {
    static_assert(BLOCK == 512, "two waves per SIMD");
    // every wave of the workgroup passes the same number of barriers: the trip count comes from
    // the workgroup's FIRST wave; a wave whose iteration index runs past the end idles through
    const uint64_t first = (uint64_t)blockIdx.x * (BLOCK / 64);
    const uint64_t trips = first < niter ? (niter - first + stride - 1) / stride : 0;
    const bool second = wave >= BLOCK / 128;  // waves w and w + 4 share a SIMD
    V x[NL];
    if (trips && it < niter) load_x(x, it);
    if (trips && second) __builtin_amdgcn_s_barrier();  // the second half starts one phase late
    for (uint64_t k = 0; k < trips; ++k) {
      if (it < niter) compute(x);
      __builtin_amdgcn_s_barrier();  // ---- phase switch: the partner half takes over the matrix pipe
      if (it < niter) store_x(x, it);
      it += stride;
      if (k + 1 < trips && it < niter) load_x(x, it);  // in flight for a whole phase before its first use
      __builtin_amdgcn_s_barrier();
    }
    if (trips && !second) __builtin_amdgcn_s_barrier();
  }
}

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
  constexpr unsigned WB = BLOCK == 512 ? 3 : (BLOCK == 256 ? 2 : 4);
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
  constexpr unsigned WB = BLOCK == 512 ? 3 : (BLOCK == 256 ? 2 : 4);
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
  constexpr unsigned WB = BLOCK == 512 ? 3 : (BLOCK == 256 ? 2 : 4);
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
  constexpr unsigned WB = BLOCK == 512 ? 3 : (BLOCK == 256 ? 2 : 4);
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
    auto plain_loop = [&]() {
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
    };
    constexpr bool kPipe = PIPE && gemm_can_pipe<T, RBW, CBW>();
    if constexpr (!kPipe) {
      plain_loop();
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


struct GenArg {
  unsigned k, c;                 // target bits, column bits (c >= 2)
  unsigned tpos[kMaxK];          // target positions, ORIGINAL order (bit j of matrix index)
  unsigned cpos[kTileBits];      // column positions ascending (lowest non-target bits)
  unsigned apos[kTileBits + kMaxK];  // all tile positions ascending
  unsigned vec_ok;               // cpos[0]==0 && cpos[1]==1: quads contiguous in memory
};

template <typename T>
__global__ void __launch_bounds__(kBlock)
apply_generic_kernel(T* __restrict__ re, T* __restrict__ im, const T* __restrict__ U,
                     const GenArg a, const uint64_t nblocks) {
  using Q = typename Vec<T>::quad;
  HQ_DYN_LDS(smem);
  const unsigned D = 1u << a.k, C = 1u << a.c, CQ = C >> 2;
  uint64_t* toff = reinterpret_cast<uint64_t*>(smem);
  uint32_t* coff = reinterpret_cast<uint32_t*>(toff + D);
  T* xr = reinterpret_cast<T*>(coff + C);
  T* xi = xr + (size_t)D * C;
  const unsigned tid = threadIdx.x;

  for (unsigned i = tid; i < D; i += kBlock) {
    uint64_t o = 0;
    for (unsigned j = 0; j < a.k; ++j) o |= (uint64_t)((i >> j) & 1u) << a.tpos[j];
    toff[i] = o;
  }
  for (unsigned i = tid; i < C; i += kBlock) {
    uint32_t o = 0;
    for (unsigned j = 0; j < a.c; ++j) o |= ((i >> j) & 1u) << a.cpos[j];
    coff[i] = o;
  }
  __syncthreads();

  const unsigned nitems = D * CQ;
  for (uint64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    uint64_t base = b;
    for (unsigned j = 0; j < a.k + a.c; ++j) {
      const uint64_t lo = (1ull << a.apos[j]) - 1;
      base = ((base & ~lo) << 1) | (base & lo);
    }
    // stage the tile
    for (unsigned e = tid; e < nitems; e += kBlock) {
      const unsigned t = e / CQ, q = e % CQ;
      const uint64_t idx = base | toff[t];
      Q vr, vi;
      if (a.vec_ok) {
        vr = *reinterpret_cast<const Q*>(re + (idx | coff[4 * q]));
        vi = *reinterpret_cast<const Q*>(im + (idx | coff[4 * q]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vr[j] = re[idx | coff[4 * q + j]];
          vi[j] = im[idx | coff[4 * q + j]];
        }
      }
      *reinterpret_cast<Q*>(xr + (size_t)t * C + 4 * q) = vr;
      *reinterpret_cast<Q*>(xi + (size_t)t * C + 4 * q) = vi;
    }
    __syncthreads();
    // dense complex (D x D) . (D x C)
    for (unsigned e = tid; e < nitems; e += kBlock) {
      const unsigned t = e / CQ, q = e % CQ;
      const T* __restrict__ Urow = U + (size_t)2 * D * t;
      Q ar = {0, 0, 0, 0}, ai = {0, 0, 0, 0};
      for (unsigned s = 0; s < D; ++s) {
        const T ur = Urow[2 * s], ui = Urow[2 * s + 1];
        const Q pr = *reinterpret_cast<const Q*>(xr + (size_t)s * C + 4 * q);
        const Q pi = *reinterpret_cast<const Q*>(xi + (size_t)s * C + 4 * q);
        ar += ur * pr - ui * pi;
        ai += ur * pi + ui * pr;
      }
      const uint64_t idx = base | toff[t];
      if (a.vec_ok) {
        *reinterpret_cast<Q*>(re + (idx | coff[4 * q])) = ar;
        *reinterpret_cast<Q*>(im + (idx | coff[4 * q])) = ai;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          re[idx | coff[4 * q + j]] = ar[j];
          im[idx | coff[4 * q + j]] = ai[j];
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------
// apply_mfma_tile (k = 5, 6): LDS-staged tile GEMM on the matrix cores.
//
// Same workgroup tile as apply_generic (2^k target rows x C = 2^c lowest non-target
// columns, both planes, 32 KiB of LDS) but the dense product is the REAL 2^(k+1)-square
// embedded matrix M = [[Ur,-Ui],[Ui,Ur]] (row/column index = plane*2^k + t) applied with
// v_mfma_*_16x16x4: wave w owns the output row blocks {w*RBW .. w*RBW+RBW-1} for every
// column of the tile, keeps its A operands (rows of M) in registers for the whole launch,
// reads B operands from LDS as 16-byte vectors (one ds_read_b128 feeds CW MFMAs: the CW
// vector components are CW column blocks) and streams the result registers straight back
// to HBM (the inputs of the tile are safe in LDS, so the update is in place).
// k = 5 needs 57 % of the f32 MFMA peak at full HBM rate, k = 6 is MFMA-bound (512 flop per
// amplitude: >= 3.5 ms at n = 30).
// ---------------------------------------------------------------------------------
template <typename T, int K>
__global__ void __launch_bounds__(kBlock)
apply_mfma_tile_kernel(T* __restrict__ re, T* __restrict__ im, const T* __restrict__ A,
                       const GenArg a, const uint64_t nblocks) {
  using V = typename Vec<T>::type;
  using Acc = typename Mfma<T>::acc;
  constexpr int CW = 1 << Vec<T>::VB;                 // components of a 16-byte vector
  constexpr int D = 1 << K, E = 2 * D, NSTEP = E / 4, NRBT = E / 16, RBW = NRBT / 4;
  constexpr int CBITS = (sizeof(T) == 4 ? 12 : 11) - K, C = 1 << CBITS, NCG = C / (16 * CW);
  static_assert(NCG >= 1 && RBW >= 1, "tile shape");
  HQ_DYN_LDS(smem);
  uint64_t* toff = reinterpret_cast<uint64_t*>(smem);
  uint32_t* coff = reinterpret_cast<uint32_t*>(toff + D);
  T* xr = reinterpret_cast<T*>(coff + C);
  T* xi = xr + (size_t)D * C;
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned q = lane >> 4, j = lane & 15;

  for (unsigned i = tid; i < (unsigned)D; i += kBlock) {
    uint64_t o = 0;
    for (unsigned b = 0; b < (unsigned)K; ++b) o |= (uint64_t)((i >> b) & 1u) << a.tpos[b];
    toff[i] = o;
  }
  for (unsigned i = tid; i < (unsigned)C; i += kBlock) {
    uint32_t o = 0;
    for (unsigned b = 0; b < (unsigned)CBITS; ++b) o |= ((i >> b) & 1u) << a.cpos[b];
    coff[i] = o;
  }
  T areg[RBW][NSTEP];
#pragma unroll
  for (int i = 0; i < RBW; ++i)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) areg[i][s] = A[((size_t)(wave * RBW + i) * NSTEP + s) * 64 + lane];
  __syncthreads();

  constexpr unsigned CV = C / CW;  // 16-byte vectors per tile row
  for (uint64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    uint64_t base = b;
    for (unsigned m = 0; m < (unsigned)(K + CBITS); ++m) {
      const uint64_t lo = (1ull << a.apos[m]) - 1;
      base = ((base & ~lo) << 1) | (base & lo);
    }
    // stage the tile: rows t, columns c (16-byte vectors when the low index bits are columns)
    for (unsigned e = tid; e < (unsigned)D * CV; e += kBlock) {
      const unsigned t = e / CV, cv = e % CV;
      const uint64_t idx = base | toff[t];
      V vr, vi;
      if (a.vec_ok) {
        vr = *reinterpret_cast<const V*>(re + (idx | coff[CW * cv]));
        vi = *reinterpret_cast<const V*>(im + (idx | coff[CW * cv]));
      } else {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          vr[c] = re[idx | coff[CW * cv + c]];
          vi[c] = im[idx | coff[CW * cv + c]];
        }
      }
      *reinterpret_cast<V*>(xr + (size_t)t * C + CW * cv) = vr;
      *reinterpret_cast<V*>(xi + (size_t)t * C + CW * cv) = vi;
    }
    __syncthreads();
    Acc acc[RBW][NCG][CW];
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
        for (int c = 0; c < CW; ++c) acc[i][cg][c] = Acc{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const unsigned kk = 4 * s + q;  // K-row of this lane group: plane*D + t
      const T* xp = (kk >= (unsigned)D ? xi : xr) + (size_t)(kk & (D - 1)) * C;
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        const V bv = *reinterpret_cast<const V*>(xp + (cg * 16 + j) * CW);
#pragma unroll
        for (int c = 0; c < CW; ++c)
#pragma unroll
          for (int i = 0; i < RBW; ++i) acc[i][cg][c] = Mfma<T>::run(areg[i][s], bv[c], acc[i][cg][c]);
      }
    }
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // D row of lane (q, j) register r inside its 16-row block: 4q+r (f32) / q+4r (f64)
        const unsigned row = 16 * (wave * RBW + i) + (sizeof(T) == 4 ? 4 * q + r : q + 4 * r);
        T* plane = row >= (unsigned)D ? im : re;
        const uint64_t idx = base | toff[row & (D - 1)];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
          const unsigned col0 = (cg * 16 + j) * CW;
          V y;
#pragma unroll
          for (int c = 0; c < CW; ++c) y[c] = acc[i][cg][c][r];
          if (a.vec_ok) {
            *reinterpret_cast<V*>(plane + (idx | coff[col0])) = y;
          } else {
#pragma unroll
            for (int c = 0; c < CW; ++c) plane[idx | coff[col0 + c]] = y[c];
          }
        }
      }
    __syncthreads();  // LDS is restaged by the next tile
  }
}

// ---------------------------------------------------------------------------------
// apply_naive: out-of-place, one thread per output amplitude (tiny states only)
// ---------------------------------------------------------------------------------
struct NaiveArg {
  unsigned k;
  unsigned tpos[kMaxK];
};

template <typename T>
__global__ void __launch_bounds__(kBlock)
apply_naive_kernel(const T* __restrict__ in_re, const T* __restrict__ in_im,
                   T* __restrict__ out_re, T* __restrict__ out_im, const T* __restrict__ U,
                   const NaiveArg a, const uint64_t size) {
  const uint64_t x = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (x >= size) return;
  const unsigned D = 1u << a.k;
  unsigned t = 0;
  uint64_t mask = 0;
  for (unsigned j = 0; j < a.k; ++j) {
    t |= (unsigned)((x >> a.tpos[j]) & 1u) << j;
    mask |= 1ull << a.tpos[j];
  }
  const uint64_t b = x & ~mask;
  T ar = 0, ai = 0;
  for (unsigned s = 0; s < D; ++s) {
    uint64_t idx = b;
    for (unsigned j = 0; j < a.k; ++j) idx |= (uint64_t)((s >> j) & 1u) << a.tpos[j];
    const T ur = U[2 * ((size_t)t * D + s)], ui = U[2 * ((size_t)t * D + s) + 1];
    const T pr = in_re[idx], pi = in_im[idx];
    ar += ur * pr - ui * pi;
    ai += ur * pi + ui * pr;
  }
  out_re[x] = ar;
  out_im[x] = ai;
}

}  // namespace hq
