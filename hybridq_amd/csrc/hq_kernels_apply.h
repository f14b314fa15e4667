// hq_kernels_apply.h -- apply_U kernels (reference: /root/reference/include/U.h:28-102, 123-202): VALU butterflies,
// matrix-core role kernels (k <= 4, k = 5/6), the generic LDS-tile kernel, the LDS tile GEMM for k = 5/6 and the
// tiny-state fallback.  The cache-blocked many-gates-per-pass kernels are in hq_kernels_blocked.h, the k >= 7 tile GEMM in
// hq_kernels_gemm.h (both include this file).  Overview in hq_kernels_common.h.
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
//
// TWOB (only meaningful where the table exceeds 64 KiB = complex128 k = 6): true = the second LDS base address of round 5
// (no scratch, operand pipeline), false = the instantiation rounds 2-3 ran on hardware (plain indexing, operands read where
// they are used, ~52 B/lane of scratch).  The host picks (HQ_BIG_TWOBASE, default 0 until a hardware run has seen the new form).
template <typename T, int KBITS, int VMASK, bool NT, int BLOCK, bool PHASED, bool TWOB = true>
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
  constexpr bool kTableAbove64K = (size_t)NRB * NG * 64 > (size_t)kHalf;
  constexpr bool kTwoBases = kTableAbove64K && TWOB;
  // (TWOB = false is the code of rounds 2-4 to the letter: there only complex128 k = 6 WITHOUT a component target -- 32 loaded
  // vectors -- had no registers for a second operand pair; with one (VMASK = 1, 16 vectors) it pipelined its operands already)
  constexpr bool kOperandPipe = TWOB || !(sizeof(T) == 8 && NL == 32);
  typedef __attribute__((address_space(3))) const V LdsCV;
  const unsigned al_lo = (unsigned)reinterpret_cast<uintptr_t>(Al);  // LDS byte address of this lane's slot of row 0
  unsigned al_hi = al_lo + (kTwoBases ? 65536u : 0u);
#ifndef HQ_ASAN  // (host builds: the address is used as it is)
  if constexpr (kTwoBases) asm volatile("" : "+v"(al_hi));
#endif
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
        if constexpr (kOperandPipe) {
          if (g + 1 < NGRP || cf + 1 < NCB) {
            n0 = Aop((rbn * NG + sgn) * 64);
            n1 = Aop(((rbn + 1) * NG + sgn) * 64);
          }
        } else {
          a0 = Al[(rb * NG + sg) * 64];
          a1 = Al[((rb + 1) * NG + sg) * 64];
          n0 = a0;
          n1 = a1;
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
