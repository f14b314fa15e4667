// hq_kernels_aux.h -- streaming helpers: to_complex (reference: /root/reference/include/python_U.cpp:114-123), initial
// states, reductions, device side of Measure / Projection.
#pragma once
#include "hq_kernels_common.h"

namespace hq {

// ---------------------------------------------------------------------------------
// to_complex, init_state, norm2
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
interleave_kernel(const T* __restrict__ re, const T* __restrict__ im, T* __restrict__ out,
                  const uint64_t size) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += stride) {
    out[2 * i] = re[i];
    out[2 * i + 1] = im[i];
  }
}

// 4 elements per thread, 16-byte accesses; size must be a multiple of 4 and pointers
// 16/32-byte aligned (checked by the host).
template <typename T>
__global__ void __launch_bounds__(kBlock)
interleave4_kernel(const T* __restrict__ re, const T* __restrict__ im, T* __restrict__ out,
                   const uint64_t nquads) {
  using Q = typename Vec<T>::quad;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nquads; i += stride) {
    const Q r = reinterpret_cast<const Q*>(re)[i];
    const Q m = reinterpret_cast<const Q*>(im)[i];
    Q o0 = {r[0], m[0], r[1], m[1]};
    Q o1 = {r[2], m[2], r[3], m[3]};
    reinterpret_cast<Q*>(out)[2 * i] = o0;  // (measured, n = 30: non-temporal stores 4.6-4.9 TB/s, non-temporal loads 5.2, two quads in flight 5.3 = this form)
    reinterpret_cast<Q*>(out)[2 * i + 1] = o1;
  }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
init_state_kernel(T* __restrict__ re, T* __restrict__ im, const uint64_t size, const int kind,
                  const uint64_t basis, const T amp) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += stride) {
    re[i] = kind == 1 ? amp : (i == basis ? (T)1 : (T)0);
    im[i] = 0;
  }
}

// Product state of '0' / '1' / '+' / '-' factors (hybridq/circuit/simulation/utils.py:99-153 builds it
// on the host with kron + parity + transpose): amplitude of index X is 0 unless the '0'/'1' bits of
// X match, else (-1)^popcount(X & minus_mask) * 2^(-#pm/2).  X = hi_bits | local index, so a shard
// of a multi-GPU state (hi_bits = rank << n_local) is written by the same kernel.
template <typename T>
__global__ void __launch_bounds__(kBlock)
init_product_kernel(T* __restrict__ re, T* __restrict__ im, const uint64_t nquads, const uint64_t hi_bits,
                    const uint64_t mask01, const uint64_t val01, const uint64_t mask_minus, const T amp) {
  using Q = typename Vec<T>::quad;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nquads; i += stride) {
    Q r, z = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint64_t x = hi_bits | (4 * i + c);
      const T v = (__popcll(x & mask_minus) & 1) ? -amp : amp;
      r[c] = ((x & mask01) == val01) ? v : (T)0;
    }
    __builtin_nontemporal_store(r, reinterpret_cast<Q*>(re) + i);
    __builtin_nontemporal_store(z, reinterpret_cast<Q*>(im) + i);
  }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
norm2_kernel(const T* __restrict__ re, const T* __restrict__ im, const uint64_t size,
             double* __restrict__ out) {
  HQ_LDS double part[kBlock / 64];
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  double acc = 0;
  using V = typename Vec<T>::type;
  constexpr int VE = 1 << Vec<T>::VB;
  if (size % (2 * VE) == 0 && reinterpret_cast<uintptr_t>(re) % 16 == 0 && reinterpret_cast<uintptr_t>(im) % 16 == 0) {
    // 16-byte non-temporal loads, two vector pairs in flight per thread, two accumulators (5.4 -> 6 TB/s)
    const V* __restrict__ vr = reinterpret_cast<const V*>(re);
    const V* __restrict__ vi = reinterpret_cast<const V*>(im);
    const uint64_t nvec = size / VE;
    double acc2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += 2 * stride) {
      const uint64_t i2 = i + stride < nvec ? i + stride : i;  // (nvec is a multiple of 2: clamped repeats are skipped below)
      const V r0 = __builtin_nontemporal_load(vr + i), m0 = __builtin_nontemporal_load(vi + i);
      const V r1 = __builtin_nontemporal_load(vr + i2), m1 = __builtin_nontemporal_load(vi + i2);
#pragma unroll
      for (int c = 0; c < VE; ++c) acc += (double)r0[c] * (double)r0[c] + (double)m0[c] * (double)m0[c];
      if (i2 != i) {
#pragma unroll
        for (int c = 0; c < VE; ++c) acc2 += (double)r1[c] * (double)r1[c] + (double)m1[c] * (double)m1[c];
      }
    }
    acc += acc2;
  } else {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += stride) {
      const double r = re[i], m = im[i];
      acc += r * r + m * m;
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int w = 0; w < kBlock / 64; ++w) s += part[w];
    atomicAdd(out, s);
  }
}

// ---------------------------------------------------------------------------------
// probabilities / project: device side of the Measure and Projection functional gates
// (hybridq/gate/measure.py:25-125, gate/projection.py:25-119), so that circuits containing
// them need no D2H round trip of the state.
// ---------------------------------------------------------------------------------
struct BitsArg {
  unsigned k;
  unsigned pos[kMaxK];  // bit j of the outcome index <-> index bit pos[j]
};

template <typename T>
__global__ void __launch_bounds__(kBlock)
probabilities_kernel(const T* __restrict__ re, const T* __restrict__ im, const uint64_t size,
                     const BitsArg ba, double* __restrict__ out /* 2^k, pre-zeroed */) {
  HQ_DYN_LDS(smem);
  double* bins = reinterpret_cast<double*>(smem);
  const unsigned nb = 1u << ba.k;
  for (unsigned i = threadIdx.x; i < nb; i += kBlock) bins[i] = 0.0;
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t x = (uint64_t)blockIdx.x * kBlock + threadIdx.x; x < size; x += stride) {
    unsigned t = 0;
    for (unsigned j = 0; j < ba.k; ++j) t |= (unsigned)((x >> ba.pos[j]) & 1ull) << j;
    const double r = re[x], m = im[x];
    atomicAdd(&bins[t], r * r + m * m);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < nb; i += kBlock)
    if (bins[i] != 0.0) atomicAdd(&out[i], bins[i]);
}

// Streaming variant for n >= 16: a workgroup walks chunks of 2^16 amplitudes as 16-byte vectors
// (index = chunk : it[6 bits] : thread[8 bits] : component[CB bits]).  Measured bits in the
// component / thread / chunk fields are constant per register, thread or chunk; only measured
// bits in the 6 `it` bits change inside a chunk, and the loop is ordered so that they form the
// OUTER loop: a thread sums a whole inner loop in registers and issues one LDS atomic per
// (outer value, component class) instead of one per amplitude.
template <typename T>
__global__ void __launch_bounds__(kBlock)
probabilities_stream_kernel(const T* __restrict__ re, const T* __restrict__ im, const unsigned n,
                            const BitsArg ba, double* __restrict__ out /* 2^k, pre-zeroed */) {
  using V = typename Vec<T>::type;
  constexpr unsigned CB = Vec<T>::VB, NC = 1u << CB;
  constexpr unsigned TB = 8, IB = 6, CHUNK = CB + TB + IB;  // bits of the thread / it fields; 2^CHUNK amplitudes per chunk
  HQ_DYN_LDS(smem);
  double* bins = reinterpret_cast<double*>(smem);
  const unsigned nb = 1u << ba.k;
  for (unsigned i = threadIdx.x; i < nb; i += kBlock) bins[i] = 0.0;
  __syncthreads();
  // outcome contributions of the fields
  unsigned comp_t[NC];
#pragma unroll
  for (unsigned c = 0; c < NC; ++c) comp_t[c] = 0;
  unsigned thr_t = 0, it_meas[IB], n_it_meas = 0, it_free[IB], n_it_free = 0, it_out[IB];
  for (unsigned b = 0; b < IB; ++b) {
    bool measured = false;
    for (unsigned j = 0; j < ba.k; ++j)
      if (ba.pos[j] == CB + TB + b) { it_meas[n_it_meas] = b; it_out[n_it_meas] = j; ++n_it_meas; measured = true; }
    if (!measured) it_free[n_it_free++] = b;
  }
  for (unsigned j = 0; j < ba.k; ++j) {
    const unsigned p = ba.pos[j];
    if (p < CB) {
#pragma unroll
      for (unsigned c = 0; c < NC; ++c) comp_t[c] |= ((c >> p) & 1u) << j;
    } else if (p < CB + TB) {
      thr_t |= ((threadIdx.x >> (p - CB)) & 1u) << j;
    }
  }
  const V* __restrict__ vre = reinterpret_cast<const V*>(re);
  const V* __restrict__ vim = reinterpret_cast<const V*>(im);
  const uint64_t nchunks = 1ull << (n - CHUNK);
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    unsigned hi_t = 0;
    for (unsigned j = 0; j < ba.k; ++j)
      if (ba.pos[j] >= CHUNK) hi_t |= (unsigned)((chunk >> (ba.pos[j] - CHUNK)) & 1ull) << j;
    const uint64_t vbase = (chunk << (TB + IB)) + threadIdx.x;
    for (unsigned om = 0; om < (1u << n_it_meas); ++om) {
      unsigned it0 = 0, it_t = 0;
      for (unsigned b = 0; b < n_it_meas; ++b) {
        it0 |= ((om >> b) & 1u) << it_meas[b];
        it_t |= ((om >> b) & 1u) << it_out[b];
      }
      double acc[NC];
#pragma unroll
      for (unsigned c = 0; c < NC; ++c) acc[c] = 0.0;
      for (unsigned f = 0; f < (1u << n_it_free); ++f) {
        unsigned it = it0;
        for (unsigned b = 0; b < n_it_free; ++b) it |= ((f >> b) & 1u) << it_free[b];
        const V r = __builtin_nontemporal_load(vre + vbase + ((uint64_t)it << TB));
        const V m = __builtin_nontemporal_load(vim + vbase + ((uint64_t)it << TB));
#pragma unroll
        for (unsigned c = 0; c < NC; ++c) acc[c] += (double)r[c] * (double)r[c] + (double)m[c] * (double)m[c];
      }
      const unsigned t0 = hi_t | thr_t | it_t;
      // components that fall into the same outcome are summed first
#pragma unroll
      for (unsigned c = 0; c < NC; ++c) {
        bool first = true;
        double sum = acc[c];
#pragma unroll
        for (unsigned d = 0; d < NC; ++d)
          if (d != c && comp_t[d] == comp_t[c]) {
            if (d < c) first = false; else sum += acc[d];
          }
        if (first) atomicAdd(&bins[t0 | comp_t[c]], sum);
      }
    }
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < nb; i += kBlock)
    if (bins[i] != 0.0) atomicAdd(&out[i], bins[i]);
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
project_kernel(T* __restrict__ re, T* __restrict__ im, const uint64_t size, const uint64_t mask,
               const uint64_t want, const T scale) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t x = (uint64_t)blockIdx.x * kBlock + threadIdx.x; x < size; x += stride) {
    const bool keep = (x & mask) == want;
    re[x] = keep ? re[x] * scale : (T)0;
    im[x] = keep ? im[x] * scale : (T)0;
  }
}

// <a|b> = sum conj(a) b on split planes, accumulated in double: out[0] += sum(ar*br + ai*bi),
// out[1] += sum(ar*bi - ai*br)   (expectation values, simulation.py:1125-1216)
template <typename T>
__global__ void __launch_bounds__(kBlock)
vdot_kernel(const T* __restrict__ are, const T* __restrict__ aim, const T* __restrict__ bre,
            const T* __restrict__ bim, const uint64_t size, double* __restrict__ out) {
  HQ_LDS double part[2][kBlock / 64];
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  double sr = 0, si = 0;
  using V = typename Vec<T>::type;
  constexpr int VE = 1 << Vec<T>::VB;
  const bool al = size % VE == 0 && reinterpret_cast<uintptr_t>(are) % 16 == 0 && reinterpret_cast<uintptr_t>(aim) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(bre) % 16 == 0 && reinterpret_cast<uintptr_t>(bim) % 16 == 0;
  if (al) {  // 16-byte non-temporal loads of the four planes (as norm2_kernel)
    const uint64_t nvec = size / VE;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
      const V ar = __builtin_nontemporal_load(reinterpret_cast<const V*>(are) + i), ai = __builtin_nontemporal_load(reinterpret_cast<const V*>(aim) + i);
      const V br = __builtin_nontemporal_load(reinterpret_cast<const V*>(bre) + i), bi = __builtin_nontemporal_load(reinterpret_cast<const V*>(bim) + i);
#pragma unroll
      for (int c = 0; c < VE; ++c) {
        sr += (double)ar[c] * (double)br[c] + (double)ai[c] * (double)bi[c];
        si += (double)ar[c] * (double)bi[c] - (double)ai[c] * (double)br[c];
      }
    }
  } else {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += stride) {
      const double ar = are[i], ai = aim[i], br = bre[i], bi = bim[i];
      sr += ar * br + ai * bi;
      si += ar * bi - ai * br;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    sr += __shfl_down(sr, o, 64);
    si += __shfl_down(si, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    part[0][threadIdx.x >> 6] = sr;
    part[1][threadIdx.x >> 6] = si;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int w = 0; w < kBlock / 64; ++w) { a += part[0][w]; b += part[1][w]; }
    atomicAdd(&out[0], a);
    atomicAdd(&out[1], b);
  }
}

}  // namespace hq
