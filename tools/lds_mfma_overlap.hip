// Developer microbenchmark: do LDS traffic (ds_read_b128 / ds_write_b128) and f32 MFMAs of DIFFERENT waves on the
// same SIMD overlap on gfx950?  512-thread workgroups, 64 KiB of LDS each (two per CU, like the cache-blocked
// kernel): waves 0-3 (one per SIMD) run MFMA bursts, waves 4-7 stream their LDS slots.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_mfma_overlap.hip -o /tmp/lds_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bit 0: MFMA waves work, bit 1: LDS waves work, bit 2: every wave does both (read, MFMA, write per iteration)
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode, float a, int flags, unsigned m0, unsigned m1, const unsigned* __restrict__ desc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4* tile = reinterpret_cast<f32x4*>(smem);
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (unsigned i = threadIdx.x; i < 4096; i += 512) tile[i] = f32x4{1, 2, 3, 4};
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x4 x[4] = {{1, 1, 1, 1}, {1, 1, 1, 1}, {1, 1, 1, 1}, {1, 1, 1, 1}};
  const unsigned slot = (wave & 3) * 1024 + lane;  // 4 x 64 vectors per wave, conflict-free
  if ((mode & 4) && (flags & 16)) {
    // the same LDS traffic and pipe time per iteration with v_mfma_f32_32x32x2_f32: 8 instructions of 64 cycles
    f32x16 big[2];
    for (int i = 0; i < 16; ++i) { big[0][i] = 0; big[1][i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = tile[slot + ((r * 64 + it * 256) & 1023 & ~63u)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        big[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[s][0], big[0], 0, 0, 0);
        big[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[s][2], big[1], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[slot + ((r * 64 + it * 256) & 1023 & ~63u)] = f32x4{big[r & 1][r], big[r & 1][r + 4], big[r & 1][r + 8], big[r & 1][r + 12]};
      if ((flags & 1) && (it & 1)) __syncthreads();
    }
    acc[0][0] += big[0][0] + big[1][0];
  } else if ((mode & 4) && (flags & 32)) {
    // MERGED: the two iterations of a "gate" as one body -- 8 reads, 32 MFMAs on 8 accumulators, 8 writes -- and the barrier
    f32x4 acc2[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 y[4];
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = tile[slot + ((r * 64 + it * 256) & 1023 & ~63u)];
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = tile[slot + ((r * 64 + (it + 1) * 256) & 1023 & ~63u)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x[s][c], acc[c], 0, 0, 0);
          acc2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, y[s][c], acc2[c], 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[slot + ((r * 64 + it * 256) & 1023 & ~63u)] = acc[r];
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[slot + ((r * 64 + (it + 1) * 256) & 1023 & ~63u)] = acc2[r];
      if (flags & 1) __syncthreads();
    }
    acc[0][0] += acc2[0][0];
  } else if (mode & 4) {
    // flags bit 0: workgroup barrier every 2 iterations ("per gate"); bit 1: ~25 dependent VALU ops of address
    // arithmetic per iteration; bit 2: a "gate prologue" every 2 iterations (8 ds_read_b32 + ~50 dependent VALU ops)
    unsigned salt = 0;
    float pro = 0;
    for (int it = 0; it < iters; ++it) {
      if ((flags & 8) && !(it & 1)) {  // two dependent scalar loads (a descriptor fetch, then fields chosen by it)
        const unsigned i0 = desc[(it >> 1) & 7];
        const unsigned i1 = desc[8 + (i0 & 7)];
        salt += i1 & 0;  // the table holds values with zero low bits
        if (i1 == 12345u) pro += 1.0f;
      }
      if ((flags & 4) && !(it & 1)) {
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) t[r] = reinterpret_cast<float*>(smem)[lane + r * 64 + (it & 3) * 512];
        unsigned u = lane + salt;
#pragma unroll
        for (int r = 0; r < 50; ++r) u = ((u & m0) << 1) ^ (u & m1) ^ r;
        salt = u & 0;  // m0 = m1 = 0xffffffff at run time would change it; the host passes masks that keep it 0
#pragma unroll
        for (int r = 0; r < 8; ++r) pro += t[r];
      }
      unsigned sl = slot;
      if (flags & 2) {
        unsigned u = slot + it;
#pragma unroll
        for (int r = 0; r < 25; ++r) u = ((u & m0) << 1) | (u & m1);
        sl = slot + (u & 0) + salt;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = tile[sl + ((r * 64 + it * 256) & 1023 & ~63u)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x[s][c], acc[c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[sl + ((r * 64 + it * 256) & 1023 & ~63u)] = acc[r];
      if ((flags & 1) && (it & 1)) __syncthreads();
    }
    acc[0][0] += pro;
  } else if (wave < 4) {
    if (mode & 1)
      for (int it = 0; it < 2 * iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x[s][c], acc[c], 0, 0, 0);
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < 2 * iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = tile[slot + ((r * 64 + it * 256) & 1023 & ~63u)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x[r][0] += 1.0f;
          tile[slot + ((r * 64 + it * 256) & 1023 & ~63u)] = x[r];
        }
      }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + x[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 512 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  const int iters = 4000;
  const char* names[] = {"", "MFMA waves only", "LDS waves only", "MFMA + LDS waves", "every wave: read, MFMA, write"};
  struct Case { int mode, flags; const char* name; };
  const Case cases[] = {{1, 0, "MFMA waves only"}, {2, 0, "LDS waves only"}, {3, 0, "MFMA + LDS waves"},
                        {4, 0, "every wave: read, MFMA, write"}, {4, 1, " + barrier / 2 it"}, {4, 2, " + 25 VALU / it"},
                        {4, 4, " + prologue / 2 it"}, {4, 3, " + barrier + VALU"}, {4, 5, " + barrier + prologue"},
                        {4, 7, " + barrier + VALU + prologue"}, {4, 8, " + 2 dependent s_loads / 2 it"},
                        {4, 9, " + barrier + s_loads"}, {4, 15, " + barrier + VALU + prologue + s_loads"},
                        {4, 16, "32x32x2: read, MFMA, write"}, {4, 17, "32x32x2 + barrier / 2 it"},
                        {4, 32, "merged pair: 8 rd, 32 MFMA, 8 wr"}, {4, 33, "merged pair + barrier / pair"}};
  unsigned* desc;
  hipMalloc(&desc, 64 * 4);
  hipMemset(desc, 0, 64 * 4);
  for (const Case& c : cases) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(512), dim3(512), 64 * 1024, 0, out, iters, c.mode, 1.0f, c.flags, 0x0u, 0x3ffu, desc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(512), dim3(512), 64 * 1024, 0, out, iters, c.mode, 1.0f, c.flags, 0x0u, 0x3ffu, desc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per CU and "iteration pair": 16 wave-iterations of 16 MFMAs (2048 pipe cycles), 128 KiB of LDS traffic
    std::printf("%-32s %8.3f ms   %7.0f cycles@2.4GHz per 16 wave-iterations\n", c.name, ms, ms * 2.4e6 / iters);
  }
  return 0;
}
