/*
 * hq_oracle.c -- CPU restatement of the HybridQ state-vector evolution core.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the "port"
 * CPU baseline.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from it.  The product path
 * (hybridq_amd/) never links, imports or falls back to it.
 *
 * Parity pin: checked against (a) the reference C++ core compiled from
 * /root/reference/include/python_{U,swap}.cpp into oracle/_ref/ (tests/test_oracle.py,
 * only where /root/reference exists) and (b) the committed golden vectors under
 * tests/golden/ which were produced by that compiled reference and by the
 * reference's Python driver (tests/golden/make_golden.py).
 *
 * It is written from the index semantics of the reference, not from its
 * template machinery:
 *
 *   apply_U   follows include/U.h:28-102 (k<=4) and U.h:123-202 (generic k):
 *             for every base index b with (b & mask)==0 and tile row t,
 *             idx(b,t) = b | sum_j ((t>>j)&1) << pos[j];
 *             out[idx(b,t)] = sum_s U[t][s] * in[idx(b,s)]   (complex,
 *             accumulated in float_type in the order s = 0..2^k-1 with
 *             re += Ur*xr - Ui*xi ; im += Ur*xi + Ui*xr, U.h:93-94,192-193).
 *             U is row-major interleaved (re,im) (U.h:63-64,190-191).
 *             Unlike the reference there is no "pos >= log2_pack_size"
 *             restriction (U.h:48-54): any distinct positions < n are accepted.
 *   swap      follows include/swap.h:28-95:
 *             new[x] = old[(x & ~(2^s-1)) | sum_i ((x>>i)&1) << pos[i]].
 *   to_complex follows include/python_U.cpp:114-123.
 *
 * Exported symbols mirror include/python_U.cpp:127-154 and
 * include/python_swap.cpp:68-99 so the same ctypes binding drives this
 * library, oracle/_ref/ and the HIP library.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef HQO_LOG2_PACK_SIZE
#define HQO_LOG2_PACK_SIZE 3 /* what the reference reports for AVX2 builds (Makefile:63-72) */
#endif

#define HQO_CHUNK 16 /* contiguous low-index run processed together (vectorisable) */

static int check_positions(const unsigned int *pos, unsigned int n, unsigned int k) {
  if (k > n) return 1;
  uint64_t seen = 0;
  for (unsigned int i = 0; i < k; ++i) {
    if (pos[i] >= n || pos[i] >= 64) return 1;
    if (seen & (1ull << pos[i])) return 1;
    seen |= 1ull << pos[i];
  }
  return 0;
}

/* insert zero bits at the (ascending) positions sp[0..k) into x */
static inline uint64_t deposit_zeros(uint64_t x, const unsigned int *sp, unsigned int k) {
  for (unsigned int i = 0; i < k; ++i) {
    const uint64_t lo = (1ull << sp[i]) - 1;
    x = ((x & ~lo) << 1) | (x & lo);
  }
  return x;
}

static void sort_positions(const unsigned int *pos, unsigned int k, unsigned int *sp) {
  memcpy(sp, pos, k * sizeof(unsigned int));
  for (unsigned int i = 1; i < k; ++i) {
    unsigned int v = sp[i], j = i;
    while (j > 0 && sp[j - 1] > v) { sp[j] = sp[j - 1]; --j; }
    sp[j] = v;
  }
}

#define DEFINE_APPLY_U(NAME, T)                                                         \
  int NAME(T *psi_re, T *psi_im, const T *U, const unsigned int *pos,                   \
           const unsigned int n, const unsigned int k) {                                \
    if (k == 0) return 0; /* python_U.cpp:38-39 */                                      \
    if (check_positions(pos, n, k)) return 1;                                           \
    const size_t dim = (size_t)1 << k;                                                  \
    unsigned int sp[64];                                                                \
    sort_positions(pos, k, sp);                                                         \
    /* offsets of the 2^k tile rows relative to the base index */                       \
    uint64_t *off = (uint64_t *)malloc(dim * sizeof(uint64_t));                         \
    if (!off) return 1;                                                                 \
    for (size_t t = 0; t < dim; ++t) {                                                  \
      uint64_t o = 0;                                                                   \
      for (unsigned int j = 0; j < k; ++j) o |= (uint64_t)((t >> j) & 1u) << pos[j];    \
      off[t] = o;                                                                       \
    }                                                                                   \
    /* contiguous run below the lowest target position */                              \
    const unsigned int low = sp[0];                                                     \
    const size_t run = (size_t)1 << (low < 4 ? low : 4); /* <= HQO_CHUNK */             \
    const uint64_t n_tiles = (uint64_t)1 << (n - k);                                    \
    const uint64_t n_outer = n_tiles / run;                                             \
    int fail = 0;                                                                       \
    _Pragma("omp parallel")                                                             \
    {                                                                                   \
      T *xr = (T *)malloc(dim * HQO_CHUNK * sizeof(T));                                 \
      T *xi = (T *)malloc(dim * HQO_CHUNK * sizeof(T));                                 \
      if (!xr || !xi) {                                                                 \
        _Pragma("omp atomic write") fail = 1;                                           \
      }                                                                                 \
      _Pragma("omp barrier")                                                            \
      if (!fail) {                                                                      \
        _Pragma("omp for schedule(static)")                                             \
        for (uint64_t o = 0; o < n_outer; ++o) {                                        \
          const uint64_t base = deposit_zeros(o * run, sp, k);                          \
          for (size_t t = 0; t < dim; ++t) {                                            \
            const T *sr = psi_re + base + off[t];                                       \
            const T *si = psi_im + base + off[t];                                       \
            for (size_t l = 0; l < run; ++l) {                                          \
              xr[t * HQO_CHUNK + l] = sr[l];                                            \
              xi[t * HQO_CHUNK + l] = si[l];                                            \
            }                                                                           \
          }                                                                             \
          for (size_t t = 0; t < dim; ++t) {                                            \
            T ar[HQO_CHUNK], ai[HQO_CHUNK];                                             \
            for (size_t l = 0; l < run; ++l) ar[l] = ai[l] = (T)0;                      \
            const T *Urow = U + 2 * t * dim;                                            \
            for (size_t s = 0; s < dim; ++s) {                                          \
              const T ur = Urow[2 * s], ui = Urow[2 * s + 1];                           \
              const T *pr = xr + s * HQO_CHUNK, *pi = xi + s * HQO_CHUNK;               \
              for (size_t l = 0; l < run; ++l) {                                        \
                ar[l] += ur * pr[l] - ui * pi[l];                                       \
                ai[l] += ur * pi[l] + ui * pr[l];                                       \
              }                                                                         \
            }                                                                           \
            T *dr = psi_re + base + off[t];                                             \
            T *di = psi_im + base + off[t];                                             \
            for (size_t l = 0; l < run; ++l) {                                          \
              dr[l] = ar[l];                                                            \
              di[l] = ai[l];                                                            \
            }                                                                           \
          }                                                                             \
        }                                                                               \
      }                                                                                 \
      free(xr);                                                                         \
      free(xi);                                                                         \
    }                                                                                   \
    free(off);                                                                          \
    return fail;                                                                        \
  }

DEFINE_APPLY_U(apply_U_float32, float)
DEFINE_APPLY_U(apply_U_float64, double)

#define DEFINE_TO_COMPLEX(NAME, T)                                              \
  int NAME(T *psi_re, T *psi_im, T *psi_out, const unsigned int size) {         \
    _Pragma("omp parallel for schedule(static)")                                \
    for (size_t i = 0; i < (size_t)size; ++i) {                                 \
      psi_out[2 * i + 0] = psi_re[i];                                           \
      psi_out[2 * i + 1] = psi_im[i];                                           \
    }                                                                           \
    return 0;                                                                   \
  }

DEFINE_TO_COMPLEX(to_complex64, float)
DEFINE_TO_COMPLEX(to_complex128, double)

unsigned int get_log2_pack_size(void) { return HQO_LOG2_PACK_SIZE; }

/* swap: pos must be a permutation of 0..s-1 (the reference does not check;
 * anything else reads outside the chunk there, swap.h:38,86). */
static int check_swap_positions(const unsigned int *pos, unsigned int n, unsigned int s) {
  if (s > n || s > 30) return 1;
  uint64_t seen = 0;
  for (unsigned int i = 0; i < s; ++i) {
    if (pos[i] >= s) return 1;
    if (seen & (1ull << pos[i])) return 1;
    seen |= 1ull << pos[i];
  }
  return 0;
}

#define DEFINE_SWAP(NAME, T)                                                            \
  int NAME(T *array, const unsigned int *pos, const unsigned int n,                     \
           const unsigned int s) {                                                      \
    if (s == 0) return 0; /* python_swap.cpp:35-36 */                                   \
    if (check_swap_positions(pos, n, s)) return 1;                                      \
    const size_t chunk = (size_t)1 << s;                                                \
    uint32_t *src = (uint32_t *)malloc(chunk * sizeof(uint32_t));                       \
    if (!src) return 1;                                                                 \
    for (size_t x = 0; x < chunk; ++x) {                                                \
      uint32_t y = 0;                                                                   \
      for (unsigned int i = 0; i < s; ++i) y |= (uint32_t)((x >> i) & 1u) << pos[i];    \
      src[x] = y;                                                                       \
    }                                                                                   \
    const uint64_t n_chunks = (uint64_t)1 << (n - s);                                   \
    int fail = 0;                                                                       \
    _Pragma("omp parallel")                                                             \
    {                                                                                   \
      T *buf = (T *)malloc(chunk * sizeof(T));                                          \
      if (!buf) {                                                                       \
        _Pragma("omp atomic write") fail = 1;                                           \
      }                                                                                 \
      _Pragma("omp barrier")                                                            \
      if (!fail) {                                                                      \
        _Pragma("omp for schedule(static)")                                             \
        for (uint64_t c = 0; c < n_chunks; ++c) {                                       \
          T *a = array + c * chunk;                                                     \
          for (size_t x = 0; x < chunk; ++x) buf[x] = a[src[x]];                        \
          memcpy(a, buf, chunk * sizeof(T));                                            \
        }                                                                               \
      }                                                                                 \
      free(buf);                                                                        \
    }                                                                                   \
    free(src);                                                                          \
    return fail;                                                                        \
  }

DEFINE_SWAP(swap_float32, float)
DEFINE_SWAP(swap_float64, double)
DEFINE_SWAP(swap_int32, int32_t)
DEFINE_SWAP(swap_int64, int64_t)
DEFINE_SWAP(swap_uint32, uint32_t)
DEFINE_SWAP(swap_uint64, uint64_t)
