/*
 * hq_hip.h -- C ABI of libhq_hip.so, the MI355X (gfx950) state-vector evolution core.
 *
 * Part 1 is the drop-in boundary: exactly the symbols the reference's ctypes
 * bindings look up in hybridq.so / hybridq_swap.so, with the same argument
 * meaning, in-place semantics and "int, 0 = ok" error convention.
 * Part 2 are extensions the reference has no counterpart for (stream control,
 * 64-bit counts, device-side state helpers).
 *
 * Pointer kinds.  `psi_re/psi_im/array/psi_out` may be DEVICE pointers (the
 * normal case: the state lives in HBM; the call enqueues kernels on the
 * library's stream and returns without synchronising) or HOST pointers
 * (compatibility path for the unmodified reference Python: the planes are
 * staged H2D, processed and copied back before the call returns).  The kind is
 * detected with hipPointerGetAttributes.  `U` and `pos` are always HOST
 * pointers and are only read during the call (they are Python temporaries in
 * the reference: hybridq/circuit/simulation/simulation.py:633-644).
 *
 * Environment (read once): HQ_LOG2_PACK_SIZE (1..5, what get_log2_pack_size()
 * reports), HQ_APPLY_MODE / HQ_NONTEMPORAL (initial hq_set_apply_mode values),
 * HQ_PROGRAM_MB (table buffer of a recorded program, default 64),
 * HQ_PROGRAM_GRAPH (0: replay programs as a launch loop instead of a hipGraph);
 * measurement switches: HQ_GEMM_TB (tile bits of the k >= 7 kernel), HQ_BLOCKED_ALDS (0: operand and address tables of
 * blocked passes stay in global memory / are computed per gate), HQ_BLOCKED_GRID (cap, a power of two, on the resident
 * workgroups of a blocked pass), HQ_BLOCKED_PREF, HQ_GEMM_PREF, HQ_SWAP_PREF (0: no register prefetch of the next tile in
 * the cache-blocked / k >= 7 / low-bit-swap kernels), HQ_BIG_PHASED, HQ_BIG_GRID (k = 5, 6 kernel), HQ_SWAP_TWO_PASS (0: one
 * 128 KiB-tile pass or gather + copy for s > 13), HQ_VMM_FREE_VA (1: hq_free also releases the virtual range).
 *
 * OPT-IN kernel variants (all default 0 = the code the driver's GPU tests have run; each is a template parameter of the
 * same build, exercised against the oracle in every setting by tests/test_gpu_round4.py and timed by bench.py's
 * `blocked_variants` leg, so that one hardware run decides them):
 *   HQ_BLOCKED_PIPE=1    cache-blocked inner gates: LDS operands requested one wave-iteration ahead of the matrix cores
 *   HQ_BLOCKED_GROUPS=1  cache-blocked passes: barrier-free wave groups instead of a workgroup barrier after every inner gate
 *   HQ_BLOCKED_DIRECT=1  tile movement of a blocked pass folded into its first gate (implies HQ_BLOCKED_PIPE=1)
 *   HQ_BLOCKED_BIG=1     128 KiB tiles on one 1024-thread workgroup per CU (implies HQ_BLOCKED_PIPE=1)
 *   HQ_GEMM_PIPE=1       k >= 7 tile GEMM: operands requested ahead of the matrix cores in the K loop
 *   HQ_BIG_TWOBASE=1     complex128 k = 6 role kernel: second LDS base address (no scratch) + operand pipeline
 * HQ_BLOCKED_R3 (default 1): a cache-blocked pass that uses none of the variants above is launched on the kernel family of
 * the last commit that ran on hardware (csrc/hq_kernels_blocked_r3.h: instruction for instruction that commit's binaries,
 * tools/isa_vs_round.py); 0: on the PIPE = false instantiations of the current family (the code the variants are built on).
 * HQ_BLOCKED_SELFCHECK (default 3): how many of the first blocked passes of a process THAT RUN ONE OF THE OPT-IN
 * VARIANTS are cross-checked against the default kernels on a scratch state (hq_blocked_selfcheck); a default run never
 * checks anything, and a check that cannot run (no memory, stream under capture) is skipped, never an error.
 */
#ifndef HQ_HIP_H
#define HQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libhq_hip.so is built with -fvisibility=hidden: the prototypes below are its ENTIRE dynamic symbol table (like the
 * reference's hybridq.so / hybridq_swap.so, which export their C symbols and nothing else: python_U.cpp:127-154,
 * python_swap.cpp:68-99); tests/test_abi.py compares `nm -D` with this header in both directions. */
#pragma GCC visibility push(default)

/* ------------------------------------------------------------------------- */
/* Part 1 -- reference boundary                                               */
/* ------------------------------------------------------------------------- */

/* Replaces get_log2_pack_size(), /root/reference/include/python_U.cpp:129
 * (bound at hybridq/utils/dot.py:49-50).  The caller promises every target
 * position >= this value (simulation.py:559, dot.py:217-221) and treats 0 as
 * "library missing" (simulation.py:393-397).  The GPU kernels accept ANY
 * position, so this returns 1 by default (env HQ_LOG2_PACK_SIZE or
 * hq_set_log2_pack_size() override it, e.g. 3 to reproduce the reference's
 * exact call sequence). */
unsigned int get_log2_pack_size(void);

/* Replace apply_U_float32/64, python_U.cpp:131-143 (bound at dot.py:53-62;
 * called at simulation.py:640-646 and dot.py:305).  Semantics of
 * include/U.h:28-102,123-202: for every base index b with all target bits
 * clear and every row t,  out[b | dep(t)] = sum_s U[t][s] * in[b | dep(s)],
 * dep(t) = sum_j ((t>>j)&1) << pos[j].  U: row-major 2^k x 2^k, interleaved
 * (re,im).  pos: k distinct positions < n_qubits in any order.  In place.
 * Returns 0 on success, 1 on invalid arguments (U.h:34-36,48-54) or a HIP
 * error.  n_pos == 0 is a no-op returning 0 (python_U.cpp:38-39).  Limits:
 * n_pos <= 10 (what dot.py:236 allows). */
int apply_U_float32(float *psi_re, float *psi_im, const float *U, const unsigned int *pos,
                    unsigned int n_qubits, unsigned int n_pos);
int apply_U_float64(double *psi_re, double *psi_im, const double *U, const unsigned int *pos,
                    unsigned int n_qubits, unsigned int n_pos);

/* Replace to_complex64/128, python_U.cpp:145-153 (bound at dot.py:64-71;
 * called at simulation.py:671-674, dot.py:111): out[2i]=re[i], out[2i+1]=im[i]. */
int to_complex64(float *psi_re, float *psi_im, float *psi_out, unsigned int size);
int to_complex128(double *psi_re, double *psi_im, double *psi_out, unsigned int size);

/* Replace swap_{float,int,uint}{32,64}, /root/reference/include/python_swap.cpp:70-98
 * (bound at hybridq/utils/transpose.py:52-58; called at simulation.py:623-630,
 * 658-663, dot.py:291-317, transpose.py:148).  Semantics of include/swap.h:28-95:
 * new[x] = old[(x & ~(2^s-1)) | sum_i ((x>>i)&1) << pos[i]], s = n_pos; pos must
 * be a permutation of 0..s-1 (the reference does not check; we return 1). */
int swap_float32(float *array, const unsigned int *pos, unsigned int n_qubits, unsigned int n_pos);
int swap_float64(double *array, const unsigned int *pos, unsigned int n_qubits, unsigned int n_pos);
int swap_int32(int *array, const unsigned int *pos, unsigned int n_qubits, unsigned int n_pos);
int swap_int64(long *array, const unsigned int *pos, unsigned int n_qubits, unsigned int n_pos);
int swap_uint32(unsigned int *array, const unsigned int *pos, unsigned int n_qubits,
                unsigned int n_pos);
int swap_uint64(unsigned long *array, const unsigned int *pos, unsigned int n_qubits,
                unsigned int n_pos);

/* ------------------------------------------------------------------------- */
/* Part 2 -- extensions (no reference counterpart)                            */
/* ------------------------------------------------------------------------- */

/* HIP stream (hipStream_t as void*) that device-pointer calls enqueue on.
 * Default: the null stream.  Pass torch.cuda.current_stream().cuda_stream to
 * order the kernels with torch / RCCL work. */
int hq_set_stream(void *hip_stream);
/* hipStreamSynchronize on the library's stream. */
int hq_sync(void);
/* What get_log2_pack_size() reports (must be >= 1). */
int hq_set_log2_pack_size(unsigned int v);
/* Last error message of this library (never NULL). */
const char *hq_last_error(void);
/* Number of visible HIP devices (0 if none / runtime failure). */
int hq_device_count(void);
/* Kernel variant selection for A/B measurements: name in {"auto","mfma","direct","tile",
 * "gemm","generic","naive"} selects the apply_U kernel family -- auto = matrix-core role
 * kernels for k <= 6, tile GEMM for k = 7..10, VALU kernels below their size limits;
 * "tile" = LDS tile GEMM for k = 5,6; "direct" = VALU butterflies k <= 3; "generic" = VALU
 * LDS tile any k (a forced family that cannot run a call falls back to auto); {"nt=auto","nt=0","nt=1"} the non-temporal policy;
 * {"dummy=auto","dummy=comp","dummy=low","dummy=high"} the placement of identity digits in the
 * matrix-core kernel.  Returns 1 for an unknown name. */
int hq_set_apply_mode(const char *name);
/* Name of the kernel family the last apply_U call dispatched to. */
const char *hq_last_kernel(void);
/* Full template instantiation name of that kernel, as rocprofv3 prints it (e.g.
 * "apply_mfma_f32_kernel<4, 0, 2, true>"). */
const char *hq_last_kernel_desc(void);

/* 64-bit-count variants (the reference's `unsigned int size` overflows at
 * n = 32, python_U.cpp:116,145,150). */
int hq_to_complex64(float *psi_re, float *psi_im, float *psi_out, uint64_t size);
int hq_to_complex128(double *psi_re, double *psi_im, double *psi_out, uint64_t size);

/* ---- State memory owned by the library (hq_alloc_state / hq_free_state; counterpart of the aligned planes
 * of hybridq/circuit/simulation/simulation.py:491-494).  Any device memory may be passed to the other entry points; this
 * allocator exists because WHERE the planes live is worth 70 % -> 80 % of the HBM peak for the streaming gate kernels
 * (DESIGN.md section 2).
 *
 * hq_alloc_state: both planes of an n-qubit state in ONE allocation -- *psi_re at the base, *psi_im 2^n elements plus a
 * 12 KiB pad later (the pad keeps the two streams of a kernel out of step in the memory-channel hash), both 32-byte
 * aligned (U.h:34-36).  States of >= 256 MiB get a TUNED PLACEMENT: the library maps 2-16 MiB physical granules itself
 * (HIP virtual-memory management), draws up to 8 placements (env HQ_STATE_TRIES), probes each with 12 gate applications
 * on the library's stream and keeps the fastest (~0.3 s per draw at n = 30; stops after three draws once one streams
 * >= 6.25 TB/s and is a clear winner).  hq_free_state keeps ONE winning placement per state size in a pool: the next
 * hq_alloc_state of that size returns it at once -- no second search, no second virtual range (a retired range is never
 * handed back to the driver: unmap + immediate remap kept stale translations on this stack; HQ_VMM_FREE_VA=1 overrides).
 * A request of another size, a failed hipMalloc inside the library or hq_state_pool_trim() release the pool.
 * flags: */
#define HQ_STATE_PLAIN 1     /* hipMalloc memory: what hq_ipc_export can export (peer-to-peer transport) */
#define HQ_STATE_NO_SEARCH 2 /* one mapped placement, no probing */
#define HQ_STATE_NO_POOL 4   /* do not take a pooled placement */
int hq_alloc_state(unsigned int n_qubits, int float_bits /* 32 | 64 */, int flags, void **psi_re, void **psi_im);
int hq_free_state(void *psi_re);
/* JSON text about the placement of a state (psi_re) or of the last allocation (NULL): the draws with their probe times,
 * the layout kept, its probe rate in TB/s, "from_pool". */
int hq_state_info(const void *psi_re, char *buf, uint64_t cap);
int hq_state_pool_trim(void);

/* Raw device memory.  hq_alloc flags bit 0: physically contiguous VRAM (returns 1 when the driver cannot find a
 * contiguous range).  hq_alloc_mapped: n_granules physical granules of `granule` bytes (a multiple of the driver
 * minimum, returned in *granule_min when non-NULL) are created in sequence and granule i is mapped at virtual slot
 * va_slot[i] (a permutation of 0..n_granules-1) of ONE contiguous virtual range.  hq_alloc_scattered:
 * ceil(bytes / granule) granules in creation order (seed 0) or shuffled by `seed`.  Every failure releases what was
 * created.  Free with hq_free. */
int hq_alloc(void **dev_ptr, uint64_t bytes, int flags);
int hq_free(void *dev_ptr);
int hq_alloc_mapped(void **dev_ptr, uint64_t granule, uint64_t n_granules, const uint32_t *va_slot, uint64_t *granule_min);
int hq_alloc_scattered(void **dev_ptr, uint64_t bytes, uint64_t granule, uint64_t seed);

/* Device-side initial states (counterpart of prepare_state,
 * hybridq/circuit/simulation/utils.py:106-113): kind 0 -> |basis> (re[basis]=1),
 * kind 1 -> uniform |+...+> (re[i] = 2^{-n/2}).  Device pointers only. */
int hq_init_state_float32(float *psi_re, float *psi_im, unsigned int n_qubits, int kind,
                          uint64_t basis);
int hq_init_state_float64(double *psi_re, double *psi_im, unsigned int n_qubits, int kind,
                          uint64_t basis);

/* Product states of '0' / '1' / '+' / '-' factors written on the device (the mixed-string branch of
 * prepare_state, hybridq/circuit/simulation/utils.py:115-153, which builds a 2^n kron on the
 * host): with X = hi_bits | x for the local index x in [0, 2^n_local),
 *   re[x] = ((X & mask01) == val01) ? (-1)^popcount(X & mask_minus) * 2^(-n_pm/2) : 0,  im[x] = 0.
 * mask01 / val01: index bits holding a '0'/'1' character and their values; mask_minus: bits holding
 * '-'; n_pm: number of '+' and '-' characters.  hi_bits = 0 on one GPU; rank << n_local for the
 * shard of a multi-GPU state.  Device pointers only, n_local >= 2. */
int hq_init_product_state_float32(float *psi_re, float *psi_im, unsigned int n_local, uint64_t hi_bits,
                                  uint64_t mask01, uint64_t val01, uint64_t mask_minus, unsigned int n_pm);
int hq_init_product_state_float64(double *psi_re, double *psi_im, unsigned int n_local, uint64_t hi_bits,
                                  uint64_t mask01, uint64_t val01, uint64_t mask_minus, unsigned int n_pm);

/* Out-of-place permutation of ARBITRARY index bits of an array of 2^n 4-byte
 * (_32) or 8-byte (_64) elements: dst[x] = src[pi(x)], where bit i of x moves to
 * bit perm[i] of pi(x) (perm = a permutation of 0..n-1, any number of moved bits).
 * Generalises swap_* (low bits only, in place) to the whole index; the multi-GPU
 * driver uses it to bring qubits into the exchange slots.  Device pointers only. */
int hq_permute_bits_32(const void *src, void *dst, const unsigned int *perm, unsigned int n);
int hq_permute_bits_64(const void *src, void *dst, const unsigned int *perm, unsigned int n);

/* ---- Multi-GPU: high-qubit shards and the qubit exchange (the reference has no counterpart,
 * hybridq/circuit/simulation/simulation.py:379-380).  One process per GPU; rank r of G = 2^g holds the
 * 2^n_local amplitudes whose top g index bits equal r.  A gate on a global qubit is preceded by ONE
 * hq_exchange_* call that swaps the g global index bits with the top g local ones.
 *
 * Transports (pick one per process after hipSetDevice):
 *   hq_shard_init_rccl   RCCL: rank 0 obtains 128 bytes from hq_shard_unique_id, every rank receives them
 *                        out of band (the Python driver uses torch.distributed) and calls init (collective,
 *                        ncclCommInitRank).  hq_shard_attach_rccl adopts an ncclComm_t the caller owns.
 *                        librccl is dlopen()ed (env HQ_RCCL_LIBRARY overrides the search), so the library
 *                        has no link-time dependency on it.
 *   hq_shard_init_p2p    peer-to-peer stores over xGMI: every rank maps the other ranks' planes with
 *                        hq_ipc_export / hq_ipc_open (HIP IPC, dmabuf) and registers, per local plane, the
 *                        address of the same plane on every rank (hq_shard_p2p_register; entry `rank` is
 *                        the local address).  The caller MUST bracket hq_exchange_* with barriers: no rank
 *                        may start before all ranks finished using their dst planes, none may touch dst
 *                        before all ranks' exchange has completed (hq_sync + barrier).
 * hq_exchange_*: src/dst = the rank's two shard buffers (split planes, device pointers).  `perm` (NULL =
 * none) is a local bit permutation applied on the way (dst bit i <- src bit perm[i], as hq_permute_bits):
 * the eviction that moves the outgoing qubits to the top g local bits costs no pass of its own.  Effect:
 * with P = permuted src cut into G chunks by its top g local bits, chunk j of rank r becomes chunk r of
 * rank j.  *result_in_src tells where the exchanged shard is: 0 = dst planes, 1 = src planes (RCCL with a
 * permutation packs src -> dst and transfers dst -> src).  Asynchronous on the library stream; both
 * planes travel in one ncclGroup (all 2(G-1) transfers of a GPU at once: xGMI is point to point), the
 * self chunk is copied by an own kernel, and with a permutation the transfer of the re plane overlaps
 * the packing of the im plane on a second stream. */
int hq_shard_unique_id(void *id128);
/* Binds librccl (dlopen + symbol lookup) and nothing else: the part of the RCCL start-up a rank can fail ALONE, so that
 * the ranks can agree on it before anybody enters the collective hq_shard_init_rccl (which blocks until every rank has
 * arrived; it does not hold the library lock meanwhile, and hq_shard_free from another thread cancels it). */
int hq_shard_load_rccl(void);
int hq_shard_init_rccl(unsigned int world, unsigned int rank, const void *id128);
int hq_shard_attach_rccl(void *nccl_comm, unsigned int world, unsigned int rank);
int hq_shard_init_p2p(unsigned int world, unsigned int rank);
int hq_shard_p2p_register(const void *local_plane, void *const *peer_planes);
int hq_shard_info(unsigned int *world, unsigned int *rank, int *transport /* 0 none, 1 rccl, 2 p2p */);
int hq_shard_free(void);
/* Plumbing check that needs no second GPU: one grouped ncclSend + ncclRecv of `bytes` (device buffers)
 * with this rank as its own peer, through the same stream / event ordering as the exchange.  Needs a
 * communicator (a one-rank hq_shard_init_rccl with a unique id is legal). */
int hq_shard_rccl_selftest(const void *src, void *dst, uint64_t bytes);
/* Ranks of the library's RCCL communicator as RCCL itself reports them (ncclCommCount); 0 when the transport is not RCCL.
 * What a multi-GPU record cites to show that the exchange really ran between `count` processes over RCCL. */
int hq_shard_comm_count(int *count);
int hq_ipc_export(const void *dev_ptr, void *handle64, uint64_t *offset);
int hq_ipc_open(const void *handle64, uint64_t offset, void **dev_ptr);
int hq_ipc_close(void *dev_ptr, uint64_t offset);
int hq_exchange_float32(float *src_re, float *src_im, float *dst_re, float *dst_im, unsigned int n_local,
                        const unsigned int *perm, int *result_in_src);
int hq_exchange_float64(double *src_re, double *src_im, double *dst_re, double *dst_im, unsigned int n_local,
                        const unsigned int *perm, int *result_in_src);
/* The same exchange in 2^sub_bits ROUNDS (sub_bits <= 6, n_local >= 2g + 2 + sub_bits): round s moves piece s (of
 * 2^sub_bits) of every chunk, both planes -- all 4(G-1) transfers of a round (a send and a receive per peer and plane) in
 * ONE ncclGroup, so every xGMI link is busy in every round; only round 0 of an exchange with a folded permutation uses one
 * group per plane (the re plane leaves while im is still being packed) -- and the library stream does NOT wait for the transfers: hq_exchange_round_wait(s) makes it wait for
 * round s, after which pieces [chunk j][piece s] of the result planes (where *result_in_src says) are final and the
 * caller may work on them while the later rounds are still on the wire (hybridq_amd.dist: the local gates attached to
 * an exchange run on the pieces as they land).  The caller waits for EVERY round, the last one at the latest before it
 * touches either plane pair in any other way.  The pack pass stays folded in (a plane's rounds start when that plane has
 * been packed; the rounds of re overlap the pack of im).  *n_rounds: the number of rounds actually used -- 2^sub_bits on
 * the RCCL transport, 1 on the peer-to-peer transport (whose stores the caller brackets with barriers as for
 * hq_exchange_*) and on a single rank; hq_exchange_round_wait(0) is then a no-op in stream order.  A call that is
 * rejected (bad arguments, no transport) changes no state and leaves NO round to wait for: hq_exchange_round_wait then
 * fails for every round, as it does after hq_shard_free. */
int hq_exchange_rounds_float32(float *src_re, float *src_im, float *dst_re, float *dst_im, unsigned int n_local,
                               const unsigned int *perm, unsigned int sub_bits, int *result_in_src, unsigned int *n_rounds);
int hq_exchange_rounds_float64(double *src_re, double *src_im, double *dst_re, double *dst_im, unsigned int n_local,
                               const unsigned int *perm, unsigned int sub_bits, int *result_in_src, unsigned int *n_rounds);
int hq_exchange_round_wait(unsigned int round);

/* sum_i re[i]^2 + im[i]^2 accumulated in double, written to *out (host).
 * Synchronises the stream.  Device pointers only. */
int hq_norm2_float32(const float *psi_re, const float *psi_im, uint64_t size, double *out);
int hq_norm2_float64(const double *psi_re, const double *psi_im, uint64_t size, double *out);

/* Device side of the Measure / Projection functional gates (hybridq/gate/measure.py:25-125,
 * gate/projection.py:25-119).  probabilities: out[t] = sum of |psi[x]|^2 over the x whose
 * bits pos[0..k) spell t (bit j of t <-> index bit pos[j]); `out` is a HOST array of 2^k
 * doubles, the call synchronises.  project: psi[x] *= scale where those bits spell `state`,
 * psi[x] = 0 elsewhere (asynchronous).  Device pointers only, k <= 10. */
int hq_probabilities_float32(const float *psi_re, const float *psi_im, unsigned int n_qubits,
                             const unsigned int *pos, unsigned int n_pos, double *out);
int hq_probabilities_float64(const double *psi_re, const double *psi_im, unsigned int n_qubits,
                             const unsigned int *pos, unsigned int n_pos, double *out);
int hq_project_float32(float *psi_re, float *psi_im, unsigned int n_qubits, const unsigned int *pos,
                       unsigned int n_pos, uint64_t state, double scale);
int hq_project_float64(double *psi_re, double *psi_im, unsigned int n_qubits, const unsigned int *pos,
                       unsigned int n_pos, uint64_t state, double scale);

/* <a|b> = sum_i conj(a_i) b_i on split planes, accumulated in double: out[0] = real part,
 * out[1] = imaginary part (HOST array of 2 doubles; the call synchronises).  Device side of
 * expectation_value (hybridq/circuit/simulation/simulation.py:1125-1216). */
int hq_vdot_float32(const float *a_re, const float *a_im, const float *b_re, const float *b_im,
                    uint64_t size, double *out);
int hq_vdot_float64(const double *a_re, const double *a_im, const double *b_re, const double *b_im,
                    uint64_t size, double *out);

/* Many gates in ONE HBM pass (device pointers): the state is processed in tiles of
 * 2^tile_bits amplitudes spanned by the index bits tile_pos[0..tile_bits) (ascending; must
 * start with the vector-component bits: 0,1 for float32, 0 for float64; 10 <= tile_bits <= 14
 * (float32) / 13 (float64); 64 KiB of LDS = 13 / 12).  Each tile is staged in LDS, the n_gates
 * gates are applied to it in order, and it is written back.  Gate g has k_all[g] (1..4)
 * targets; its positions (GLOBAL index bits, all members of tile_pos, pos[0] = LSB of the
 * matrix index as in apply_U) and its row-major interleaved matrix are the next entries of
 * pos_all / U_all.  Result identical to calling apply_U_* gate by gate. */
int hq_apply_blocked_float32(float *psi_re, float *psi_im, unsigned int n_qubits,
                             const unsigned int *tile_pos, unsigned int tile_bits, unsigned int n_gates,
                             const float *U_all, const unsigned int *pos_all, const unsigned int *k_all);
int hq_apply_blocked_float64(double *psi_re, double *psi_im, unsigned int n_qubits,
                             const unsigned int *tile_pos, unsigned int tile_bits, unsigned int n_gates,
                             const double *U_all, const unsigned int *pos_all, const unsigned int *k_all);

/* The OPT-IN cache-blocked kernel variants (pipelined inner gates, barrier-free wave groups, direct first gate, 1024-thread
 * tiles; environment section above) are cross-checked on the device against the default kernels for the first
 * HQ_BLOCKED_SELFCHECK (default 3) passes of a process whose launch really selects one of them; a variant whose result
 * differs by more than a few units in the last place is switched off for the process with a warning on stderr.  A parity
 * check, not a race detector; a check that cannot run is skipped (one note on stderr), it never fails the caller's apply.
 * *runs / *failures: checks done / failed so far; *switches: bit 0 pipelined gates, bit 1 barrier-free groups, bit 2
 * direct first gate, bit 3 1024-thread tiles currently ON (all 0 on a default run).  Any pointer may be NULL. */
int hq_blocked_selfcheck(int *runs, int *failures, int *switches);

/* Compiled circuits.  Between hq_program_begin() and hq_program_end() every DEVICE-pointer call
 * of apply_U_*, hq_apply_blocked_*, swap_* (LDS path), hq_permute_bits_*, hq_to_complex*,
 * hq_init_state_* and hq_project_* is RECORDED instead of executed: its launch parameters and a
 * private copy of its matrices / operand tables go into the program.  hq_program_run() replays
 * the recorded launches on the library's stream with no planning, no uploads and no host work
 * per gate -- from the second run on as ONE hipGraph launch (HQ_PROGRAM_GRAPH=0: plain launch
 * loop).  The program is bound to the plane pointers it was recorded with.  Calls that return a
 * value to the host or need staging (host pointers, norm2, probabilities, vdot) cannot be
 * recorded and return 1.  Table space: 64 MiB per program (env HQ_PROGRAM_MB). */
int hq_program_begin(void);
int hq_program_end(void **handle);
int hq_program_size(void *handle);   /* recorded launches, -1 for NULL */
int hq_program_run(void *handle);
int hq_program_free(void *handle);

/* ---- host-side planning (no device work; hybridq_amd/csrc/hq_plan.hip) --------------------------------------------
 * The cache-blocked schedule of a run of matrix gates: list scheduling of the dependency DAG into passes over LDS tiles
 * and fusion of each pass's gates (the reference's greedy rule, hybridq/circuit/utils.py:606-669, on gates that commute
 * to `commute_tol`).  No reference counterpart for the schedule itself (the reference applies one fused gate per pass,
 * simulation.py:522-646); hybridq_amd/blocking.py is the same algorithm in Python.
 *   gate g acts on k[g] POSITIONS (index bits), listed with the matrix's most significant qubit first (gate.qubits
 *   order); U holds the matrices one after the other, row-major, interleaved (re, im) doubles.
 *   inner_max: widest fused inner gate (0 = keep the gates as they are, 255 = 3 with a widening round to 4 where cheaper).
 * hq_plan_read fills caller-allocated arrays sized by hq_plan_counts: op_kind[n_ops] (0 plain gate, 1 blocked pass),
 * op_first_gate[n_ops + 1] (its gates are [first, next first)), op_tile[n_ops * tile_bits] (ascending positions of a
 * pass's tile), gate_k[n_gates], gate_positions[n_positions] (most significant first), U[2 * n_matrix_elems]. */
int hq_plan_blocked(unsigned int n_qubits, unsigned int n_gates, const unsigned int *k, const unsigned int *positions,
                    const double *U, unsigned int tile_bits, unsigned int low_bits, unsigned int inner_max,
                    unsigned int min_gates, unsigned int tries, unsigned int fusion_orders, unsigned int elem_bytes,
                    uint64_t seed, double commute_tol, void **plan);
int hq_plan_counts(const void *plan, unsigned int *n_ops, unsigned int *n_gates, uint64_t *n_positions,
                   uint64_t *n_matrix_elems, unsigned int *tile_bits);
int hq_plan_read(const void *plan, unsigned int *op_kind, unsigned int *op_first_gate, unsigned int *op_tile,
                 unsigned int *gate_k, unsigned int *gate_positions, double *U);
int hq_plan_free(void *plan);
/* fusion.fuse = the reference's utils.compress + to_matrix_gate (hybridq/circuit/utils.py:467-685, 419-464) with its options;
 * `qubits` are integer ids below 62 whose ORDER is the order of the labels (a fused gate's qubits come out sorted, most
 * significant first); commute_tol 1e-5 reproduces the reference.  The plan holds one plain op per fused gate.
 * Limits (all three planners): gates of 1..10 qubits, max_n_qubits <= 10, max_n_qubits_matrix is clamped to 12, and the
 * commutator of two gates is only evaluated when their union has at most 12 qubits (wider: treated as not commuting).
 * Nothing thrown inside a planner crosses the ABI: out-of-memory and the like come back as return code 1 + hq_last_error. */
int hq_plan_fuse(unsigned int n_qubits, unsigned int n_gates, const unsigned int *k, const unsigned int *qubits, const double *U,
                 unsigned int max_n_qubits, int use_matrix_commutation, unsigned int max_n_qubits_matrix, uint64_t exclude_mask,
                 double commute_tol, void **plan);
/* fusion.simplify (the reference's utils.simplify, hybridq/circuit/utils.py:825-866 + insert_from_left :122-208) on gates
 * given as in hq_plan_blocked, `qubits` being integer ids below 62: out_index[0 .. *out_count) = the surviving gates in
 * their new order (out_index holds n_gates entries). */
int hq_plan_simplify(unsigned int n_qubits, unsigned int n_gates, const unsigned int *k, const unsigned int *qubits,
                     const double *U, double atol, int use_matrix_commutation, unsigned int max_n_qubits_matrix,
                     int remove_id_gates, unsigned int *out_index, unsigned int *out_count);

/* Diagnostics (no reference counterpart).  hq_pointer_info: the raw hipPointerGetAttributes answer for `p` (memory type,
 * owning device, HIP error code); returns 0 when the runtime knows the pointer.  hq_vmm_remap (tools/placement_remap.py): the
 * SAME physical granules of a buffer from hq_alloc_mapped / hq_alloc_scattered mapped in another order into a fresh virtual
 * range (granule i -> slot va_slot[i]); the old range is retired, *new_ptr replaces dev_ptr (free it with hq_free). */
int hq_pointer_info(const void *p, int *type, int *device, int *err);
int hq_vmm_remap(void *dev_ptr, const uint32_t *va_slot, void **new_ptr);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* HQ_HIP_H */
